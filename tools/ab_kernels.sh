#!/bin/bash
# A/B of alternative libqoc builds under rocprofv3: average duration of the bench kernels.  ab_kernels.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf $R/gpurun_out/ab_prof
  QOC_HIP_LIBRARY=$R/$lib rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_prof -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-single ${AB_ARGS:-} > /dev/null 2>&1
  echo "== $lib"
  python $R/tools/rocpd_kernel_stats.py $(ls $R/gpurun_out/ab_prof/*/*_results.db | head -1) | head -5 | tail -4 | cut -c1-110
done
rm -rf $R/gpurun_out/ab_prof
