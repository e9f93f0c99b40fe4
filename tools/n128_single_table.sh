#!/bin/bash
# Runs on the GPU box: kernel table of ONE trajectory of n = 128 (k = 6, 500 slices) and n = 200 on the GEMM path.  Output -> stdout
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/n128_tab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 128 200; do
  echo "== n $n"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/n$n -- python $R/tools/single_big_n.py $n > $O/n$n.txt 2>/dev/null
  cat $O/n$n.txt
  python $R/tools/rocpd_kernel_stats.py $(ls $O/n$n/*/*_results.db | head -1) 2>&1 | head -24
  rm -rf $O/n$n
done
