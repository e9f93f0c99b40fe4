#!/bin/bash
# Runs on the GPU box (gpurun): regenerates the measurement files that profiles/ keeps.  Output -> gpurun_out/collect/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/collect
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python tools/bench_configs.py > $O/secondary_configs.txt 2>&1
python tools/path_sweep.py > $O/path_sweep.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_c2_single -- python $R/bench.py --seeds-per-gpu 1 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_c3_single -- python $R/tools/bench_configs.py c3 > /dev/null 2>&1
cd $R
for d in prof_bench prof_c2_single prof_c3_single; do
  python tools/rocpd_kernel_stats.py $(ls $O/$d/*/*_results.db | head -1) > $O/$d.txt 2>&1
done
rm -rf $O/prof_bench $O/prof_c2_single $O/prof_c3_single
ls -la $O
