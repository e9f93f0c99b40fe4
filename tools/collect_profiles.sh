#!/bin/bash
# Runs on the GPU box (gpurun): regenerates the measurement files that profiles/ keeps.  Output -> gpurun_out/collect/
# Usage: collect_profiles.sh [quick]   (quick: bench line + its kernel table only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/collect
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
# kernel table of the driver command's timed workload (its dominant-kernel average must reproduce roofline.avg_launch_ms); the secondary configs and the
# live PMC passes of the line (which start rocprofv3 themselves) are left out of this run: their kernels have their own tables below
rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-live-pmc > $O/bench_under_rocprof.json 2>/dev/null
cd $R
python tools/rocpd_kernel_stats.py $(ls $O/prof_bench/*/*_results.db | head -1) > $O/prof_bench.txt 2>&1
rm -rf $O/prof_bench
if [ "${1:-}" = "quick" ]; then ls -la $O; exit 0; fi
python tools/bench_configs.py > $O/secondary_configs.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c2_single -- python $R/bench.py --seeds-per-gpu 1 --steps 50 --warmup 5 --no-cpu-baseline --no-single --no-secondary --no-live-pmc > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_c3_single -- python $R/tools/bench_configs.py c3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_c3x64 -- python $R/tools/bench_configs.py c3x64 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_c5 -- python $R/tools/bench_configs.py c5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_c3x256 -- python $R/tools/c3_batches.py 256 > /dev/null 2>&1
cd $R
for d in prof_c2_single prof_c3_single prof_c3x64 prof_c3x256 prof_c5; do
  python tools/rocpd_kernel_stats.py $(ls $O/$d/*/*_results.db | head -1) > $O/$d.txt 2>&1
done
rm -rf $O/prof_c2_single $O/prof_c3_single $O/prof_c3x64 $O/prof_c3x256 $O/prof_c5
ls -la $O
