#!/bin/bash
# Runs on the GPU box (gpurun): the round-6 measurement files of the workgroup-resident path and of a Grape() caller's wall time.  Output -> gpurun_out/collect/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/collect
mkdir -p $O
cd $R
python tools/small_n_latency.py > $O/small_n_latency.txt 2>&1
python tools/grape_walltime.py 1000 > $O/grape_walltime.txt 2>&1
if [ -x /opt/conda/bin/python3.9 ]; then      # save=True needs h5py: the image's conda interpreter has it (tests/test_h5_log.py runs its bodies there too)
  echo "# the same under /opt/conda/bin/python3.9 (h5py): save=True with update_step 100" >> $O/grape_walltime.txt
  LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 /opt/conda/bin/python3.9 -W ignore tools/grape_walltime.py 1000 >> $O/grape_walltime.txt 2>&1
fi
# phase stamps: libraries built with -DQOC_SMALL_TIMING, one per translation unit of instances (python tools/build_variant.py timing_<u> qoc_small_<u> -DQOC_SMALL_TIMING)
if [ -f quantum-optimal-control_amd/lib_timing_a1/libqoc_hip.so ]; then
  { for v in a1 a b; do QOC_HIP_LIBRARY=quantum-optimal-control_amd/lib_timing_$v/libqoc_hip.so python tools/small_phase_timing.py 2>&1 | grep "iters=200:\|state-regulariser\|exchange A:"; done
    QOC_HIP_LIBRARY=quantum-optimal-control_amd/lib_timing_c/libqoc_hip.so python tools/small_phase_timing_src.py 2>&1 | grep -v "iters=1000"; } > $O/small_phase_timing.txt
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c1 -- python $R/tools/bench_configs.py c1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_n8 -- python $R/tools/bench_configs.py n8small > /dev/null 2>&1
cd $R
for d in prof_c1 prof_n8; do
  python tools/rocpd_kernel_stats.py $(ls $O/$d/*/*_results.db | head -1) > $O/$d.txt 2>&1
  rm -rf $O/$d
done
ls -la $O
