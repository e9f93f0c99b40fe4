#!/bin/bash
# Runs on the GPU box: does a SHORTER half chunk bring pass B of k_mfma_downup (its re-read of the half chunk's propagators) back into L2?
# Bench workload (C2 x 64 control sets) at 16 / 32 / 64 chunks per control set: half chunks of 16 / 8 / 4 slices = 256 / 128 / 64 KB per wave between the passes.
# Per chunk count: kernel times (rocprofv3 --kernel-trace --stats) and FETCH_SIZE / WRITE_SIZE (separate PMC passes) of k_mfma_downup and k_mfma_expm_inplace.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sweeps_reuse
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 16 32 64; do
  echo "== chunks $c"
  rocprofv3 --kernel-trace --stats -d $O/t -- python $R/bench.py --chunks $c --steps 10 --warmup 2 --no-cpu-baseline --no-single --no-secondary --no-live-pmc 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_per_step', d['ms_per_step'], 'chunks', d['config']['chunks'])"
  python $R/tools/rocpd_kernel_stats.py $(ls $O/t/*/*_results.db | head -1) 2>&1 | grep "downup\|expm_inplace\|bnd_scan\|^kernel"
  rm -rf $O/t
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d $O/$ctr -- python $R/bench.py --chunks $c --steps 4 --warmup 1 --no-cpu-baseline --no-single --no-secondary --no-live-pmc > /dev/null 2>&1
    python $R/tools/rocpd_pmc_stats.py $(ls $O/$ctr/*/*_results.db | head -1) 2>&1 | grep "downup\|expm_inplace"
    rm -rf $O/$ctr
  done
done
