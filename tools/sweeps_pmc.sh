#!/bin/bash
# Runs on the GPU box: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the sweep kernels of the bench workload, the two
# separate kernels (QOC_UPDOWN=0) against k_mfma_downup.  Output -> gpurun_out/sweeps_pmc.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sweeps_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  echo "== QOC_UPDOWN=$v"
  for c in FETCH_SIZE WRITE_SIZE; do
    QOC_EXPERIMENTAL=1 QOC_UPDOWN=$v rocprofv3 --pmc $c --kernel-trace -d $O/$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single --no-secondary --no-live-pmc > /dev/null 2>&1
    python $R/tools/rocpd_pmc_stats.py $(ls $O/$c/*/*_results.db | head -1) 2>&1 | grep "forward2\|backward3\|downup\|bnd_scan\|expm_inplace\|^kernel"
    rm -rf $O/$c
  done
done
