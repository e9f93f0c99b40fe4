#!/bin/bash
# Runs on the GPU box: shader clock under the C5 products (GRBM_GUI_ACTIVE / kernel duration) and matrix-pipe busy share of k_zgemm_wg
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c5_clock
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --stats -d $O/$c -- python $R/tools/bench_configs.py c5 > /dev/null 2>&1
  python $R/tools/rocpd_pmc_stats.py $(ls $O/$c/*/*_results.db | head -1) 2>&1 | grep -i "zgemm_wg" | head -2
  python $R/tools/rocpd_kernel_stats.py $(ls $O/$c/*/*_results.db | head -1) 2>&1 | grep -i "zgemm_wg" | head -1
  rm -rf $O/$c
done
