#!/bin/bash
# Runs on the GPU box: MFMA-utilisation PMC passes for the bench kernels (one counter per pass, kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_mfma
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $O/available_mfma_counters.txt
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single --no-secondary --no-live-pmc > /dev/null 2>&1
  python $R/tools/rocpd_pmc_stats.py $(ls $O/$c/*/*_results.db | head -1) > $O/$c.txt 2>&1
  rm -rf $O/$c
done
head -3 $O/*.txt
