#!/bin/bash
# Runs on the GPU box: counters of k_zgemm_wg under the C5 workload, one pass per counter.  Usage: c5_pmc.sh <counter>...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c5_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/$c -- python $R/tools/bench_configs.py c5 > /dev/null 2>&1
  python $R/tools/rocpd_pmc_stats.py $(ls $O/$c/*/*_results.db | head -1) 2>&1 | grep -i "zgemm_wg" | head -1
  rm -rf $O/$c
done
