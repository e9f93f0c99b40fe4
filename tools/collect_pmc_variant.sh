#!/bin/bash
# Runs on the GPU box: one PMC pass per counter (kernel-trace only) for a bench variant.  Usage: collect_pmc_variant.sh <variant> <counter>...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
O=$R/gpurun_out/pmc_v$V
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single --variant $V > /dev/null 2>&1
  python $R/tools/rocpd_pmc_stats.py $(ls $O/$c/*/*_results.db | head -1) 2>&1 | grep -i "expm" | head -2
  rm -rf $O/$c
done
