#!/bin/bash
# Runs on the GPU box: kernel table of the bench workload for several chunk counts (what the sweeps gain from more, shorter chunks).
# Usage: chunks_kernel_table.sh <chunks>...   Output -> gpurun_out/chunks_kernel_table.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/chunks_tab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  echo "== chunks $c"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/c$c -- python $R/bench.py --steps 30 --warmup 5 --chunks $c --no-cpu-baseline --no-single > $O/c$c.json 2>/dev/null
  python -c "import json; j=json.load(open('$O/c$c.json')); print('ms_per_step', j['ms_per_step'], 'chunks', j['config']['chunks'])"
  python $R/tools/rocpd_kernel_stats.py $(ls $O/c$c/*/*_results.db | head -1) 2>&1 | head -12
  rm -rf $O/c$c
done
