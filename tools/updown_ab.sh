#!/bin/bash
# Runs on the GPU box: bench workload with the fused sweep kernel off / on (and forced chunk lengths), kernel tables.  Output -> gpurun_out/updown_ab.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/updown_tab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # label, bench chunks (0 = planned), env...
  local label=$1; shift
  local ch=$1; shift
  echo "== $label"
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/$label -- python $R/bench.py --steps 40 --warmup 10 --chunks $ch --no-cpu-baseline --no-single > $O/$label.json 2>/dev/null
  python -c "import json; j=json.load(open('$O/$label.json')); print('ms_per_step', j['ms_per_step'], 'chunks', j['config']['chunks'], 'best_fidelity', j['best_fidelity'])"
  python $R/tools/rocpd_kernel_stats.py $(ls $O/$label/*/*_results.db | head -1) 2>&1 | head -9
  rm -rf $O/$label
}
run off 0 QOC_EXPERIMENTAL=1 QOC_UPDOWN=0
run on 0 QOC_UPDOWN=1
for c in "$@"; do run on_c$c $c QOC_UPDOWN=1; done
