#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/exp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  QOC_EXP=$e rocprofv3 --kernel-trace --stats -d $O/p$e -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
  echo "== QOC_EXP=$e"
  python $R/tools/rocpd_kernel_stats.py $(ls $O/p$e/*/*_results.db | head -1) 2>&1 | grep "backward\|forward"
done
