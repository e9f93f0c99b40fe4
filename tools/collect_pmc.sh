#!/bin/bash
# Runs on the GPU box: HBM-traffic PMC passes for the bench kernels (separate passes, as MI355X_MICROARCH.md prescribes).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single --no-secondary --no-live-pmc > /dev/null 2>&1
  python $R/tools/rocpd_pmc_stats.py $(ls $O/$c/*/*_results.db | head -1) > $O/$c.txt 2>&1
  rm -rf $O/$c
done
cat $O/FETCH_SIZE.txt $O/WRITE_SIZE.txt
python $R/tools/make_pmc_traffic_json.py $O
