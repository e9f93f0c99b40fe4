#!/bin/bash
# Runs on the GPU box: kernel table and timeline (last iteration) of C3 x SEEDS on the AUTO route.  Usage: tools/c3_timeline.sh [seeds=64] [n_last=24]
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-64}; NL=${2:-24}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_c3 -- python $R/tools/c3_batches.py $S 2>/dev/null | grep -v Taylor
DB=$(ls $R/gpurun_out/p_c3/*/*_results.db | head -1)
python $R/tools/rocpd_kernel_stats.py $DB 2>&1 | head -14
python $R/tools/kernel_timeline.py $DB $NL
rm -rf $R/gpurun_out/p_c3
