#!/bin/bash
# Runs on the GPU box: C3 x 64 against the number of assembly windows beside the forward chain (QOC_ASM_WINDOWS), the size of the window in front
# (QOC_ASM_SPLIT16 / 16) and the number of workgroups of the assembly launches beside the chain (QOC_ASM_TAIL_WGS)
R=${GRAFT_REPO_ROOT:-/root/repo}
export QOC_EXPERIMENTAL=1     # the library's A/B switches only count beside it
for sp in ${SPLITS:-1 2 4}; do for nw in ${WINDOWS:-2 3 4 6 9}; do for wg in ${WGS:-8192}; do
  echo -n "split16=$sp windows=$nw tail_wgs=$wg : "; QOC_ASM_SPLIT16=$sp QOC_ASM_WINDOWS=$nw QOC_ASM_TAIL_WGS=$wg timeout 40 python $R/tools/c3_batches.py ${1:-64} 2>&1 | grep -v Taylor | sed 's/.*aggregate, *//'
done; done; done
