#!/bin/bash
# A/B of alternative libqoc builds on one secondary configuration: ab_config.sh <bench_configs hook> lib1.so lib2.so ...  ("product" = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
hook=$1; shift
for lib in "$@"; do
  if [ "$lib" = product ]; then unset QOC_HIP_LIBRARY; else export QOC_HIP_LIBRARY=$R/$lib; fi
  printf '%-52s ' "$lib"; python $R/tools/bench_configs.py $hook 2>&1 | tail -1 | sed 's/.*seeds=/seeds=/'
done
