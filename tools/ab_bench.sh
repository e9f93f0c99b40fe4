#!/bin/bash
# A/B of alternative builds of libqoc_hip.so on the GPU box: ab_bench.sh lib1.so lib2.so ... (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  QOC_HIP_LIBRARY=$R/$lib python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single ${AB_ARGS:-} | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print('%-44s %9.0f it/s  %.4f ms/step  %s %.4f ms/launch  frac %.3f' % ('$lib', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac']))"
done
