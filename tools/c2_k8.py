#!/usr/bin/env python
"""C2-size gate with 8 controls x 64 seeds (NT = 2, costate sweep + slice-parallel gradient kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
for k in (6, 8):
    bench_configs.run('C2 size, k=%d x64' % k, cases.case_c2(n=32, k=k, steps=500, m=8, taylor=(5, 3), seed=2), 64, 10)
