#!/usr/bin/env python
"""Padded sizes (n = 20, 24, 27), 500 slices: AUTO (latency mode up to 16 control sets / 4608 seed-slices) against the batch kernels, which work on the active
strips since round 4 -- where is the crossover now?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
for n in (20, 24, 27, 32):
    for B in (2, 4, 6, 8, 9, 12):
        c = cases.case_c2(n=n, k=4, steps=500, m=8, taylor=(5, 3), seed=2)
        bench_configs.run('n=%d x%d AUTO' % (n, B), c, B, 20)
        bench_configs.run('n=%d x%d batch kernels' % (n, B), c, B, 20, path=2, variant=8)
