#!/usr/bin/env python
"""Unitary gates of 17 <= n <= 32 levels x 64 control sets WITH forbidden levels + dwdt (k = 4, 500 slices, m = 8, (T, s) = (5, 3)):
the regularised NT = 2 batch route (k_mfma_forward2 / k_mfma_bwd_offsets2 / k_mfma_backward3) on padded sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
for n in (17, 20, 24, 27, 28, 32):
    c = cases.case_c2(n=n, k=4, steps=500, m=8, taylor=(5, 3), seed=2)
    c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [n - 1, n - 2]}
    bench_configs.run('n=%d x64 + dwdt + forbidden (active strips %d of 8)' % (n, (n + 3) // 4), c, 64, 20)
