#!/usr/bin/env python
"""Unitary gates of 17 <= n <= 32 levels x 64 control sets (k = 4, 500 slices, m = 8, (T, s) = (5, 3)): the NT = 2 batch kernels work on the
ACTIVE 4-row strips ceil(n / 4) of the padded 32 x 32 matrices (k_mfma_expm_inplace / k_mfma_downup templated on QA, round 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
for n in (17, 20, 21, 24, 25, 27, 28, 29, 32):
    bench_configs.run('n=%d x64 (active strips %d of 8)' % (n, (n + 3) // 4), cases.case_c2(n=n, k=4, steps=500, m=8, taylor=(5, 3), seed=2), 64, 20)
c = cases.case_c2(n=27, k=6, steps=500, m=8, taylor=(5, 3), seed=2)
c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [26, 25]}
bench_configs.run('three qutrits n=27 k=6 + dwdt + forbidden x64', c, 64, 20)
bench_configs.run('three qutrits n=27 k=6 + dwdt + forbidden x1', c, 1, 20)
