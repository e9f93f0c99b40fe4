#!/usr/bin/env python
"""ONE control set (the reference's own use; AUTO = latency mode of the MFMA path) over n = 2 .. 32, k = 4, 500 slices, (T, s) = (5, 3): the latency-mode
kernels work on the padded 32 x 32 problem and skip its all-zero strips (k_mfma_expm_slice2 templated on the active strips, round 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
bench_configs.run('C1 single qubit', cases.case_c1(), 1, 50)
for n in (4, 8, 12, 16, 17, 20, 24, 28, 32):
    bench_configs.run('n=%d one control set' % n, cases.case_c2(n=n, k=4, steps=500, m=min(8, n), taylor=(5, 3), seed=2), 1, 50)
c = cases.case_c2(n=20, k=4, steps=500, m=8, taylor=(5, 3), seed=2)
c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [18, 19]}
bench_configs.run('n=20 one control set + dwdt + forbidden', c, 1, 50)
