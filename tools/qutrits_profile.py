#!/usr/bin/env python
"""Profiling hook: three qutrits (n = 27, k = 6 or 4) with dwdt + forbidden levels x 64 control sets -- what the reference's transmon examples look like."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
c = cases.case_c2(n=27, k=k, steps=500, m=8, taylor=(5, 3), seed=2)
c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [26, 25]}
bench_configs.run('three qutrits n=27 k=%d + dwdt + forbidden x64' % k, c, 64, 20)
