#!/usr/bin/env python
"""C3 x 64 without its state regularisers (dwdt only): the direct route runs the backward chain beside the forward one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
c = cases.case_c3()
c['reg_coeffs'] = {'dwdt': 1e-3}
bench_configs.run('C3 x64, dwdt only', c, 64, 5)
bench_configs.run('C3 x128, dwdt only', c, 128, 5)
bench_configs.run('C3 x256, dwdt only', c, 256, 5)
