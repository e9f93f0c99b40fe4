#!/usr/bin/env python
"""C3 route crossover at a finer grid of control sets (after the DPP chain took the three-multiplication form): see tools/c3_route_sweep.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
for reg, grid in ((True, (18, 20, 22, 24, 26, 28)), (False, (10, 11, 12, 13, 14, 16))):
    c = cases.case_c3()
    if not reg:
        c['reg_coeffs'] = {'dwdt': 1e-3}
    for seeds in grid:
        for chunks, name in ((2, 'propagator'), (1, 'direct')):
            bench_configs.run('C3%s x%d %s' % ('' if reg else ' (no forbidden levels)', seeds, name), c, seeds, 5, path=4, chunks=chunks)
