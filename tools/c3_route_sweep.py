#!/usr/bin/env python
"""C3 (state transfer, n = 64, k = 6, 1000 slices, forbidden levels + dwdt): propagator route against direct route (DPP Taylor chains) of the GEMM path over the number
of control sets -- where AUTO's ST_DIRECT_FROM (csrc/qoc_engine.hip) for n > 32 comes from.  Also one case without a state regulariser."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
for reg in (True, False):
    c = cases.case_c3()
    if not reg:
        c['reg_coeffs'] = {'dwdt': 1e-3}
    for seeds in (8, 12, 16, 24, 32, 48):
        for chunks, name in ((2, 'propagator'), (1, 'direct')):
            bench_configs.run('C3%s x%d %s' % ('' if reg else ' (no forbidden levels)', seeds, name), c, seeds, 5, path=4, chunks=chunks)
