#!/usr/bin/env python
"""State transfer of 32 < n <= 48 levels with k <= 4 controls (m = 1, T = 10, 500 slices): MFMA batch kernels (NT = 3) against the GEMM path's direct route
(k_gemm_taylor_chain_dpp) over the number of control sets -- AUTO's st_big thresholds for n > 32 (csrc/qoc_engine.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
from tests.golden import cases
from tests.helpers import oracle_system
from st_path_sweep_lib import ms
for reg in (False, True):
    print('# %s' % ('dwdt + two forbidden levels' if reg else 'dwdt only'))
    print('   n  k steps seeds :  MFMA batch  GEMM direct  GEMM propagator')
    for n, k, steps in ((40, 4, 500), (48, 4, 500)):
        c = cases.case_c3(n=n, k=k, steps=steps, taylor=(10, 0), seed=3)
        if not reg:
            c['reg_coeffs'] = {'dwdt': 1e-3}
        sp = oracle_system(c)
        for B in (16, 20, 24, 28, 32, 40, 48, 64, 128):
            mb, _ = ms(sp, B, 2, 0)
            gd, _ = ms(sp, B, 4, 0, chunks=1)
            gp, _ = ms(sp, B, 4, 0, chunks=2) if B <= 32 else (float('nan'), 0)
            print('%4d %2d %5d %5d : %10.4f %10.4f %10.4f' % (n, k, steps, B, mb, gd, gp), flush=True)
