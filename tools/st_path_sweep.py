#!/usr/bin/env python
"""State transfer (m = 1, T = 10 Taylor terms, dwdt + two forbidden levels as in C3): the GEMM path (what AUTO took up to round 3) against the MFMA path
(round 4: K_t of degree T - 1 + thin sweeps; latency mode / batch kernels), ms per iteration of the batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system
import time


def ms(sp, B, path, variant):
    try:
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=True,
                                   reg_coeffs=sp.reg_coeffs, n_seeds=B, path=path, variant=variant)
    except hip_engine.QocError:
        return float('nan'), -1
    eng.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (B, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    t0 = time.perf_counter(); eng.iterate(p, 2); eng.sync()
    per = max((time.perf_counter() - t0) / 2, 1e-5)
    eng.iterate(p, max(1, min(2000, int(0.3 / per)))); eng.sync()
    it = max(5, min(3000, int(0.4 / per)))
    t0 = time.perf_counter(); eng.iterate(p, it); eng.sync()
    el = (time.perf_counter() - t0) / it * 1e3
    path_used = eng.path
    eng.close()
    return el, path_used


shapes = [(8, 2, 300), (16, 4, 500), (20, 4, 500), (27, 4, 500), (32, 4, 500), (48, 4, 500), (64, 6, 1000)]
regs = [False, True] if len(sys.argv) < 2 else [sys.argv[1] == 'reg']
for reg in regs:
    print('# %s' % ('dwdt + two forbidden levels' if reg else 'dwdt only'))
    print('   n  k steps seeds :     AUTO   GEMM(4)  MFMA batch  MFMA latency')
    for n, k, steps in shapes:
        c = cases.case_c3(n=n, k=k, steps=steps, taylor=(10, 0), seed=3)
        if not reg:
            c['reg_coeffs'] = {'dwdt': 1e-3}
        sp = oracle_system(c)
        for B in (1, 2, 4, 8, 16, 64, 256):
            if n >= 48 and B > 64:
                continue
            a, pa = ms(sp, B, 0, 0)
            g, _ = ms(sp, B, 4, 0)
            mb, _ = ms(sp, B, 2, 0)
            ml, _ = ms(sp, B, 2, 5) if B <= 16 else (float('nan'), -1)
            print('%4d %2d %5d %5d : %8.4f (path %d) %8.4f %8.4f %8.4f' % (n, k, steps, B, a, pa, g, mb, ml), flush=True)
