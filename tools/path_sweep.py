#!/usr/bin/env python
"""MFMA (register-resident) vs GEMM (tiled launches) crossover for mid-size unitary problems; feeds the AUTO rule of
qoc_create.  Usage: python tools/path_sweep.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
import numpy as np  # noqa: E402

from quantum_optimal_control.core import hip_engine  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.helpers import oracle_system  # noqa: E402


def ms_per_iter(c, n_seeds, path, iters=3):
    sp = oracle_system(c)
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms,
                               sp.scaling, reg_coeffs=sp.reg_coeffs, n_seeds=n_seeds, path=path)
    rng = np.random.default_rng(0)
    eng.set_base(rng.normal(0, 1 / np.sqrt(sp.steps), (n_seeds, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    eng.iterate(p, 1); eng.sync()
    t0 = time.perf_counter()
    eng.iterate(p, iters); eng.sync()
    el = (time.perf_counter() - t0) / iters * 1e3
    eng.close()
    return el


if __name__ == '__main__':
    print('%4s %6s %6s %10s %10s' % ('n', 'steps', 'seeds', 'mfma ms', 'gemm ms'))
    for n in (24, 32, 40, 48, 56, 64):
        for steps in (100, 500):
            for seeds in (1, 4, 16, 64):
                c = cases.case_c2(n=n, k=4, steps=steps, m=8, taylor=(5, 3), seed=2)
                a = ms_per_iter(c, seeds, 2)
                b = ms_per_iter(c, seeds, 4)
                print('%4d %6d %6d %10.3f %10.3f %s' % (n, steps, seeds, a, b, 'GEMM' if b < a else ''), flush=True)

    print('state transfer: fused mat-vec kernels (path 3) vs GEMM path propagator route (chunks > 1) vs direct route (chunks = 1)')
    print('%4s %6s %6s %10s %10s %10s' % ('n', 'steps', 'seeds', 'fused ms', 'propag ms', 'direct ms'))

    def st_ms(c, n_seeds, path, chunks=0, iters=2):
        sp = oracle_system(c)
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms,
                                   sp.scaling, state_transfer=True, reg_coeffs=sp.reg_coeffs, n_seeds=n_seeds, path=path,
                                   chunks=chunks)
        rng = np.random.default_rng(0)
        eng.set_base(rng.normal(0, 1 / np.sqrt(sp.steps), (n_seeds, sp.k, sp.steps)))
        p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
        eng.iterate(p, 1); eng.sync()
        t0 = time.perf_counter()
        eng.iterate(p, iters); eng.sync()
        el = (time.perf_counter() - t0) / iters * 1e3
        eng.close()
        return el

    for n in (16, 32, 64):
        for steps in (200, 1000):
            for seeds in (1, 4, 8, 16, 32, 64, 128, 256):
                c = cases.case_c3(n=n, k=6, steps=steps, taylor=(10, 0))
                a = st_ms(c, seeds, 3)
                b = st_ms(c, seeds, 4, chunks=2)
                d = st_ms(c, seeds, 4, chunks=1)
                best = min((a, 'fused'), (b, 'propagator'), (d, 'direct'))[1]
                print('%4d %6d %6d %10.3f %10.3f %10.3f %s' % (n, steps, seeds, a, b, d, best), flush=True)
