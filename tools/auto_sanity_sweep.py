#!/usr/bin/env python
"""AUTO over a grid of problem shapes: ms per iteration and ns per (seed x slice), to spot shapes that fall onto a slow route."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system

def run(c, seeds, iters):
    sp = oracle_system(c)
    e = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=sp.state_transfer,
                             reg_coeffs=sp.reg_coeffs, one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=seeds)
    e.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (seeds, sp.k, sp.steps)))
    p = e.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    e.iterate(p, 3); e.sync()
    tw = time.perf_counter()                                  # warm-up: an idle GPU needs tens of milliseconds to reach its working clocks
    while time.perf_counter() - tw < 0.3:
        e.iterate(p, 2); e.sync()
    t0 = time.perf_counter(); e.iterate(p, iters); e.sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    out = (ms, e.path, e.chunks)
    e.close()
    return out

if __name__ == '__main__':
    print('%-14s %5s %3s %6s %3s %6s : %10s %5s %7s %12s' % ('mode', 'n', 'k', 'steps', 'm', 'seeds', 'ms/iter', 'path', 'chunks', 'ns/seed/slice'))
    for n, k, steps, m in ((4, 2, 200, 2), (9, 2, 300, 4), (16, 3, 500, 4), (20, 3, 400, 8), (32, 4, 500, 8), (32, 8, 500, 16), (40, 4, 500, 8), (64, 4, 500, 8),
                           (64, 4, 500, 32), (96, 4, 300, 8), (128, 6, 500, 8), (200, 4, 200, 8)):
        for seeds in (1, 8, 64):
            if n >= 96 and seeds == 64:
                continue
            c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(5, 3), seed=2)
            ms, path, ch = run(c, seeds, 10 if n >= 96 else 30)
            print('%-14s %5d %3d %6d %3d %6d : %10.4f %5d %7d %12.1f' % ('unitary', n, k, steps, m, seeds, ms, path, ch, ms * 1e6 / (seeds * steps)), flush=True)
    for n, k, steps, m in ((8, 2, 300, 1), (24, 3, 500, 2), (32, 4, 500, 1), (32, 4, 500, 4), (64, 6, 1000, 1), (64, 6, 1000, 4), (100, 4, 400, 1)):
        for seeds in (1, 8, 64, 256):
            if n >= 100 and seeds >= 64:
                continue
            c = cases.case_c3(n=n, k=k, steps=steps, taylor=(10, 0))
            if m != 1:
                continue
            ms, path, ch = run(c, seeds, 10 if n >= 100 else 20)
            print('%-14s %5d %3d %6d %3d %6d : %10.4f %5d %7d %12.1f' % ('state transfer', n, k, steps, m, seeds, ms, path, ch, ms * 1e6 / (seeds * steps)), flush=True)
