#!/usr/bin/env python
"""One C2 trajectory WITH state regularisers (forbidden levels, dwdt): the routes AUTO can take."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import time
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system

def run(c, seeds, path, variant, iters=100):
    sp = oracle_system(c)
    e = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs=sp.reg_coeffs,
                             one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=seeds, path=path, variant=variant)
    e.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (seeds, sp.k, sp.steps)))
    p = e.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    e.iterate(p, 10); e.sync()
    t0 = time.perf_counter(); e.iterate(p, iters); e.sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    pth = e.path
    e.close()
    return ms, pth

if __name__ == '__main__':
    for n, k, steps in ((32, 4, 500), (9, 2, 300), (27, 3, 1000), (48, 4, 500), (64, 4, 500)):
        c = cases.case_c2(n=n, k=k, steps=steps, m=min(n, 8), taylor=(5, 3), seed=3)
        c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [n - 2, n - 1]}
        for seeds in (1, 2, 4, 8):
            print('n=%-2d k=%d steps=%-4d seeds=%d + dwdt + forbidden: AUTO %.4f ms (path %d)   GEMM route %.4f ms   latency mode with sources %.4f ms   batch kernels %.4f ms'
                  % ((n, k, steps, seeds) + run(c, seeds, 0, 0) + (run(c, seeds, 4, 0)[0], run(c, seeds, 2, 5)[0], run(c, seeds, 2, 7 if n > 32 else (4 if n > 16 else 0))[0])), flush=True)
