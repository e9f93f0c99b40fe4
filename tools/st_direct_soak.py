#!/usr/bin/env python
"""Soak of the direct state-transfer route: thousands of asynchronously queued iterations (two streams with CU masks, events between the assembly windows and the
chain launches) on the packed, the active-column and the > 128-control-set forms -- prints ms per iteration, the plan, and whether every loss is finite."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system
for n, seeds, iters in ((64, 64, 3000), (40, 32, 3000), (64, 200, 600)):
    sp = oracle_system(cases.case_c3(n=n))
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=True,
                               reg_coeffs=sp.reg_coeffs, Vs=sp.Vs, n_seeds=seeds)
    eng.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (seeds, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    t0 = time.perf_counter()
    eng.iterate(p, iters); eng.sync()
    el = time.perf_counter() - t0
    s = eng.scalars()
    print('n=%d x%d: %d iterations in %.1f s (%.3f ms each), plan %s, loss[0] %.6f, all finite %s' % (n, seeds, iters, el, el / iters * 1e3, eng.plan, s['loss'][0], bool(np.all(np.isfinite(s['loss'])))), flush=True)
    eng.close()
