#!/usr/bin/env python
"""Phase stamps (see tools/small_phase_timing.py) of the state-regulariser flow: two qutrits with forbidden levels, with 1 / 2 slices per row and several workgroup counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system


def run(name, c, groups=0):
    sp = oracle_system(c)
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=sp.state_transfer,
                               reg_coeffs=sp.reg_coeffs, one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=1, path=5, chunks=groups)
    eng.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (1, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    eng.iterate(p, 1000); eng.sync()
    print('== %s: plan=%s' % (name, eng.plan), flush=True)
    eng.iterate(p, 200); eng.sync()
    eng.close()


c = cases.case_c2(n=9, k=4, steps=300, m=4, taylor=(5, 3), seed=2)
c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [8, 5]}
for g in (0,):
    run('two qutrits + forbidden, groups=%d' % g, c, g)
c2 = dict(c); c2['reg_coeffs'] = {'dwdt': 1e-3}
run('two qutrits, no state regulariser', c2)
