#!/usr/bin/env python
"""Where an iteration of k_small_iter spends its cycles: a library built with -DQOC_SMALL_TIMING (python tools/build_variant.py timing qoc_small
-DQOC_SMALL_TIMING) makes workgroup 0 print the shader-clock stamps of its phase boundaries.  Run as
    QOC_HIP_LIBRARY=quantum-optimal-control_amd/lib_timing/libqoc_hip.so python tools/small_phase_timing.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system


def run(name, c, seeds=1, groups=0, rows=0):
    sp = oracle_system(c)
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=sp.state_transfer,
                               reg_coeffs=sp.reg_coeffs, one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=seeds, path=5, chunks=groups, variant=rows)
    rng = np.random.default_rng(0)
    eng.set_base(rng.normal(0, 1 / np.sqrt(sp.steps), (seeds, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    eng.iterate(p, 2000); eng.sync()                      # clocks up
    print('== %s: n=%d k=%d steps=%d T=%d s=%d seeds=%d plan=%s' % (name, sp.n, sp.k, sp.steps, sp.exp_terms, sp.scaling, seeds, eng.plan), flush=True)
    eng.iterate(p, 200); eng.sync()
    eng.close()


def main():
    run('C1', cases.case_c1())
    run('C1 rows=16', cases.case_c1(), rows=16)
    run('C1 x64', cases.case_c1(), 64)
    for n in (4, 8):
        run('n=%d x 500' % n, cases.case_c2(n=n, k=4, steps=500, m=min(8, n), taylor=(5, 3), seed=2))
    run('n=8 x 500, 16 workgroups', cases.case_c2(n=8, k=4, steps=500, m=8, taylor=(5, 3), seed=2), groups=16)
    c = cases.case_c2(n=9, k=4, steps=300, m=4, taylor=(5, 3), seed=2)
    c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [8, 5]}
    run('two qutrits + forbidden', c)


if __name__ == '__main__':
    main()
