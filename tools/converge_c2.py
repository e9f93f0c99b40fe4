#!/usr/bin/env python
"""End-to-end use of the engine on the C2 workload: 64 random restarts optimised to convergence on one GPU with the
device-resident Adam loop (qoc_run_adam), reference hyper-parameters.  Prints fidelity statistics over the restarts and
the wall-clock time -- evidence that the measured iterations are the real optimisation, not a stripped-down loop."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
import numpy as np  # noqa: E402

from quantum_optimal_control.core import hip_engine  # noqa: E402
from quantum_optimal_control.parallel_seeds import restart_guesses  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.helpers import oracle_system  # noqa: E402

if __name__ == '__main__':
    max_it = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    sp = oracle_system(cases.case_c2())
    B = 64
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
                               reg_coeffs={}, n_seeds=B)
    eng.set_base(restart_guesses(sp.k, sp.steps, 0, B))
    p = eng.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-4, min_grad=1e-25, max_iterations=max_it,
                        poll_every=100)
    print('C2: n=%d k=%d steps=%d m=%d (T,s)=(%d,%d), %d restarts, path=%d, conv_target=1e-4, max_iterations=%d'
          % (sp.n, sp.k, sp.steps, sp.m, sp.exp_terms, sp.scaling, B, eng.path, max_it))
    t0 = time.perf_counter()
    done_it = 0
    while True:
        eng.iterate(p, 250)
        s = eng.scalars()
        done_it = int(np.max(s['iterations']))
        el = time.perf_counter() - t0
        loss = np.sort(s['loss'])
        print('t=%6.2fs  max iterations %5d  converged %2d/%d  loss best %.2e  median %.2e  worst %.2e'
              % (el, done_it, int(np.sum(s['done'])), B, loss[0], loss[B // 2], loss[-1]), flush=True)
        if np.all(s['done']) or done_it >= max_it:
            break
    total_it = int(np.sum(s['iterations']))
    print('total %d seed-iterations in %.2f s = %.0f GRAPE iterations/s (whole optimisation, polling included)'
          % (total_it, el, total_it / el))
    eng.close()
