"""Wall time of a plain Grape() call (one trajectory of the C2 workload, 1000 Adam iterations, progress line every 100)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
import numpy as np
from quantum_optimal_control.main_grape.grape import Grape
from tests.golden import cases
from tests.helpers import grape_kwargs

c = cases.case_c2()
conv = {'rate': 0.01, 'update_step': 100, 'max_iterations': 1000, 'conv_target': 1e-12, 'learning_rate_decay': 2500}
for rep in range(2):
    np.random.seed(c['np_seed'])
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        uks, Uf = Grape(convergence=conv, method='Adam', **grape_kwargs(c))
    el = time.perf_counter() - t0
    last = [l for l in buf.getvalue().splitlines() if l.startswith('Error')][-1]
    print('call %d: %.3f s wall for 1000 iterations (%.0f it/s incl. setup, polling, read-back); %s' % (rep, el, 1000 / el, last))
