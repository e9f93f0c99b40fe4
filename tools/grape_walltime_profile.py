"""cProfile of the second Grape() call of tools/grape_walltime.py (host-side overhead around the 1000 iterations)."""
import contextlib, io, os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
import numpy as np
from quantum_optimal_control.main_grape.grape import Grape
from tests.golden import cases
from tests.helpers import grape_kwargs

c = cases.case_c2()
conv = {'rate': 0.01, 'update_step': 100, 'max_iterations': 1000, 'conv_target': 1e-12, 'learning_rate_decay': 2500}
def call():
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        Grape(convergence=conv, method='Adam', **grape_kwargs(c))
call()
pr = cProfile.Profile(); pr.enable(); call(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
