import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system
for (n,k,T) in [(64,6,10),(64,6,2),(64,1,10),(64,1,2),(16,6,10),(16,1,2)]:
    c = cases.case_c3(n=n, k=k, steps=1000, taylor=(T,0)); c['reg_coeffs']={}
    sp = oracle_system(c)
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=True, reg_coeffs={}, n_seeds=8)
    eng.set_base(np.zeros((8,k,1000)))
    p = eng.adam_params(max_iterations=10**9, conv_target=-1.0, min_grad=-1.0)
    eng.iterate(p,1); eng.sync(); t0=time.perf_counter(); eng.iterate(p,3); eng.sync(); el=(time.perf_counter()-t0)/3
    print('n=%d k=%d T=%d : %.2f ms/iter = %.2f us per slice (fwd+bwd)' % (n,k,T,el*1e3, el*1e6/1000))
    eng.close()
