import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/quantum-optimal-control_amd')
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system
def run(seeds, iters=20):
    sp = oracle_system(cases.case_c3())
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=True,
                               reg_coeffs=sp.reg_coeffs, one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=seeds, path=4, chunks=1)
    rng = np.random.default_rng(0)
    eng.set_base(rng.normal(0, 1 / np.sqrt(sp.steps), (seeds, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    eng.iterate(p, 10); eng.sync()
    t0 = time.perf_counter(); eng.iterate(p, iters); eng.sync(); el = time.perf_counter() - t0
    s = eng.scalars(); eng.close()
    return el / iters * 1e3, s['loss'].copy()
for seeds in (64, 256):
    os.environ['QOC_EXPERIMENTAL'] = '1'                 # the library's A/B switches only count beside it
    os.environ['QOC_CHAIN_DPP'] = '1'; t1, l1 = run(seeds)
    os.environ['QOC_CHAIN_DPP'] = '0'; t0, l0 = run(seeds)
    print('C3 x %d: dpp %.3f ms, butterfly %.3f ms per iteration; max |loss difference| after 30 Adam iterations %.2e' % (seeds, t1, t0, np.max(np.abs(l1 - l0))))
