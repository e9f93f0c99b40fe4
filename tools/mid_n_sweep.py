#!/usr/bin/env python
"""32 < n <= 64, a few seeds: AUTO against the GEMM path, the MFMA batch kernels and (n <= 48) the latency mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system

def run(c, seeds, path, variant, iters=50):
    sp = oracle_system(c)
    try:
        e = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs={},
                                 n_seeds=seeds, path=path, variant=variant)
    except hip_engine.QocError:
        return float('nan')
    e.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (seeds, sp.k, sp.steps)))
    p = e.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    e.iterate(p, 5); e.sync()
    t0 = time.perf_counter(); e.iterate(p, iters); e.sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    e.close()
    return ms

if __name__ == '__main__':
  for n, k, steps in ((40, 4, 500), (48, 4, 500), (48, 3, 2000), (64, 4, 500), (64, 6, 1000)):
      c = cases.case_c2(n=n, k=k, steps=steps, m=8, taylor=(5, 3), seed=2)
      for seeds in (1, 2, 4, 8, 16):
          print('n=%-2d k=%d steps=%-4d seeds=%-2d : AUTO %.4f ms   GEMM %.4f ms   MFMA batch %.4f ms   latency mode %.4f ms'
                % (n, k, steps, seeds, run(c, seeds, 0, 0), run(c, seeds, 4, 0), run(c, seeds, 2, 7), run(c, seeds, 2, 5)), flush=True)
