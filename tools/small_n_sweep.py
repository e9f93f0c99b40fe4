#!/usr/bin/env python
"""Small Hilbert spaces (n <= 16): batch kernels (NT = 1) against the latency mode padded to 32, for a few seeds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system

def run(c, seeds, path, variant, iters=100):
    sp = oracle_system(c)
    e = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs=sp.reg_coeffs,
                             n_seeds=seeds, path=path, variant=variant)
    e.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (seeds, sp.k, sp.steps)))
    p = e.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    e.iterate(p, 10); e.sync()
    t0 = time.perf_counter(); e.iterate(p, iters); e.sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    ch = e.chunks
    e.close()
    return ms, ch

if __name__ == "__main__":
  for n, k, steps, m in ((2, 1, 100, 2), (4, 2, 200, 4), (9, 2, 300, 4), (9, 2, 1000, 4), (16, 4, 500, 8), (16, 4, 2000, 8), (27, 3, 4000, 8)):
      c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(5, 3), seed=3)
      for seeds in (1, 2, 4, 8, 16):
          row = [run(c, seeds, 2, 0)[0] if n <= 16 else float('nan'), run(c, seeds, 4, 0)[0]]
          lat, ch = run(c, seeds, 2, 5)
          print('n=%-2d k=%d steps=%-4d seeds=%-2d : batch NT=1 %.4f ms   GEMM %.4f ms   latency mode %.4f ms (chunks %d)' % (n, k, steps, seeds, row[0], row[1], lat, ch), flush=True)
