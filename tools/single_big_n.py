#!/usr/bin/env python
"""One trajectory of a large Hilbert space (n from argv, k = 6, 500 slices, (T, s) = (5, 3)) on the GEMM path: ms per iteration."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
import numpy as np  # noqa: E402
from quantum_optimal_control.core import hip_engine  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.helpers import oracle_system  # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    sp = oracle_system(cases.case_c2(n=n, k=6, steps=500, m=8, taylor=(5, 3), seed=2))
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs={}, n_seeds=1)
    eng.set_base(sp.base0[None])
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    eng.iterate(p, 30); eng.sync()
    t0 = time.perf_counter()
    eng.iterate(p, 50); eng.sync()
    print('n = %d, one trajectory: %.3f ms per iteration (path %d, chunks %d)' % (n, (time.perf_counter() - t0) / 50 * 1e3, eng.path, eng.chunks))
    eng.close()
