#!/usr/bin/env python
"""One C2 trajectory (the reference's own use: a single control set per Grape() call): iterations/s of the AUTO path, the GEMM
latency route and the MFMA latency mode (variant 5)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
import numpy as np  # noqa: E402

import bench  # noqa: E402
from quantum_optimal_control.core import hip_engine  # noqa: E402

if __name__ == '__main__':
    c, Hs, U0, V, W, dt = bench.build_problem()
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for name, path, variant in (('AUTO', 0, 0), ('GEMM latency route', 4, 0), ('MFMA latency mode (variant 5)', 2, 5), ('MFMA batch kernels', 2, 0)):
        e = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], bench.SLICES, bench.TAYLOR[0], bench.TAYLOR[1], reg_coeffs={},
                                 n_seeds=seeds, path=path, variant=variant)
        e.set_base(bench.seed_bases(0, seeds))
        p = e.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25, max_iterations=10 ** 9, poll_every=10 ** 9)
        e.iterate(p, 20); e.sync()
        t0 = time.perf_counter()
        e.iterate(p, 300); e.sync()
        el = (time.perf_counter() - t0) / 300
        s = e.scalars()
        print('%-32s seeds=%d path=%d chunks=%-3d: %8.1f it/s per trajectory, %.4f ms per iteration, loss[0]=%.9f uscale=%.6f'
              % (name, seeds, e.path, e.chunks, 1.0 / el, el * 1e3, s['loss'][0], s['unitary_scale'][0]), flush=True)
        e.close()
