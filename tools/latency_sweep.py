#!/usr/bin/env python
"""C2 with few seeds: ms per iteration of the three candidates AUTO chooses between (latency mode / GEMM latency route / batch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
import bench  # noqa: E402
from quantum_optimal_control.core import hip_engine  # noqa: E402

if __name__ == '__main__':
    c, Hs, U0, V, W, dt = bench.build_problem()
    print('%6s %14s %14s %14s %14s' % ('seeds', 'AUTO', 'MFMA latency', 'GEMM route', 'MFMA batch'))
    for seeds in (1, 2, 4, 8, 12, 16, 20, 24, 32, 48, 64):
        row = []
        for path, variant in ((0, 0), (2, 5), (4, 0), (2, 0)):
            e = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], bench.SLICES, bench.TAYLOR[0], bench.TAYLOR[1], reg_coeffs={},
                                     n_seeds=seeds, path=path, variant=variant)
            e.set_base(bench.seed_bases(0, seeds))
            p = e.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25, max_iterations=10 ** 9, poll_every=10 ** 9)
            e.iterate(p, 20); e.sync()
            t0 = time.perf_counter()
            e.iterate(p, 100); e.sync()
            row.append((time.perf_counter() - t0) / 100 * 1e3)
            e.close()
        print('%6d %11.4f ms %11.4f ms %11.4f ms %11.4f ms' % (seeds, row[0], row[1], row[2], row[3]), flush=True)
