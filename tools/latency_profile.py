#!/usr/bin/env python
"""Profiling hook: one C2 trajectory on a given (path, variant) -- run under rocprofv3 --kernel-trace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
import bench  # noqa: E402
from quantum_optimal_control.core import hip_engine  # noqa: E402

if __name__ == '__main__':
    path, variant, seeds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1
    c, Hs, U0, V, W, dt = bench.build_problem()
    e = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], bench.SLICES, bench.TAYLOR[0], bench.TAYLOR[1], reg_coeffs={},
                             n_seeds=seeds, path=path, variant=variant)
    e.set_base(bench.seed_bases(0, seeds))
    p = e.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25, max_iterations=10 ** 9, poll_every=10 ** 9)
    e.iterate(p, 100)
    e.sync()
    e.close()
