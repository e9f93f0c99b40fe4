#!/usr/bin/env python
"""Profiling hook: one trajectory of an n-level unitary problem (500 slices, k = 4, m = 8) on AUTO -- run under rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
import numpy as np
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system
n = int(sys.argv[1])
sp = oracle_system(cases.case_c2(n=n, k=4, steps=500, m=8, taylor=(5, 3), seed=2))
e = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs={}, n_seeds=1)
e.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (1, sp.k, sp.steps)))
p = e.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
e.iterate(p, 100); e.sync(); e.close()
