#!/usr/bin/env python
"""48 < n <= 64 batches: GEMM path against the MFMA batch kernels (NT = 4), for the AUTO policy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
from mid_n_sweep import run
from tests.golden import cases
for n, k, steps in ((64, 4, 500), (56, 4, 500), (64, 6, 200), (64, 8, 1000)):
    c = cases.case_c2(n=n, k=k, steps=steps, m=8, taylor=(5, 3), seed=2)
    for seeds in (8, 16, 32, 64, 128):
        print('n=%-2d k=%d steps=%-4d seeds=%-3d : AUTO %.3f ms   GEMM %.3f ms   MFMA batch %.3f ms'
              % (n, k, steps, seeds, run(c, seeds, 0, 0, 8), run(c, seeds, 4, 0, 8), run(c, seeds, 2, 7, 8)), flush=True)
