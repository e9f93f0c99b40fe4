#!/usr/bin/env python
"""One C2 trajectory with the speed_up regulariser (+ dwdt): the routes AUTO can take."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
from tests.golden import cases
from c2_forbidden_single import run

if __name__ == '__main__':
    for n, k, steps in ((32, 4, 500),):
        c = cases.case_c2(n=n, k=k, steps=steps, m=8, taylor=(5, 3), seed=3)
        c['reg_coeffs'] = {'dwdt': 1e-3, 'speed_up': 0.1}
        for seeds in (1, 4, 64):
            print('n=%-2d k=%d steps=%-4d seeds=%-2d + dwdt + speed_up: AUTO %.4f ms (path %d)   GEMM route %.4f ms   latency mode with sources %.4f ms   batch kernels %.4f ms'
                  % ((n, k, steps, seeds) + run(c, seeds, 0, 0, 30) + (run(c, seeds, 4, 0, 30)[0], run(c, seeds, 2, 5, 30)[0], run(c, seeds, 2, 0, 30)[0])), flush=True)
