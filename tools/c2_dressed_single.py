#!/usr/bin/env python
"""One C2-size trajectory with DRESSED forbidden levels + dwdt: the routes AUTO can take."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import numpy as np
from quantum_optimal_control.helper_functions import grape_functions as gf
from tests.golden import cases
from c2_forbidden_single import run

if __name__ == '__main__':
    for n, k, steps in ((32, 4, 500), (27, 3, 1000)):
        c = cases.case_c2(n=n, k=k, steps=steps, m=8, taylor=(5, 3), seed=3)
        w, v, did = gf.get_dressed_info(c['H0'])
        c['dressed_info'] = dict(eigenvectors=v, dressed_id=did, eigenvalues=w, is_dressed=True)
        c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [n - 2, n - 1], 'forbid_dressed': True}
        for seeds in (1, 2, 4):
            print('n=%-2d k=%d steps=%-4d seeds=%d + dwdt + DRESSED forbidden: AUTO %.4f ms (path %d)   GEMM route %.4f ms   latency mode with sources %.4f ms   batch kernels %.4f ms'
                  % ((n, k, steps, seeds) + run(c, seeds, 0, 0) + (run(c, seeds, 4, 0)[0], run(c, seeds, 2, 5)[0], run(c, seeds, 2, 0)[0])), flush=True)
