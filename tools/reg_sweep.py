#!/usr/bin/env python
"""C2 (n=32 k=4 steps=500 m=8) with each regulariser switched on alone: ms per iteration on AUTO for 1 and 64 control sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
from quantum_optimal_control.helper_functions import grape_functions as gf
from tests.golden import cases
from c2_forbidden_single import run

if __name__ == '__main__':
    regs = [('none', {}), ('amplitude', {'amplitude': 0.1}), ('envelope', {'envelope': 0.1}), ('dwdt', {'dwdt': 1e-3}), ('dwdt + d2wdt2', {'dwdt': 1e-3, 'd2wdt2': 1e-3}),
            ('bandpass', {'bandpass': 0.1, 'band': [0.5, 2.0]}), ('forbidden x2', {'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [30, 31]}),
            ('dressed forbidden x2', {'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [30, 31], 'forbid_dressed': True}),
            ('speed_up', {'speed_up': 0.1}),
            ('all of them', {'amplitude': 0.1, 'envelope': 0.1, 'dwdt': 1e-3, 'd2wdt2': 1e-3, 'bandpass': 0.1, 'band': [0.5, 2.0],
                             'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [30, 31], 'forbid_dressed': True, 'speed_up': 0.1})]
    for name, reg in regs:
        c = cases.case_c2(n=32, k=4, steps=500, m=8, taylor=(5, 3), seed=3)
        if reg.get('forbid_dressed'):
            w, v, did = gf.get_dressed_info(c['H0'])
            c['dressed_info'] = dict(eigenvectors=v, dressed_id=did, eigenvalues=w, is_dressed=True)
        c['reg_coeffs'] = reg
        a1, a64 = run(c, 1, 0, 0, 50), run(c, 64, 0, 0, 20)
        print('%-22s : 1 control set %.4f ms (path %d)   64 control sets %.4f ms (path %d)' % (name, a1[0], a1[1], a64[0], a64[1]), flush=True)
