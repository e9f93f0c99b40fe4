"""Shared test utilities: build an OracleSystem from a tests/golden/cases.py recipe."""
import os

import numpy as np

from oracle import grape_oracle as go
from tests.golden import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def resolve_dressed(c, fixture=None):
    """'from_H0' recipes take the dressed_info the reference computed (stored in the fixture as INPUT data)."""
    d = c['dressed_info']
    if isinstance(d, str) and d == 'from_H0':
        fx = fixture if fixture is not None else load_golden('sysparams_dressed.npz')
        d = dict(eigenvectors=fx['dressed_eigenvectors'], dressed_id=[int(i) for i in fx['dressed_id']],
                 eigenvalues=fx['dressed_eigenvalues'], is_dressed=True)
    return d


def oracle_system(c, fixture=None, seed_numpy=True):
    if seed_numpy:
        np.random.seed(c['np_seed'])
    return go.OracleSystem(c['H0'], c['Hops'], c['U'], c['total_time'], c['steps'], c['states_concerned_list'],
                           U0=c['U0'], reg_coeffs=c['reg_coeffs'], dressed_info=resolve_dressed(c, fixture),
                           maxA=c['maxA'], initial_guess=c['initial_guess'], state_transfer=c['state_transfer'],
                           Taylor_terms=c['Taylor_terms'])


def grape_kwargs(c, fixture=None):
    """kwargs for quantum_optimal_control.main_grape.grape.Grape from a recipe."""
    return dict(H0=c['H0'], Hops=c['Hops'], Hnames=c['Hnames'], U=c['U'], total_time=c['total_time'],
                steps=c['steps'], states_concerned_list=c['states_concerned_list'], U0=c['U0'],
                reg_coeffs=c['reg_coeffs'], dressed_info=resolve_dressed(c, fixture), maxA=c['maxA'],
                initial_guess=c['initial_guess'], state_transfer=c['state_transfer'],
                Taylor_terms=c['Taylor_terms'], save=False, show_plots=False)
