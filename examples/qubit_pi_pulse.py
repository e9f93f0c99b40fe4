#!/usr/bin/env python
"""Single-qubit pi pulse (BASELINE config C1): the same call a user of the reference makes, on the MI355X engine.

    python examples/qubit_pi_pulse.py [--iterations N]

Prints the reference's progress lines and the final gate fidelity |tr(U_target^dagger U)|^2 / 4."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'quantum-optimal-control_amd'))
from quantum_optimal_control.main_grape.grape import Grape  # noqa: E402


def main(iterations=300, quiet=False):
    sz = np.array([[1, 0], [0, -1]], dtype=complex)
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    H0 = 2 * np.pi * 0.05 * sz / 2                     # 50 MHz detuning, GHz / ns units
    Hops, Hnames = [2 * np.pi * sx / 2], ['x']
    U = sx                                             # target: X gate
    convergence = {'rate': 0.05, 'update_step': 50, 'max_iterations': iterations, 'conv_target': 1e-6,
                   'learning_rate_decay': 500}
    np.random.seed(1)
    uks, U_final = Grape(H0, Hops, Hnames, U, total_time=10.0, steps=100, states_concerned_list=[0, 1],
                         convergence=convergence, reg_coeffs={'dwdt': 0.01}, maxA=[0.2], method='Adam',
                         show_plots=not quiet, save=False)
    fidelity = abs(np.trace(U.conj().T @ U_final)) ** 2 / 4
    print('pulse shape %s, max |u| = %.4f, gate fidelity = %.6f' % (uks.shape, np.max(np.abs(uks)), fidelity))
    return fidelity


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--iterations', type=int, default=300)
    main(ap.parse_args().iterations)
