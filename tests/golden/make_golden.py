#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING the reference (build container only).

What can be imported from /root/reference without TensorFlow / Python 2:
  * quantum_optimal_control/helper_functions/grape_functions.py  -- as is (pure NumPy/SciPy)
  * quantum_optimal_control/core/system_parameters.py            -- after a mechanical lib2to3 pass applied to a
    SCRATCH COPY under a temp dir OUTSIDE the repo (print statements / xrange), with a no-op stand-in for the HDF5
    logger class (save=False, so it is never called).
No reference source is copied into the repo: only input recipes (tests/golden/cases.py, ours) and the numeric
outputs of the reference functions are stored.  The TF graph itself cannot run here, so these
fixtures pin rows a1-a5 (pre-processing) and the helper builders; a6-a14 are pinned by oracle/tf_graph_emulation.py.

Run:  python tests/golden/make_golden.py        (needs /root/reference; writes tests/golden/*.npz)
"""
import importlib
import io
import os
import shutil
import sys
import tempfile
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/quantum_optimal_control'
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "quantum-optimal-control_amd"))

from tests.golden import cases  # noqa: E402


def import_reference():
    scratch = tempfile.mkdtemp(prefix='qoc_ref_py3_')
    pkg = os.path.join(scratch, 'quantum_optimal_control')
    os.makedirs(os.path.join(pkg, 'core'))
    os.makedirs(os.path.join(pkg, 'helper_functions'))
    for d in ('', 'core', 'helper_functions'):
        open(os.path.join(pkg, d, '__init__.py'), 'w').close()
    shutil.copy(os.path.join(REF, 'helper_functions', 'grape_functions.py'),
                os.path.join(pkg, 'helper_functions', 'grape_functions.py'))
    shutil.copy(os.path.join(REF, 'core', 'system_parameters.py'), os.path.join(pkg, 'core', 'system_parameters.py'))
    with open(os.path.join(pkg, 'helper_functions', 'data_management.py'), 'w') as f:
        f.write('class H5File(object):\n    def __init__(self, *a, **k):\n        raise RuntimeError("save=False")\n')
    from lib2to3.main import main as two_to_three
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        two_to_three('lib2to3.fixes', ['-w', '-n', os.path.join(pkg, 'core', 'system_parameters.py')])
    # the reference package has the same top-level name as this repo's drop-in package (which tests/golden/cases.py imports its recipes
    # from): forget ours, or the import below would silently resolve to THIS repo's modules
    for name in [m for m in sys.modules if m == 'quantum_optimal_control' or m.startswith('quantum_optimal_control.')]:
        del sys.modules[name]
    sys.path.insert(0, scratch)
    gf = importlib.import_module('quantum_optimal_control.helper_functions.grape_functions')
    sp = importlib.import_module('quantum_optimal_control.core.system_parameters')
    assert gf.__file__.startswith(scratch) and sp.__file__.startswith(scratch), 'not the reference: %s %s' % (gf.__file__, sp.__file__)
    return gf, sp, scratch


def run_case(sp_mod, gf, name, c):
    n = len(c['H0'])
    # Grape() defaulting, main_grape/grape.py:89-101
    U0 = np.identity(n) if c['U0'] is None else c['U0']
    if c['maxA'] is None:
        if c['initial_guess'] is None:
            maxAmp = 4 * np.ones(len(c['Hops']))
        else:
            maxAmp = 1.5 * np.max(np.abs(c['initial_guess'])) * np.ones(len(c['Hops']))
    else:
        maxAmp = c['maxA']
    dressed = c['dressed_info']
    extra = {}
    if isinstance(dressed, str) and dressed == 'from_H0':
        w_c, v_c, dressed_id = gf.get_dressed_info(c['H0'])
        dressed = {'eigenvectors': v_c, 'dressed_id': dressed_id, 'eigenvalues': w_c, 'is_dressed': True}
        extra = dict(dressed_eigenvectors=v_c, dressed_id=np.array(dressed_id), dressed_eigenvalues=w_c)
    guess = c['initial_guess']
    if guess is not None:
        # modern NumPy cannot evaluate the reference's `self.u0 != []` (system_parameters.py:274) for an ndarray;
        # a list of 1-D rows takes the same branch on every NumPy version and is numerically identical.
        guess = [np.asarray(row) for row in guess]
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        S = sp_mod.SystemParameters(c['H0'], c['Hops'], c['Hnames'], c['U'], U0, c['total_time'], c['steps'],
                                    c['states_concerned_list'], dressed, maxAmp, None, guess, False,
                                    1e-4, c['state_transfer'], False, c['reg_coeffs'], False, None,
                                    c['Taylor_terms'], True, True, False, False, False)
    out = dict(dt=S.dt, state_num=S.state_num, exp_terms=S.exp_terms, scaling=S.scaling,
               matrix_list=np.asarray(S.matrix_list, dtype=np.float64),
               initial_vectors=np.asarray(S.initial_vectors, dtype=np.float64),
               initial_unitary=np.asarray(S.initial_unitary, dtype=np.float64),
               one_minus_gauss=np.asarray(S.one_minus_gauss), ops_weight_base=np.asarray(S.ops_weight_base),
               ops_max_amp=np.asarray(S.ops_max_amp, dtype=np.float64), **extra)
    if c['Taylor_terms'] is None:
        out['exps'] = np.asarray(S.exps)
        out['scalings'] = np.asarray(S.scalings)
    if c['state_transfer']:
        out['target_vectors'] = np.asarray(S.target_vectors, dtype=np.float64)
    else:
        out['target_unitary'] = np.asarray(S.target_unitary, dtype=np.float64)
    if c['initial_guess'] is not None:
        out['u0_base'] = np.asarray(S.u0_base)
    np.savez_compressed(os.path.join(HERE, 'sysparams_%s.npz' % name), **out)
    print('  sysparams_%s: T=%d s=%d' % (name, S.exp_terms, S.scaling),
          ('exps=%s scalings=%s' % (list(S.exps), list(S.scalings))) if c['Taylor_terms'] is None else '')


def run_helpers(gf):
    rng = np.random.default_rng(42)
    M = rng.normal(size=(3, 3)) + 1j * rng.normal(size=(3, 3))
    v = rng.normal(size=4) + 1j * rng.normal(size=4)
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    sx = np.array([[0, 1], [1, 0]], dtype=float)
    a3 = np.diag(np.sqrt(np.arange(1, 3)), 1)
    Hd = np.diag([0.0, 1.0, 2.1, 2.9]) + 0.1 * (np.ones((4, 4)) - np.eye(4))
    w_c, v_c, dressed_id = gf.get_dressed_info(Hd)
    Ugate = np.array([[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=complex)
    Hops, Hnames, amps = gf.append_separate_krons(a3 + a3.T, 'x', 2, 3, [], [], [], amp=2.5)
    out = dict(
        in_M=M, in_v=v, c_to_r_mat=gf.c_to_r_mat(M), c_to_r_vec=gf.c_to_r_vec(v),
        qft2=gf.qft(2), hadamard2=gf.Hadamard(2), concerned_2_3=np.array(gf.concerned(2, 3)),
        transmon_gate_cnot_3=gf.transmon_gate(cnot, 3), rz=np.array(gf.rz(0.7)), rx=np.array(gf.rx(0.7)),
        kron_all=gf.kron_all(sx, 3, np.eye(2)), multi_kron=gf.multi_kron(sx, 3),
        nn_chain_kron=gf.nn_chain_kron(a3 + a3.T, np.eye(3), 3, 3),
        sep_kron_ops=np.array(Hops), sep_kron_names=np.array(Hnames), sep_kron_amps=np.array(amps),
        in_Hd=Hd, dressed_w=w_c, dressed_v=v_c, dressed_id=np.array(dressed_id),
        sort_ev=gf.sort_ev(v_c, dressed_id), state_index_2=gf.get_state_index(2, dressed_id),
        in_Ugate=Ugate, dressed_unitary=gf.dressed_unitary(Ugate, v_c, dressed_id),
        bin_5_6=np.array(gf.Bin(5, 6)), basis_7_3_3=np.array(gf.Basis(7, 3, 3)), baseN_11_3=np.array(gf.baseN(11, 3)),
    )
    np.savez_compressed(os.path.join(HERE, 'helpers.npz'), **out)
    print('  helpers: %d arrays' % len(out))


if __name__ == '__main__':
    gf, sp_mod, scratch = import_reference()
    try:
        print('reference imported from scratch copy', scratch, 'numpy', np.__version__)
        for name, fn in cases.ALL_CASES.items():
            run_case(sp_mod, gf, name, fn())
        run_helpers(gf)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
