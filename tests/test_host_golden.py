"""Host-side mirror of the reference interface (rows a1-a5, a17) vs fixtures captured from the reference."""
import io
import contextlib

import numpy as np
import pytest

from quantum_optimal_control.core.convergence import Convergence
from quantum_optimal_control.core.system_parameters import SystemParameters
from quantum_optimal_control.helper_functions import grape_functions as gf
from tests.golden import cases
from tests.helpers import load_golden, resolve_dressed


def build_sys_para(c, fx=None):
    n = len(c['H0'])
    U0 = np.identity(n) if c['U0'] is None else c['U0']
    if c['maxA'] is None:
        maxA = (4 * np.ones(len(c['Hops'])) if c['initial_guess'] is None
                else 1.5 * np.max(np.abs(c['initial_guess'])) * np.ones(len(c['Hops'])))
    else:
        maxA = c['maxA']
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        return SystemParameters(c['H0'], c['Hops'], c['Hnames'], c['U'], U0, c['total_time'], c['steps'],
                                c['states_concerned_list'], resolve_dressed(c, fx), maxA, None, c['initial_guess'],
                                False, 1e-4, c['state_transfer'], False, c['reg_coeffs'], False, None,
                                c['Taylor_terms'], True, True, False, False, False)


@pytest.mark.parametrize('name', list(cases.ALL_CASES))
def test_system_parameters_bit_exact(name):
    c = cases.ALL_CASES[name]()
    fx = load_golden('sysparams_%s.npz' % name)
    S = build_sys_para(c, fx)
    assert S.dt == float(fx['dt']) and S.state_num == int(fx['state_num'])
    assert (S.exp_terms, S.scaling) == (int(fx['exp_terms']), int(fx['scaling']))
    if c['Taylor_terms'] is None:
        assert list(S.exps) == list(fx['exps']) and list(S.scalings) == list(fx['scalings'])
    np.testing.assert_array_equal(S.matrix_list, fx['matrix_list'])
    np.testing.assert_array_equal(np.array(S.initial_vectors), fx['initial_vectors'])
    np.testing.assert_array_equal(S.initial_unitary, fx['initial_unitary'])
    np.testing.assert_array_equal(S.one_minus_gauss, fx['one_minus_gauss'])
    np.testing.assert_array_equal(S.ops_weight_base, fx['ops_weight_base'])
    if c['state_transfer']:
        np.testing.assert_array_equal(np.array(S.target_vectors), fx['target_vectors'])
    else:
        np.testing.assert_array_equal(S.target_unitary, fx['target_unitary'])
    if c['initial_guess'] is not None:
        np.testing.assert_array_equal(S.u0_base, fx['u0_base'])
    # complex stack handed to the engine is the un-embedded matrix_list
    Hs, U0, V, W, Vs = S.engine_inputs()
    n = S.state_num
    for i in range(len(Hs)):
        np.testing.assert_array_equal(gf.c_to_r_mat(Hs[i]), fx['matrix_list'][i])


def test_helper_functions_match_reference():
    fx = load_golden('helpers.npz')
    np.testing.assert_array_equal(gf.c_to_r_mat(fx['in_M']), fx['c_to_r_mat'])
    np.testing.assert_array_equal(gf.c_to_r_vec(fx['in_v']), fx['c_to_r_vec'])
    did = [int(i) for i in fx['dressed_id']]
    np.testing.assert_array_equal(gf.sort_ev(fx['dressed_v'], did), fx['sort_ev'])
    assert gf.get_state_index(2, did) == int(fx['state_index_2'])
    np.testing.assert_allclose(gf.dressed_unitary(fx['in_Ugate'], fx['dressed_v'], did), fx['dressed_unitary'], atol=1e-15)
    w, v, d2 = gf.get_dressed_info(fx['in_Hd'])
    assert d2 == did
    np.testing.assert_allclose(np.sort(w.real), np.sort(fx['dressed_w'].real), atol=1e-12)


def test_convergence_defaults():
    conv = Convergence(None, 'ns', {})
    assert (conv.rate, conv.update_step, conv.evol_save_step, conv.conv_target, conv.max_iterations,
            conv.learning_rate_decay, conv.min_grad) == (0.01, 100, 100, 1e-8, 5000, 2500, 1e-25)
    conv = Convergence(None, 'ns', {'rate': 0.5, 'min_grad': 1e-9})
    assert conv.rate == 0.5 and conv.min_grad == 1e-9 and conv.max_iterations == 5000


def test_builder_helpers_match_reference():
    """Row f2: caller-side builders vs outputs of the reference's own functions (tests/golden/helpers.npz)."""
    fx = load_golden('helpers.npz')
    np.testing.assert_allclose(gf.qft(2), fx['qft2'], atol=1e-15)
    np.testing.assert_array_equal(gf.Hadamard(2), fx['hadamard2'])
    assert gf.concerned(2, 3) == list(fx['concerned_2_3'])
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    np.testing.assert_array_equal(gf.transmon_gate(cnot, 3), fx['transmon_gate_cnot_3'])
    np.testing.assert_array_equal(np.array(gf.rz(0.7)), fx['rz'])
    np.testing.assert_array_equal(np.array(gf.rx(0.7)), fx['rx'])
    sx = np.array([[0, 1], [1, 0]], dtype=float)
    np.testing.assert_array_equal(gf.kron_all(sx, 3, np.eye(2)), fx['kron_all'])
    np.testing.assert_array_equal(gf.multi_kron(sx, 3), fx['multi_kron'])
    a3 = np.diag(np.sqrt(np.arange(1, 3)), 1)
    np.testing.assert_array_equal(gf.nn_chain_kron(a3 + a3.T, np.eye(3), 3, 3), fx['nn_chain_kron'])
    Hops, Hnames, amps = gf.append_separate_krons(a3 + a3.T, 'x', 2, 3, [], [], [], amp=2.5)
    np.testing.assert_array_equal(np.array(Hops), fx['sep_kron_ops'])
    assert list(Hnames) == [str(s) for s in fx['sep_kron_names']] and list(amps) == list(fx['sep_kron_amps'])
    assert gf.Bin(5, 6) == str(fx['bin_5_6']) and gf.Basis(7, 3, 3) == str(fx['basis_7_3_3'])
    assert gf.baseN(11, 3) == str(fx['baseN_11_3']) and gf.baseN(0, 5) == '0'
    assert gf.hamming_distance(0b101101) == 4


def test_convergence_plot_summary_writes_the_reference_panels(tmp_path):
    """Row f4 (UI): the summary figure of convergence.py:121-222 -- error curves, final operator (re / im), pulses, one
    population panel per concerned state with the forbidden-level sum -- drawn from stub read-back, off-screen."""
    pytest.importorskip('matplotlib')
    from quantum_optimal_control.core.convergence import Convergence

    class Sys(object):
        pass

    n, k, steps, m = 4, 2, 20, 2
    sp = Sys()
    sp.state_transfer, sp.use_inter_vecs, sp.dt, sp.steps, sp.ops_len = False, True, 0.1, steps, k
    sp.ops_max_amp, sp.Hnames, sp.states_concerned_list = [1.0, 2.0], ['x', 'y'], [0, 1]
    sp.draw_list, sp.draw_names, sp.dressed_info = [], [], None
    sp.reg_coeffs = {'states_forbidden_list': [3], 'forbidden_coeff_list': [1.0]}
    rng = np.random.default_rng(0)

    class Anly(object):
        def get_final_state(self):
            return np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))[0]

        def get_inter_vecs(self):
            p = rng.uniform(size=(m, n, steps + 1))
            return list(p / p.sum(axis=1, keepdims=True))

        def get_ops_weight(self):
            return np.sin(rng.normal(size=(k, steps)))

    conv = Convergence(sp, 'ns', {'max_iterations': 100})
    for it in range(0, 30, 10):
        conv.record(it, 0.5 / (1 + it), 0.6 / (1 + it))
    out = tmp_path / 'summary.png'
    fig = conv.plot_summary(0.02, 0.03, Anly(), unitary_metric=0.99999, filename=str(out))
    assert out.exists() and out.stat().st_size > 10000
    titles = [ax.get_title() for ax in fig.axes]
    assert any(t.startswith('Error = 2.00e-02; Other errors = 1.00e-02') for t in titles)
    assert 'operator: real' in titles and 'operator: imaginary' in titles and 'Optimized pulse' in titles
    assert titles.count('Evolution') == m
