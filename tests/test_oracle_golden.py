"""Oracle pre-processing (rows a1-a5) vs fixtures captured from the reference's own NumPy code (tests/golden)."""
import numpy as np
import pytest

from oracle import grape_oracle as go
from tests.golden import cases
from tests.helpers import load_golden, oracle_system


@pytest.mark.parametrize('name', list(cases.ALL_CASES))
def test_system_parameters_match_reference(name):
    c = cases.ALL_CASES[name]()
    fx = load_golden('sysparams_%s.npz' % name)
    sp = oracle_system(c, fx)
    assert sp.dt == float(fx['dt'])
    assert sp.n == int(fx['state_num'])
    assert (sp.exp_terms, sp.scaling) == (int(fx['exp_terms']), int(fx['scaling']))
    if c['Taylor_terms'] is None:
        assert list(sp.exps) == list(fx['exps'])
        assert list(sp.scalings) == list(fx['scalings'])
    # matrix_list is bit-exact: same -1j*dt*H products, same embedding
    np.testing.assert_array_equal(sp.matrix_list(), fx['matrix_list'])
    np.testing.assert_array_equal(go.c_to_r_mat(sp.U0), fx['initial_unitary'])
    iv = np.stack([go.c_to_r_vec(sp.V[:, j]) for j in range(sp.m)])
    np.testing.assert_array_equal(iv, fx['initial_vectors'])
    np.testing.assert_array_equal(sp.one_minus_gauss, fx['one_minus_gauss'])
    np.testing.assert_array_equal(sp.base0, fx['ops_weight_base'])      # same global NumPy RNG stream / arcsin
    np.testing.assert_array_equal(sp.maxA, fx['ops_max_amp'])
    if c['state_transfer']:
        tv = np.stack([go.c_to_r_vec(sp.W[:, j]) for j in range(sp.m)])
        np.testing.assert_array_equal(tv, fx['target_vectors'])
    else:
        np.testing.assert_array_equal(go.c_to_r_mat(sp.U_target), fx['target_unitary'])
    if c['initial_guess'] is not None:
        np.testing.assert_array_equal(sp.u0_base, fx['u0_base'])


def test_embedding_helpers_match_reference():
    fx = load_golden('helpers.npz')
    np.testing.assert_array_equal(go.c_to_r_mat(fx['in_M']), fx['c_to_r_mat'])
    np.testing.assert_array_equal(go.c_to_r_vec(fx['in_v']), fx['c_to_r_vec'])
    did = [int(i) for i in fx['dressed_id']]
    np.testing.assert_array_equal(go.sort_ev(fx['dressed_v'], did), fx['sort_ev'])
    assert go.get_state_index(2, did) == int(fx['state_index_2'])
    # RtoCMat inverts the embedding (analysis.py:18-24)
    np.testing.assert_array_equal(go.r_to_c_mat(fx['c_to_r_mat'], 3), fx['in_M'])


def test_guess_above_max_amp_raises():
    c = cases.case_guess()
    c['maxA'] = [0.1, 0.1]
    with pytest.raises(ValueError, match='Initial guess has strength > max_amp'):
        oracle_system(c)


# ---- rows a6-a14: the oracle against vectors produced by the reference's OWN graph code -------------------------------------------------
# tests/golden/graph_*.npz come from tests/golden/make_graph_golden.py: core/tensorflow_state.py + core/regularization_functions.py of the
# reference (lib2to3 scratch copy) executed against a TF1 stand-in on torch autograd (tests/golden/tf1_shim.py), every float32 tensor held
# in float64.  Loop bounds, slices, signs, the custom gradient functions and the Adam update in these numbers are the reference's text.
GRAPH_CASES = ['c1', 'small_auto_U0', 'dressed_forbidden', 'state_small', 'c3_small', 'unitary_allreg', 'state_transfer_allreg', 'c2_n8']


@pytest.mark.parametrize('name', GRAPH_CASES)
def test_oracle_matches_the_reference_graph_code(name):
    from tests.golden.make_graph_golden import graph_cases
    c = graph_cases()[name]
    fx = load_golden('graph_%s.npz' % name)
    sp = oracle_system(c)
    assert (sp.exp_terms, sp.scaling) == (int(fx['exp_terms']), int(fx['scaling']))
    np.testing.assert_array_equal(sp.base0, fx['base0'])                 # same NumPy RNG stream: the two start from the same variable
    o = go.evaluate(sp, sp.base0, want_inter=True)
    for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared'):
        assert abs(o[key] - float(fx[key])) <= 1e-12 * max(1.0, abs(float(fx[key]))), (key, o[key], float(fx[key]))
    gmax = np.max(np.abs(fx['grad_pack']))
    assert np.max(np.abs(o['grad'] - fx['grad_pack'])) <= 1e-12 * max(gmax, 1e-3), np.max(np.abs(o['grad'] - fx['grad_pack']))
    np.testing.assert_allclose(o['inter_vecs'], fx['inter_vecs'], rtol=0, atol=1e-13)
    if not sp.state_transfer:
        np.testing.assert_allclose(o['U_final'], fx['final_state'], rtol=0, atol=1e-13)
    # one TF1 Adam step (tensorflow_state.py:342-356, run_session.py:69)
    base1 = go.Adam(sp.base0.shape).step(sp.base0.copy(), o['grad'], float(fx['adam_lr']))
    np.testing.assert_allclose(base1, fx['base_after_adam'], rtol=0, atol=1e-13)
