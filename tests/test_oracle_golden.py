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
