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
# round 4: BASELINE configs 2 and 3 at their FULL sizes (C2 from the control sets of bench.py's restart seeds 0 and 63)
FULL_CASES = ['c2_full_s0', 'c2_full_s63', 'c3_full']
# one case per kernel family of the engine: 48- / 64-wide MFMA kernels, a padded size, three qutrits with six controls and forbidden levels, n = 100
FAMILY_CASES = ['fam_n40', 'fam_n64', 'fam_n20', 'fam_qutrits', 'fam_n100']


def graph_case(name):
    from tests.golden.make_graph_golden import full_size_cases, graph_cases, kernel_family_cases
    if name in FAMILY_CASES:
        return kernel_family_cases()[name]
    return full_size_cases()[name] if name in FULL_CASES else graph_cases()[name]


def picked_time_points(inter, fx_inter):
    """Fixtures that keep three time points (first, middle, last) against the full trajectory."""
    steps = inter.shape[0] - 1
    return inter if fx_inter.shape[0] == steps + 1 else inter[[0, steps // 2, steps]]


@pytest.mark.parametrize('name', GRAPH_CASES + FULL_CASES + FAMILY_CASES)
def test_oracle_matches_the_reference_graph_code(name):
    c = graph_case(name)
    fx = load_golden('graph_%s.npz' % name)
    sp = oracle_system(c)
    assert (sp.exp_terms, sp.scaling) == (int(fx['exp_terms']), int(fx['scaling']))
    if c.get('base0') is None:
        np.testing.assert_array_equal(sp.base0, fx['base0'])             # same NumPy RNG stream: the two start from the same variable
    else:
        np.testing.assert_array_equal(c['base0'], fx['base0'])
        sp.base0 = np.array(fx['base0'])
    o = go.evaluate(sp, sp.base0, want_inter=True)
    o['inter_vecs'] = picked_time_points(o['inter_vecs'], fx['inter_vecs'])
    for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared'):
        assert abs(o[key] - float(fx[key])) <= 1e-12 * max(1.0, abs(float(fx[key]))), (key, o[key], float(fx[key]))
    gmax = np.max(np.abs(fx['grad_pack']))
    assert np.max(np.abs(o['grad'] - fx['grad_pack'])) <= 1e-12 * max(gmax, 1e-3), np.max(np.abs(o['grad'] - fx['grad_pack']))
    np.testing.assert_allclose(o['inter_vecs'], fx['inter_vecs'], rtol=0, atol=1e-13)
    if not sp.state_transfer:
        np.testing.assert_allclose(o['U_final'], fx['final_state'], rtol=0, atol=1e-13)
    # one TF1 Adam step (tensorflow_state.py:342-356, run_session.py:69)
    # (full sizes: the first step is lr g / (|g| + 1e-8), whose slope at the few entries with |g| ~ 1e-8 is ~1e6 -- 1e-17 of gradient
    # round-off becomes 1e-11 of the variable there; measured 1.7e-12 on 2 of 2000 entries)
    base1 = go.Adam(sp.base0.shape).step(sp.base0.copy(), o['grad'], float(fx['adam_lr']))
    np.testing.assert_allclose(base1, fx['base_after_adam'], rtol=0, atol=1e-11 if name in FULL_CASES else 1e-13)


# ---- tier 2: the reference's text at the reference's OWN precision ---------------------------------------------------------
# tests/golden/graph32_*.npz: the same generator with `tf.float32` = torch.float32 (make_graph_golden.py --fp32).  What the real reference
# computes differs from the fp64 restatement by float32 round-off accumulated over the slices; measured at generation time (fp64 run of the
# same text against the fp32 run): scalars <= 2.3e-6, gradient <= 1.6e-5 max|g|, vectors <= 6.1e-6 (all at full C2 size, the worst case).
# The bounds below are those with a margin of ~5.  The variable after an Adam step is NOT compared: the first TF1-Adam step is
# lr * g / (|g| + eps'), i.e. the sign of g -- entries with |g| below the float32 error flip (8.6e-4 at C2).
FP32_CASES = ['c1', 'c2_n8', 'unitary_allreg', 'state_transfer_allreg', 'dressed_forbidden', 'c2_full_s0', 'c3_full']
T2_SCALAR, T2_GRAD, T2_VEC = 1e-5, 1e-4, 5e-5


def assert_tier2(o, fx, state_transfer):
    worst = 0.0
    for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared'):
        d = abs(o[key] - float(fx[key])) / max(1.0, abs(float(fx[key])))
        assert d <= T2_SCALAR, (key, o[key], float(fx[key]))
        worst = max(worst, d)
    gmax = np.max(np.abs(fx['grad_pack']))
    dg = np.max(np.abs(o['grad'] - fx['grad_pack'])) / max(gmax, 1e-3)
    assert dg <= T2_GRAD, dg
    np.testing.assert_allclose(o['inter_vecs'], fx['inter_vecs'], rtol=0, atol=T2_VEC)
    if not state_transfer:
        np.testing.assert_allclose(o['U_final'], fx['final_state'], rtol=0, atol=T2_VEC)
    return max(worst, dg)


@pytest.mark.parametrize('name', FP32_CASES)
def test_oracle_within_float32_roundoff_of_the_reference_text_at_its_own_precision(name):
    c = graph_case(name)
    fx = load_golden('graph32_%s.npz' % name)
    sp = oracle_system(c)
    base = np.array(c['base0']) if c.get('base0') is not None else sp.base0
    # the float32 run starts from the float32 rounding of the variable (tf.constant(..., dtype=tf.float32), tensorflow_state.py:176)
    np.testing.assert_array_equal(fx['base0'], base.astype(np.float32).astype(np.float64))
    o = go.evaluate(sp, base, want_inter=True)
    o['inter_vecs'] = picked_time_points(o['inter_vecs'], fx['inter_vecs'])
    d = assert_tier2(o, fx, sp.state_transfer)
    assert d > 1e-10                                                     # and it really was a float32 run
