"""Oracle parity at the EXACT configuration bench.py times ("Next round" 1).

bench.py resolves C2 x 64 seeds per GPU to path = MFMA, the in-place-image exponential kernel, 16 chunks of 32 slices and the one-wave
sweep kernel k_mfma_downup; the other
full-size tests use 2 seeds (63 chunks of 8) or compare HIP batches with each other.  Here the 64-seed batch itself is compared with
the oracle (core/tensorflow_state.py:204-242,323-356; core/run_session.py:47-69) for seeds {0, 31, 63}: one evaluation (loss, U_final,
gradient) and three iterations of the device loop.  The expected values are committed fixtures (tests/golden/make_bench_golden.py),
so the test costs no CPU seconds on the GPU box.  The same for ONE control set with dwdt + forbidden levels (AUTO = latency mode).
"""
import numpy as np
import pytest

import bench
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu

L_ATOL = 1e-11          # loss, scalars
U_ATOL = 1e-11          # U_final entries (|entries| <= 1)
G_RTOL = 1e-10          # gradient, relative to max |g|


def bench_engine(n_seeds, reg=None, **kw):
    from quantum_optimal_control.core import hip_engine
    c, Hs, U0, V, W, dt = bench.build_problem()
    return hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], bench.SLICES, bench.TAYLOR[0], bench.TAYLOR[1],
                                reg_coeffs=reg or {}, n_seeds=n_seeds, **kw)


def compare_evaluation(eng, r, g, rows):
    Uf = eng.get_final_unitary()
    for i, b in enumerate(rows):
        for key in ('loss', 'reg_loss', 'grad_squared', 'unitary_scale'):
            assert abs(r[key][b] - g[key][i]) <= L_ATOL * max(1.0, abs(g[key][i])), (key, b, r[key][b], g[key][i])
        gmax = np.max(np.abs(g['grad'][i]))
        assert np.max(np.abs(r['grad'][b] - g['grad'][i])) <= G_RTOL * gmax, (b, np.max(np.abs(r['grad'][b] - g['grad'][i])), gmax)
        np.testing.assert_allclose(Uf[b], g['U_final'][i], rtol=0, atol=U_ATOL)


def compare_three_iterations(eng, g, rows):
    its = eng.run_adam(eng.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25, max_iterations=3, poll_every=3))
    assert np.all(its == 3)
    base, sc, Uf = eng.get_base(), eng.scalars(), eng.get_final_unitary()
    for i, b in enumerate(rows):
        np.testing.assert_allclose(base[b], g['adam_base'][i], rtol=0, atol=1e-10)
        assert abs(sc['loss'][b] - g['adam_loss'][i]) <= 1e-10
        np.testing.assert_allclose(Uf[b], g['adam_U_final'][i], rtol=0, atol=1e-10)


def test_bench_batch_configuration_against_the_oracle():
    g = load_golden('c2_bench_batch.npz')
    seeds = [int(s) for s in g['seeds']]
    eng = bench_engine(bench.SEEDS_PER_GPU)
    # what `python bench.py` resolves to: MFMA path, 16 chunks of 32 slices, the in-place-image kernel of the exponentials
    assert (eng.path, eng.chunks) == (2, 16)
    bases = bench.seed_bases(0, bench.SEEDS_PER_GPU)
    eng.set_base(bases)
    eng.profile_enable(True)
    r = eng.evaluate()
    assert eng.profile_read()['kernel'] == 'k_mfma_expm_inplace'
    eng.profile_enable(False)
    compare_evaluation(eng, r, g, seeds)
    eng.set_base(bases)
    compare_three_iterations(eng, g, seeds)
    eng.close()


def test_single_control_set_with_regularisers_against_the_oracle():
    from tests.golden.make_bench_golden import LAT_REG
    g = load_golden('c2_bench_single_regularised.npz')
    eng = bench_engine(1, reg=LAT_REG)
    assert eng.path == 2
    base = bench.seed_bases(0, 1)
    eng.set_base(base)
    eng.profile_enable(True)
    r = eng.evaluate()
    assert 'slice2' in eng.profile_read()['kernel']           # AUTO = latency mode
    eng.profile_enable(False)
    compare_evaluation(eng, r, g, [0])
    eng.set_base(base)
    compare_three_iterations(eng, g, [0])
    eng.close()


def test_bench_batch_configuration_against_the_reference_text():
    """The 64-seed bench engine against tests/golden/graph_c2_full_s{0,63}.npz: the reference's own graph text (core/tensorflow_state.py:204-242,
    323-356 on the TF1 stand-in, tests/golden/make_graph_golden.py) evaluated on the control sets of restart seeds 0 and 63 -- the oracle is
    not in between.  One evaluation and the reference's one optimiser step (run_session.py:69)."""
    fx = {s: load_golden('graph_c2_full_s%d.npz' % s) for s in (0, 63)}
    eng = bench_engine(bench.SEEDS_PER_GPU)
    assert (eng.path, eng.chunks) == (2, 16)
    bases = bench.seed_bases(0, bench.SEEDS_PER_GPU)
    for s in fx:
        np.testing.assert_array_equal(bases[s], fx[s]['base0'])
    eng.set_base(bases)
    r = eng.evaluate()
    Uf, inter = eng.get_final_unitary(), eng.get_inter_vecs()
    for s, f in fx.items():
        for key in ('loss', 'reg_loss', 'grad_squared', 'unitary_scale'):
            assert abs(r[key][s] - float(f[key])) <= L_ATOL * max(1.0, abs(float(f[key]))), (key, s, r[key][s], float(f[key]))
        gmax = np.max(np.abs(f['grad_pack']))
        assert np.max(np.abs(r['grad'][s] - f['grad_pack'])) <= G_RTOL * gmax
        np.testing.assert_allclose(Uf[s], f['final_state'], rtol=0, atol=U_ATOL)
        iv = inter[s] if f['inter_vecs'].shape[0] == bench.SLICES + 1 else inter[s][[0, bench.SLICES // 2, bench.SLICES]]
        np.testing.assert_allclose(iv, f['inter_vecs'], rtol=0, atol=U_ATOL)
    eng.adam_step(float(fx[0]['adam_lr']))
    base = eng.get_base()
    for s, f in fx.items():
        np.testing.assert_allclose(base[s], f['base_after_adam'], rtol=0, atol=1e-9)
    eng.close()


@pytest.mark.parametrize('n_seeds,rows', [(1, (0,)), (64, (0, 31, 63)), (256, (0, 127, 255))], ids=['c3_single_trajectory', 'c3_x64', 'c3_x256'])
def test_secondary_c3_engines_against_the_oracle(n_seeds, rows):
    """The C3 engines whose times the bench line prints under `secondary` ("Next round" 2), built by bench._engine_for exactly as bench.secondary_configs
    builds them -- one trajectory (propagator route), 64 control sets (packed Taylor chains, assembly beside the forward chain on masked CUs) and 256 (no overlap) --
    against committed oracle values of the named control sets (tests/golden/make_bench_golden.py c3): one evaluation and three iterations of the device loop.
    Reference: core/tensorflow_state.py:77-133, 244-261; core/regularization_functions.py:28-35, 71-85; core/run_session.py:47-69."""
    from quantum_optimal_control.helper_functions import synthetic_systems
    g = load_golden('c3_bench_batch.npz')
    at = {int(s): i for i, s in enumerate(g['sets'])}
    c3 = synthetic_systems.case_c3()
    eng, sp = bench._engine_for(c3, n_seeds, 0)
    try:
        assert eng.plan['path'] == 'gemm' and eng.plan['route'] == ('propagator' if n_seeds == 1 else 'direct')
        if n_seeds > 1:
            assert eng.plan['taylor_chain'] == 'packed'
        bases = np.random.default_rng(0).normal(0, 1 / np.sqrt(c3['steps']), (n_seeds, len(c3['Hops']), c3['steps']))
        r = eng.evaluate()
        inter = eng.get_inter_vecs()
        for b in rows:
            i = at[b]
            for key in ('loss', 'reg_loss', 'grad_squared', 'unitary_scale'):
                assert abs(r[key][b] - g[key][i]) <= L_ATOL * max(1.0, abs(g[key][i])), (key, b, r[key][b], g[key][i])
            gmax = np.max(np.abs(g['grad'][i]))
            assert np.max(np.abs(r['grad'][b] - g['grad'][i])) <= G_RTOL * gmax, (b, np.max(np.abs(r['grad'][b] - g['grad'][i])), gmax)
            np.testing.assert_allclose(inter[b][-1], g['final_vecs'][i], rtol=0, atol=U_ATOL)
        del inter
        eng.set_base(bases)
        its = eng.run_adam(eng.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25, max_iterations=3, poll_every=3))
        assert np.all(its == 3)
        base, sc = eng.get_base(), eng.scalars()
        for b in rows:
            i = at[b]
            np.testing.assert_allclose(base[b], g['adam_base'][i], rtol=0, atol=1e-10)
            assert abs(sc['loss'][b] - g['adam_loss'][i]) <= 1e-10 and abs(sc['reg_loss'][b] - g['adam_reg_loss'][i]) <= 1e-10
    finally:
        eng.close()
