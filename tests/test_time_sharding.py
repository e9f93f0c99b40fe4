"""Time-axis sharding of ONE large trajectory (the alternative for BASELINE config 5; csrc/qoc_gemm_ts.h).

Rank r of G owns a run of time chunks: it forms the propagators, sweeps and gradients of its slices, the ranks exchange G rank products
(all-gather) and the gradient columns (all-reduce) per iteration.  With `time_rank = -1` ONE engine emulates all G ranks on one GPU -- the
decomposition (index ranges, rank products, the two chains over them, the column-wise gradient) is what these tests pin against the CPU oracle
(reference: core/tensorflow_state.py:25-65, 204-242, 323-356) and against the unsharded engine; the real mode replaces two no-ops by RCCL calls
(world size 1 on this box: tests/rccl_world1_script.py)."""
import numpy as np
import pytest

from oracle import grape_oracle as go
from tests.golden import cases
from tests.helpers import oracle_system
from tests.test_hip_parity import check_eval

pytestmark = pytest.mark.gpu


def problem(n=100, k=3, steps=48, m=4, reg=None):
    c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(5, 2), seed=21)
    c['total_time'] = 2.0
    if reg:
        c['reg_coeffs'] = reg
    return c


def engine(sp, **kw):
    from quantum_optimal_control.core import hip_engine
    return hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs=sp.reg_coeffs,
                                one_minus_gauss=sp.one_minus_gauss, n_seeds=1, **kw)


@pytest.mark.parametrize('G', [1, 2, 3, 5, 12])
def test_emulated_time_shards_against_the_oracle_and_the_unsharded_engine(G):
    sp = oracle_system(problem(reg={'dwdt': 0.05, 'amplitude': 0.1}))
    base = 1.5 * sp.base0 + 0.1
    ref = engine(sp, path=4)
    assert ref.plan['path'] == 'gemm' and int(ref.plan['chunks']) == 12
    ref.set_base(base[None])
    r0 = ref.evaluate()
    U0, iv0 = ref.get_final_unitary()[0], ref.get_inter_vecs()[0]
    ref.close()
    eng = engine(sp, time_shards=G, time_rank=-1)
    assert eng.plan['path'] == 'gemm' and int(eng.plan['time_shards']) == G and int(eng.plan['time_rank']) == -1
    eng.set_base(base[None])
    check_eval(eng, sp, [base])                                  # loss, reg_loss, unitary_scale, grad_squared, gradient, inter_vecs, U_final
    r = eng.evaluate()
    gmax = np.max(np.abs(r0['grad']))
    assert np.max(np.abs(r['grad'] - r0['grad'])) <= 1e-12 * gmax
    assert abs(r['loss'][0] - r0['loss'][0]) <= 1e-13
    np.testing.assert_allclose(eng.get_final_unitary()[0], U0, rtol=0, atol=1e-13)
    np.testing.assert_allclose(eng.get_inter_vecs()[0], iv0, rtol=0, atol=1e-13)
    eng.close()


def test_emulated_time_shards_follow_the_unsharded_adam_loop():
    sp = oracle_system(problem(n=128, k=2, steps=40, m=8))
    out = []
    for kw in (dict(path=4), dict(time_shards=4, time_rank=-1)):
        eng = engine(sp, **kw)
        eng.set_base(sp.base0[None])
        its = eng.run_adam(eng.adam_params(rate=0.02, learning_rate_decay=100, conv_target=1e-12, min_grad=1e-30, max_iterations=4, poll_every=4))
        assert int(its[0]) == 4
        out.append((eng.get_base()[0], eng.scalars()['loss'][0], eng.get_final_unitary()[0]))
        eng.close()
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-11)
    assert abs(out[1][1] - out[0][1]) <= 1e-12
    np.testing.assert_allclose(out[1][2], out[0][2], rtol=0, atol=1e-12)
    o = go.run_adam(sp, dict(rate=0.02, learning_rate_decay=100, conv_target=1e-12, min_grad=1e-30, max_iterations=4), base=sp.base0)
    np.testing.assert_allclose(out[1][0], o['base'], rtol=0, atol=1e-10)


def test_time_sharding_refuses_what_it_does_not_cover():
    from quantum_optimal_control.core import hip_engine
    sp = oracle_system(problem())
    for kw, what in ((dict(time_shards=13, time_rank=-1), 'number of chunks'), (dict(time_shards=2, time_rank=2), 'time_rank'),
                     (dict(time_shards=2, time_rank=-1, path=2), 'GEMM path')):
        with pytest.raises(hip_engine.QocError, match=what):
            engine(sp, **kw)
    with pytest.raises(hip_engine.QocError, match='state regulariser'):
        engine(oracle_system(problem(reg={'forbidden_coeff_list': [2.0], 'states_forbidden_list': [99]})), time_shards=2, time_rank=-1)
    with pytest.raises(hip_engine.QocError, match='N >= 128'):
        engine(oracle_system(problem(n=40)), time_shards=2, time_rank=-1)
    sp2 = oracle_system(problem())
    with pytest.raises(hip_engine.QocError, match='one control set'):
        hip_engine.HipEngine(sp2.Hs, sp2.U0, sp2.V, sp2.W, sp2.maxA, sp2.dt, sp2.total_time, sp2.steps, sp2.exp_terms, sp2.scaling, reg_coeffs={}, n_seeds=2,
                             time_shards=2, time_rank=-1)
    # a rank of a real run without its communicator fails at the first exchange, with a message
    eng = engine(sp, time_shards=2, time_rank=0)
    eng.set_base(sp.base0[None])
    with pytest.raises(hip_engine.QocError, match='communicator'):
        eng.evaluate()
    eng.close()
