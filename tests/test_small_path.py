"""GPU parity of the workgroup-resident path (QOC_PATH_SMALL = 5, csrc/qoc_small_kernel.h) against the CPU oracle, through the C ABI.

The path runs whole iterations inside one launch: a row of 16 lanes per time slice, matrices column-per-lane in registers, the chain as a
product tree in LDS, several workgroups per control set (qoc_config.chunks pins how many) exchanging subtree products and partial sums.
Tolerances are those of tests/test_hip_parity.py (fp64 against the oracle: vectors 1e-12, gradient 1e-11 max|g|, scalars 1e-12 rel).
Reference: core/tensorflow_state.py:25-46, 204-242, 323-356; core/regularization_functions.py:15-45, 69-95; core/run_session.py:47-69."""
import numpy as np
import pytest

from oracle import grape_oracle as go
from tests.golden import cases
from tests.helpers import oracle_system
from tests.test_hip_parity import check_eval, make_engine

pytestmark = pytest.mark.gpu

SMALL = 5
PULSE_REG = {'amplitude': 0.3, 'envelope': 0.2, 'dwdt': 0.1, 'd2wdt2': 0.05}


def small_cases():
    out = [('c1', cases.case_c1()), ('small_auto_U0', cases.case_small_auto()), ('guess', cases.case_guess()),
           ('dressed_forbidden', cases.case_dressed()), ('state_small', cases.case_state_small()), ('big_auto_n12', cases.case_big_auto()),
           ('c3_n12_forbidden', cases.case_c3(n=12, k=3, steps=30, taylor=(10, 0)))]
    c = cases.case_c2(n=8, k=4, steps=50, m=8, taylor=(6, 2), seed=3); out.append(('c2_n8', c))
    c = cases.case_c2(n=4, k=2, steps=12, m=3, taylor=(6, 1), seed=2); c['reg_coeffs'] = dict(PULSE_REG); c['total_time'] = 2.0
    out.append(('unitary_pulse_regs', c))
    c = cases.case_c2(n=4, k=2, steps=37, m=3, taylor=(6, 1), seed=2); c['total_time'] = 2.0
    c['reg_coeffs'] = dict(PULSE_REG, forbidden_coeff_list=[3.0, 2.0], states_forbidden_list=[3, 2], speed_up=0.7)
    out.append(('unitary_allreg_no_band', c))
    c = cases.case_c3(n=6, k=3, steps=15, taylor=(8, 0)); c['total_time'] = 1.0
    c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [4, 5], 'speed_up': 0.3, 'amplitude': 0.2}
    out.append(('state_transfer_allreg', c))
    out.append(('m1_single_vector', cases.case_c2(n=8, k=1, steps=9, m=1, taylor=(7, 1), seed=5)))
    out.append(('one_step', cases.case_c2(n=5, k=2, steps=1, m=2, taylor=(6, 0), seed=6)))
    out.append(('n3_T4_s0', cases.case_c2(n=3, k=1, steps=17, m=2, taylor=(4, 0), seed=9)))
    out.append(('n7_k8', cases.case_c2(n=7, k=8, steps=33, m=7, taylor=(5, 1), seed=17)))
    c = cases.case_c2(n=9, k=4, steps=70, m=4, taylor=(7, 2), seed=19)
    c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [3.0, 2.0, 1.0], 'states_forbidden_list': [8, 5, 2]}
    out.append(('two_qutrits_forbidden', c))
    out.append(('n10', cases.case_c2(n=10, k=2, steps=21, m=5, taylor=(6, 2), seed=20)))
    c = cases.case_c2(n=11, k=3, steps=19, m=11, taylor=(5, 2), seed=21); c['reg_coeffs'] = {'amplitude': 0.4}
    out.append(('n11_amplitude', c))
    c = cases.case_c2(n=12, k=3, steps=19, m=12, taylor=(5, 2), seed=22); c['reg_coeffs'] = {'speed_up': 0.4}
    out.append(('n12_speed_up', c))
    out.append(('T1_no_products', cases.case_c2(n=6, k=2, steps=11, m=3, taylor=(1, 0), seed=23)))
    # all seven regularisers at once; the bandpass DFT needs the whole pulse in ONE workgroup
    c = cases.case_c2(n=4, k=2, steps=12, m=3, taylor=(6, 1), seed=2); c['total_time'] = 2.0
    c['reg_coeffs'] = dict(PULSE_REG, forbidden_coeff_list=[3.0, 2.0], states_forbidden_list=[3, 2], speed_up=0.7, bandpass=0.4, band=[0.5, 2.0])
    out.append(('unitary_allreg_bandpass', c))
    c = cases.case_c2(n=3, k=3, steps=101, m=3, taylor=(5, 1), seed=29); c['total_time'] = 10.0; c['reg_coeffs'] = {'bandpass': 0.2, 'band': [0.3, 3.0], 'dwdt': 0.05}
    out.append(('n3_bandpass_101_slices', c))
    return out


@pytest.mark.parametrize('groups', [0, 1, 2, 5], ids=['auto_groups', 'one_workgroup', 'two_workgroups', 'five_workgroups'])
@pytest.mark.parametrize('name,c', small_cases(), ids=[n for n, _ in small_cases()])
def test_small_path_eval_parity(name, c, groups):
    sp = oracle_system(c)
    rng = np.random.default_rng(123)
    bases = [sp.base0, 2.5 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.3, 3 * sp.base0]
    from quantum_optimal_control.core import hip_engine
    try:
        eng = make_engine(sp, n_seeds=len(bases), path=SMALL, chunks=groups)
    except hip_engine.QocError as err:
        if (groups in (1, 2) or (groups == 5 and sp.n > 10) or (groups > 1 and 'bandpass' in sp.reg_coeffs)) and 'a pulse that fits' in str(err):
            pytest.skip('the pulse needs more workgroups than pinned (or, n > 10: the trees of this many workgroups more LDS than there is)')
        raise
    assert eng.path == SMALL and eng.plan['path'] == 'small'
    if groups:
        assert int(eng.plan['workgroups']) == groups
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    check_eval(eng, sp, bases)                 # a second evaluation of the same engine (exchange epochs, LDS state) gives the same
    eng.close()


@pytest.mark.parametrize('rows,name,groups', [(16, 'c1', 0), (32, 'c1', 0), (16, 'small_auto', 3), (32, 'small_auto', 0), (8, 'big_auto', 0), (8, 'big_auto', 12), (16, 'big_auto', 0),
                                              (16, 'dressed', 2), (32, 'guess', 0)])
def test_small_path_rows_per_workgroup(rows, name, groups):
    """qoc_config.variant pins the rows of 16 lanes per workgroup (8 / 16 / 32: the instances AUTO chooses among by its cost model); 8 rows with 20 workgroups:
    8 rows: the instances of n > 10."""
    sp = oracle_system(cases.ALL_CASES[name]())
    rng = np.random.default_rng(7)
    bases = [sp.base0, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) - 0.1]
    eng = make_engine(sp, n_seeds=len(bases), path=SMALL, chunks=groups, variant=rows)
    assert eng.plan['path'] == 'small' and int(eng.plan['rows']) == rows
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


@pytest.mark.parametrize('groups', [1, 3])
@pytest.mark.parametrize('name', ['c1', 'dressed', 'state_small'])
def test_small_path_adam_loop(name, groups):
    """The loop of run_session.start_adam_optimizer (run_session.py:47-69) INSIDE the launch: iteration counting, learning-rate schedule,
    TF1 Adam, per-control-set stop rule; bursts of 7 iterations per launch."""
    sp = oracle_system(cases.ALL_CASES[name]())
    conv = dict(rate=0.05, max_iterations=40, learning_rate_decay=100, conv_target=1e-12, min_grad=1e-25)
    ref = go.run_adam(sp, conv)
    eng = make_engine(sp, n_seeds=1, path=SMALL, chunks=groups)
    eng.set_base(sp.base0[None])
    its = eng.run_adam(eng.adam_params(poll_every=7, **conv))
    assert its[0] == ref['iterations'] == 40
    np.testing.assert_allclose(eng.get_base()[0], ref['base'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(eng.get_uks()[0], ref['uks'], rtol=0, atol=1e-10)
    if not sp.state_transfer:
        np.testing.assert_allclose(eng.get_final_unitary()[0], ref['U_final'], rtol=0, atol=1e-10)
    s = eng.scalars()
    assert abs(s['loss'][0] - ref['loss']) < 1e-10 and abs(s['reg_loss'][0] - ref['reg_loss']) < 1e-10
    assert abs(s['unitary_scale'][0] - ref['unitary_scale']) < 1e-10
    eng.close()
    # conv_target stop: per control set, independently, in the middle of a burst
    conv2 = dict(rate=0.05, max_iterations=500, learning_rate_decay=100, conv_target=0.5 if name == 'c1' else 0.9, min_grad=1e-25)
    base_b = 0.1 * np.ones_like(sp.base0)
    ref_a, ref_b = go.run_adam(sp, conv2), go.run_adam(sp, conv2, base=base_b)
    eng = make_engine(sp, n_seeds=2, path=SMALL, chunks=groups)
    eng.set_base(np.stack([sp.base0, base_b]))
    its = eng.run_adam(eng.adam_params(poll_every=5, **conv2))
    assert list(its) == [ref_a['iterations'], ref_b['iterations']]
    np.testing.assert_allclose(eng.get_base()[0], ref_a['base'], atol=1e-10)
    np.testing.assert_allclose(eng.get_base()[1], ref_b['base'], atol=1e-10)
    s = eng.scalars()
    assert list(s['done']) == [1, 1]
    assert abs(s['loss'][0] - ref_a['loss']) < 1e-10 and abs(s['loss'][1] - ref_b['loss']) < 1e-10
    eng.close()


def test_small_path_explicit_step_and_iterate():
    sp = oracle_system(cases.case_small_auto())
    eng = make_engine(sp, path=SMALL)
    eng.set_base(sp.base0[None])
    opt = go.Adam(sp.base0.shape)
    base = sp.base0.copy()
    for lr in (0.01, 0.02, 0.005):
        g = go.evaluate(sp, base)['grad']
        base = opt.step(base, g, lr)
        eng.evaluate()
        eng.adam_step(lr)
    np.testing.assert_allclose(eng.get_base()[0], base, atol=1e-13)
    eng.close()
    # qoc_iterate(iters): exactly `iters` loop iterations in one launch == iters single-iteration launches
    conv = dict(rate=0.03, max_iterations=1000, learning_rate_decay=50, conv_target=1e-14, min_grad=1e-30)
    a, b = make_engine(sp, path=SMALL), make_engine(sp, path=SMALL)
    for e in (a, b):
        e.set_base(sp.base0[None])
    a.iterate(a.adam_params(**conv), 12); a.sync()
    for _ in range(12):
        b.iterate(b.adam_params(**conv), 1)
    b.sync()
    np.testing.assert_allclose(a.get_base(), b.get_base(), rtol=0, atol=1e-13)      # (the learning rate and beta^t advance by running products inside a launch)
    assert a.scalars()['iterations'][0] == 12 == b.scalars()['iterations'][0]
    ref = go.run_adam(sp, dict(conv, max_iterations=12))
    np.testing.assert_allclose(a.get_base()[0], ref['base'], atol=1e-11)
    # uks the last evaluation ran on (run_session.py:75-91 logs them beside its loss): the 12th evaluation's, one Adam step behind the variable
    assert not np.allclose(a.get_uks(evaluated=True), a.get_uks())
    a.close(); b.close()


@pytest.mark.parametrize('n,steps,reg', [(6, 500, None), (4, 500, None), (9, 300, 'forbidden'), (3, 190, 'speed_up')])
def test_small_path_long_pulse_many_workgroups(n, steps, reg):
    """A pulse over 10 .. 32 workgroups (AUTO's own choice), bursts of iterations inside one launch: the deferred stop rule (the partial sums of iteration i
    travel with the exchange of iteration i + 1) and the two alternating exchange buffers -- against the oracle's loop, and twice on the device, bit for bit."""
    c = cases.case_c2(n=n, k=3, steps=steps, m=min(n, 4), taylor=(5, 2), seed=31 + n)
    if reg == 'forbidden':
        c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [n - 1, n - 4]}
    elif reg == 'speed_up':
        c['reg_coeffs'] = {'speed_up': 0.3, 'amplitude': 0.1}
    sp = oracle_system(c)
    conv = dict(rate=0.02, max_iterations=24, learning_rate_decay=100, conv_target=1e-14, min_grad=1e-30)
    ref = go.run_adam(sp, conv)
    out = []
    for _ in range(2):
        eng = make_engine(sp, n_seeds=2, path=SMALL)
        assert int(eng.plan['workgroups']) >= 10
        eng.set_base(np.stack([sp.base0, 0.5 * sp.base0]))
        p = eng.adam_params(poll_every=10, **conv)
        its = eng.run_adam(p)
        assert list(its) == [24, 24]
        out.append((eng.get_base(), eng.scalars(), eng.evaluate()['grad']))
        eng.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][2], out[1][2])
    np.testing.assert_allclose(out[0][0][0], ref['base'], rtol=0, atol=1e-10)
    assert abs(out[0][1]['loss'][0] - ref['loss']) < 1e-10 and abs(out[0][1]['reg_loss'][0] - ref['reg_loss']) < 1e-10
    # a stop in the middle of a burst, found one exchange late: the step taken past it is undone
    hist = go.run_adam(sp, conv, history=True)['history'][:, 0]
    first = [i for i in range(2, 20) if hist[i] < hist[:i].min()]
    if not first:
        return
    target = 0.5 * (hist[first[0]] + hist[:first[0]].min())
    conv3 = dict(conv, max_iterations=500, conv_target=target)
    r3 = go.run_adam(sp, conv3)
    assert r3['iterations'] == first[0]
    eng = make_engine(sp, n_seeds=1, path=SMALL)
    eng.set_base(sp.base0[None])
    its = eng.run_adam(eng.adam_params(poll_every=24, **conv3))
    assert its[0] == r3['iterations']
    np.testing.assert_allclose(eng.get_base()[0], r3['base'], rtol=0, atol=1e-10)
    s = eng.scalars()
    assert s['done'][0] == 1 and abs(s['loss'][0] - r3['loss']) < 1e-10 and abs(s['grad_squared'][0] - r3['grad_squared']) < 1e-9 * max(1.0, r3['grad_squared'])
    np.testing.assert_allclose(eng.get_uks(evaluated=True)[0], r3['uks'], rtol=0, atol=1e-10)
    eng.close()
