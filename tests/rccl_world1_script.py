"""Body of tests/test_seed_sharding.py::test_rccl_communicator_world1_on_the_engine_stream (fresh interpreter, no pytest, no
torch unless argv[1] == 'torch-first')."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
if len(sys.argv) > 1 and sys.argv[1] == 'torch-first':
    import torch  # noqa: F401  (binds libamdhip64.so.7 to torch's private copy before libqoc_hip.so is loaded)
from quantum_optimal_control.core import hip_engine  # noqa: E402
from quantum_optimal_control.parallel_seeds import SeedShard, open_comm  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.helpers import oracle_system  # noqa: E402

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'torch-first':
        try:
            hip_engine.comm_unique_id()
        except hip_engine.QocError as exc:
            assert 'private HIP runtime' in str(exc), exc
            print('OK refused')
            sys.exit(0)
        # torch was imported but our library still got the system runtime (load order differs): the transport must work then
        print('OK refused (not applicable: system HIP runtime in use)')
        sys.exit(0)
    assert open_comm(rank=0, world=1, device=0) is None            # a single process needs no communicator
    sp = oracle_system(cases.case_c2(n=8, k=2, steps=12, m=4, taylor=(4, 1), seed=2))
    B = 5
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
                               reg_coeffs={}, n_seeds=B)
    eng.set_base(np.random.default_rng(0).normal(0, 0.3, (B, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 6, conv_target=-1.0, min_grad=-1.0)
    comm = hip_engine.QocComm(hip_engine.comm_unique_id(), 1, 0, 0)
    assert 'rccl' in comm.library, comm.library
    eng.iterate(p, 3)                                        # no sync: the gather is ordered behind these on the stream
    rows = comm.all_gather_scalar(eng, hip_engine.SCALAR_LOSS, B + 2)
    assert rows.shape == (1, B + 2)
    np.testing.assert_array_equal(rows[0, :B], eng.scalars()['loss'])
    assert np.all(rows[0, B:] == 0.0)
    sh = SeedShard(B, 0, 1)
    np.testing.assert_array_equal(comm.all_gather([1.5, -2.0]), [[1.5, -2.0]])
    np.testing.assert_array_equal(comm.all_reduce_max([3.0, -1.0]), [3.0, -1.0])
    np.testing.assert_array_equal(comm.broadcast(np.arange(6.0).reshape(2, 3), 0), np.arange(6.0).reshape(2, 3))
    comm.barrier()
    np.testing.assert_array_equal(sh.all_gather_engine_scalar(eng, hip_engine.SCALAR_LOSS, None), eng.scalars()['loss'])
    try:
        comm.all_gather_scalar(eng, 7, B)
        raise SystemExit('unknown scalar index was accepted')
    except hip_engine.QocError:
        pass
    eng.close()
    # time-axis sharding with its two RCCL calls on the engine's stream (world 1: the all-gather and the all-reduce are identities, but they run):
    # a "one rank of one" engine against the unsharded one, three iterations of the device loop
    spL = oracle_system(cases.case_c2(n=100, k=2, steps=32, m=4, taylor=(5, 2), seed=9))
    res = []
    for kw in (dict(path=4), dict(time_shards=1, time_rank=0, time_comm=comm)):
        e2 = hip_engine.HipEngine(spL.Hs, spL.U0, spL.V, spL.W, spL.maxA, spL.dt, spL.total_time, spL.steps, spL.exp_terms, spL.scaling, reg_coeffs={}, n_seeds=1, **kw)
        e2.set_base(spL.base0[None])
        e2.iterate(e2.adam_params(rate=0.02, max_iterations=10 ** 6, conv_target=-1.0, min_grad=-1.0), 3)
        if 'time_comm' in kw:                                # the communicator must outlive the engine that enqueues collectives on it: the C ABI refuses
            lib = hip_engine.load_library()                  # (at the C ABI: QocComm.close() itself closes the engines that hold it first)
            if lib.qoc_comm_destroy(comm._h) == 0:
                raise SystemExit('qoc_comm_destroy went through while a time-sharded engine still held the communicator')
            assert 'still use this communicator' in lib.qoc_last_error().decode(), lib.qoc_last_error()
        res.append((e2.get_base()[0].copy(), e2.scalars()['loss'][0], e2.plan, e2.get_inter_vecs()[0].copy()))
        e2.close()
    assert res[1][2].get('time_shards') == '1' and res[1][2].get('time_rank') == '0', res[1][2]
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=0, atol=1e-12)
    assert abs(res[1][1] - res[0][1]) < 1e-12
    np.testing.assert_allclose(res[1][3], res[0][3], rtol=0, atol=1e-12)      # inter_vecs: gathered over the (one) rank
    # and through the Python entry point: GrapeTimeSharded(comm=...) = Grape(...) for a world of one
    import contextlib
    import io
    from quantum_optimal_control.main_grape.grape import Grape, GrapeTimeSharded
    cL = cases.case_c2(n=100, k=2, steps=32, m=4, taylor=(5, 2), seed=9)
    kw = dict(H0=cL['H0'], Hops=cL['Hops'], Hnames=cL['Hnames'], U=cL['U'], total_time=cL['total_time'], steps=cL['steps'],
              states_concerned_list=cL['states_concerned_list'], maxA=cL['maxA'], reg_coeffs={}, Taylor_terms=cL['Taylor_terms'], save=False, show_plots=False,
              convergence={'rate': 0.02, 'update_step': 2, 'max_iterations': 3, 'conv_target': 1e-12, 'learning_rate_decay': 100})
    outs = []
    for fn, extra in ((Grape, {}), (GrapeTimeSharded, {'comm': comm})):
        np.random.seed(cL['np_seed'])
        with contextlib.redirect_stdout(io.StringIO()):
            outs.append(fn(**kw, **extra))
    np.testing.assert_allclose(np.asarray(outs[1][0]), np.asarray(outs[0][0]), rtol=0, atol=1e-11)
    np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=0, atol=1e-11)
    comm.close()
    print('OK rccl world1 via', comm.library)
