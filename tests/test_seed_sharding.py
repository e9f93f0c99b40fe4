"""Row (e) multi-GPU: seed block partition + the one all-gather / broadcast, exercised with world_size 2 on gloo."""
import os
import socket
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_partition_covers_every_seed_once():
    from quantum_optimal_control.parallel_seeds import SeedShard
    for total, world in [(512, 8), (10, 4), (3, 8), (1, 1), (7, 2)]:
        seen = []
        for r in range(world):
            sh = SeedShard(total, r, world)
            seen += list(range(sh.first, sh.first + sh.count))
            assert abs(sh.count - total / world) < 1
        assert seen == list(range(total))
    with pytest.raises(ValueError):
        SeedShard(4, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
    import torch.distributed as dist
    from quantum_optimal_control.parallel_seeds import SeedShard, restart_guesses, select_best
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    total = 5                                            # ragged: rank 0 owns 3 seeds, rank 1 owns 2
    sh = SeedShard(total, rank, world)
    guesses = restart_guesses(2, 6, sh.first, sh.count)
    local_fid = np.array([0.1 * (sh.first + i) if (sh.first + i) != 3 else 0.99 for i in range(sh.count)])
    fid = sh.all_gather(local_fid, dist)
    best = select_best(fid)
    win = sh.broadcast_from_owner(best, lambda i: guesses[i], (2, 6), dist)
    q.put((rank, fid.tolist(), best, win.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_and_winner_broadcast_world2_gloo():
    import torch.multiprocessing as mp
    from quantum_optimal_control.parallel_seeds import restart_guesses
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [0.0, 0.1, 0.2, 0.99, 0.4]
    for rank, fid, best, win in res:
        np.testing.assert_allclose(fid, expect)
        assert best == 3
        np.testing.assert_array_equal(np.array(win), restart_guesses(2, 6, 3, 1)[0])   # seed 3 lives on rank 1


def test_restart_guesses_are_reproducible_and_independent_of_sharding():
    from quantum_optimal_control.parallel_seeds import restart_guesses
    whole = restart_guesses(3, 10, 0, 6)
    parts = np.concatenate([restart_guesses(3, 10, 0, 4), restart_guesses(3, 10, 4, 2)])
    np.testing.assert_array_equal(whole, parts)
    assert abs(np.std(whole) - 1 / np.sqrt(10)) < 0.1


@pytest.mark.gpu
def test_grape_sharded_two_ranks_one_gpu():
    """GrapeSharded: 6 restarts over 2 ranks (gloo rendezvous, both on GPU 0) = the single-process Grape(restarts=6) result on
    every rank (all-gather of the rank-best losses + broadcast of the winner)."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533',
                        os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sharded_script.py')],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'OK sharded rank 0' in r.stdout and 'OK sharded rank 1' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_bench_two_ranks_prints_one_aggregate_line():
    """The driver's multi-GPU invocation of bench.py (torch.distributed.run, one rank per GPU) with the test hook that puts both
    ranks on GPU 0 over gloo: rank 0 prints exactly one JSON line whose value aggregates the seeds of both ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', QOC_BENCH_BACKEND='gloo', QOC_BENCH_SAME_DEVICE='1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29547', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--seeds-per-gpu', '8', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['total_seeds'] == 16 and j['steps'] == 3 and j['value'] > 0 and j['scaling'] == 'weak'


def _rdzv_worker(rank, world, key, directory, q):
    sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
    from quantum_optimal_control.parallel_seeds import rendezvous
    calls = []

    def payload():
        calls.append(1)
        return bytes(range(128))
    got = rendezvous(rank, world, payload, key=key, directory=directory, timeout=60)
    q.put((rank, got, len(calls)))


def test_file_rendezvous_world2_hands_rank0_payload_to_every_rank(tmp_path):
    """The RCCL id travels from rank 0 to the other ranks through a file on the node (no torch in the product path)."""
    import multiprocessing as mp
    from quantum_optimal_control.parallel_seeds import rendezvous_cleanup
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 2, 'unit_test', str(tmp_path), q)) for r in (1, 0)]   # reader first
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == (0, bytes(range(128)), 1) and res[1] == (1, bytes(range(128)), 0)    # only rank 0 evaluates the payload
    rendezvous_cleanup(0, 2, key='unit_test', directory=str(tmp_path))
    assert not os.listdir(str(tmp_path))


def test_grape_sharded_refuses_more_ranks_than_restarts_on_every_rank():
    """the verdict must be collective -- rank 0 (which would own a restart) raises too instead of entering a
    collective the empty ranks never join."""
    from quantum_optimal_control.main_grape.grape import GrapeSharded

    class FakeComm(object):
        world, rank, device = 8, 0, 0
    with pytest.raises(ValueError, match='more ranks'):
        GrapeSharded(None, None, None, None, 1.0, 4, [0], restarts=3, comm=FakeComm())


@pytest.mark.gpu
def test_rccl_communicator_world1_on_the_engine_stream():
    """The RCCL branch behind the C ABI executes: id, ncclCommInitRank, all-gather of the engine's device-resident losses on the
    engine's stream, max all-reduce, broadcast, barrier (world size 1 is all a one-GPU box can run; RCCL refuses two ranks on
    one device).  Runs in a fresh torch-free interpreter, like bench.py: the RCCL transport refuses a process whose HIP runtime
    is PyTorch's private copy (second half of the test)."""
    import subprocess
    script = os.path.join(ROOT, 'tests', 'rccl_world1_script.py')
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and 'OK rccl world1' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    r = subprocess.run([sys.executable, script, 'torch-first'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and 'OK refused' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def _file_comm_worker(rank, world, key, directory, out):
    os.environ['QOC_RDZV_DIR'] = directory
    os.environ['QOC_TRANSPORT'] = 'file'
    from quantum_optimal_control import parallel_seeds
    comm = parallel_seeds.open_comm(rank=rank, world=world, device=0, key=key)
    assert isinstance(comm, parallel_seeds.FileComm) and comm.library.startswith('files')
    shard = parallel_seeds.SeedShard(5, rank, world)
    local = 10.0 * rank + np.arange(shard.count)
    res = dict(gather=shard.all_gather(local, comm=comm),
               vmax=comm.all_reduce_max([float(rank), -float(rank)]),
               bcast=shard.broadcast_from_owner(4, lambda i: np.full((2, 3), 100.0 + i), (2, 3), comm=comm))
    for _ in range(5):
        comm.barrier()
    comm.close()
    out.put((rank, res))


def _failing_init_worker(rank, world, key, directory, out, bad=1):
    os.environ['QOC_RDZV_DIR'] = directory
    os.environ.pop('QOC_TRANSPORT', None)
    from quantum_optimal_control import parallel_seeds
    from quantum_optimal_control.core import hip_engine

    class FakeComm(object):                                  # stands for a communicator whose ncclCommInitRank fails on rank 1 only
        closed = False
        library = 'fake rccl'

        def __init__(self, uid, world_, rank_, device):
            assert uid == bytes(range(7, 7 + hip_engine.COMM_ID_BYTES))      # the id rank 0 made reached this rank unchanged
            if rank_ == bad:
                raise hip_engine.QocError('ncclCommInitRank: unhandled system error')

        def close(self):
            FakeComm.closed = True

        def barrier(self):
            pass

        def all_gather(self, values):
            return np.tile(np.asarray(values, dtype=np.float64), (world, 1))

    def probe(device):                                       # bad <= -2: rank -bad - 2 fails BEFORE the collective (no librccl, bad device index)
        if bad <= -2 and rank == -bad - 2:
            raise hip_engine.QocError('qoc_comm_probe: librccl not loadable')

    hip_engine.comm_probe = probe
    hip_engine.comm_unique_id = lambda: bytes(range(7, 7 + hip_engine.COMM_ID_BYTES))
    hip_engine.QocComm = FakeComm
    comm = parallel_seeds.open_comm(rank=rank, world=world, device=0, key=key)
    res = dict(kind=type(comm).__name__, library=comm.library, closed=FakeComm.closed, reason=comm.fallback_reason,
               gather=comm.all_gather([float(rank)]).reshape(-1))
    comm.close()
    out.put((rank, res))


def test_rccl_init_failure_on_one_rank_sends_every_rank_to_the_file_transport(tmp_path):
    """open_comm: rank 0 can hand out an RCCL id, but the communicator cannot be initialised on rank 1 -- the ranks agree through
    files, every one of them ends up on the file transport (the healthy ranks drop their RCCL communicator), and the job goes on."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_failing_init_worker, args=(r, world, 'fail_%d' % os.getpid(), str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r]['kind'] == 'FileComm' and 'rank(s) [1]' in got[r]['library'], got[r]
        assert got[r]['closed'] == (r != 1)
        np.testing.assert_array_equal(got[r]['gather'], [0.0, 1.0, 2.0])
    assert os.listdir(str(tmp_path)) == []
    # and when every rank initialises: the RCCL communicator is what open_comm returns, the agreement files are gone
    procs = [ctx.Process(target=_failing_init_worker, args=(r, world, 'fine_%d' % os.getpid(), str(tmp_path), out, -1)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(got[r]['kind'] == 'FakeComm' and got[r]['reason'] is None for r in range(world))
    assert os.listdir(str(tmp_path)) == []
    # a rank that fails BEFORE ncclCommInitRank (librccl not loadable, bad device) must not leave the others waiting
    # inside it: the local checks are agreed on first, nobody creates a communicator, everybody takes the file transport
    procs = [ctx.Process(target=_failing_init_worker, args=(r, world, 'early_%d' % os.getpid(), str(tmp_path), out, -4)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r]['kind'] == 'FileComm' and 'rank(s) [2]' in got[r]['library'] and got[r]['reason'], got[r]
        assert got[r]['closed'] is False
        np.testing.assert_array_equal(got[r]['gather'], [0.0, 1.0, 2.0])
    assert os.listdir(str(tmp_path)) == []


def test_file_transport_world3_matches_the_collective_contract(tmp_path):
    """The fallback transport of parallel_seeds.open_comm (no RCCL available / QOC_TRANSPORT=file): all-gather in global seed
    order, max-reduce, broadcast from the owner, barriers; nothing left behind in the rendezvous directory."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_file_comm_worker, args=(r, world, 'test_%d' % os.getpid(), str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.array([0.0, 1.0, 10.0, 11.0, 20.0])             # counts 2, 2, 1
    for r in range(world):
        np.testing.assert_array_equal(got[r]['gather'], expect)
        np.testing.assert_array_equal(got[r]['vmax'], [2.0, 0.0])
        np.testing.assert_array_equal(got[r]['bcast'], np.full((2, 3), 100.0))      # seed 4 is local index 0 of rank 2
    assert os.listdir(str(tmp_path)) == []


@pytest.mark.gpu
def test_bench_gpus2_file_transport_fallback():
    """`python bench.py --gpus 2` when RCCL cannot be used (forced here with QOC_TRANSPORT=file; both ranks on GPU 0): the bench
    line still carries n_gpus 2 and says which transport gathered the fidelities."""
    import json
    import subprocess
    env = dict(os.environ, QOC_TRANSPORT='file', QOC_BENCH_SAME_DEVICE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'QOC_BENCH_BACKEND'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--seeds-per-gpu', '8', '--no-cpu-baseline', '--no-single'], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert j['n_gpus'] == 2 and j['config']['fidelities_gathered'] == 16 and j['config']['transport'].startswith('files')


@pytest.mark.gpu
def test_bench_gpus2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) must start two ranks itself and print n_gpus 2.
    On the one-GPU box both ranks share GPU 0 through the gloo test hook."""
    import json
    import subprocess
    env = dict(os.environ, QOC_BENCH_BACKEND='gloo', QOC_BENCH_SAME_DEVICE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--seeds-per-gpu', '8', '--no-cpu-baseline', '--no-single'], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['ranks_seen'] == 2 and j['config']['fidelities_gathered'] == 16


@pytest.mark.gpu
def test_bench_gpus8_rank_bookkeeping_on_one_gpu():
    """VERDICT r2 #6c: the 8-rank bench line before an 8-GPU node exists -- eight self-started ranks share GPU 0 (same-device hook)
    and exchange through the file transport: 8 ranks seen, 8 x 64 = 512 fidelities gathered in global seed order (bench.py asserts
    the local block), and the line says in so many words that RCCL was NOT what gathered them."""
    import json
    import subprocess
    env = dict(os.environ, QOC_TRANSPORT='file', QOC_BENCH_SAME_DEVICE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'QOC_BENCH_BACKEND'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-single'], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    cfg = j['config']
    assert j['n_gpus'] == 8 and cfg['ranks_seen'] == 8 and cfg['fidelities_gathered'] == 512 and cfg['total_seeds'] == 512
    assert cfg['transport'].startswith('files') and cfg['transport_fallback'] is True and cfg['rccl_error'] == 'QOC_TRANSPORT=file'
    assert cfg['chunks'] == 16 and cfg['path'] == 2                      # every rank runs the single-GPU bench configuration


# ---- round 4: 8-GPU readiness -------------------------------------------------------------------------------
def _require_rccl_worker(rank, world, key, directory, out):
    """open_comm(require_rccl=True) with a communicator whose ncclCommInitRank fails on rank 1: EVERY rank must raise (exit code 3 here, as
    bench.py does), not drop to the file transport."""
    os.environ['QOC_RDZV_DIR'] = directory
    os.environ.pop('QOC_TRANSPORT', None)
    from quantum_optimal_control import parallel_seeds
    from quantum_optimal_control.core import hip_engine

    class FakeComm(object):
        library = 'fake rccl'

        def __init__(self, uid, world_, rank_, device):
            if rank_ == 1:
                raise hip_engine.QocError('ncclCommInitRank: unhandled system error')

        def close(self):
            pass

    hip_engine.comm_probe = lambda device: None
    hip_engine.comm_unique_id = lambda: bytes(range(hip_engine.COMM_ID_BYTES))
    hip_engine.QocComm = FakeComm
    try:
        parallel_seeds.open_comm(rank=rank, world=world, device=0, key=key, require_rccl=True)
    except parallel_seeds.RcclRequired as exc:
        out.put((rank, str(exc)))
        sys.exit(3)
    out.put((rank, 'no exception'))


def test_require_rccl_turns_a_forced_rccl_failure_into_a_nonzero_exit_on_every_rank(tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_require_rccl_worker, args=(r, 2, 'req_%d' % os.getpid(), str(tmp_path), out)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 3
    for r in range(2):
        assert 'RCCL is required' in got[r] and 'rank(s) [1]' in got[r], got[r]
    assert os.listdir(str(tmp_path)) == []                   # and the agreement files are gone


def test_local_rank_to_device_mapping():
    """torch.distributed.run exports no HIP_VISIBLE_DEVICES: every rank sees all 8 GPUs and takes device = LOCAL_RANK; a launcher that narrowed
    each rank's view to one device gives device 0; more ranks than devices go round-robin."""
    from quantum_optimal_control.parallel_seeds import device_for_rank
    assert [device_for_rank(r, 8) for r in range(8)] == list(range(8))
    assert [device_for_rank(r, 1) for r in range(8)] == [0] * 8
    assert [device_for_rank(r, 4) for r in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    with pytest.raises(RuntimeError, match='no HIP device'):
        device_for_rank(0, 0)


def test_file_transport_refuses_a_planted_symlink_and_keys_differ_between_launches(tmp_path):
    from quantum_optimal_control import parallel_seeds
    victim = tmp_path / 'victim'
    victim.mkdir(mode=0o700)
    (victim / 'precious').write_text('x')
    link = tmp_path / ('qoc_fc_%d_planted' % os.getuid())
    os.symlink(str(victim), str(link))
    with pytest.raises(PermissionError, match='not a private directory'):
        parallel_seeds.FileComm(0, 2, 'planted', directory=str(tmp_path))
    assert (victim / 'precious').exists()
    # the default key carries the launcher's pid AND its start time: a recycled pid of a crashed launch does not collide
    port, pid, start = parallel_seeds.launch_key().split('_')
    assert int(pid) == os.getppid() and int(start) > 0
    # close() removes only what the communicator wrote
    comm = parallel_seeds.FileComm(0, 1, 'solo', directory=str(tmp_path))
    foreign = os.path.join(comm._dir, 'somebody_elses_file')
    open(foreign, 'w').close()
    comm.barrier()
    comm.close()
    assert os.path.exists(foreign)


def test_grape_sharded_plans_for_the_largest_shard_of_its_launch(monkeypatch):
    """the default plan of GrapeSharded is ceil(restarts / ranks) of the launch at hand -- never smaller than the batch an
    engine holds -- not restarts / GPUs of the node."""
    from quantum_optimal_control.main_grape import grape as G
    seen = {}

    def fake_grape(*a, **kw):
        seen.update(kw)
        return None
    monkeypatch.setattr(G, 'Grape', fake_grape)

    class FakeComm(object):
        world, rank, device = 2, 1, 1

        def all_gather(self, values):                      # an interrupted run on every rank: GrapeSharded returns None after the gather
            return np.full((2, len(values)), np.inf)
    assert G.GrapeSharded(None, None, None, None, 1.0, 4, [0], restarts=65, comm=FakeComm()) is None
    assert seen['plan_seeds'] == 33 and seen['restarts'] == 32 and seen['_first_seed'] == 33 and seen['_device'] == 1
    G.GrapeSharded(None, None, None, None, 1.0, 4, [0], restarts=64)
    assert seen['plan_seeds'] == 64 and seen['restarts'] == 64
    G.GrapeSharded(None, None, None, None, 1.0, 4, [0], restarts=64, plan_seeds=8)
    assert seen['plan_seeds'] == 8


@pytest.mark.gpu
def test_bench_driver_form_fails_when_rccl_cannot_start():
    """The driver's N > 1 form (torch.distributed.run) requires RCCL by default: with the RCCL library made unloadable both ranks exit non-zero
    and no JSON line is printed; --allow-file-transport brings the flagged fallback line back."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', QOC_RCCL_LIBRARY='/nonexistent/librccl.so', QOC_BENCH_SAME_DEVICE='1')
    for k in ('QOC_TRANSPORT', 'QOC_BENCH_BACKEND', 'QOC_REQUIRE_RCCL'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29561',
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--seeds-per-gpu', '8', '--no-cpu-baseline', '--no-single']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode != 0, r.stdout[-1500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert 'RCCL is required' in r.stderr
    r = subprocess.run(cmd + ['--allow-file-transport'], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    cfg = j['config']
    assert cfg['transport_fallback'] is True and cfg['ranks_seen'] == 2 and len(cfg['per_rank']) == 2
    assert all(q['device'] == 0 and q['ms_total'] > 0 and q['device_name'] for q in cfg['per_rank'])


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_multi_gpu_selftest_plumbing_world2_gloo_without_a_gpu():
    """tools/multi_gpu_selftest.py (the first-contact check of a multi-GPU node) under its CPU hook: two ranks, gloo, no engine -- partition, all-gather and
    the comparison of every gathered row run; a mismatch would be a non-zero exit of the launcher."""
    import subprocess
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                        str(_free_port()), os.path.join(ROOT, 'tools', 'multi_gpu_selftest.py'), '--transport', 'gloo', '--no-gpu'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('ALL STAGES OK') == 2, r.stdout[-2000:]


@pytest.mark.gpu
def test_multi_gpu_selftest_world1_and_two_ranks_on_one_gpu():
    """The same script on the GPU box: a world of one (every stage, the RCCL collectives of the time-sharded engine on a one-rank communicator) and two
    ranks sharing GPU 0 over gloo (stages i and ii: every gathered loss against the rank's own evaluation, GrapeSharded against Grape bit for bit)."""
    import subprocess
    script = os.path.join(ROOT, 'tools', 'multi_gpu_selftest.py')
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for tag in ('(i) ', '(ii) ', '(iii) time-sharded', 'peer access', 'ALL STAGES OK'):
        assert tag in r.stdout, (tag, r.stdout[-3000:])
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                        str(_free_port()), script, '--transport', 'gloo', '--same-device'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('ALL STAGES OK') == 2 and r.stdout.count('(ii) GrapeSharded') == 2, r.stdout[-3000:]


@pytest.mark.gpu
def test_bench_line_n1_carries_secondary_configs_and_a_measured_traffic_figure():
    """The driver's N = 1 form of bench.py: exactly ONE line on stdout (the pre-processing's prints go to stderr), a JSON object with the contract's
    keys, `secondary` = BASELINE configs 3 and 5 through the same C ABI without an `error` entry, and roofline.traffic either measured in the run
    (rocprofv3 on PATH) or taken from the committed profile -- the line says which."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--prewarm', '5', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    out = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in out, key
    assert out['n_gpus'] == 1 and out['steps'] == 3 and out['dtype'] == 'f64' and out['vs_baseline'] is None
    sec = out['secondary']
    for name, sets in (('c3_single_trajectory', 1), ('c3_x64', 64), ('c3_x256', 256), ('c5_single_trajectory', 1)):
        assert 'error' not in sec[name], sec[name]
        assert sec[name]['control_sets'] == sets and sec[name]['ms_per_iteration'] > 0
    assert sec['c3_x64']['plan'].get('taylor_chain') == 'packed' and sec['c3_single_trajectory']['plan'].get('route') == 'propagator'
    roof = out['roofline']
    assert roof['bound'] == 'mfma' and 0.3 < roof['frac'] < 1.0
    assert roof['traffic'] is None or roof['traffic'] > 1e8
    assert 'measured in this run' in roof['traffic_source'] or 'committed profile' in roof['traffic_source']


# ---- round 6: first-contact hardening of the N > 1 start-up ("Next round" 7) ----------------------------------------------------------------

def test_launch_key_prefers_an_explicit_key_then_the_elastic_run_id(monkeypatch):
    from quantum_optimal_control import parallel_seeds as ps
    for k in ('QOC_RDZV_KEY', 'TORCHELASTIC_RUN_ID'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('MASTER_PORT', '29512')
    by_parent = ps.launch_key()
    assert by_parent.startswith('29512_%d_' % os.getppid())
    monkeypatch.setenv('TORCHELASTIC_RUN_ID', 'none')                # torch.distributed.run's default for a static rendezvous: not a key
    assert ps.launch_key() == by_parent
    monkeypatch.setenv('TORCHELASTIC_RUN_ID', 'job 4711/a')
    assert ps.launch_key() == 'ejob_4711_a_29512'
    monkeypatch.setenv('QOC_RDZV_KEY', 'slurm-99.0')
    assert ps.launch_key() == 'kslurm-99.0'


def test_rendezvous_gives_up_soon_and_names_file_and_key(monkeypatch, tmp_path):
    from quantum_optimal_control import parallel_seeds as ps
    monkeypatch.setenv('QOC_RDZV_DIR', str(tmp_path))
    monkeypatch.setenv('QOC_RDZV_TIMEOUT', '1')
    monkeypatch.setenv('QOC_RDZV_KEY', 'nobody-writes-this')
    assert ps.rendezvous_timeout() == 1.0
    t0 = time.time()
    with pytest.raises(TimeoutError) as err:
        ps.rendezvous(1, 2, lambda: b'x')
    assert time.time() - t0 < 10
    msg = str(err.value)
    assert 'qoc_rdzv_' in msg and 'knobody-writes-this' in msg and 'QOC_RDZV_KEY' in msg and str(tmp_path) in msg
    fc = ps.FileComm(1, 2, 'knobody-writes-this_c0', timeout=1.0)
    with pytest.raises(TimeoutError) as err:
        fc.all_gather([1.0])
    assert 'rank 0' in str(err.value) and 'QOC_RDZV_KEY' in str(err.value)


def test_ranks_without_a_common_parent_meet_through_qoc_rdzv_key(tmp_path):
    """A launcher that wraps every rank in its own shell (srun, a job script per GPU): os.getppid() differs between the ranks, QOC_RDZV_KEY is what they share.
    World 2 on the file transport (no GPU involved): both ranks gather each other's rows; without the key they would wait for files nobody writes."""
    import subprocess
    code = ("import os, sys; sys.path[:0] = [%r, %r]\n"
            "import numpy as np\n"
            "from quantum_optimal_control import parallel_seeds as ps\n"
            "c = ps.open_comm(call_index=0)\n"
            "rows = c.all_gather([float(c.rank) + 0.5, float(os.getppid())])\n"
            "assert rows.shape == (2, 2) and rows[0, 0] == 0.5 and rows[1, 0] == 1.5 and rows[0, 1] != rows[1, 1], rows\n"
            "c.close(); print('RANK_OK', c.rank)\n") % (ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'))
    script = tmp_path / 'rank.py'
    script.write_text(code)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29571', QOC_TRANSPORT='file',
                   QOC_RDZV_DIR=str(tmp_path), QOC_RDZV_KEY='job-%d' % os.getpid(), QOC_RDZV_TIMEOUT='30')
        env.pop('TORCHELASTIC_RUN_ID', None)
        procs.append(subprocess.Popen(['bash', '-c', '%s %s; exit $?' % (sys.executable, script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and 'RANK_OK %d' % r in so, (so[-800:], se[-1500:])
