#!/usr/bin/env python
"""First-contact self-test of the multi-GPU paths: run it BEFORE trusting a scaling number from a new node.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/multi_gpu_selftest.py
        N GPUs, one rank per GPU, RCCL over xGMI behind the C ABI (the form bench.py runs in)
  python tools/multi_gpu_selftest.py
        one process, one GPU: every stage with a world of one (the RCCL calls still run, on a one-rank communicator)
  ... tools/multi_gpu_selftest.py --transport gloo [--same-device] [--no-gpu]
        test hooks: torch.distributed (gloo, host tensors) instead of RCCL, so that N ranks can share ONE GPU (--same-device) or run
        without any GPU (--no-gpu: only the partition / gather / broadcast plumbing, on synthetic per-seed values)

Per rank it prints the RCCL library in use, its HIP device and whether that device can address every other rank's device directly
(hipDeviceCanAccessPeer), then runs
  (i)   the seed-sharded all-gather of bench.py: every rank optimises its block of the restarts, the per-seed losses are all-gathered from the
        engines' device arrays, and EVERY row is compared with what this rank computes for that seed itself (bit for bit: the kernels are
        planned for the same batch on every rank);
  (ii)  GrapeSharded(restarts=16) against Grape(restarts=16) of a single process, bit for bit;
  (iii) one GrapeTimeSharded iteration sequence at n = 128 (the time axis of ONE trajectory over the ranks: two collectives per iteration,
        csrc/qoc_gemm_ts.h) against the unsharded engine, <= 1e-12   (RCCL only: the file / gloo transports cannot carry it).
Exit code 0 only when every stage that ran agreed on every rank; any mismatch or exception -> non-zero on that rank (and the launcher fails).
"""
import argparse
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC, before the HIP runtime starts
import numpy as np  # noqa: E402

from quantum_optimal_control import parallel_seeds  # noqa: E402
from quantum_optimal_control.helper_functions import synthetic_systems  # noqa: E402
from quantum_optimal_control.parallel_seeds import SeedShard  # noqa: E402


def say(rank, text):
    print('[selftest rank %d] %s' % (rank, text), flush=True)


def engine_inputs(c):
    """(Hs, U0, V, W, dt) of a synthetic_systems.case_c2 recipe (unitary gate on the first m levels), as bench.py builds them."""
    n, m, steps = len(c['H0']), len(c['states_concerned_list']), c['steps']
    dt = c['total_time'] / steps
    Hs = np.stack([-1j * dt * c['H0']] + [-1j * dt * h for h in c['Hops']])
    V = np.eye(n, dtype=complex)[:, :m]
    return Hs, np.eye(n, dtype=complex), V, c['U'] @ V, dt


def stage_gather(rank, world, device, comm, dist, no_gpu):
    """(i) bench.py's exchange on a small problem: 4 seeds per rank."""
    per, k, steps = 4, 2, 24
    total = per * world
    shard = SeedShard(total, rank, world)
    guesses = parallel_seeds.restart_guesses(k, steps, 0, total)             # the restart streams of ALL seeds (any rank can form any of them)
    if no_gpu:
        mine = np.array([np.sum(np.sin(g)) for g in guesses[shard.first:shard.first + shard.count]])
        want = np.array([np.sum(np.sin(g)) for g in guesses])
        got = shard.all_gather(mine, dist=dist, comm=comm)
    else:
        from quantum_optimal_control.core import hip_engine
        c = synthetic_systems.case_c2(n=8, k=k, steps=steps, m=4, taylor=(4, 1), seed=2)
        Hs, U0, V, W, dt = engine_inputs(c)

        def losses(first, count, gather=False):
            eng = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], steps, 4, 1, reg_coeffs={}, n_seeds=count, device=device, plan_seeds=per)
            eng.set_base(guesses[first:first + count])
            eng.iterate(eng.adam_params(rate=0.02, max_iterations=10 ** 6, conv_target=-1.0, min_grad=-1.0), 5)
            if gather:
                out = shard.all_gather_engine_scalar(eng, hip_engine.SCALAR_LOSS, comm)       # device to device on the engine's stream
            else:
                out = eng.scalars()['loss']
            eng.close()
            return out
        if comm is not None and hasattr(comm, '_h'):
            got = losses(shard.first, shard.count, gather=True)
        else:
            got = shard.all_gather(losses(shard.first, shard.count), dist=dist, comm=comm)
        # every block of the partition once more on THIS rank's GPU, with the same batch size and plan: the same kernels, so the same bits
        want = np.concatenate([losses(SeedShard(total, r, world).first, SeedShard(total, r, world).count) for r in range(world)])
    assert got.shape == (total,), got.shape
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, 'gathered losses differ from this rank\'s own evaluation of seeds %s: %s vs %s' % (bad[:4], got[bad[:4]], want[bad[:4]])
    say(rank, '(i) all-gather of %d per-seed losses: every row equals this rank\'s own evaluation of that seed  OK' % total)


def stage_grape_sharded(rank, world, device, comm, dist):
    """(ii) GrapeSharded(restarts=16) == Grape(restarts=16) of one process, bit for bit."""
    from quantum_optimal_control.core import hip_engine  # noqa: F401
    from quantum_optimal_control.main_grape.grape import Grape, GrapeSharded
    restarts = 16
    c = synthetic_systems.case_c2(n=4, k=2, steps=20, m=2, taylor=(4, 1), seed=5)
    kw = dict(H0=c['H0'], Hops=c['Hops'], Hnames=c['Hnames'], U=c['U'], total_time=c['total_time'], steps=c['steps'], states_concerned_list=c['states_concerned_list'],
              maxA=c['maxA'], reg_coeffs={}, Taylor_terms=c['Taylor_terms'], save=False, show_plots=False, method='Adam',
              convergence={'rate': 0.05, 'update_step': 10, 'max_iterations': 30, 'conv_target': 1e-12, 'learning_rate_decay': 100})
    plan = max(SeedShard(restarts, 0, world).counts)
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        a = GrapeSharded(restarts=restarts, dist=dist, comm=comm, device=device, **kw)
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        b = Grape(restarts=restarts, plan_seeds=plan, _device=device, **kw)
    assert a is not None and b is not None
    assert np.array_equal(np.asarray(a[0]), np.asarray(b[0])), 'uks differ by %g' % np.max(np.abs(np.asarray(a[0]) - np.asarray(b[0])))
    assert np.array_equal(np.asarray(a[1]), np.asarray(b[1])), 'U_final differs by %g' % np.max(np.abs(np.asarray(a[1]) - np.asarray(b[1])))
    say(rank, '(ii) GrapeSharded(restarts=%d) over %d rank(s) == Grape(restarts=%d) in one process, bit for bit  OK' % (restarts, world, restarts))


def stage_time_sharded(rank, world, device, comm):
    """(iii) one trajectory of n = 128 cut along the time axis over the ranks against the unsharded engine."""
    from quantum_optimal_control.core import hip_engine
    c = synthetic_systems.case_c2(n=128, k=3, steps=96, m=4, taylor=(5, 2), seed=9)
    Hs, U0, V, W, dt = engine_inputs(c)
    base = np.random.default_rng(3).normal(0, 1 / np.sqrt(c['steps']), (1, 3, c['steps']))
    own = None
    if comm is None:                                                             # world of one: a one-rank RCCL communicator still runs both collectives
        own = comm = hip_engine.QocComm(hip_engine.comm_unique_id(), 1, 0, device)
    res = []
    for kw in (dict(path=hip_engine.PATH_GEMM), dict(time_shards=comm.world, time_rank=comm.rank, time_comm=comm)):
        eng = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], c['steps'], 5, 2, reg_coeffs={}, n_seeds=1, device=device, **kw)
        eng.set_base(base)
        eng.iterate(eng.adam_params(rate=0.02, max_iterations=10 ** 6, conv_target=-1.0, min_grad=-1.0), 3)
        res.append((eng.get_base()[0].copy(), float(eng.scalars()['loss'][0]), dict(eng.plan)))
        eng.close()
    assert res[1][2].get('time_shards') == str(comm.world), res[1][2]
    err = float(np.max(np.abs(res[1][0] - res[0][0])))
    assert err <= 1e-12 and abs(res[1][1] - res[0][1]) <= 1e-12, 'time-sharded run differs: controls %g, loss %g' % (err, abs(res[1][1] - res[0][1]))
    say(rank, '(iii) time-sharded trajectory (n = 128, %d shard(s), 3 Adam iterations) against the unsharded engine: max |d base| = %.1e  OK' % (comm.world, err))
    if own is not None:
        own.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--transport', choices=['rccl', 'gloo'], default='rccl')
    ap.add_argument('--same-device', action='store_true', help='test hook: every rank on HIP device 0 (gloo transport only: RCCL refuses two ranks on one device)')
    ap.add_argument('--no-gpu', action='store_true', help='test hook: no engine at all, only the partition / gather plumbing (gloo transport)')
    args = ap.parse_args()
    rank, local_rank, world = parallel_seeds.launch_env()
    t_start = time.perf_counter()
    comm = dist = None
    device = 0
    if not args.no_gpu:
        from quantum_optimal_control.core import hip_engine
        ndev = hip_engine.device_count()
        device = 0 if args.same_device else parallel_seeds.device_for_rank(local_rank, ndev)
        info = hip_engine.device_info(device)
        say(rank, 'HIP device %d of %d visible: %s, %d CUs, %.0f GB' % (device, ndev, info['name'], info['compute_units'], info['hbm_bytes'] / 2 ** 30))
    if world > 1 and args.transport == 'gloo':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    elif world > 1:
        t0 = time.perf_counter()
        comm = parallel_seeds.open_comm(rank=rank, world=world, device=device, require_rccl=True)      # raises on every rank when RCCL cannot start
        say(rank, 'RCCL communicator up in %.2f s: %s' % (time.perf_counter() - t0, comm.library))
    if not args.no_gpu:
        # who can address whom: the devices of all ranks (gathered), then hipDeviceCanAccessPeer from this rank's device to each of them
        devs = SeedShard(world, rank, world).all_gather(np.array([float(device)]), dist=dist, comm=comm).astype(int) if world > 1 else np.array([device])
        peers = ['%d:%s' % (int(p), 'self' if int(p) == device else ('yes' if hip_engine.device_peer_access(device, int(p)) else 'NO')) for p in devs]
        say(rank, 'peer access from device %d to the ranks\' devices: %s' % (device, ' '.join(peers)))
        if world > 1 and args.transport == 'rccl':
            assert len(set(int(p) for p in devs)) == world, 'two ranks share a HIP device: %s' % devs
    stage_gather(rank, world, device, comm, dist, args.no_gpu)
    if not args.no_gpu:
        stage_grape_sharded(rank, world, device, comm, dist)
        if args.transport == 'rccl':
            stage_time_sharded(rank, world, device, comm)
        else:
            say(rank, '(iii) skipped: the time-sharded engine needs the RCCL transport')
    if comm is not None:
        comm.barrier()
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    say(rank, 'ALL STAGES OK in %.1f s' % (time.perf_counter() - t_start))


if __name__ == '__main__':
    main()
