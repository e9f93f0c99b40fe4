#!/usr/bin/env python
"""bench.py -- GRAPE iterations/s (forward chain + first-order backward + regularisers + Adam) on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from rank 0.
  * workload (BASELINE.json configs[1] / SURVEY.md 8d "C2"): n=32, k=4, time slices=500, m=8, (T,s)=(5,3), fp64
    complex, synthetic Hamiltonians from numpy.random.default_rng(0); SEEDS_PER_GPU independent random-restart control
    sets per GPU (config 4: 512 seeds / 8 GPUs = 64 per GPU) -> weak scaling, no data-path collective; the final
    per-seed fidelities are all-gathered once over RCCL after the last iteration (inside the timed region).
  * a "step" = one GRAPE iteration of every seed on the GPU (all inputs resident in HBM).
  * value = (seeds on all GPUs) * K / wall time, wall = max over ranks, barrier + device sync on both sides.
  * roofline: dominant kernel (k_mfma_expm_chunk4w: matrix exponentials + chunk products) timed with hipEvents on the
    engine's stream in a separate short pass; algorithmic FLOPs per launch from SURVEY.md 8d.
  * cpu_baseline: the compiled C restatement of the oracle (oracle/qoc_oracle.c, OpenMP, one seed per thread, ONE
    evaluation per iteration) timed on a bounded sample on this host -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

N, K_OPS, SLICES, M, TAYLOR = 32, 4, 500, 8, (5, 3)
SEEDS_PER_GPU = 64
FP64_MATRIX_PEAK_TFLOPS = 78.6      # MI355X public fp64 matrix (= vector) peak; MI355X_MICROARCH.md lists no fp64 row


def build_problem():
    from tests.golden import cases
    c = cases.case_c2(n=N, k=K_OPS, steps=SLICES, m=M, taylor=TAYLOR, seed=0)
    H0, Hops, U = c['H0'], c['Hops'], c['U']
    dt = c['total_time'] / SLICES
    Hs = np.stack([-1j * dt * H0] + [-1j * dt * h for h in Hops])
    V = np.eye(N, dtype=complex)[:, :M]
    W = U @ V
    return c, Hs, np.eye(N, dtype=complex), V, W, dt


def seed_bases(first, count):
    return np.stack([np.random.default_rng(1000 + first + i).normal(0, 1 / np.sqrt(SLICES), (K_OPS, SLICES))
                     for i in range(count)])


def cpu_baseline(budget_s=15.0):
    """CPU oracle timed on a bounded sample of the same workload: whole iterations (evaluate + Adam) of one seed per
    thread, all host threads busy (seeds are the parallel axis on the CPU too).  Compiled C port (oracle/qoc_oracle.c)
    when its .so is present, else the NumPy oracle on one thread."""
    from tests.helpers import oracle_system
    from tests.golden import cases
    sp = oracle_system(cases.case_c2(n=N, k=K_OPS, steps=SLICES, m=M, taylor=TAYLOR, seed=0))
    try:
        from oracle import c_port
        threads = c_port.max_threads()
        bases = seed_bases(0, threads)
        t0 = time.perf_counter()
        c_port.iterate(sp, bases, 1, nthreads=threads)                      # calibration + warm-up
        per_it = time.perf_counter() - t0
        iters = int(max(2, min(200, budget_s / max(per_it, 1e-3))))
        t0 = time.perf_counter()
        c_port.iterate(sp, bases, iters, nthreads=threads)
        el = time.perf_counter() - t0
        return {'value': threads * iters / el, 'unit': 'GRAPE iterations/s', 'cores': int(threads), 'kind': 'port',
                'sample': '%d iterations x %d seeds (one seed per thread, OpenMP) of the same C2 workload in %.1f s; '
                          'compiled C restatement oracle/qoc_oracle.c, one evaluation per iteration, fp64 complex; '
                          'host has %d logical CPUs' % (iters, threads, el, os.cpu_count()),
                'per_thread_value': iters / el}
    except OSError:
        pass
    from oracle import grape_oracle as go
    try:
        import threadpoolctl
        ctx = threadpoolctl.threadpool_limits(limits=1)
    except Exception:
        ctx = None
    base = seed_bases(0, 1)[0]
    opt = go.Adam(base.shape)
    go.evaluate(sp, base)
    t0 = time.perf_counter()
    its = 0
    while True:
        r = go.evaluate(sp, base)
        base = opt.step(base, r['grad'], 0.01)
        its += 1
        if time.perf_counter() - t0 > budget_s or its >= 200:
            break
    el = time.perf_counter() - t0
    return {'value': its / el, 'unit': 'GRAPE iterations/s', 'cores': 1, 'kind': 'port',
            'sample': '%d iterations of 1 seed of the same C2 workload in %.1f s (NumPy complex128 oracle, BLAS limited '
                      'to one thread)' % (its, el)}


def cpu_baseline_reference_ops(max_evals=6, budget_s=10.0):
    """BASELINE.md "B-faithful": the reference's own op sequence -- real-embedded 2n x 2n float32 matrices, one Defun per
    slice with recompute inside the gradient function, TWO graph evaluations per iteration (run_session.py:53-54, 69) --
    emulated node for node in torch-CPU (oracle/tf_graph_emulation.py).  Not TensorFlow (absent)."""
    import torch
    from oracle import tf_graph_emulation as tfe
    from tests.helpers import oracle_system
    from tests.golden import cases
    sp = oracle_system(cases.case_c2(n=N, k=K_OPS, steps=SLICES, m=M, taylor=TAYLOR, seed=0))
    torch.set_num_threads(min(16, torch.get_num_threads()))        # 2n x 2n = 64 x 64 matmuls: more threads only thrash
    base = seed_bases(0, 1)[0]
    tfe.evaluate_graph(sp, base, dtype=torch.float32)               # warm-up
    t0 = time.perf_counter()
    evals = 0
    while evals < max_evals and time.perf_counter() - t0 < budget_s:
        tfe.evaluate_graph(sp, base, dtype=torch.float32)
        evals += 1
    el = time.perf_counter() - t0
    return {'value': evals / el / 2.0, 'unit': 'GRAPE iterations/s', 'cores': int(torch.get_num_threads()),
            'kind': 'port', 'sample': '%d fp32 real-embedded graph evaluations (fwd + custom-gradient bwd) of 1 seed of C2 in '
                                      '%.1f s; 2 evaluations per reference iteration; torch-CPU emulation of the TF graph' % (evals, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--seeds-per-gpu', type=int, default=SEEDS_PER_GPU)
    ap.add_argument('--chunks', type=int, default=0)
    ap.add_argument('--path', type=int, default=0)
    ap.add_argument('--variant', type=int, default=0, help='MFMA path: kernel of the exponentials (qoc_config.variant)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--groups', type=int, default=1, help='split the seeds of this GPU over G engines/streams')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # test hook: QOC_BENCH_BACKEND=gloo + QOC_BENCH_SAME_DEVICE=1 runs N ranks on ONE GPU to exercise this code path
        backend = os.environ.get('QOC_BENCH_BACKEND', 'nccl')
        if os.environ.get('QOC_BENCH_SAME_DEVICE') == '1':
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from quantum_optimal_control.core import hip_engine
    from quantum_optimal_control.parallel_seeds import SeedShard

    c, Hs, U0, V, W, dt = build_problem()
    B = args.seeds_per_gpu
    shard = SeedShard(total_seeds=B * world, rank=rank, world=world)
    G = max(1, args.groups)
    gsh = [SeedShard(shard.count, g, G) for g in range(G)]
    engs = []
    for g in range(G):
        e = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], SLICES, TAYLOR[0], TAYLOR[1],
                                 reg_coeffs={}, n_seeds=gsh[g].count, device=local_rank, path=args.path,
                                 chunks=args.chunks, variant=args.variant)
        e.set_base(seed_bases(shard.first + gsh[g].first, gsh[g].count))
        engs.append(e)
    eng = engs[0]
    params = eng.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25,
                             max_iterations=10 ** 9, poll_every=10 ** 9)

    class _Multi(object):
        def iterate(self, p, n):
            for _ in range(n):
                for e in engs:
                    e.iterate(p, 1)

        def sync(self):
            for e in engs:
                e.sync()

        def scalars(self):
            parts = [e.scalars() for e in engs]
            return {k: np.concatenate([q[k] for q in parts]) for k in parts[0]}

    multi = _Multi()

    def barrier():
        multi.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    multi.iterate(params, args.warmup)
    barrier()
    t0 = time.perf_counter()
    multi.iterate(params, args.steps)
    multi.sync()
    sc = multi.scalars()
    fidelity = shard.all_gather(1.0 - sc['loss'], dist)          # RCCL all-gather of the final fidelities
    if dist is not None:
        import torch
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    assert int(np.sum(sc['done'])) == 0 and np.all(sc['iterations'] == args.warmup + args.steps), \
        'a seed stopped early: timed work would be incomplete'
    if dist is not None:
        import torch
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant kernel, hipEvents on the engine stream (separate pass, rank 0) -----------------
    roof = None
    if rank == 0:
        eng.profile_enable(True)
        eng.iterate(params, min(10, args.steps))
        pr = eng.profile_read()
        eng.profile_enable(False)
        T, s = TAYLOR
        flops_per_launch = gsh[0].count * SLICES * ((T - 1 + s) + 1) * 8.0 * N ** 3   # expm GEMMs + chain GEMM, SURVEY 8d
        avg_ms = pr['total_ms'] / max(1, pr['launches'])
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        # HBM traffic of the dominant kernel: PMC counters need rocprofv3, so the value comes from the committed PMC pass of
        # this same command (profiles/r01_pmc_traffic.json) when the workload matches; null otherwise
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
            w = pm['workload']
            if (w['seeds_per_gpu'], w['chunks'], pm['kernel']) == (gsh[0].count, eng.chunks, pr['kernel']):
                traffic = pm['hbm_bytes_per_launch']
        except Exception:
            traffic = None
        roof = {'bound': 'mfma', 'kernel': pr['kernel'], 'achieved': achieved, 'peak': FP64_MATRIX_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': achieved / FP64_MATRIX_PEAK_TFLOPS, 'traffic': traffic,
                'traffic_unit': 'bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_traffic.txt)',
                'avg_launch_ms': avg_ms, 'launches': pr['launches'], 'flops_per_launch': flops_per_launch,
                'measured_mfma_f64_ceilings_TFLOPs': {'v_mfma_f64_16x16x4': 48.2, 'v_mfma_f64_4x4x4_4b': 73.0},
                # `achieved` counts the ALGORITHMIC flops (4 real products per complex product, plain Taylor); the
                # kernel issues 3 real MFMA products per complex product (Karatsuba form) = 3/4 of that on the matrix pipe
                'executed_mfma_TFLOPs': 0.75 * achieved,
                'executed_frac_of_measured_4x4x4_ceiling': 0.75 * achieved / 73.0}
    # ---- latency of ONE trajectory of the same workload (what a plain Grape() call runs), outside the timed region ----
    single = None
    if rank == 0:
        e1 = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], SLICES, TAYLOR[0], TAYLOR[1],
                                  reg_coeffs={}, n_seeds=1, device=local_rank)
        e1.set_base(seed_bases(0, 1))
        e1.iterate(params, 3); e1.sync()
        t1 = time.perf_counter()
        e1.iterate(params, 50); e1.sync()
        el1 = (time.perf_counter() - t1) / 50
        single = {'value': 1.0 / el1, 'unit': 'GRAPE iterations/s', 'ms_per_iteration': el1 * 1e3, 'path': e1.path,
                  'note': 'one control set (n_seeds=1, AUTO path) of the same C2 workload, 50 iterations; not part of `value`'}
        e1.close()
    total_seeds = B * world
    value = total_seeds * args.steps / elapsed
    if rank == 0:
        out = {
            'metric': 'GRAPE iterations/sec (fwd+bwd+Adam) at n=32 k=4 steps=500', 'value': value,
            'unit': 'GRAPE iterations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'C2 3-transmon-size unitary gate: n=32 k=4 steps=500 m=8 Taylor(T,s)=(5,3), '
                                   '%d independent control seeds per GPU (aggregate over seeds), reg_coeffs={}' % B,
                       'seeds_per_gpu': B, 'total_seeds': total_seeds, 'path': eng.path, 'chunks': eng.chunks,
                       'stream_groups': G,
                       'parallelism': 'seed-sharded x%d, RCCL all-gather of final fidelities' % world},
            'per_seed_iterations_per_s': args.steps / elapsed,
            'single_trajectory': single,
            'best_fidelity': float(np.max(fidelity)),
            'roofline': roof,
        }
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only (contract)
            out['cpu_baseline'] = cpu_baseline()
            try:
                out['cpu_baseline_reference_ops'] = cpu_baseline_reference_ops()
            except Exception as exc:                         # torch missing etc.: the primary baseline stands
                out['cpu_baseline_reference_ops'] = {'error': repr(exc)}
        print(json.dumps(out))
    for e in engs:
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
