#!/usr/bin/env python
"""BASELINE config 5 (n = 512, k = 8, 2000 slices) with the pulse sharded along the TIME axis (csrc/qoc_gemm_ts.h).

  python tools/c5_time_sharded.py [G]                                   one GPU: G time shards EMULATED in one engine (default 8) against the plain engine
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/c5_time_sharded.py
                                                                        N GPUs: one rank per GPU, RCCL over xGMI (two collectives per iteration)
  QOC_HIP_LIBRARY=<a -DQOC_DEBUG build of qoc_engine> python tools/c5_time_sharded.py 8 rank-time
                                                                        one GPU: what ONE rank of G computes per iteration (the other ranks' share and
                                                                        the two collectives left out: a lower bound of the sharded iteration time)
Prints ms per iteration (max over ranks).  QOC_C5_N / QOC_C5_STEPS shrink the problem for a quick run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import numpy as np
from quantum_optimal_control import parallel_seeds
from quantum_optimal_control.core import hip_engine
from tests.golden import cases
from tests.helpers import oracle_system

n, steps = int(os.environ.get('QOC_C5_N', 512)), int(os.environ.get('QOC_C5_STEPS', 2000))
sp = oracle_system(cases.case_c2(n=n, k=8, steps=steps, m=8, taylor=(5, 3), seed=2))
rank, local, world = parallel_seeds.launch_env()


def run(label, iters=3, **kw):
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs={}, n_seeds=1, **kw)
    eng.set_base(sp.base0[None])
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    eng.iterate(p, 2); eng.sync()
    t0 = time.perf_counter()
    eng.iterate(p, iters); eng.sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    out = (ms, eng.scalars()['loss'][0], eng.plan)
    eng.close()
    return out


if world > 1:
    comm = parallel_seeds.open_comm(require_rccl=True)
    ms, loss, plan = run('rank', device=comm.device, time_shards=world, time_rank=rank, time_comm=comm)
    ms = float(comm.all_reduce_max([ms])[0])
    if rank == 0:
        print('n=%d steps=%d: %d time shards over RCCL (%s): %.2f ms per iteration (max over ranks), loss %.9f, %s' % (n, steps, world, comm.library, ms, loss, plan))
    comm.barrier(); comm.close()
else:
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    if len(sys.argv) > 2 and sys.argv[2] == 'rank-time':
        for r in sorted({0, G // 2, G - 1}):
            os.environ['QOC_TS_ONLY_RANK'] = str(r)
            ms, _, _ = run('rank %d' % r, time_shards=G, time_rank=-1)
            print('n=%d steps=%d rank %d of %d alone (no exchange) : %8.2f ms per iteration' % (n, steps, r, G, ms))
        sys.exit(0)
    ms0, loss0, plan0 = run('plain')
    ms1, loss1, plan1 = run('emulated', time_shards=G, time_rank=-1)
    print('n=%d steps=%d plain engine               : %8.2f ms per iteration, loss %.12f, chunks %s' % (n, steps, ms0, loss0, plan0['chunks']))
    print('n=%d steps=%d %2d time shards, emulated    : %8.2f ms per iteration, loss %.12f (all ranks one after the other on ONE GPU: the sum of their work)' % (n, steps, G, ms1, loss1))
    assert abs(loss0 - loss1) < 1e-9
