"""Body of tests/test_seed_sharding.py::test_grape_sharded_two_ranks_one_gpu: two ranks (gloo) share GPU 0, each optimises
half of the restarts; every rank must return the pair a single-process Grape(restarts=total) returns."""
import contextlib
import io
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
from quantum_optimal_control.main_grape.grape import Grape, GrapeSharded  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.helpers import grape_kwargs  # noqa: E402

if __name__ == '__main__':
    dist.init_process_group('gloo')
    rank = dist.get_rank()
    c = cases.case_c1()
    conv = {'rate': 0.05, 'update_step': 10, 'max_iterations': 30, 'conv_target': 1e-12, 'learning_rate_decay': 100}
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks, Uf = GrapeSharded(convergence=conv, method='Adam', restarts=6, dist=dist, device=0, **grape_kwargs(c))
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks1, Uf1 = Grape(convergence=conv, method='Adam', restarts=6, plan_seeds=3, **grape_kwargs(c))
    # bit for bit: both runs plan their kernels for the same batch (qoc_config.plan_seeds; GrapeSharded's default is the largest shard of ITS launch,
    # ceil(6 / 2) = 3, and the single process is told the same), whatever the rank count
    assert np.array_equal(uks, uks1) and np.array_equal(Uf, Uf1), (rank, np.max(np.abs(uks - uks1)))
    print('OK sharded rank %d' % rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()
