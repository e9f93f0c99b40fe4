"""Grape() -- drop-in entry point of the MI355X-native GRAPE engine.

Signature, defaults, return value and error behaviour follow the reference's main_grape/grape.py:19-139; the
TensorFlow graph + session underneath are replaced by libqoc_hip.so (hand-written HIP kernels, complex fp64).
Differences that a caller can observe are listed in INTEGRATION.md (no live matplotlib figure; `use_gpu=False`
still runs on the MI355X because there is no CPU engine; HDF5 logging needs h5py).
"""
import os
import time

import numpy as np

from quantum_optimal_control.core.convergence import Convergence
from quantum_optimal_control.core.hip_state import HipState
from quantum_optimal_control.core.run_session import run_session
from quantum_optimal_control.core.system_parameters import SystemParameters

_TIME_UNITS = {"GHz": "ns", "MHz": "us", "KHz": "ms", "Hz": "s"}


def _next_free_log(data_path, file_name):
    number = 0
    while os.path.exists(os.path.join(data_path, str(number).zfill(5) + "_" + file_name + ".h5")):
        number += 1
    return os.path.join(data_path, str(number).zfill(5) + "_" + file_name + ".h5")


def _dump_inputs(file_path, H0, Hops, Hnames, U, total_time, steps, states_concerned_list, use_gpu, sparse_H,
                 sparse_U, sparse_K, maxA, initial_guess, method, convergence, reg_coeffs, dressed_info):
    from quantum_optimal_control.helper_functions.data_management import H5File
    with H5File(file_path) as hf:
        for key, val in (('H0', H0), ('Hops', Hops), ('Hnames', Hnames), ('U', U), ('total_time', total_time),
                         ('steps', steps), ('states_concerned_list', states_concerned_list), ('use_gpu', use_gpu),
                         ('sparse_H', sparse_H), ('sparse_U', sparse_U), ('sparse_K', sparse_K)):
            hf.add(key, data=val)
        if maxA is not None:
            hf.add('maxA', data=maxA)
        if initial_guess is not None:
            hf.add('initial_guess', data=initial_guess)
        hf.add('method', method)
        group = hf.create_group('convergence')
        for key, val in convergence.items():          # like the reference this needs a dict when save=True
            group.create_dataset(key, data=val)
        for name, mapping in (('reg_coeffs', reg_coeffs), ('dressed_info', dressed_info)):
            if mapping is not None:
                group = hf.create_group(name)
                for key, val in mapping.items():
                    group.create_dataset(key, data=val)


def Grape(H0, Hops, Hnames, U, total_time, steps, states_concerned_list, convergence=None, U0=None, reg_coeffs=None,
          dressed_info=None, maxA=None, use_gpu=True, sparse_H=True, sparse_U=False, sparse_K=False, draw=None,
          initial_guess=None, show_plots=True, unitary_error=1e-4, method='Adam', state_transfer=False,
          no_scaling=False, freq_unit='GHz', file_name=None, save=True, data_path=None, Taylor_terms=None,
          use_inter_vecs=True, restarts=1, plan_seeds=None, time_comm=None, _first_seed=0, _device=0, _return_session=False):
    """Reference signature (main_grape/grape.py:19) plus one optional extension: ``restarts=B`` optimises B control sets at
    once on the GPU -- the first is the reference's own initial guess (same NumPy RNG draw / ``initial_guess``), the others
    are independent N(0, 1/sqrt(steps)) restarts -- and returns the (uks, U_final) of the best final fidelity."""
    grape_start_time = time.time()
    time_unit = _TIME_UNITS[freq_unit]                  # KeyError on an unknown unit, as in the reference
    if use_gpu:
        sparse_H = sparse_U = sparse_K = False          # dense kernels only

    file_path = None
    if save:
        if file_name is None:
            raise ValueError('Grape function input: file_name, is not specified.')
        if data_path is None:
            raise ValueError('Grape function input: data_path, is not specified.')
        file_path = _next_free_log(data_path, file_name)
        print("data saved at: " + str(file_path))
        _dump_inputs(file_path, H0, Hops, Hnames, U, total_time, steps, states_concerned_list, use_gpu, sparse_H,
                     sparse_U, sparse_K, maxA, initial_guess, method, convergence, reg_coeffs, dressed_info)

    if U0 is None:
        U0 = np.identity(len(H0))
    if convergence is None:
        convergence = {'rate': 0.01, 'update_step': 100, 'max_iterations': 5000, 'conv_target': 1e-8,
                       'learning_rate_decay': 2500}
    if maxA is None:
        if initial_guess is None:
            maxAmp = 4 * np.ones(len(Hops))
        else:
            maxAmp = 1.5 * np.max(np.abs(initial_guess)) * np.ones(len(Hops))
    else:
        maxAmp = maxA

    sys_para = SystemParameters(H0, Hops, Hnames, U, U0, total_time, steps, states_concerned_list, dressed_info,
                                maxAmp, draw, initial_guess, show_plots, unitary_error, state_transfer, no_scaling,
                                reg_coeffs, save, file_path, Taylor_terms, use_gpu, use_inter_vecs, sparse_H, sparse_U,
                                sparse_K)
    # plan_seeds (extension): the batch size the engine plans its path / kernels / chunking for instead of `restarts` (None: its own batch,
    # the fastest choice for this process).  GrapeSharded passes hip_engine.plan_seeds_for(all restarts), so that a restart evolves
    # bit-identically under any rank count; Grape(restarts=R, plan_seeds=hip_engine.plan_seeds_for(R)) reproduces a sharded run in one process.
    # time_comm (extension): a hip_engine.QocComm whose ranks share ONE large trajectory along the time axis (GrapeTimeSharded below)
    tfs = HipState(sys_para, n_seeds=max(1, int(restarts)), device=_device if time_comm is None else time_comm.device, first_seed=_first_seed,
                   plan_seeds=0 if plan_seeds is None else int(plan_seeds), time_comm=time_comm)   # constants -> HBM
    graph = tfs.build_graph()
    conv = Convergence(sys_para, time_unit, convergence)
    try:
        SS = run_session(tfs, graph, conv, sys_para, method, show_plots=sys_para.show_plots, use_gpu=use_gpu)
        if save:
            from quantum_optimal_control.helper_functions.data_management import H5File
            with H5File(file_path) as hf:
                hf.add('wall_clock_time', data=np.array(time.time() - grape_start_time))
            print("data saved at: " + str(file_path))
        if _return_session:
            return SS.uks, SS.Uf, float(SS.l)
        return SS.uks, SS.Uf
    except KeyboardInterrupt:
        if save:
            from quantum_optimal_control.helper_functions.data_management import H5File
            with H5File(file_path) as hf:
                hf.add('wall_clock_time', data=np.array(time.time() - grape_start_time))
            print("data saved at: " + str(file_path))
        return None
    finally:
        tfs.close()


def GrapeTimeSharded(*args, comm=None, **kwargs):
    """`Grape(...)` for ONE large control problem (hundreds of levels, thousands of slices: BASELINE config 5) on all GPUs of a node: the pulse is
    cut along the TIME axis, rank r of `comm` (parallel_seeds.open_comm(): RCCL over xGMI behind the C ABI) forms the propagators, sweeps and
    gradients of its run of time chunks, and two small collectives per iteration -- an all-gather of one n x n product per rank, an all-reduce of the
    gradient array -- keep the ranks in lock step (csrc/qoc_gemm_ts.h).  Every rank runs the same optimiser on the whole pulse and returns
    the same (uks, U_final); only rank 0 writes the run log.  comm = None: a plain Grape call.  Unitary mode, no forbidden-level / speed_up term,
    n > 96, at most 8 states of interest; the reference has no counterpart (one device: main_grape/grape.py:106-109)."""
    if comm is None:
        return Grape(*args, **kwargs)
    from quantum_optimal_control.core import hip_engine
    if not isinstance(comm, hip_engine.QocComm):
        # parallel_seeds.open_comm() falls back to a host file transport when RCCL cannot start; that transport cannot carry the two collectives
        # every iteration enqueues on the engine's stream.  Say so before any rank builds an engine (open it with require_rccl=True to fail earlier).
        raise TypeError('GrapeTimeSharded needs an RCCL communicator (hip_engine.QocComm, e.g. parallel_seeds.open_comm(require_rccl=True)); got %s%s'
                        % (type(comm).__name__, (': ' + str(getattr(comm, 'fallback_reason', ''))) if getattr(comm, 'fallback_reason', None) else ''))
    if comm.rank != 0:
        kwargs['save'] = False
        kwargs['show_plots'] = False
    return Grape(*args, time_comm=comm, **kwargs)


def GrapeSharded(*args, restarts=8, dist=None, comm=None, **kwargs):
    """`Grape(...)` with `restarts` control sets block-partitioned over the ranks of a one-node job (one process per GPU).
    Transport: `comm` = a `hip_engine.QocComm` (RCCL over xGMI behind the C ABI; `parallel_seeds.open_comm()` builds it from the
    launcher's RANK/LOCAL_RANK/WORLD_SIZE) or `dist` = an initialised `torch.distributed` module (gloo in the CPU tests); neither
    = a single process.  Every rank optimises its own restarts on its own GPU -- no data-path collective --, the best final
    losses are all-gathered once, and the winner's (uks, U_final) is broadcast, so every rank returns the same pair.
    Global restart g starts from the same point whatever the number of ranks (restart 0 = the reference's own draw), and every engine of the
    launch plans its kernels for the same batch: `plan_seeds`, default = the LARGEST shard of this launch, ceil(restarts / ranks) -- identical on all
    ranks, and never smaller than what an engine holds, so AUTO picks what is fastest for the launch at hand.  Bit-identity ACROSS launches with
    different rank counts is opt-in: pass the same `plan_seeds` to each of them (`hip_engine.plan_seeds_for(restarts)` = restarts / GPUs of the
    node is the natural choice) -- the result is then bit for bit that of `Grape(restarts=restarts, plan_seeds=<the same>)` in one process
    (tests/sharded_script.py checks array_equal); an engine that holds more control sets than it plans for says so on stderr when that changes
    the kernels it runs."""
    from quantum_optimal_control.parallel_seeds import SeedShard
    if comm is not None:
        world, rank = comm.world, comm.rank
    elif dist is not None:
        world, rank = dist.get_world_size(), dist.get_rank()
    else:
        world, rank = 1, 0
    shard = SeedShard(total_seeds=int(restarts), rank=rank, world=world)
    if min(shard.counts) == 0:           # the same verdict on EVERY rank: nobody is left waiting in a collective
        raise ValueError('GrapeSharded: more ranks (%d) than restarts (%d)' % (world, restarts))
    default_device = comm.device if comm is not None else (os.environ.get('LOCAL_RANK', 0) if dist is not None else 0)
    device = int(kwargs.pop('device', default_device))
    if rank != 0:
        kwargs['save'] = False                                       # only rank 0 may write the run log
    kwargs.setdefault('plan_seeds', max(shard.counts))
    out = Grape(*args, restarts=shard.count, _first_seed=shard.first, _device=device, _return_session=True, **kwargs)
    if world == 1:
        return None if out is None else out[:2]
    # one value per RANK (its best restart): gather, pick the winner, broadcast its pulse and unitary.  A rank whose run was
    # interrupted still takes part (loss = +inf), so the others are not left waiting.
    ranks = SeedShard(total_seeds=world, rank=rank, world=world)     # one slot per rank
    loss = np.inf if out is None else out[2]
    losses = ranks.all_gather(np.array([loss]), dist=dist, comm=comm)
    if not np.all(np.isfinite(losses)):
        return None
    best = int(np.argmin(losses))
    uks, Uf = out[0], out[1]
    uks = ranks.broadcast_from_owner(best, lambda i: np.asarray(uks, dtype=np.float64), np.shape(uks), dist=dist, comm=comm)
    if not isinstance(Uf, list):
        shape = np.shape(Uf)
        flat = ranks.broadcast_from_owner(best, lambda i: np.ascontiguousarray(Uf, dtype=np.complex128).view(np.float64),
                                          shape[:-1] + (2 * shape[-1],), dist=dist, comm=comm)
        Uf = np.ascontiguousarray(flat).view(np.complex128)
    return uks, Uf
