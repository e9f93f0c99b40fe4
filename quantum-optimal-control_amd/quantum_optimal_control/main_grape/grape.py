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
          use_inter_vecs=True, restarts=1, _first_seed=0, _device=0, _return_session=False):
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
    tfs = HipState(sys_para, n_seeds=max(1, int(restarts)), device=_device, first_seed=_first_seed)   # constants -> HBM
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


def GrapeSharded(*args, restarts=8, dist=None, **kwargs):
    """`Grape(...)` with `restarts` control sets block-partitioned over the ranks of a torch.distributed process group
    (one process per GPU; `dist` = the initialised `torch.distributed` module, or None for a single process).  Every rank
    optimises its own restarts on its own GPU -- no data-path collective --, the best final losses are all-gathered once
    (RCCL when the backend is "nccl"), and the winner's (uks, U_final) is broadcast, so every rank returns the same pair.
    Global restart g starts from the same point whatever the number of ranks (restart 0 = the reference's own draw)."""
    from quantum_optimal_control.parallel_seeds import SeedShard
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    shard = SeedShard(total_seeds=int(restarts), rank=rank, world=world)
    if shard.count == 0:
        raise ValueError('GrapeSharded: more ranks (%d) than restarts (%d)' % (world, restarts))
    device = int(kwargs.pop('device', os.environ.get('LOCAL_RANK', 0) if dist is not None else 0))
    if rank != 0:
        kwargs['save'] = False                                       # only rank 0 may write the run log
    out = Grape(*args, restarts=shard.count, _first_seed=shard.first, _device=device, _return_session=True, **kwargs)
    if out is None:
        return None
    uks, Uf, loss = out
    if dist is None or world == 1:
        return uks, Uf
    # one value per RANK (its best restart): gather, pick the winner, broadcast its pulse and unitary
    import torch
    dev = shard._device(dist)
    mine = torch.tensor([loss], dtype=torch.float64, device=dev)
    allv = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(allv, mine)
    best = int(np.argmin([float(v.item()) for v in allv]))
    t = torch.from_numpy(np.ascontiguousarray(uks, dtype=np.float64)).to(dev)
    dist.broadcast(t, src=best)
    uks = t.cpu().numpy()
    if not isinstance(Uf, list):
        tu = torch.from_numpy(np.ascontiguousarray(Uf).view(np.float64).copy()).to(dev)
        dist.broadcast(tu, src=best)
        Uf = tu.cpu().numpy().view(np.complex128)
    return uks, Uf
