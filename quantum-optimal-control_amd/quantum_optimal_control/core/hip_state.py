"""HipState -- the engine-side counterpart of the reference's TensorflowState (core/tensorflow_state.py:11-394).

TensorflowState builds a static TF graph whose tensors run_session fetches; HipState uploads the same constants
to the MI355X once (``build_graph`` keeps the reference's method name) and exposes the same fetch set as plain
methods backed by the C ABI of libqoc_hip.so.
"""
import numpy as np

from quantum_optimal_control.core import hip_engine


class HipState(object):

    def __init__(self, sys_para, n_seeds=1, device=0, path=hip_engine.PATH_AUTO, chunks=0, first_seed=0, plan_seeds=0, time_comm=None):
        self.sys_para = sys_para
        self.n_seeds = n_seeds
        self.first_seed = first_seed      # global index of this engine's first restart (seed-sharded runs)
        self.device = device
        self.path = path
        self.chunks = chunks
        self.plan_seeds = plan_seeds      # batch size AUTO plans for (0 = n_seeds): qoc_config.plan_seeds
        self.time_comm = time_comm        # hip_engine.QocComm of a time-sharded run (one trajectory over the GPUs of a node), or None
        self.engine = None

    def build_graph(self):
        sp = self.sys_para
        rc = sp.reg_coeffs
        if rc is None:
            # the reference evaluates `'amplitude' in None` here (regularization_functions.py:15)
            raise TypeError("argument of type 'NoneType' is not iterable")
        if 'd2wdt2' in rc and 'dwdt' not in rc:
            raise NameError("name 'new_weights' is not defined")         # regularization_functions.py:38-45
        if 'bandpass' in rc and not sp.use_gpu:
            raise ValueError('currently does not support bandpass reg for CPU (no CPU kernel for FFT)')
        needs_inter = ('forbidden_coeff_list' in rc) or ('speed_up' in rc)
        if needs_inter and not sp.state_transfer and not sp.use_inter_vecs:
            raise TypeError("'NoneType' object is not iterable")         # tfs.inter_vecs is None (:71, :381)
        Hs, U0, V, W, Vs = sp.engine_inputs()
        self.engine = hip_engine.HipEngine(
            Hs, U0, V, W, sp.ops_max_amp, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
            state_transfer=sp.state_transfer, reg_coeffs=rc,
            one_minus_gauss=sp.one_minus_gauss if 'envelope' in rc else None, Vs=Vs,
            n_seeds=self.n_seeds, device=self.device, path=self.path, chunks=self.chunks, plan_seeds=self.plan_seeds,
            time_shards=0 if self.time_comm is None else self.time_comm.world, time_rank=-1 if self.time_comm is None else self.time_comm.rank,
            time_comm=self.time_comm)
        base = np.asarray(sp.ops_weight_base, dtype=np.float64)
        if base.ndim == 2 and (self.n_seeds > 1 or self.first_seed > 0):
            # global restart 0 = the reference's own starting point; restart g > 0 = the reproducible stream of index g,
            # whatever rank / batch it lands in
            from quantum_optimal_control.parallel_seeds import restart_guesses
            if self.first_seed == 0:
                extra = restart_guesses(base.shape[0], base.shape[1], 1, self.n_seeds - 1)
                base = np.concatenate([base[None], extra], axis=0)
            else:
                base = restart_guesses(base.shape[0], base.shape[1], self.first_seed, self.n_seeds)
        elif base.ndim == 2:
            base = base[None]
        self.engine.set_base(base)
        print("Graph built!")
        return self.engine

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None
