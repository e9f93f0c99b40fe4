"""ctypes binding of libqoc_hip.so (C ABI declared in include/qoc.h).

This is the only place the Python host touches the device: the engine object plays the role the TensorFlow
graph + session play in the reference (core/tensorflow_state.py + the session.run fetches of
core/run_session.py:53-54,66-69,119-127 and core/analysis.py:26-41).  There is NO CPU fallback: if the shared
library or a HIP device is missing, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('QOC_HIP_LIBRARY') or os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libqoc_hip.so'))   # override: A/B builds

PATH_AUTO, PATH_GENERIC, PATH_MFMA, PATH_ST_FUSED, PATH_GEMM, PATH_SMALL = 0, 1, 2, 3, 4, 5


class QocConfig(C.Structure):
    _fields_ = [('n', C.c_int32), ('k', C.c_int32), ('steps', C.c_int32), ('m', C.c_int32),
                ('taylor_terms', C.c_int32), ('scaling', C.c_int32), ('state_transfer', C.c_int32),
                ('n_seeds', C.c_int32), ('dt', C.c_double), ('total_time', C.c_double),
                ('has_amplitude', C.c_int32), ('has_envelope', C.c_int32), ('has_dwdt', C.c_int32),
                ('has_d2wdt2', C.c_int32), ('has_speed_up', C.c_int32), ('has_bandpass', C.c_int32),
                ('c_amplitude', C.c_double), ('c_envelope', C.c_double), ('c_dwdt', C.c_double),
                ('c_d2wdt2', C.c_double), ('c_speed_up', C.c_double), ('c_bandpass', C.c_double),
                ('band_lo', C.c_int32), ('band_hi', C.c_int32), ('n_forbidden', C.c_int32),
                ('forbid_dressed', C.c_int32), ('device', C.c_int32), ('path', C.c_int32), ('chunks', C.c_int32),
                ('variant', C.c_int32), ('plan_seeds', C.c_int32), ('time_shards', C.c_int32), ('time_rank', C.c_int32), ('reserved', C.c_int32 * 3)]


class QocAdamParams(C.Structure):
    _fields_ = [('rate', C.c_double), ('learning_rate_decay', C.c_double), ('conv_target', C.c_double),
                ('min_grad', C.c_double), ('max_iterations', C.c_int32), ('poll_every', C.c_int32)]


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)
_lib = None

_SIGNATURES = {
    'qoc_create': (C.c_int, [C.POINTER(QocConfig), _DP, _DP, _DP, _DP, _DP, _DP, _IP, _DP, _DP, C.POINTER(C.c_void_p)]),
    'qoc_destroy': (C.c_int, [C.c_void_p]),
    'qoc_set_base': (C.c_int, [C.c_void_p, _DP]),
    'qoc_get_base': (C.c_int, [C.c_void_p, _DP]),
    'qoc_eval': (C.c_int, [C.c_void_p, _DP, _DP, _DP, _DP, _DP]),
    'qoc_adam_step': (C.c_int, [C.c_void_p, _DP]),
    'qoc_run_adam': (C.c_int, [C.c_void_p, C.POINTER(QocAdamParams), _IP]),
    'qoc_iterate': (C.c_int, [C.c_void_p, C.POINTER(QocAdamParams), C.c_int32]),
    'qoc_sync': (C.c_int, [C.c_void_p]),
    'qoc_get_scalars': (C.c_int, [C.c_void_p, _DP, _DP, _DP, _DP, _IP, _IP]),
    'qoc_get_uks': (C.c_int, [C.c_void_p, _DP]),
    'qoc_get_uks_evaluated': (C.c_int, [C.c_void_p, _DP]),
    'qoc_get_final_unitary': (C.c_int, [C.c_void_p, _DP]),
    'qoc_get_inter_vecs': (C.c_int, [C.c_void_p, _DP]),
    'qoc_profile_enable': (C.c_int, [C.c_void_p, C.c_int32]),
    'qoc_profile_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), _DP]),
    'qoc_time_iterations': (C.c_int, [C.c_void_p, C.POINTER(QocAdamParams), C.c_int32, _DP]),
    'qoc_path_in_use': (C.c_int, [C.c_void_p]),
    'qoc_chunks_in_use': (C.c_int, [C.c_void_p]),
    'qoc_plan_describe': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    'qoc_set_time_comm': (C.c_int, [C.c_void_p, C.c_void_p]),
    'qoc_comm_unique_id': (C.c_int, [C.c_void_p]),
    'qoc_comm_probe': (C.c_int, [C.c_int32]),
    'qoc_comm_create': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'qoc_comm_destroy': (C.c_int, [C.c_void_p]),
    'qoc_comm_world': (C.c_int, [C.c_void_p]),
    'qoc_comm_rank': (C.c_int, [C.c_void_p]),
    'qoc_comm_library': (C.c_char_p, []),
    'qoc_comm_all_gather_scalar': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _DP]),
    'qoc_comm_all_gather_f64': (C.c_int, [C.c_void_p, _DP, C.c_int32, _DP]),
    'qoc_comm_all_reduce_max_f64': (C.c_int, [C.c_void_p, _DP, C.c_int32]),
    'qoc_comm_broadcast_f64': (C.c_int, [C.c_void_p, _DP, C.c_int64, C.c_int32]),
    'qoc_comm_barrier': (C.c_int, [C.c_void_p]),
    'qoc_device_count': (C.c_int, []),
    'qoc_device_info': (C.c_int, [C.c_int32, C.c_char_p, C.c_int32, _IP, C.POINTER(C.c_int64)]),
    'qoc_device_peer_access': (C.c_int, [C.c_int32, C.c_int32, _IP]),
    'qoc_last_error': (C.c_char_p, []),
    'qoc_version': (C.c_char_p, []),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library():
    """dlopen the in-tree libqoc_hip.so and declare every prototype of include/qoc.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('libqoc_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
                          'g.build()"` (hipcc --offload-arch=gfx950); there is no CPU fallback.' % LIB_PATH)
    # one process per GPU under a launcher: RCCL shares device memory between the ranks through dmabuf IPC handles, the only kind this host
    # driver supports, and the HIP runtime reads the switch when it starts -- i.e. possibly at the first call into the library loaded here
    if int(os.environ.get('WORLD_SIZE', '1') or 1) > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def library_loaded():
    return _lib is not None


class QocError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise QocError('libqoc_hip: status %d: %s' % (rc, load_library().qoc_last_error().decode()))


def _dp(a):
    return None if a is None else a.ctypes.data_as(_DP)


def _c128(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.complex128))
    assert a.shape == tuple(shape), (a.shape, shape)
    return a


def device_count():
    return load_library().qoc_device_count()


def device_info(device=0):
    lib = load_library()
    name = C.create_string_buffer(256)
    cus = C.c_int32()
    mem = C.c_int64()
    _check(lib.qoc_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=mem.value)


def device_peer_access(device, peer):
    """True when HIP device `device` can address the memory of device `peer` directly (hipDeviceCanAccessPeer)."""
    lib = load_library()
    can = C.c_int32()
    _check(lib.qoc_device_peer_access(int(device), int(peer), C.byref(can)))
    return bool(can.value)


def reg_config(reg_coeffs, total_time):
    """Translate the reference's reg_coeffs dict (core/regularization_functions.py:15-88) into qoc_config fields."""
    rc = {} if reg_coeffs is None else reg_coeffs
    out = {}
    for key, name in (('amplitude', 'amplitude'), ('envelope', 'envelope'), ('dwdt', 'dwdt'), ('d2wdt2', 'd2wdt2'),
                      ('speed_up', 'speed_up'), ('bandpass', 'bandpass')):
        out['has_' + name] = int(key in rc)
        out['c_' + name] = float(rc[key]) if key in rc else 0.0
    if 'bandpass' in rc:
        band_id = (np.array(rc['band']) * total_time).astype(int)              # regularization_functions.py:59-61
        out['band_lo'], out['band_hi'] = int(band_id[0]), int(band_id[1])
    if 'forbidden_coeff_list' in rc:
        pairs = list(zip(rc['forbidden_coeff_list'], rc['states_forbidden_list']))   # zip truncation as in :81
        out['forbidden_coeffs'] = np.array([float(c) for c, _ in pairs], dtype=np.float64)
        out['forbidden_states'] = np.array([int(s) for _, s in pairs], dtype=np.int32)
    return out


COMM_ID_BYTES = 128
SCALAR_LOSS, SCALAR_REG_LOSS, SCALAR_GRAD_SQUARED, SCALAR_UNITARY_SCALE = 0, 1, 2, 3


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it, the launcher hands it to the other ranks: parallel_seeds.rendezvous)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().qoc_comm_unique_id(buf))
    return buf.raw


def plan_seeds_for(total_restarts):
    """The batch size AUTO should plan for when `total_restarts` control sets may be sharded over the GPUs of this node: what a
    full-node run gives each GPU.  It depends on the node, not on how many ranks a launch uses, so a restart evolves bit-identically
    under any rank count (and a full-node run keeps every GPU's 1024 SIMDs busy)."""
    return max(1, -(-int(total_restarts) // max(1, device_count())))


def comm_probe(device=0):
    """Local preconditions of QocComm (librccl loadable, device usable); raises QocError with the reason.  No collective."""
    _check(load_library().qoc_comm_probe(int(device)))


class QocComm(object):
    """One RCCL communicator per process (one process per GPU) behind the C ABI (include/qoc.h, multi-GPU section).
    No torch involved: the collectives run from the engine's device buffers on the engine's HIP stream."""

    def __init__(self, unique_id, world, rank, device=0):
        assert len(unique_id) == COMM_ID_BYTES
        self._lib = load_library()
        self._h = C.c_void_p()
        self.world, self.rank, self.device = int(world), int(rank), int(device)
        _check(self._lib.qoc_comm_create(C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES), self.world, self.rank,
                                         self.device, C.byref(self._h)))
        self.library = self._lib.qoc_comm_library().decode()

    def _attach(self, engine):
        import weakref
        if not hasattr(self, '_engines'):
            self._engines = weakref.WeakSet()
        self._engines.add(engine)

    def _detach(self, engine):
        if hasattr(self, '_engines'):
            self._engines.discard(engine)

    def close(self):
        """Destroys the communicator.  Time-sharded engines that still hold it (HipEngine(time_comm=...)) are closed FIRST -- the C ABI refuses to destroy a
        communicator an engine still uses, and a close() in a caller's finally block after an engine-side exception must neither leak the communicator nor
        mask that exception with a second one."""
        if getattr(self, '_h', None) is not None and self._h.value:
            for eng in list(getattr(self, '_engines', ())):
                eng.close()
            _check(self._lib.qoc_comm_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather_scalar(self, engine, which, width):
        """[world][width] rows of one per-seed scalar array of `engine` (device to device on the engine's stream)."""
        out = np.empty((self.world, int(width)))
        _check(self._lib.qoc_comm_all_gather_scalar(self._h, engine._h, int(which), int(width), _dp(out)))
        return out

    def all_gather(self, values):
        values = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
        out = np.empty((self.world, values.shape[0]))
        _check(self._lib.qoc_comm_all_gather_f64(self._h, _dp(values), values.shape[0], _dp(out)))
        return out

    def all_reduce_max(self, values):
        buf = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1)).copy()
        _check(self._lib.qoc_comm_all_reduce_max_f64(self._h, _dp(buf), buf.shape[0]))
        return buf

    def broadcast(self, array, root):
        buf = np.ascontiguousarray(np.asarray(array, dtype=np.float64)).copy()
        _check(self._lib.qoc_comm_broadcast_f64(self._h, _dp(buf.reshape(-1)), buf.size, int(root)))
        return buf

    def barrier(self):
        _check(self._lib.qoc_comm_barrier(self._h))


class HipEngine(object):
    """Device-resident GRAPE problem: constants in HBM, n_seeds control sets, one HIP stream."""

    def __init__(self, Hs, U0, V, W, maxA, dt, total_time, steps, taylor_terms, scaling, state_transfer=False,
                 reg_coeffs=None, one_minus_gauss=None, Vs=None, n_seeds=1, device=0, path=PATH_AUTO, chunks=0, variant=0, plan_seeds=0,
                 time_shards=0, time_rank=-1, time_comm=None):
        lib = load_library()
        self._lib = lib
        self._h = C.c_void_p()
        Hs = np.ascontiguousarray(np.asarray(Hs, dtype=np.complex128))
        k = Hs.shape[0] - 1
        n = Hs.shape[1]
        V = np.ascontiguousarray(np.asarray(V, dtype=np.complex128))
        m = V.shape[1]
        self.n, self.k, self.m, self.steps, self.n_seeds = n, k, m, int(steps), int(n_seeds)
        self.state_transfer = bool(state_transfer)
        W = _c128(W, (n, m))
        U0a = None if U0 is None else _c128(U0, (n, n))
        maxA = np.ascontiguousarray(np.asarray(maxA, dtype=np.float64))
        assert maxA.shape == (k,)
        rcfg = reg_config(reg_coeffs, total_time)
        cfg = QocConfig()
        cfg.n, cfg.k, cfg.steps, cfg.m = n, k, int(steps), m
        cfg.taylor_terms, cfg.scaling = int(taylor_terms), int(scaling)
        cfg.state_transfer, cfg.n_seeds = int(bool(state_transfer)), int(n_seeds)
        cfg.dt, cfg.total_time = float(dt), float(total_time)
        for name in ('amplitude', 'envelope', 'dwdt', 'd2wdt2', 'speed_up', 'bandpass'):
            setattr(cfg, 'has_' + name, rcfg['has_' + name])
            setattr(cfg, 'c_' + name, rcfg['c_' + name])
        cfg.band_lo, cfg.band_hi = rcfg.get('band_lo', 0), rcfg.get('band_hi', 0)
        fs = rcfg.get('forbidden_states')
        fc = rcfg.get('forbidden_coeffs')
        cfg.n_forbidden = 0 if fs is None else len(fs)
        use_vs = Vs is not None and cfg.n_forbidden > 0
        cfg.forbid_dressed = int(use_vs)
        cfg.device, cfg.path, cfg.chunks, cfg.variant = int(device), int(path), int(chunks), int(variant)
        cfg.time_shards, cfg.time_rank = int(time_shards), int(time_rank)     # one trajectory sharded along the time axis (csrc/qoc_gemm_ts.h); -1: emulated in this engine
        cfg.plan_seeds = int(plan_seeds)       # 0: plan for n_seeds; > 0: the batch AUTO plans for (sharded restarts: see plan_seeds_for)
        omg = None
        if one_minus_gauss is not None:
            omg = np.ascontiguousarray(np.asarray(one_minus_gauss, dtype=np.float64))
            assert omg.shape == (k, int(steps))
        Vsa = _c128(Vs, (n, n)) if use_vs else None
        _check(lib.qoc_create(C.byref(cfg), _dp(Hs.view(np.float64)), None if U0a is None else _dp(U0a.view(np.float64)),
                              _dp(V.view(np.float64)), _dp(W.view(np.float64)), _dp(maxA), _dp(omg),
                              None if fs is None else fs.ctypes.data_as(_IP), _dp(fc),
                              None if Vsa is None else _dp(Vsa.view(np.float64)), C.byref(self._h)))
        self.path = lib.qoc_path_in_use(self._h)
        self.chunks = lib.qoc_chunks_in_use(self._h)
        buf = C.create_string_buffer(256)
        _check(lib.qoc_plan_describe(self._h, buf, 256))
        self.plan = dict(kv.split('=', 1) for kv in buf.value.decode().split())
        if time_comm is not None:
            _check(lib.qoc_set_time_comm(self._h, time_comm._h))
            self._time_comm = time_comm                # keep it alive as long as the engine
            time_comm._attach(self)                    # ... and let it close this engine before it goes (QocComm.close)

    # -- lifetime ---------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.qoc_destroy(self._h)
            self._h = C.c_void_p()
            tc = getattr(self, '_time_comm', None)
            if tc is not None:
                tc._detach(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- trainable variable ---------------------------------------------------------------------------------------
    def _seed_shape(self):
        return (self.n_seeds, self.k, self.steps)

    def set_base(self, base):
        base = np.ascontiguousarray(np.asarray(base, dtype=np.float64)).reshape(self._seed_shape())
        _check(self._lib.qoc_set_base(self._h, _dp(base)))

    def get_base(self):
        out = np.empty(self._seed_shape())
        _check(self._lib.qoc_get_base(self._h, _dp(out)))
        return out

    # -- evaluation / optimisation ------------------------------------------------------------------------------
    def evaluate(self, want_grad=True):
        B = self.n_seeds
        loss, reg, g2, us = (np.empty(B) for _ in range(4))
        grad = np.empty(self._seed_shape()) if want_grad else None
        _check(self._lib.qoc_eval(self._h, _dp(loss), _dp(reg), _dp(g2), _dp(us), _dp(grad)))
        return dict(loss=loss, reg_loss=reg, grad_squared=g2, unitary_scale=us, grad=grad)

    def adam_step(self, lr):
        lr = np.ascontiguousarray(np.broadcast_to(np.asarray(lr, dtype=np.float64), (self.n_seeds,)))
        _check(self._lib.qoc_adam_step(self._h, _dp(lr)))

    @staticmethod
    def adam_params(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25, max_iterations=5000,
                    poll_every=100):
        p = QocAdamParams()
        p.rate, p.learning_rate_decay = float(rate), float(learning_rate_decay)
        p.conv_target, p.min_grad = float(conv_target), float(min_grad)
        p.max_iterations, p.poll_every = int(max_iterations), int(poll_every)
        return p

    def run_adam(self, params):
        its = np.empty(self.n_seeds, dtype=np.int32)
        _check(self._lib.qoc_run_adam(self._h, C.byref(params), its.ctypes.data_as(_IP)))
        return its

    def iterate(self, params, iters):
        _check(self._lib.qoc_iterate(self._h, C.byref(params), int(iters)))

    def sync(self):
        _check(self._lib.qoc_sync(self._h))

    def scalars(self):
        B = self.n_seeds
        loss, reg, g2, us = (np.empty(B) for _ in range(4))
        its = np.empty(B, dtype=np.int32)
        done = np.empty(B, dtype=np.int32)
        _check(self._lib.qoc_get_scalars(self._h, _dp(loss), _dp(reg), _dp(g2), _dp(us), its.ctypes.data_as(_IP),
                                         done.ctypes.data_as(_IP)))
        return dict(loss=loss, reg_loss=reg, grad_squared=g2, unitary_scale=us, iterations=its, done=done)

    # -- read-back ------------------------------------------------------------------------------------------------
    def get_uks(self, evaluated=False):
        """maxA*sin(base) of the current variable, or (evaluated=True) the controls the last evaluation ran on."""
        out = np.empty(self._seed_shape())
        fn = self._lib.qoc_get_uks_evaluated if evaluated else self._lib.qoc_get_uks
        _check(fn(self._h, _dp(out)))
        return out

    def get_final_unitary(self):
        out = np.empty((self.n_seeds, self.n, self.n), dtype=np.complex128)
        _check(self._lib.qoc_get_final_unitary(self._h, _dp(out.view(np.float64))))
        return out

    def get_inter_vecs(self):
        out = np.empty((self.n_seeds, self.steps + 1, self.n, self.m), dtype=np.complex128)
        _check(self._lib.qoc_get_inter_vecs(self._h, _dp(out.view(np.float64))))
        return out

    # -- measurement ----------------------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        _check(self._lib.qoc_profile_enable(self._h, int(on)))

    def profile_read(self):
        name = C.c_char_p()
        launches = C.c_int64()
        ms = C.c_double()
        _check(self._lib.qoc_profile_read(self._h, C.byref(name), C.byref(launches), C.byref(ms)))
        return dict(kernel=name.value.decode(), launches=launches.value, total_ms=ms.value)

    def time_iterations(self, params, iters):
        ms = C.c_double()
        _check(self._lib.qoc_time_iterations(self._h, C.byref(params), int(iters), C.byref(ms)))
        return ms.value
