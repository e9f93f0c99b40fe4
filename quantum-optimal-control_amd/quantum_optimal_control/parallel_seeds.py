"""Seed-parallel outer loop: independent random-restart control sets sharded over the GPUs of one node.

The reference optimises exactly one control set per Grape() call and draws its initial guess from NumPy's global RNG
(core/system_parameters.py:272-284); random restarts are an embarrassingly-parallel loop around it.  Here the
restarts become the leading `n_seeds` dimension of one engine per GPU (one process per GPU), block-partitioned over
ranks.  Seeds never interact, so there is NO data-path collective: the only exchange is one all-gather of the
per-seed scalars (fidelity, iterations) at the end, followed by an optional broadcast of the winner's controls.

Two transports, same partition and same results:
  * `comm`  -- a `hip_engine.QocComm`: RCCL over xGMI behind the C ABI (include/qoc.h); the all-gather runs device to
               device on the engine's own HIP stream.  No torch in the process.  The 128-byte RCCL id travels from rank 0
               to the other ranks through `rendezvous()` below (a file on the node, one node per job).
  * `dist`  -- an initialised `torch.distributed` module (gloo in the CPU tests): host tensors.
"""
import os
import time

import numpy as np


def launch_env():
    """(rank, local_rank, world) as torch.distributed.run / bench.py's own launcher export them."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def launch_key():
    """What the ranks of ONE launch share and no other launch does, in this order of preference:
      1. QOC_RDZV_KEY -- whatever the launcher says (any launcher that wraps every rank in its own shell -- srun, mpirun, a job script per GPU -- has no common parent
         process: give all ranks of the launch one value, e.g. the job id);
      2. TORCHELASTIC_RUN_ID (+ master port) when torch.distributed.run was given a real --rdzv-id (its default for a static rendezvous is the literal "none");
      3. master port + the launcher's PID (the ranks of one torch.distributed.run / bench.py launch are siblings: os.getppid() is the same number on all of them) + the
         launcher's start time in clock ticks (/proc/<pid>/stat field 22: a recycled PID of a crashed earlier launch gives another key, so its left-over files are
         never read as fresh exchanges)."""
    explicit = os.environ.get('QOC_RDZV_KEY')
    if explicit:
        return 'k' + ''.join(ch if (ch.isalnum() or ch in '-_.') else '_' for ch in explicit)[:96]
    run_id = os.environ.get('TORCHELASTIC_RUN_ID', '')
    if run_id and run_id.lower() != 'none':
        return 'e%s_%s' % (''.join(ch if (ch.isalnum() or ch in '-_.') else '_' for ch in run_id)[:64], os.environ.get('MASTER_PORT', '0'))
    ppid = os.getppid()
    start = '0'
    try:
        with open('/proc/%d/stat' % ppid) as f:
            start = f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return '%s_%d_%s' % (os.environ.get('MASTER_PORT', '0'), ppid, start)


def device_for_rank(local_rank, visible_devices):
    """HIP device index of a rank: LOCAL_RANK when the rank sees all GPUs of the node (torch.distributed.run exports no
    HIP_VISIBLE_DEVICES: every rank sees 8 devices and must pick its own), device 0 when the launcher already narrowed the rank's view to ONE
    device (HIP_VISIBLE_DEVICES=<local rank>), LOCAL_RANK modulo the count otherwise (more ranks than visible GPUs share them round-robin)."""
    local_rank, visible_devices = int(local_rank), int(visible_devices)
    if visible_devices <= 0:
        raise RuntimeError('no HIP device visible to local rank %d' % local_rank)
    if visible_devices == 1:
        return 0
    return local_rank % visible_devices


def rendezvous_timeout():
    """Seconds a rank waits for the OTHER ranks of its launch to show up (QOC_RDZV_TIMEOUT, default 60): the ranks of one node start within seconds of each other, so
    a longer silence means the key differs between them (see launch_key) or a rank died -- say so soon, with the names involved."""
    try:
        return max(1.0, float(os.environ.get('QOC_RDZV_TIMEOUT', '60')))
    except ValueError:
        return 60.0


def _where(path, key):
    return ('%s (key %r: QOC_RDZV_KEY=%r TORCHELASTIC_RUN_ID=%r MASTER_PORT=%r parent pid %d; every rank of a launch must derive the same key -- set QOC_RDZV_KEY '
            'when the ranks do not share a parent process; QOC_RDZV_TIMEOUT changes the wait)' % (path, key, os.environ.get('QOC_RDZV_KEY'),
            os.environ.get('TORCHELASTIC_RUN_ID'), os.environ.get('MASTER_PORT'), os.getppid()))


def rendezvous(rank, world, make_payload, key=None, directory=None, timeout=None):
    """Hand `make_payload()` (bytes, evaluated on rank 0 only) to every rank of a one-node job through a file.

    key: anything all ranks agree on and no other live job shares; default = master port + the launcher's PID (the
    ranks of one torch.distributed.run / bench.py launch are siblings, so os.getppid() is the same number on all of
    them and differs between concurrent launches)."""
    if world == 1:
        return make_payload()
    if key is None:
        key = launch_key()
    if timeout is None:
        timeout = rendezvous_timeout()
    directory = directory or os.environ.get('QOC_RDZV_DIR', '/tmp')
    path = os.path.join(directory, 'qoc_rdzv_%d_%s' % (os.getuid(), key))
    if rank == 0:
        payload = make_payload()
        tmp = '%s.%d.tmp' % (path, os.getpid())
        with open(tmp, 'wb') as f:
            f.write(payload)
        os.replace(tmp, path)                     # atomic: a reader sees nothing or everything
        return payload
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                data = f.read()
            if data:
                return data
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError('rendezvous: rank %d of %d saw no file from rank 0 within %.0f s: %s' % (rank, world, timeout, _where(path, key)))
        time.sleep(0.01)


def rendezvous_cleanup(rank, world, key=None, directory=None):
    """Rank 0 removes the rendezvous file (call after every rank holds the payload, e.g. behind a barrier)."""
    if world == 1 or rank != 0:
        return
    if key is None:
        key = launch_key()
    directory = directory or os.environ.get('QOC_RDZV_DIR', '/tmp')
    try:
        os.remove(os.path.join(directory, 'qoc_rdzv_%d_%s' % (os.getuid(), key)))
    except OSError:
        pass


_NO_RCCL = b'NO_RCCL:'
_OPEN_CALLS = 0          # open_comm calls of this process: every call of one launch gets its own FileComm directory


def _private_dir(path):
    """Directory only this user can write (0700), created if missing; anything somebody else planted under the (predictable) name is refused --
    lstat, not stat: a symbolic link to a private directory of this user would pass an ownership check that follows it, and close() removes files there."""
    import stat
    parent = os.path.dirname(os.path.abspath(path))
    if parent and not os.path.isdir(parent):
        os.makedirs(parent, exist_ok=True)           # a QOC_RDZV_DIR that does not exist yet (or is nested): only the LEAF is the private directory checked below
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if stat.S_ISLNK(st.st_mode) or not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise PermissionError('%s exists and is not a private directory of uid %d' % (path, os.getuid()))


class FileComm(object):
    """Host-side stand-in with QocComm's interface for a one-node job whose RCCL cannot be opened (no librccl next to the
    HIP runtime, a process bound to PyTorch's private runtime, ...): the ranks exchange small float64 arrays through files
    in a per-launch directory.  The seeds of a restart batch never interact, so the only traffic of the data path is one
    gather of per-seed scalars and one broadcast of the winner -- this transport costs milliseconds where RCCL costs
    microseconds, and nothing inside the iterations."""

    def __init__(self, rank, world, key, reason='', directory=None, timeout=600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.device = int(os.environ.get('LOCAL_RANK', 0))        # GrapeSharded reads the GPU of this rank from its communicator
        self.library = 'files (host)' + ((': ' + reason) if reason else '')
        self._dir = os.path.join(directory or os.environ.get('QOC_RDZV_DIR', '/tmp'), 'qoc_fc_%d_%s' % (os.getuid(), key))
        self._key = key
        _private_dir(self._dir)
        self._seq = 0

    def _path(self, seq, rank):
        return os.path.join(self._dir, '%d_%d.npy' % (seq, rank))

    def _exchange(self, array):
        """Every rank contributes one array; returns the list of all of them, in rank order."""
        seq, self._seq = self._seq, self._seq + 1
        tmp = self._path(seq, self.rank) + '.tmp'
        with open(tmp, 'wb') as f:
            np.save(f, np.ascontiguousarray(np.asarray(array, dtype=np.float64)))
        os.replace(tmp, self._path(seq, self.rank))          # atomic: a reader sees nothing or everything
        rows, t0 = [], time.time()
        for r in range(self.world):
            while True:
                try:
                    with open(self._path(seq, r), 'rb') as f:
                        rows.append(np.load(f))
                    break
                except (FileNotFoundError, ValueError, EOFError):
                    if time.time() - t0 > self.timeout:
                        raise TimeoutError('FileComm: rank %d of %d saw nothing from rank %d in exchange %d within %.0f s: %s' % (
                            self.rank, self.world, r, seq, self.timeout, _where(self._path(seq, r), self._key)))
                    time.sleep(0.001)
        # whoever wrote exchange seq - 1 had read all of seq - 2, and this rank has just seen every file of seq - 1 or later
        if seq >= 2:
            try:
                os.remove(self._path(seq - 2, self.rank))
            except OSError:
                pass
        return rows

    def all_gather(self, values):
        return np.stack(self._exchange(np.asarray(values, dtype=np.float64).reshape(-1)))

    def all_gather_scalar(self, engine, which, width):
        s = engine.scalars()[('loss', 'reg_loss', 'grad_squared', 'unitary_scale')[which]]
        buf = np.zeros(int(width))
        buf[:s.shape[0]] = s
        return self.all_gather(buf)

    def all_reduce_max(self, values):
        return np.max(np.stack(self._exchange(np.asarray(values, dtype=np.float64).reshape(-1))), axis=0)

    def broadcast(self, array, root):
        array = np.asarray(array, dtype=np.float64)
        rows = self._exchange(array if self.rank == int(root) else np.zeros(0))
        return rows[int(root)].reshape(array.shape)

    def barrier(self):
        self._exchange(np.zeros(1))

    def close(self):
        if self._dir is None:
            return
        self.barrier()
        # a rank says goodbye only after it has read the last exchange; whoever sees every goodbye is the last one out: it removes the files of
        # THIS transport -- <seq>_<rank>.npy and bye_<rank> of the ranks of this communicator, nothing else that may sit in the directory -- and
        # the then empty directory (two ranks may both see all goodbyes: removals that find nothing are fine; rmdir fails on foreign files)
        import re
        open(os.path.join(self._dir, 'bye_%d' % self.rank), 'wb').close()
        try:
            byes = ['bye_%d' % r for r in range(self.world)]
            if all(os.path.exists(os.path.join(self._dir, x)) for x in byes):
                mine = re.compile(r'^(\d+_(\d+)\.npy(\.tmp)?|bye_(\d+))$')
                for x in os.listdir(self._dir):
                    hit = mine.match(x)
                    if hit and int(hit.group(2) or hit.group(4)) < self.world:
                        try:
                            os.remove(os.path.join(self._dir, x))
                        except OSError:
                            pass
                os.rmdir(self._dir)
        except OSError:
            pass
        self._dir = None


class RcclRequired(RuntimeError):
    """open_comm(require_rccl=True) / QOC_REQUIRE_RCCL=1: some rank could not use RCCL and a file-transport line must not stand in for it."""


def open_comm(rank=None, world=None, device=None, key=None, require_rccl=None, call_index=None):
    """Communicator for this rank (None for a single process): RCCL behind the C ABI (hip_engine.QocComm), or -- when some rank
    cannot use RCCL, or QOC_TRANSPORT=file -- the file transport above (FileComm, same interface), with a warning on stderr and
    `comm.fallback_reason` set.  Collective: every rank of the launch calls it.

    Order of events (a rank must never wait inside ncclCommInitRank for a rank that failed before reaching it):
      1. every rank checks its LOCAL preconditions (librccl loadable, device usable: hip_engine.comm_probe) -- no collective;
      2. the ranks agree on the outcome through the file transport; one failure sends everybody to it;
      3. rank 0 creates the RCCL id and hands it out through the same transport; every rank enters ncclCommInitRank;
      4. the ranks agree again (the initialisation itself may fail on some rank: IPC handles, device binding).

    require_rccl (default: QOC_REQUIRE_RCCL=1 in the environment): no fallback -- every rank raises RcclRequired with the reason (the verdict is
    collective, so nobody is left waiting).  A first run on a new multi-GPU node should fail loudly on an xGMI / RCCL problem.
    call_index: which open_comm call of the launch this is (each gets its own exchange directory); default = a per-process counter, which
    assumes every rank has called open_comm equally often -- pass it (or QOC_COMM_CALL) when that is not so.
    HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, the only kind this host driver supports) must be in the environment BEFORE the HIP runtime
    starts: hip_engine.load_library() sets it for multi-rank launches (WORLD_SIZE > 1); a process that loaded the library earlier without
    it is told so here."""
    global _OPEN_CALLS
    import sys
    from quantum_optimal_control.core import hip_engine
    if require_rccl is None:
        require_rccl = os.environ.get('QOC_REQUIRE_RCCL', '0') == '1'
    if 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ:
        if hip_engine.library_loaded():
            sys.stderr.write('quantum_optimal_control.parallel_seeds: WARNING: libqoc_hip was loaded before HSA_ENABLE_IPC_MODE_LEGACY=0 was set; '
                             'if the HIP runtime has already started, RCCL cannot exchange IPC handles (export it before the first engine call)\n')
        os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    erank, elocal, eworld = launch_env()
    rank = erank if rank is None else rank
    world = eworld if world is None else world
    device = elocal if device is None else device
    if world == 1:
        return None
    if key is None:
        key = launch_key()
    if call_index is None:
        call_index = os.environ.get('QOC_COMM_CALL')
    if call_index is None:
        call_index, _OPEN_CALLS = _OPEN_CALLS, _OPEN_CALLS + 1
    # the agreement below is a matter of seconds: a rank that waits longer is waiting for files nobody writes (other key, dead rank); the data-phase exchanges of
    # a fallback transport (ranks finish minutes apart) get the long timeout back
    files = FileComm(rank, world, '%s_c%d' % (key, int(call_index)), timeout=rendezvous_timeout())
    files.fallback_reason = None

    def fall_back(reason):
        if (rank == 0 or require_rccl) and reason != 'QOC_TRANSPORT=file':
            describe_node(sys.stderr, device)            # what a first run on a new multi-GPU node needs beside the reason: devices, peer access, librccl
        if require_rccl:
            try:
                files.close()
            except Exception:
                pass
            raise RcclRequired('RCCL is required (require_rccl / QOC_REQUIRE_RCCL=1) and is not usable: %s' % reason)
        files.library = 'files (host): ' + reason
        files.fallback_reason = reason
        files.timeout = 600.0
        if rank == 0 and os.environ.get('QOC_TRANSPORT', 'rccl') != 'file':
            sys.stderr.write('quantum_optimal_control.parallel_seeds: WARNING: RCCL is NOT in use, the ranks exchange their results '
                             'through files (%s)\n' % reason)
        return files

    # 1 + 2: local checks, then agreement
    why = ''
    if os.environ.get('QOC_TRANSPORT', 'rccl') == 'file':
        why = 'QOC_TRANSPORT=file'
    else:
        try:
            hip_engine.comm_probe(device)
        except Exception as exc:                    # no librccl, PyTorch's private HIP runtime, bad device index ...
            why = str(exc)[:400]
    ok = files.all_gather([0.0 if why else 1.0]).reshape(-1)
    if not np.all(ok > 0.5):
        bad = [int(r) for r in np.nonzero(ok <= 0.5)[0]]
        return fall_back(why if (why and len(bad) == world) else 'RCCL unusable on rank(s) %s%s' % (bad, ('; rank %d: %s' % (rank, why)) if why else ''))
    # 3: the id (128 bytes travel as 128 small floats; a failure to create it is announced with a NaN)
    uid_row = np.full(hip_engine.COMM_ID_BYTES, np.nan)
    if rank == 0:
        try:
            uid_row = np.frombuffer(hip_engine.comm_unique_id(), dtype=np.uint8).astype(np.float64)
        except Exception as exc:
            why = str(exc)[:400]
    uid_row = files.broadcast(uid_row, 0)
    if np.any(np.isnan(uid_row)):
        return fall_back('rank 0 could not create the RCCL id' + (': ' + why if why else ''))
    uid = uid_row.astype(np.uint8).tobytes()
    comm = None
    try:
        comm = hip_engine.QocComm(uid, world, rank, device)
    except Exception as exc:
        why = 'rank %d: %s' % (rank, exc)
    # 4
    ok = files.all_gather([1.0 if comm is not None else 0.0]).reshape(-1)
    if np.all(ok > 0.5):
        files.close()
        comm.fallback_reason = None
        comm.barrier()
        return comm
    bad = [int(r) for r in np.nonzero(ok <= 0.5)[0]]
    if comm is not None:
        try:
            comm.close()
        except Exception:
            pass
    return fall_back('RCCL initialisation failed on rank(s) %s%s' % (bad, ('; ' + why) if why else ''))


def describe_node(out, device=None):
    """Devices, the hipDeviceCanAccessPeer matrix (what RCCL's device-to-device transports between two ranks of a node need) and the RCCL library of this process,
    one block of text: printed when RCCL cannot start (tools/multi_gpu_selftest.py prints the same on purpose)."""
    from quantum_optimal_control.core import hip_engine
    try:
        n = hip_engine.device_count()
        out.write('quantum_optimal_control.parallel_seeds: node as rank %s (HIP device %s) sees it: %d HIP device(s); HSA_ENABLE_IPC_MODE_LEGACY=%r HIP_VISIBLE_DEVICES=%r '
                  'ROCR_VISIBLE_DEVICES=%r QOC_RCCL_LIBRARY=%r\n' % (os.environ.get('RANK', '0'), device, n, os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                                                                      os.environ.get('HIP_VISIBLE_DEVICES'), os.environ.get('ROCR_VISIBLE_DEVICES'), os.environ.get('QOC_RCCL_LIBRARY')))
        for d in range(n):
            info = hip_engine.device_info(d)
            row = ''.join('1' if hip_engine.device_peer_access(d, p) else '0' for p in range(n))
            out.write('  device %d: %s, %d CUs, %.0f GB; can access peers: %s\n' % (d, info['name'], info['compute_units'], info['hbm_bytes'] / 2.0 ** 30, row))
        lib = hip_engine.load_library().qoc_comm_library().decode()
        out.write('  RCCL library in use: %s\n' % (lib or '(none opened yet)'))
    except Exception as exc:                             # diagnostics must never replace the error they accompany
        out.write('quantum_optimal_control.parallel_seeds: (node description failed: %r)\n' % (exc,))
    out.flush()


def _dist_device(dist):
    """Where the tensors of a `dist=` exchange live: the current GPU for the nccl (= RCCL) backend, the host otherwise (gloo)."""
    import torch
    try:
        if dist.get_backend() == 'nccl':
            return torch.device('cuda', torch.cuda.current_device())
    except Exception:
        pass
    return torch.device('cpu')


class SeedShard(object):
    """Block partition of `total_seeds` over `world` ranks: rank r owns [first, first + count)."""

    def __init__(self, total_seeds, rank=0, world=1):
        if not (0 <= rank < world):
            raise ValueError('rank %d outside world %d' % (rank, world))
        self.total, self.rank, self.world = int(total_seeds), int(rank), int(world)
        base, extra = divmod(self.total, self.world)
        self.counts = [base + (1 if r < extra else 0) for r in range(self.world)]
        self.firsts = [int(np.sum(self.counts[:r])) for r in range(self.world)]
        self.first, self.count = self.firsts[self.rank], self.counts[self.rank]

    def _rows_to_global(self, rows):
        return np.concatenate([np.asarray(rows[r])[:self.counts[r]] for r in range(self.world)])

    def all_gather(self, local_values, dist=None, comm=None):
        """Gather one float64 per seed from every rank, returned in global seed order (length total_seeds)."""
        local_values = np.asarray(local_values, dtype=np.float64).reshape(-1)
        assert local_values.shape[0] == self.count
        if self.world == 1 or (dist is None and comm is None):
            return local_values.copy()
        width = max(self.counts)
        if comm is not None:
            buf = np.zeros(width)
            buf[:self.count] = local_values
            return self._rows_to_global(comm.all_gather(buf))
        import torch
        dev = _dist_device(dist)
        buf = torch.zeros(width, dtype=torch.float64)
        buf[:self.count] = torch.from_numpy(local_values)
        buf = buf.to(dev)
        out = [torch.zeros(width, dtype=torch.float64, device=dev) for _ in range(self.world)]
        dist.all_gather(out, buf)
        return self._rows_to_global([o.cpu().numpy() for o in out])

    def all_gather_engine_scalar(self, engine, which, comm):
        """The same gather straight from the engine's device array `which` (hip_engine.SCALAR_*), device to device on
        the engine's stream behind the iterations already enqueued (RCCL); global seed order."""
        if self.world == 1 or comm is None:
            s = engine.scalars()
            return np.asarray(s[('loss', 'reg_loss', 'grad_squared', 'unitary_scale')[which]], dtype=np.float64).copy()
        return self._rows_to_global(comm.all_gather_scalar(engine, which, max(self.counts)))

    def owner_of(self, seed):
        for r in range(self.world):
            if self.firsts[r] <= seed < self.firsts[r] + self.counts[r]:
                return r
        raise IndexError(seed)

    def broadcast_from_owner(self, seed, local_array_fn, shape, dist=None, comm=None):
        """Broadcast a float64 array (e.g. the winner's uks) from the rank that owns `seed` to all ranks."""
        owner = self.owner_of(seed)
        if self.world == 1 or (dist is None and comm is None):
            return np.asarray(local_array_fn(seed - self.first), dtype=np.float64).reshape(shape)
        if self.rank == owner:
            mine = np.ascontiguousarray(local_array_fn(seed - self.first), dtype=np.float64).reshape(shape)
        else:
            mine = np.zeros(shape)
        if comm is not None:
            return comm.broadcast(mine, owner).reshape(shape)
        import torch
        t = torch.from_numpy(mine.copy()).to(_dist_device(dist))
        dist.broadcast(t, src=owner)
        return t.cpu().numpy()


def restart_guesses(k, steps, first_seed, count, base_seed=1000):
    """Initial optimisation variables N(0, 1/sqrt(steps)) (same law as system_parameters.py:279-282), one
    independent, reproducible stream per global seed index."""
    return np.stack([np.random.default_rng(base_seed + first_seed + i).normal(0, 1. / np.sqrt(steps), (k, steps))
                     for i in range(count)]) if count > 0 else np.zeros((0, k, steps))


def select_best(fidelities):
    """Index of the best seed (ties -> lowest index, deterministic on every rank)."""
    return int(np.argmax(np.asarray(fidelities)))
