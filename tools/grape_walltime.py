#!/usr/bin/env python
"""What a drop-in caller of Grape() sees (main_grape/grape.py:19, 104-129): wall time of the whole call for BASELINE configs 1, 2 and 3 -- one control set, Adam,
progress line every 100 iterations -- split into the engine's creation (qoc_create: code object, arenas, uploads), the device loop (qoc_iterate + the sync of the
poll), the polls' scalars, the read-backs of the end results, and everything else (the host pre-processing of SystemParameters, prints, HDF5 appends when saving).
First call of the process (cold: HIP runtime start, 16 MB code object) and second call (warm), save=False and -- when the interpreter has h5py -- save=True.

    python tools/grape_walltime.py [iterations]          -> profiles/r06_grape_walltime.txt is this tool's output on one MI355X
"""
import contextlib, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
import numpy as np
from quantum_optimal_control.core import hip_engine
from quantum_optimal_control.main_grape.grape import Grape
from tests.golden import cases
from tests.helpers import grape_kwargs

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
BUCKET = {}


def timed(cls, name, bucket):
    orig = getattr(cls, name)

    def wrap(self, *a, **kw):
        t0 = time.perf_counter()
        try:
            return orig(self, *a, **kw)
        finally:
            BUCKET[bucket] = BUCKET.get(bucket, 0.0) + time.perf_counter() - t0
    setattr(cls, name, wrap)


timed(hip_engine.HipEngine, '__init__', 'create')
LAST = {}
_init = hip_engine.HipEngine.__init__


def _init_and_note(self, *a, **kw):
    _init(self, *a, **kw)
    LAST['plan'] = self.plan


hip_engine.HipEngine.__init__ = _init_and_note
_scalars = hip_engine.HipEngine.scalars


def _scalars_and_note(self):
    s = _scalars(self)
    LAST['iterations'] = int(s['iterations'][0])
    return s


hip_engine.HipEngine.scalars = _scalars_and_note
timed(hip_engine.HipEngine, 'scalars', 'polls')
for n in ('iterate', 'run_adam', 'sync', 'evaluate', 'adam_step'):
    timed(hip_engine.HipEngine, n, 'device loop')
for n in ('get_uks', 'get_final_unitary', 'get_inter_vecs', 'get_base', 'close'):
    timed(hip_engine.HipEngine, n, 'read-back')
timed(hip_engine.HipEngine, 'set_base', 'create')

try:
    import h5py  # noqa: F401
    HAVE_H5 = True
except ImportError:
    HAVE_H5 = False


def one_call(c, save, plan_note):
    conv = {'rate': 0.01, 'update_step': 100, 'max_iterations': ITER, 'conv_target': 1e-14, 'learning_rate_decay': 2500}
    kw = grape_kwargs(c)
    kw['save'] = save
    tmp = None
    if save:
        tmp = tempfile.mkdtemp(prefix='qoc_walltime_')
        kw.update(file_name='walltime', data_path=tmp)
    BUCKET.clear()
    np.random.seed(c['np_seed'])
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        Grape(convergence=conv, method='Adam', **kw)
    el = time.perf_counter() - t0
    its = LAST.get('iterations', -1)
    other = el - sum(BUCKET.values())
    plan_note.append(LAST.get('plan'))
    if tmp:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return el, its, dict(BUCKET), other


def report(name, c, per_iteration_us):
    print('== %s: %d Adam iterations, update_step 100 (device loop alone: %.1f us per iteration = %.3f s)' % (name, ITER, per_iteration_us, ITER * per_iteration_us * 1e-6), flush=True)
    plans = []
    for save in ([False, True] if HAVE_H5 else [False]):
        for rep in range(2):
            el, its, b, other = one_call(c, save, plans)
            print('   save=%-5s call %d (%s): %8.3f s wall, %d iterations | create %.3f | device loop %.3f | polls %.3f | read-back %.3f | host pre-processing, prints%s %.3f'
                  % (save, rep, 'cold' if (rep == 0 and not save) else 'warm', el, its, b.get('create', 0), b.get('device loop', 0), b.get('polls', 0), b.get('read-back', 0),
                     ', HDF5' if save else '', other), flush=True)
    if not HAVE_H5:
        print('   (save=True not measured: this interpreter has no h5py)')
    print('   plan: %s' % (plans[0],))


report('C1 qubit (n=2, k=1, 100 slices)', cases.case_c1(), 8.2)
report('C2 (n=32, k=4, 500 slices, m=8)', cases.case_c2(), 73.0)
c3 = cases.case_c3()
report('C3 state transfer (n=64, k=6, 1000 slices, dwdt + forbidden)', c3, 397.0)
c8 = cases.case_c2(n=8, k=4, steps=500, m=8, taylor=(5, 3), seed=2)
report('two transmons (n=8, k=4, 500 slices)', c8, 24.5)
