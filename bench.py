#!/usr/bin/env python
"""bench.py -- GRAPE iterations/s (forward chain + first-order backward + regularisers + Adam) on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from rank 0.
  * workload (BASELINE.json configs[1] / SURVEY.md 8d "C2"): n=32, k=4, time slices=500, m=8, (T,s)=(5,3), fp64
    complex, synthetic Hamiltonians from numpy.random.default_rng(0); SEEDS_PER_GPU independent random-restart control
    sets per GPU (config 4: 512 seeds / 8 GPUs = 64 per GPU) -> weak scaling, no data-path collective; the final
    per-seed fidelities are all-gathered once over RCCL (C ABI, qoc_comm_all_gather_scalar: device to device on the
    engine's stream) after the last iteration, inside the timed region.
  * launch: one process per GPU.  Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks come from
    RANK / LOCAL_RANK / WORLD_SIZE; a plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE) starts the N ranks itself.
    No torch in the process: the RCCL id travels through parallel_seeds.rendezvous (a file on the node).
  * a "step" = one GRAPE iteration of every seed on the GPU (all inputs resident in HBM).
  * value = (seeds on all GPUs) * K / wall time, wall = max over ranks, barrier + device sync on both sides.
  * roofline: dominant kernel (the MFMA-path exponential kernel: matrix exponentials + chunk products) timed with hipEvents
    on the engine's stream in a separate pass.  `achieved` / `frac` count the MFMA work the kernel EXECUTES (Paterson-Stockmeyer
    product count x 3 real products per complex product); the SURVEY.md 8d algorithmic count (plain Taylor, 4 real products)
    is reported beside it as `achieved_algorithmic` / `frac_algorithmic`.
  * cpu_baseline: the compiled C restatement of the oracle (oracle/qoc_oracle.c, OpenMP, one seed per thread, ONE
    evaluation per iteration) timed on a bounded sample on this host -- a reported baseline, not the target.
"""
import argparse
import json
import os
import socket
import subprocess
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
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PMC_TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r06_pmc_traffic.json')


def build_problem():
    from quantum_optimal_control.helper_functions import synthetic_systems
    c = synthetic_systems.case_c2(n=N, k=K_OPS, steps=SLICES, m=M, taylor=TAYLOR, seed=0)
    H0, Hops, U = c['H0'], c['Hops'], c['U']
    dt = c['total_time'] / SLICES
    Hs = np.stack([-1j * dt * H0] + [-1j * dt * h for h in Hops])
    V = np.eye(N, dtype=complex)[:, :M]
    W = U @ V
    return c, Hs, np.eye(N, dtype=complex), V, W, dt


def seed_bases(first, count):
    return np.stack([np.random.default_rng(1000 + first + i).normal(0, 1 / np.sqrt(SLICES), (K_OPS, SLICES))
                     for i in range(count)])


def executed_products_per_slice(T, s):
    """Complex n x n products per time slice as the MFMA-path exponential kernels execute them (qoc_mfma_expm.h): A^2, the
    Horner steps of the Paterson-Stockmeyer form over A^2, s squarings, one running chunk product.  (T, s) = (5, 3): 7."""
    poly = 0
    if T >= 2:
        mm = T >> 1
        poly = 1 + (mm - 1 if T % 2 == 0 else mm)
    return poly + s + 1


def oracle_problem():
    """The same C2 inputs through the CPU oracle's pre-processing (cpu_baseline leg only)."""
    from oracle import grape_oracle as go
    from quantum_optimal_control.helper_functions import synthetic_systems
    c = synthetic_systems.case_c2(n=N, k=K_OPS, steps=SLICES, m=M, taylor=TAYLOR, seed=0)
    np.random.seed(c['np_seed'])
    return go.OracleSystem(c['H0'], c['Hops'], c['U'], c['total_time'], c['steps'], c['states_concerned_list'],
                           U0=c['U0'], reg_coeffs=c['reg_coeffs'], dressed_info=None, maxA=c['maxA'],
                           initial_guess=c['initial_guess'], state_transfer=c['state_transfer'],
                           Taylor_terms=c['Taylor_terms'])


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(budget_s=15.0):
    """CPU oracle timed on a bounded sample of the same workload: whole iterations (evaluate + Adam) of one seed per
    thread, all host threads busy (seeds are the parallel axis on the CPU too).  Compiled C port (oracle/qoc_oracle.c)
    when its .so is present, else the NumPy oracle on one thread."""
    sp = oracle_problem()
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
                'per_thread_value': iters / el, 'cpu_model': cpu_model(), 'logical_cpus': os.cpu_count()}
    except OSError:
        pass
    from oracle import grape_oracle as go
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(limits=1)
    except Exception:
        pass
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
    return {'value': its / el, 'unit': 'GRAPE iterations/s', 'cores': 1, 'kind': 'port', 'cpu_model': cpu_model(), 'logical_cpus': os.cpu_count(),
            'sample': '%d iterations of 1 seed of the same C2 workload in %.1f s (NumPy complex128 oracle, BLAS limited '
                      'to one thread)' % (its, el)}


def _reference_ops_worker(args):
    """One process of the all-core B-faithful leg: `evals` fp32 real-embedded graph evaluations of its own seed on `threads` threads."""
    seed, evals, threads = args
    import torch
    from oracle import tf_graph_emulation as tfe
    torch.set_num_threads(threads)
    sp = oracle_problem()
    base = seed_bases(seed, 1)[0]
    tfe.evaluate_graph(sp, base, dtype=torch.float32)               # warm-up
    t0 = time.perf_counter()
    for _ in range(evals):
        tfe.evaluate_graph(sp, base, dtype=torch.float32)
    return time.perf_counter() - t0


def cpu_baseline_reference_ops(max_evals=6, budget_s=10.0):
    """BASELINE.md "B-faithful": the reference's own op sequence -- real-embedded 2n x 2n float32 matrices, one Defun per
    slice with recompute inside the gradient function, TWO graph evaluations per iteration (run_session.py:53-54, 69) --
    emulated node for node in torch-CPU (oracle/tf_graph_emulation.py).  Not TensorFlow (absent).
    Two readings: ONE control set on 16 threads (what a Grape() call of the reference occupies: 64 x 64 matmuls do not scale
    further), and ALL cores busy with one control set per 4-thread process (the throughput reading, comparable with `value`)."""
    import torch
    from oracle import tf_graph_emulation as tfe
    sp = oracle_problem()
    torch.set_num_threads(min(16, torch.get_num_threads()))        # 2n x 2n = 64 x 64 matmuls: more threads only thrash
    base = seed_bases(0, 1)[0]
    tfe.evaluate_graph(sp, base, dtype=torch.float32)               # warm-up
    t0 = time.perf_counter()
    evals = 0
    while evals < max_evals and time.perf_counter() - t0 < budget_s:
        tfe.evaluate_graph(sp, base, dtype=torch.float32)
        evals += 1
    el = time.perf_counter() - t0
    out = {'value': evals / el / 2.0, 'unit': 'GRAPE iterations/s', 'cores': int(torch.get_num_threads()), 'cpu_model': cpu_model(),
           'kind': 'port', 'sample': '%d fp32 real-embedded graph evaluations (fwd + custom-gradient bwd) of 1 seed of C2 in '
                                     '%.1f s; 2 evaluations per reference iteration; torch-CPU emulation of the TF graph' % (evals, el)}
    try:                                                            # all cores: one control set per process of 4 threads
        per = 4
        procs = max(1, min(64, (os.cpu_count() or 4) // per))
        w_evals = 2                                                 # 64 processes x 4 threads share the memory system: ~5 s per evaluation each
        t0 = time.perf_counter()
        kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--reference-ops-worker', '%d,%d,%d' % (i, w_evals, per)],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
        times, deadline = [], time.time() + 120.0                  # own processes, own PIDs: a straggler is killed, never waited for
        for kid in kids:
            try:
                so, _ = kid.communicate(timeout=max(1.0, deadline - time.time()))
                times.append(float(so.strip().splitlines()[-1]))
            except Exception:
                kid.kill()
                kid.communicate()
        wall = time.perf_counter() - t0
        if len(times) < procs:
            raise RuntimeError('%d of %d worker processes did not finish' % (procs - len(times), procs))
        busy = max(times)
        out['all_cores'] = {'value': procs * w_evals / busy / 2.0, 'unit': 'GRAPE iterations/s', 'cores': procs * per, 'processes': procs,
                            'sample': '%d processes x %d threads, %d evaluations each of their own control set; slowest process %.1f s '
                                      '(%.1f s with start-up)' % (procs, per, w_evals, busy, wall)}
    except Exception as exc:
        out['all_cores'] = {'error': repr(exc)}
    return out


def _engine_for(c, n_seeds, device):
    """An engine of a synthetic_systems recipe through the PRODUCT's host pre-processing (core/system_parameters.py mirror of the reference's
    SystemParameters) -- no oracle on this leg."""
    from quantum_optimal_control.core import hip_engine
    from quantum_optimal_control.core.system_parameters import SystemParameters
    import contextlib
    np.random.seed(c['np_seed'])
    U0 = np.identity(len(c['H0'])) if c['U0'] is None else c['U0']
    with contextlib.redirect_stdout(sys.stderr):                  # the reference prints its Taylor-term choice (system_parameters.py:216): stdout is the JSON line's
        sp = _system_parameters(SystemParameters, c, U0)
    Hs, U0e, V, W, Vs = sp.engine_inputs()
    eng = hip_engine.HipEngine(Hs, U0e, V, W, sp.ops_max_amp, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
                               state_transfer=sp.state_transfer, reg_coeffs=sp.reg_coeffs, Vs=Vs, n_seeds=n_seeds, device=device)
    k, steps = len(c['Hops']), c['steps']
    eng.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(steps), (n_seeds, k, steps)))
    return eng, sp


def _system_parameters(SystemParameters, c, U0):
    return SystemParameters(c['H0'], c['Hops'], c['Hnames'], c['U'], U0, c['total_time'], c['steps'], c['states_concerned_list'], None,
                          np.asarray(c['maxA'], dtype=float), None, None, False, 1e-4, c['state_transfer'], False, c['reg_coeffs'], False, None,
                          c['Taylor_terms'], True, True, False, False, False)


def _time_engine(eng, params, warm, iters):
    eng.iterate(params, warm); eng.sync()
    t0 = time.perf_counter()
    eng.iterate(params, iters); eng.sync()
    per = (time.perf_counter() - t0) / iters
    eng.profile_enable(True)
    eng.iterate(params, max(1, min(iters, 5)))
    pr = eng.profile_read()
    eng.profile_enable(False)
    sc = eng.scalars()
    assert int(np.sum(sc['done'])) == 0, 'a control set stopped inside the timed iterations'
    return per, pr, sc


def secondary_configs(device):
    """BASELINE.json configs 3 and 5 (SURVEY.md 8d "C3", "C5") through the same C ABI, AFTER the timed region, rank 0 at N = 1 only: one C3 trajectory
    (the reference's own calling mode), 64 and 256 C3 control sets, one C5 iteration.  Per entry: ms per iteration of the batch, the plan AUTO resolved, the kernel
    the engine's hipEvent bracket names with its average time, and TFLOP/s on SURVEY 8d's algorithmic count (an algorithmic rate: the kernels execute
    fewer flops -- Paterson-Stockmeyer, three-multiplication complex products -- so this is NOT a utilisation)."""
    from quantum_optimal_control.core import hip_engine
    from quantum_optimal_control.helper_functions import synthetic_systems
    params = hip_engine.HipEngine.adam_params(rate=0.01, learning_rate_decay=2500, conv_target=-1.0, min_grad=-1.0, max_iterations=10 ** 9,
                                              poll_every=10 ** 9)
    out = {}

    def entry(name, c, seeds, warm, iters, flops_alg, roof=None):
        """roof(per_iteration_s, bracketed_ms_per_iteration, eng) -> the entry's `roofline` dict: what bounds the configuration and how far below that bound it runs."""
        t0 = time.perf_counter()
        try:
            eng, sp = _engine_for(c, seeds, device)
            try:
                per, pr, sc = _time_engine(eng, params, warm, iters)
                brk = pr['total_ms'] / max(1, min(iters, 5))
                out[name] = {'ms_per_iteration': per * 1e3, 'iterations_per_s': seeds / per, 'control_sets': seeds, 'timed_iterations': iters,
                             'path': eng.path, 'plan': eng.plan, 'bracketed_kernel': pr['kernel'],
                             'bracketed_kernel_ms_per_iteration': brk, 'bracketed_launches': pr['launches'],
                             'algorithmic_TFLOPs': seeds * flops_alg / per / 1e12, 'loss0': float(sc['loss'][0]),
                             'wall_s': time.perf_counter() - t0}
                if roof is not None:
                    out[name]['roofline'] = roof(per, brk, eng)
            finally:
                eng.close()
        except Exception as exc:                                   # a secondary entry never takes the bench line down with it
            out[name] = {'error': repr(exc)}

    def frac_of(achieved, peak, unit, bound, what):
        return {'bound': bound, 'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak, 'counting': what}

    c3 = synthetic_systems.case_c3()
    T3 = c3['Taylor_terms'][0]
    f3 = c3['steps'] * 1 * 8.0 * 64 ** 2 * (3 * (T3 - 1) + 6)      # steps*m*8n^2*[3(T-1)+k]
    # one C3 trajectory: propagator route; the bracket is k_gemm_expm_fused (+ product tree): 4 products per slice of the degree-9 / no-squaring form, three real
    # products per complex one on v_mfma_f64_16x16x4 (6 n^3 flops each), against the fp64 matrix peak
    entry('c3_single_trajectory', c3, 1, 200, 400, f3,
          lambda per, brk, eng: frac_of(c3['steps'] * 4 * 6.0 * 64 ** 3 / (brk * 1e-3) / 1e12, FP64_MATRIX_PEAK_TFLOPS, 'TFLOP/s', 'mfma',
                                        'EXECUTED flops of the bracketed exponential kernel (4 products per slice x 6 n^3) / its hipEvent time; the iteration is '
                                        'launch- and chain-latency-bound beyond it (DESIGN.md 4.2)'))
    # C3 x 64: two dependent Taylor chains per control set, one workgroup each (64 of 256 CUs): executed flops of the chains (9 mat-vecs per slice, three-multiplication
    # complex MACs: 6 n^2 each, forward + backward) and of the gradient products (k wide products: 8 n^2 per control and slice) over the ITERATION, against the fp64
    # vector peak -- a latency-bound configuration, so a small number
    entry('c3_x64', c3, 64, 10, 30, f3,
          lambda per, brk, eng: frac_of(64 * c3['steps'] * (2 * (T3 - 1) * 6.0 * 64 ** 2 + 6 * 8.0 * 64 ** 2) / per / 1e12, FP64_MATRIX_PEAK_TFLOPS, 'TFLOP/s', 'mfma',
                                        'EXECUTED flops of an iteration (2 x 9 dependent 64 x 64 mat-vecs per slice at 6 n^2 + 6 gradient products at 8 n^2) / iteration time; '
                                        'one workgroup per control set: latency-bound by construction (DESIGN.md 4.2, profiles/r05_chain_micro_probe.txt)'))
    # C3 x 256: bound by the packed generators (40 KB per slice) written once by the assembly and read by both chains
    entry('c3_x256', c3, 256, 5, 15, f3,
          lambda per, brk, eng: frac_of(3 * 256 * c3['steps'] * 2560 * 16.0 / per / 1e9, HBM_PEAK_GBS, 'GB/s', 'hbm',
                                        'ALGORITHMIC bytes of an iteration (packed generators, 40 KB per slice and control set: written once, read twice) / iteration time'))
    c5 = synthetic_systems.case_c2(n=512, k=8, steps=2000, m=8, taylor=(5, 3), seed=2)
    f5 = 2000 * ((5 - 1 + 3) + 1 + 2) * 8.0 * 512 ** 3 + 2000 * 8 * 8.0 * 512 ** 2   # SURVEY 8d unitary count
    # C5: the batched products of the exponentials + product tree (k_zgemm_wg): executed_products_per_slice(5, 3) = 7 products of 6 n^3 per slice, + the wide
    # gradient product (k x m columns per slice), over the iteration
    entry('c5_single_trajectory', c5, 1, 1, 3, f5,
          lambda per, brk, eng: frac_of((2000 * executed_products_per_slice(5, 3) * 6.0 * 512 ** 3 + 2000 * 8 * 8 * 6.0 * 512 ** 2) / per / 1e12, FP64_MATRIX_PEAK_TFLOPS,
                                        'TFLOP/s', 'mfma', 'EXECUTED MFMA flops of an iteration (7 products per slice x 6 n^3 + the wide gradient product) / iteration time'))
    # the sizes the reference is used at, on the workgroup-resident path (row g2 of the round-5 verdict): BASELINE config 1 with one control set and with 64, and one
    # control set of two transmons (n = 8, k = 4, 500 slices).  Bound: the latency of the dependent chain inside ONE launch; the fraction is executed fp64 FMA flops
    # (complex products of n^2 MACs per lane row) against the vector peak -- small by construction, the number to watch is us per iteration
    from tests.golden import cases
    def small_roof(n, steps, T, s, L_products_extra):
        return lambda per, brk, eng: frac_of(eng.n_seeds * steps * (max(T - 1, 0) + s + L_products_extra) * 8.0 * n ** 3 / per / 1e12, FP64_MATRIX_PEAK_TFLOPS, 'TFLOP/s', 'latency',
                                             'EXECUTED complex-product flops of an iteration (Taylor + squarings + chain / tree / sweeps: 8 n^3 each) / iteration time; the whole iteration is one '
                                             'launch of dependent phases (csrc/qoc_small_kernel.h): us_per_iteration is the figure of merit')
    c1 = cases.case_c1()
    entry('c1_single_trajectory', c1, 1, 2000, 4000, c1['steps'] * 8.0 * 2 ** 3 * 6, small_roof(2, 100, 4, 0, 5))
    entry('c1_x64', c1, 64, 2000, 4000, c1['steps'] * 8.0 * 2 ** 3 * 6, small_roof(2, 100, 4, 0, 5))
    c8 = synthetic_systems.case_c2(n=8, k=4, steps=500, m=8, taylor=(5, 3), seed=2)
    entry('n8_single_trajectory', c8, 1, 1000, 2000, 500 * ((5 - 1 + 3) + 1 + 2) * 8.0 * 8 ** 3, small_roof(8, 500, 5, 3, 5))
    for key in ('c1_single_trajectory', 'c1_x64', 'n8_single_trajectory'):
        if 'ms_per_iteration' in out.get(key, {}):
            out[key]['us_per_iteration'] = out[key]['ms_per_iteration'] * 1e3
    out['note'] = ('BASELINE configs 3 (state transfer n=64 k=6 steps=1000 m=1 T=10, dwdt + two forbidden levels) and 5 (n=512 k=8 steps=2000 m=8 '
                   '(T,s)=(5,3)); synthetic_systems recipes, product pre-processing, AUTO path; algorithmic_TFLOPs uses SURVEY 8d counts (not a utilisation)')
    return out


class LivePmc(object):
    """roofline.traffic measured in THIS run when rocprofv3 is on PATH: two separate PMC passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md's HBM
    recipe: own runs, --kernel-trace only, gfx950 correction 2 x FETCH_SIZE, both in KB) of this same command at 3 steps, started as a background
    process chain beside the CPU baseline legs (counters are not timing-sensitive).  Falls back to the committed file."""

    def __init__(self, args):
        import shutil
        import tempfile
        self.exe = shutil.which('rocprofv3')
        self.proc = None
        if self.exe is None:
            return
        self.dir = tempfile.mkdtemp(prefix='qoc_pmc_', dir='/tmp')
        inner = '%s %s --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-single --no-secondary --no-live-pmc --seeds-per-gpu %d --chunks %d' % (
            sys.executable, os.path.abspath(__file__), args.seeds_per_gpu, args.chunks)
        cmd = ' ; '.join('%s --pmc %s --kernel-trace -d %s/%s -- %s > %s/%s.log 2>&1' % (self.exe, ctr, self.dir, ctr, inner, self.dir, ctr)
                         for ctr in ('FETCH_SIZE', 'WRITE_SIZE'))
        env = dict(os.environ, TMPDIR='/tmp')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        self.proc = subprocess.Popen(['bash', '-c', cmd], cwd='/tmp', env=env, start_new_session=True)

    @staticmethod
    def _per_kernel(db_path, counter):
        import sqlite3
        db = sqlite3.connect(db_path)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        ev = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
        info = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
        disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
        sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
        scols = [r[1] for r in db.execute('pragma table_info(%s)' % sym)]
        name_col = 'kernel_name' if 'kernel_name' in scols else 'display_name'
        q = ('select s.%s, e.value from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id join %s i on e.pmc_id = i.id '
             'where i.name = ?' % (name_col, ev, disp, sym, info))
        acc = {}
        for kname, val in db.execute(q, (counter,)):
            acc.setdefault(kname, []).append(val)
        return {k: sum(v) / len(v) for k, v in acc.items()}

    def result(self, kernel, timeout_s=150.0):
        """(bytes per launch of `kernel`, description) or (None, reason)."""
        import glob
        import shutil
        if self.proc is None:
            return None, 'rocprofv3 not on PATH'
        try:
            try:
                self.proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                # our own process group (start_new_session), never a pattern: TERM, wait for the children (a rocprofv3 GPU job must not outlive the JSON line), then KILL
                for sig, grace in ((15, 10.0), (9, 5.0)):
                    try:
                        os.killpg(self.proc.pid, sig)
                    except OSError:
                        break
                    try:
                        self.proc.wait(timeout=grace)
                        break
                    except subprocess.TimeoutExpired:
                        continue
                return None, 'PMC passes did not finish in %.0f s' % timeout_s
            vals = {}
            for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
                dbs = glob.glob(os.path.join(self.dir, ctr, '**', '*_results.db'), recursive=True)
                if not dbs:
                    return None, 'no rocpd database for %s' % ctr
                per = self._per_kernel(dbs[0], ctr)
                hit = [v for k, v in per.items() if kernel in k]
                if not hit:
                    return None, '%s: kernel %s not in the trace' % (ctr, kernel)
                vals[ctr] = max(hit)
            return int((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024), \
                'measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) of this command at 3 steps; ' \
                '2*FETCH_SIZE + WRITE_SIZE in KB (gfx950 correction); FETCH_SIZE_KB=%.1f WRITE_SIZE_KB=%.1f' % (vals['FETCH_SIZE'], vals['WRITE_SIZE'])
        except Exception as exc:
            return None, 'PMC parse failed: %r' % (exc,)
        finally:
            shutil.rmtree(self.dir, ignore_errors=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, and wait."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', QOC_BENCH_SELF_SPAWNED='1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


class _GlooTransport(object):
    """Test hook (QOC_BENCH_BACKEND=gloo): torch.distributed on host tensors, so that N ranks can share ONE GPU (RCCL refuses
    two ranks on one device).  Exercises the N-rank code path of this file on a 1-GPU box; never used by the driver."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        self.dist, self.world, self.rank = dist, world, rank

    def barrier(self):
        self.dist.barrier()

    def all_reduce_max(self, values):
        import torch
        t = torch.tensor(np.asarray(values, dtype=np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.numpy()

    def all_gather(self, values):
        import torch
        t = torch.tensor(np.asarray(values, dtype=np.float64).reshape(-1))
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.numpy() for o in out])

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--seeds-per-gpu', type=int, default=SEEDS_PER_GPU)
    ap.add_argument('--chunks', type=int, default=0)
    ap.add_argument('--path', type=int, default=0)
    ap.add_argument('--variant', type=int, default=0, help='MFMA path: kernel of the exponentials (qoc_config.variant)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-single', action='store_true', help='skip the one-trajectory latency measurement')
    ap.add_argument('--no-secondary', action='store_true', help='skip the BASELINE config 3 / config 5 entries (`secondary`)')
    ap.add_argument('--no-live-pmc', action='store_true', help='roofline.traffic from the committed PMC file instead of rocprofv3 passes in this run')
    ap.add_argument('--groups', type=int, default=1, help='split the seeds of this GPU over G engines/streams')
    ap.add_argument('--prewarm', type=int, default=100, help='untimed iterations BEFORE the W warm-up steps: a GPU that was idle takes tens of '
                                                               'milliseconds to reach its working clocks, more than W = 5 steps of 1.2 ms last')
    ap.add_argument('--require-rccl', dest='require_rccl', action='store_true', default=None,
                    help='N > 1: an RCCL start-up failure on any rank ends the run with a non-zero exit code instead of a file-transport line '
                         '(the default under a launcher, i.e. in the driver form `python -m torch.distributed.run ... bench.py --gpus N`)')
    ap.add_argument('--allow-file-transport', dest='require_rccl', action='store_false',
                    help='N > 1: let the ranks fall back to the file transport when RCCL cannot start (the line then carries transport_fallback: true)')
    ap.add_argument('--reference-ops-worker', default=None, help=argparse.SUPPRESS)   # internal: one process of the all-core B-faithful CPU leg
    args = ap.parse_args()
    if args.reference_ops_worker:
        print(_reference_ops_worker(tuple(int(x) for x in args.reference_ops_worker.split(','))), flush=True)
        return

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC, before the HIP runtime starts (also done by load_library)
    from quantum_optimal_control import parallel_seeds
    from quantum_optimal_control.core import hip_engine
    from quantum_optimal_control.parallel_seeds import SeedShard

    rank, local_rank, world = parallel_seeds.launch_env()
    if world != max(1, args.gpus):
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    backend = os.environ.get('QOC_BENCH_BACKEND', 'rccl')
    # LOCAL_RANK -> HIP device: torch.distributed.run leaves all GPUs of the node visible to every rank (device = LOCAL_RANK); a launcher that
    # narrowed the view to one device per rank (HIP_VISIBLE_DEVICES) gives device 0
    device = parallel_seeds.device_for_rank(local_rank, hip_engine.device_count())
    if os.environ.get('QOC_BENCH_SAME_DEVICE') == '1':      # test hook: N ranks on ONE GPU (with the gloo transport)
        device = 0
    local_rank = device
    comm = gloo = None
    comm_init_s = 0.0
    t_comm = time.perf_counter()
    if world > 1:
        if backend == 'gloo':
            gloo = _GlooTransport(rank, world)
        else:
            # a launcher started the ranks (the driver's form): RCCL or nothing, unless --allow-file-transport / QOC_TRANSPORT=file say otherwise
            require = args.require_rccl
            if require is None:
                require = os.environ.get('QOC_TRANSPORT', 'rccl') != 'file'      # (self-spawned ranks too, since round 6: a broken RCCL must not yield a line that only `transport_fallback` tells apart)
            try:
                comm = parallel_seeds.open_comm(rank=rank, world=world, device=device, require_rccl=bool(require))
            except parallel_seeds.RcclRequired as exc:
                sys.stderr.write('bench.py: rank %d: %s\n' % (rank, exc))
                sys.exit(3)
    transport = comm if comm is not None else gloo
    comm_init_s = time.perf_counter() - t_comm        # RCCL start-up of this rank (id exchange + ncclCommInitRank + first barrier): a straggler shows here

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

    def iterate(n):
        for _ in range(n):
            for e in engs:
                e.iterate(params, 1)

    def sync():
        for e in engs:
            e.sync()

    def barrier():
        sync()
        if transport is not None:
            transport.barrier()

    iterate(max(0, args.prewarm))          # clocks up; restarted below, so that exactly warmup + steps iterations lie behind the reported state
    sync()
    for g in range(G):
        engs[g].set_base(seed_bases(shard.first + gsh[g].first, gsh[g].count))
    iterate(args.warmup)
    barrier()
    t0 = time.perf_counter()
    iterate(args.steps)
    if comm is not None and G == 1:
        # RCCL all-gather of the final per-seed losses straight from the engine's device array, on the engine's stream
        loss_all = shard.all_gather_engine_scalar(eng, hip_engine.SCALAR_LOSS, comm)
    else:
        sync()
        loss_local = np.concatenate([e.scalars()['loss'] for e in engs])
        loss_all = shard.all_gather(loss_local, dist=None if gloo is None else gloo.dist, comm=comm)
    sync()
    elapsed = time.perf_counter() - t0
    barrier()
    fidelity = 1.0 - loss_all
    assert fidelity.shape[0] == B * world, 'all-gather returned %d fidelities for %d seeds' % (fidelity.shape[0], B * world)
    parts = [e.scalars() for e in engs]
    sc = {k: np.concatenate([q[k] for q in parts]) for k in parts[0]}
    assert int(np.sum(sc['done'])) == 0 and np.all(sc['iterations'] == args.warmup + args.steps), \
        'a seed stopped early: timed work would be incomplete'
    assert np.array_equal(fidelity[shard.first:shard.first + shard.count], 1.0 - sc['loss']), 'gathered row != local losses'
    # who ran where and how long: one row per rank (elapsed ms, HIP device index, compute units, device name as 48 bytes)
    info = hip_engine.device_info(device)
    name_bytes = np.frombuffer(info['name'].encode()[:48].ljust(48, b' '), dtype=np.uint8).astype(np.float64)
    row = np.concatenate([[elapsed * 1e3, float(device), float(info['compute_units']), comm_init_s], name_bytes])
    rows = transport.all_gather(row) if transport is not None else row[None]
    per_rank = [{'rank': r, 'ms_total': float(rows[r][0]), 'ms_per_step': float(rows[r][0]) / args.steps, 'device': int(rows[r][1]),
                 'compute_units': int(rows[r][2]), 'comm_init_s': float(rows[r][3]),
                 'device_name': bytes(rows[r][4:].astype(np.uint8)).decode(errors='replace').strip()}
                for r in range(world)]
    ms_all = [q['ms_total'] for q in per_rank]
    if transport is not None:
        elapsed = float(transport.all_reduce_max([elapsed])[0])

    # ---- roofline of the dominant kernel, hipEvents on the engine stream (separate pass, rank 0) -----------------
    roof = None
    if rank == 0:
        eng.profile_enable(True)
        eng.iterate(params, min(50, args.steps))
        pr = eng.profile_read()
        eng.profile_enable(False)
        T, s = TAYLOR
        seeds0 = gsh[0].count
        flops_alg = seeds0 * SLICES * ((T - 1 + s) + 1) * 8.0 * N ** 3      # plain Taylor + chain product, 4 real products
        n_prod = executed_products_per_slice(T, s)
        flops_exec = seeds0 * SLICES * n_prod * 6.0 * N ** 3                 # what the kernel issues: 3 real products per complex product
        avg_ms = pr['total_ms'] / max(1, pr['launches'])
        ach_exec = flops_exec / (avg_ms * 1e-3) / 1e12
        ach_alg = flops_alg / (avg_ms * 1e-3) / 1e12
        # HBM traffic of the dominant kernel: PMC counters need rocprofv3, so the value comes from the committed PMC pass of
        # this same command (profiles/r02_pmc_traffic.json) when the workload matches; null otherwise
        traffic = None
        try:
            pm = json.load(open(PMC_TRAFFIC_FILE))
            w = pm['workload']
            if (w['seeds_per_gpu'], w['chunks'], pm['kernel']) == (seeds0, eng.chunks, pr['kernel']):
                traffic = pm['hbm_bytes_per_launch']
        except Exception:
            traffic = None
        roof = {'bound': 'mfma', 'kernel': pr['kernel'], 'achieved': ach_exec, 'peak': FP64_MATRIX_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': ach_exec / FP64_MATRIX_PEAK_TFLOPS, 'traffic': traffic,
                'traffic_unit': 'bytes/launch',
                'traffic_source': 'committed profile %s (rocprofv3 PMC pass of this command; NOT measured in this run)' % os.path.relpath(PMC_TRAFFIC_FILE, ROOT),
                'avg_launch_ms': avg_ms, 'launches': pr['launches'],
                'flops_per_launch': flops_exec, 'products_per_slice_executed': n_prod,
                'counting': 'achieved/frac = EXECUTED MFMA flops (%d Paterson-Stockmeyer products per slice x 6 n^3: 3 real '
                            'products per complex product); *_algorithmic = SURVEY 8d count (%d plain-Taylor products x 8 n^3)'
                            % (n_prod, (T - 1 + s) + 1),
                'achieved_algorithmic': ach_alg, 'frac_algorithmic': ach_alg / FP64_MATRIX_PEAK_TFLOPS,
                'flops_per_launch_algorithmic': flops_alg,
                'measured_mfma_f64_ceilings_TFLOPs': {'v_mfma_f64_16x16x4': 48.2, 'v_mfma_f64_4x4x4_4b': 73.0},
                'frac_of_measured_4x4x4_ceiling': ach_exec / 73.0}
    # ---- latency of ONE trajectory of the same workload (what a plain Grape() call runs), outside the timed region ----
    single = None
    if rank == 0 and not args.no_single:
        e1 = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], SLICES, TAYLOR[0], TAYLOR[1],
                                  reg_coeffs={}, n_seeds=1, device=local_rank)
        e1.set_base(seed_bases(0, 1))
        e1.iterate(params, 10); e1.sync()
        t1 = time.perf_counter()
        e1.iterate(params, 200); e1.sync()
        el1 = (time.perf_counter() - t1) / 200
        single = {'value': 1.0 / el1, 'unit': 'GRAPE iterations/s', 'ms_per_iteration': el1 * 1e3, 'path': e1.path,
                  'note': 'one control set (n_seeds=1, AUTO path) of the same C2 workload, 200 iterations; not part of `value`'}
        e1.close()
    # ---- BASELINE configs 3 and 5, driver-visible (rank 0, N = 1, after the timed region) ------------------------------
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = secondary_configs(local_rank)
    # ---- HBM traffic of the dominant kernel, re-measured in this run (two rocprofv3 PMC passes beside the CPU legs) --------
    pmc = None
    if rank == 0 and world == 1 and roof is not None and not args.no_live_pmc:
        pmc = LivePmc(args)
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
                       'stream_groups': G, 'ranks_seen': len(per_rank), 'fidelities_gathered': int(fidelity.shape[0]),
                       'comm_library': (comm.library if comm is not None else None), 'per_rank': per_rank,
                       'devices_distinct': len(set(q['device'] for q in per_rank)),
                       # a straggling rank / a slow communicator start is visible here (SCALE_rNN.json keeps the line): max over ranks and spread of the timed region
                       'rccl_init_s': max(q['comm_init_s'] for q in per_rank) if world > 1 else None,
                       'rank_ms_total': {'min': min(ms_all), 'max': max(ms_all), 'spread_pct': 100.0 * (max(ms_all) - min(ms_all)) / max(ms_all)},
                       'transport': (comm.library if comm.library.startswith('files') else 'rccl (%s)' % comm.library) if comm is not None
                       else ('gloo (test hook)' if gloo is not None else 'single process'),
                       # an RCCL run that fell back to files must not look like an RCCL result: both keys say so explicitly
                       'transport_fallback': bool(comm is not None and comm.library.startswith('files')),
                       'rccl_error': getattr(comm, 'fallback_reason', None) if comm is not None else None,
                       'parallelism': 'seed-sharded x%d, one all-gather of final fidelities, no collective inside the iterations' % world},
            'per_seed_iterations_per_s': args.steps / elapsed, 'prewarm_steps': max(0, args.prewarm),
            'single_trajectory': single,
            'secondary': secondary,
            'best_fidelity': float(np.max(fidelity)),
            'roofline': roof,
        }
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only (contract)
            out['cpu_baseline'] = cpu_baseline()
            try:
                out['cpu_baseline_reference_ops'] = cpu_baseline_reference_ops()
            except Exception as exc:                         # torch missing etc.: the primary baseline stands
                out['cpu_baseline_reference_ops'] = {'error': repr(exc)}
        if pmc is not None:
            live, how = pmc.result(roof['kernel'])
            if live is not None:
                roof['traffic_committed_profile'] = roof['traffic']
                roof['traffic'], roof['traffic_source'] = live, how
            else:
                roof['traffic_source'] += '; live PMC pass unavailable (%s)' % how
        print(json.dumps(out), flush=True)
    for e in engs:
        e.close()
    if transport is not None:
        transport.barrier()
        transport.close()


if __name__ == '__main__':
    main()
