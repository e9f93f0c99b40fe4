#!/usr/bin/env python
"""Golden vectors for the hot path a6-a14 produced by the REFERENCE's OWN graph code (build container only).

core/tensorflow_state.py and core/regularization_functions.py of /root/reference are copied to a scratch directory OUTSIDE the repo,
passed through lib2to3 (print statements, implicit relative import) and imported there next to tests/golden/tf1_shim.py installed as the
package `tensorflow` (a TF1 stand-in on torch autograd, see its docstring).  `TensorflowState(sys_para).build_graph()` then runs the
reference's text op for op: Defun operators with their hand-written gradient functions, chain, inner products, every regulariser, TF1 Adam.
What is stored (tests/golden/graph_<case>.npz): loss, reg_loss, unitary_scale, grad_squared, grad_pack, final_state, inter_vecs and
the optimisation variable after ONE Adam step -- numbers only.  tests/test_oracle_golden.py checks oracle/grape_oracle.py against them at
fp64 (the shim computes the reference's float32 tensors in float64 on purpose).

A stand-in for TensorFlow does not make this "the reference run here" (DESIGN.md section 2 keeps saying "parity unpinned by the
reference's runtime"); it removes the human restatement of loop bounds, slices and signs between the reference text and the oracle.

Round 4 adds (a) the BASELINE sizes themselves -- graph_c2_full_s0 / _s63.npz: C2 (n=32, k=4, 500 slices, m=8, (T,s)=(5,3)) from the control sets of
bench.py's restart seeds 0 and 63; graph_c3_full.npz: C3 (n=64, k=6, 1000 slices, state transfer, dwdt + forbidden levels) -- so that the GPU tests
of those configurations compare with the reference's text directly, not with the oracle; (b) `--fp32`: the same text with `tf.float32` = torch.float32
(QOC_TF1_SHIM_FP32=1, tests/golden/tf1_shim.py), i.e. at the reference's own precision -> graph32_*.npz, the tier-2 statement of SURVEY.md 8c.

Run:  python tests/golden/make_graph_golden.py          (needs /root/reference; writes tests/golden/graph_*.npz)
      python tests/golden/make_graph_golden.py --fp32   (writes tests/golden/graph32_*.npz)
"""
import contextlib
import importlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/quantum_optimal_control'
for p in (ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

ADAM_LR = 0.05


def graph_cases():
    """name -> recipe: C1, a dressed-forbidden case, state transfer with and without regularisers, every regulariser at once, U0 != I."""
    from tests.golden import cases
    FULL_REG = {'amplitude': 0.3, 'envelope': 0.2, 'dwdt': 0.1, 'd2wdt2': 0.05, 'forbidden_coeff_list': [3.0, 2.0],
                'states_forbidden_list': [3, 2], 'speed_up': 0.7, 'bandpass': 0.4, 'band': [0.5, 2.0]}
    out = {'c1': cases.case_c1(), 'small_auto_U0': cases.case_small_auto(), 'dressed_forbidden': cases.case_dressed(),
           'state_small': cases.case_state_small(), 'c3_small': cases.ALL_CASES['c3_small']()}
    c = cases.case_c2(n=4, k=2, steps=12, m=3, taylor=(6, 1), seed=2)
    c['reg_coeffs'] = dict(FULL_REG); c['total_time'] = 2.0
    out['unitary_allreg'] = c
    c = cases.case_c3(n=6, k=3, steps=15, taylor=(8, 0)); c['total_time'] = 1.0
    c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [4, 5], 'speed_up': 0.3, 'amplitude': 0.2}
    out['state_transfer_allreg'] = c
    out['c2_n8'] = cases.case_c2(n=8, k=3, steps=20, m=4, taylor=(5, 3), seed=12)
    return out


def full_size_cases():
    """The BASELINE configurations at their full sizes.  `base0` replaces SystemParameters.ops_weight_base (an INPUT of the graph code:
    tensorflow_state.py:176 reads it) where the bench's own restart seeds are wanted; `keep_inter` = store every time point."""
    import bench
    from tests.golden import cases
    out = {}
    for seed, keep in ((0, True), (63, False)):
        c = cases.case_c2(n=bench.N, k=bench.K_OPS, steps=bench.SLICES, m=bench.M, taylor=bench.TAYLOR, seed=0)
        c['base0'] = bench.seed_bases(seed, 1)[0]; c['keep_inter'] = keep
        out['c2_full_s%d' % seed] = c
    c = cases.case_c3(); c['keep_inter'] = True            # base0: the reference's own draw under np.random.seed(np_seed)
    out['c3_full'] = c
    return out


def kernel_family_cases():
    """One case per kernel family of the engine beyond n <= 32 / the BASELINE shapes (three time points of inter_vecs kept): the 48- and 64-wide
    MFMA kernels, a padded size, three qutrits with six controls and forbidden levels (what the reference's transmon examples look like), and a
    matrix size of the launch-per-product GEMM route (which the time-sharded engine runs on as well)."""
    from tests.golden import cases
    out = {}
    out['fam_n40'] = cases.case_c2(n=40, k=3, steps=64, m=5, taylor=(5, 2), seed=31)
    out['fam_n64'] = cases.case_c2(n=64, k=4, steps=64, m=8, taylor=(5, 3), seed=32)
    out['fam_n20'] = cases.case_c2(n=20, k=4, steps=64, m=8, taylor=(5, 3), seed=33)
    c = cases.case_c2(n=27, k=6, steps=80, m=8, taylor=(5, 3), seed=34)
    c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [26, 25]}
    out['fam_qutrits'] = c
    out['fam_n100'] = cases.case_c2(n=100, k=3, steps=48, m=4, taylor=(5, 2), seed=35)
    for c in out.values():
        c['total_time'] = 20.0 * c['steps'] / 500.0
        c['keep_inter'] = False
    return out


FP32_CASES = ('c1', 'c2_n8', 'unitary_allreg', 'state_transfer_allreg', 'dressed_forbidden', 'c2_full_s0', 'c3_full')


def import_reference_graph():
    scratch = tempfile.mkdtemp(prefix='qoc_ref_graph_')
    pkg = os.path.join(scratch, 'quantum_optimal_control')
    for d in ('', 'core', 'helper_functions'):
        os.makedirs(os.path.join(pkg, d), exist_ok=True)
        open(os.path.join(pkg, d, '__init__.py'), 'w').close()
    shutil.copy(os.path.join(REF, 'helper_functions', 'grape_functions.py'), os.path.join(pkg, 'helper_functions', 'grape_functions.py'))
    py2 = []
    for f in ('system_parameters.py', 'tensorflow_state.py', 'regularization_functions.py'):
        shutil.copy(os.path.join(REF, 'core', f), os.path.join(pkg, 'core', f))
        py2.append(os.path.join(pkg, 'core', f))
    with open(os.path.join(pkg, 'helper_functions', 'data_management.py'), 'w') as f:
        f.write('class H5File(object):\n    def __init__(self, *a, **k):\n        raise RuntimeError("save=False")\n')
    # the stand-in for TensorFlow: OUR shim, installed under the name the reference imports
    tfp = os.path.join(scratch, 'tensorflow')
    os.makedirs(os.path.join(tfp, 'python', 'framework'))
    shutil.copy(os.path.join(HERE, 'tf1_shim.py'), os.path.join(tfp, '__init__.py'))
    open(os.path.join(tfp, 'python', '__init__.py'), 'w').close()
    open(os.path.join(tfp, 'python', 'framework', '__init__.py'), 'w').close()
    with open(os.path.join(tfp, 'python', 'framework', 'function.py'), 'w') as f:
        f.write('from tensorflow import Defun\n')
    open(os.path.join(tfp, 'python', 'framework', 'ops.py'), 'w').close()
    from lib2to3.main import main as two_to_three
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        two_to_three('lib2to3.fixes', ['-w', '-n'] + py2)
    # the reference package has the same top-level name as this repo's drop-in package: forget ours (the recipes are built already)
    for name in [m for m in sys.modules if m == 'quantum_optimal_control' or m.startswith('quantum_optimal_control.')]:
        del sys.modules[name]
    sys.path.insert(0, scratch)
    mods = {m: importlib.import_module('quantum_optimal_control.' + m) for m in
            ('helper_functions.grape_functions', 'core.system_parameters', 'core.tensorflow_state')}
    for m in mods.values():
        assert m.__file__.startswith(scratch), 'not the reference: %s' % m.__file__
    return mods, importlib.import_module('tensorflow'), scratch


def system_parameters(mods, c):
    """SystemParameters exactly as Grape() builds it (main_grape/grape.py:89-113); use_gpu=True so that the bandpass term is allowed."""
    gf, spm = mods['helper_functions.grape_functions'], mods['core.system_parameters']
    n = len(c['H0'])
    U0 = np.identity(n) if c['U0'] is None else c['U0']
    if c['maxA'] is None:
        maxAmp = 4 * np.ones(len(c['Hops'])) if c['initial_guess'] is None else 1.5 * np.max(np.abs(c['initial_guess'])) * np.ones(len(c['Hops']))
    else:
        maxAmp = c['maxA']
    dressed = c['dressed_info']
    if isinstance(dressed, str) and dressed == 'from_H0':
        w_c, v_c, dressed_id = gf.get_dressed_info(c['H0'])
        dressed = {'eigenvectors': v_c, 'dressed_id': dressed_id, 'eigenvalues': w_c, 'is_dressed': True}
    guess = c['initial_guess']
    if guess is not None:
        guess = [np.asarray(row) for row in guess]          # see make_golden.py
    np.random.seed(c['np_seed'])
    return spm.SystemParameters(c['H0'], c['Hops'], c['Hnames'], c['U'], U0, c['total_time'], c['steps'], c['states_concerned_list'],
                                dressed, maxAmp, None, guess, False, 1e-4, c['state_transfer'], False, c['reg_coeffs'], False, None,
                                c['Taylor_terms'], True, True, False, False, False)


def r2c_cols(M, n):
    """(2n, ...) real-embedded vectors -> complex (analysis.py:18-24 for vectors)."""
    return M[:n] + 1j * M[n:]


def run_case(mods, tf, name, c, prefix='graph'):
    tf.reset()
    with contextlib.redirect_stdout(io.StringIO()):
        S = system_parameters(mods, c)
        if c.get('base0') is not None:
            assert np.shape(S.ops_weight_base) == np.shape(c['base0'])
            S.ops_weight_base = np.array(c['base0'])
        tfs = mods['core.tensorflow_state'].TensorflowState(S)
        tfs.build_graph()
    n, steps = S.state_num, S.steps
    g = lambda t: t.detach().numpy().astype(np.float64)   # noqa: E731  (a copy; fp32 runs are stored widened)
    out = dict(base0=g(tfs.ops_weight_base), loss=float(tfs.loss), reg_loss=float(tfs.reg_loss), unitary_scale=float(tfs.unitary_scale),
               grad_squared=float(tfs.grad_squared), grad_pack=g(tfs.grad_pack)[0], exp_terms=S.exp_terms, scaling=S.scaling)
    packed = g(tfs.inter_vecs_packed)                       # (2n, steps + 1, m)
    out['inter_vecs'] = np.transpose(r2c_cols(packed, n), (1, 0, 2))       # [steps + 1][n][m]
    if not c.get('keep_inter', True) or prefix != 'graph':
        out['inter_vecs'] = out['inter_vecs'][[0, steps // 2, steps]]      # first, middle and last time point only
    if not c['state_transfer']:
        F = g(tfs.final_state)
        out['final_state'] = F[:n, :n] + 1j * F[n:, :n]     # analysis.py:18-24 (RtoCMat)
    tfs.optimizer.run(ADAM_LR)                              # session.run([optimizer], {learning_rate: lr})   run_session.py:69
    out['base_after_adam'] = g(tfs.ops_weight_base)
    out['adam_lr'] = ADAM_LR
    np.savez_compressed(os.path.join(HERE, '%s_%s.npz' % (prefix, name)), **out)
    print('  %s_%s: loss' % (prefix, name) + '  %.12f reg_loss %.12f |grad| %.3e (T, s) = (%d, %d)' % (out['loss'], out['reg_loss'], np.max(np.abs(out['grad_pack'])),
                                                                                 S.exp_terms, S.scaling))


if __name__ == '__main__':
    fp32 = '--fp32' in sys.argv[1:]
    if fp32:
        os.environ['QOC_TF1_SHIM_FP32'] = '1'               # read by the shim when the scratch copy imports it as `tensorflow`
    todo = graph_cases()                                    # recipes first: they come from THIS repo's package of the same name
    todo.update(full_size_cases())
    todo.update(kernel_family_cases())
    if fp32:
        todo = {name: todo[name] for name in FP32_CASES}
    mods, tf, scratch = import_reference_graph()
    assert tf.FP32 == fp32
    try:
        print('reference graph code imported from scratch copy', scratch, '(float32 tensors)' if fp32 else '(float32 held in float64)')
        for name, c in todo.items():
            run_case(mods, tf, name, c, prefix='graph32' if fp32 else 'graph')
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
