"""Golden vectors for the EXACT configuration bench.py times: full-size C2 (n=32, k=4, 500 slices, m=8,
(T,s)=(5,3)), restart seeds {0, 31, 63} of the 64 per GPU, and one control set of the same problem with dwdt + forbidden-level
regularisers (what a plain Grape() call runs: n_seeds=1, AUTO = latency mode).

Outputs come from the CPU oracle (oracle/grape_oracle.py, the restatement of core/tensorflow_state.py:204-242,323-356 and
core/run_session.py:47-69); they are stored so that the GPU-box tests cost no CPU seconds.  Run in the build container:

    python tests/golden/make_bench_golden.py        (C2 files)
    python tests/golden/make_bench_golden.py c3     (c3_bench_batch.npz: the C3 x 64 / x 256 engines of bench.secondary_configs)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
from oracle import grape_oracle as go  # noqa: E402
from quantum_optimal_control.helper_functions import synthetic_systems  # noqa: E402

SEEDS = (0, 31, 63)
ADAM = dict(rate=0.01, learning_rate_decay=2500, conv_target=1e-8, min_grad=1e-25, max_iterations=3)   # bench.py's Adam parameters, 3 steps
LAT_REG = {'dwdt': 0.1, 'forbidden_coeff_list': [3.0, 2.0], 'states_forbidden_list': [31, 30]}


def system(reg):
    c = synthetic_systems.case_c2(n=bench.N, k=bench.K_OPS, steps=bench.SLICES, m=bench.M, taylor=bench.TAYLOR, seed=0)
    c['reg_coeffs'] = dict(reg)
    np.random.seed(c['np_seed'])
    return go.OracleSystem(c['H0'], c['Hops'], c['U'], c['total_time'], c['steps'], c['states_concerned_list'], U0=c['U0'],
                           reg_coeffs=c['reg_coeffs'], dressed_info=None, maxA=c['maxA'], initial_guess=c['initial_guess'],
                           state_transfer=c['state_transfer'], Taylor_terms=c['Taylor_terms'])


def pack(sp, bases):
    out = {}
    for key in ('loss', 'reg_loss', 'grad_squared', 'unitary_scale', 'grad', 'U_final', 'adam_base', 'adam_loss', 'adam_U_final'):
        out[key] = []
    for base in bases:
        o = go.evaluate(sp, base)
        for key in ('loss', 'reg_loss', 'grad_squared', 'unitary_scale', 'grad', 'U_final'):
            out[key].append(o[key])
        r = go.run_adam(sp, ADAM, base=base)
        assert r['iterations'] == 3
        out['adam_base'].append(r['base']); out['adam_loss'].append(r['loss']); out['adam_U_final'].append(r['U_final'])
    return {k: np.array(v) for k, v in out.items()}


C3_SETS = (0, 31, 63, 127, 255)      # {0, 31, 63} of the 64-set and {0, 127, 255} of the 256-set C3 engine of bench.secondary_configs (the first 64 of 256 ARE the 64)


def c3_system():
    """BASELINE config 3 exactly as bench.secondary_configs builds it (synthetic_systems.case_c3: n = 64, k = 6, 1000 slices, one state vector, T = 10, dwdt + two
    forbidden levels), through the oracle's pre-processing."""
    c = synthetic_systems.case_c3()
    np.random.seed(c['np_seed'])
    return c, go.OracleSystem(c['H0'], c['Hops'], c['U'], c['total_time'], c['steps'], c['states_concerned_list'], U0=c['U0'], reg_coeffs=c['reg_coeffs'],
                              dressed_info=None, maxA=c['maxA'], initial_guess=c['initial_guess'], state_transfer=c['state_transfer'],
                              Taylor_terms=c['Taylor_terms'])


def c3_bases(n_seeds, c):
    """bench._engine_for's control sets: default_rng(0).normal(0, 1 / sqrt(steps), (n_seeds, k, steps))."""
    return np.random.default_rng(0).normal(0, 1 / np.sqrt(c['steps']), (n_seeds, len(c['Hops']), c['steps']))


def pack_c3():
    c, sp = c3_system()
    b256, b64 = c3_bases(256, c), c3_bases(64, c)
    assert np.array_equal(b256[:64], b64)
    out = {key: [] for key in ('loss', 'reg_loss', 'grad_squared', 'unitary_scale', 'grad', 'final_vecs', 'adam_base', 'adam_loss', 'adam_reg_loss')}
    for sidx in C3_SETS:
        o = go.evaluate(sp, b256[sidx], want_inter=True)
        for key in ('loss', 'reg_loss', 'grad_squared', 'unitary_scale', 'grad'):
            out[key].append(o[key])
        out['final_vecs'].append(o['inter_vecs'][-1])
        r = go.run_adam(sp, ADAM, base=b256[sidx])
        assert r['iterations'] == 3
        out['adam_base'].append(r['base']); out['adam_loss'].append(r['loss']); out['adam_reg_loss'].append(r['reg_loss'])
    return {k: np.array(v) for k, v in out.items()}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'c3':
        np.savez_compressed(os.path.join(HERE, 'c3_bench_batch.npz'), sets=np.array(C3_SETS), **pack_c3())
        print('written c3_bench_batch.npz')
        return
    bases = bench.seed_bases(0, bench.SEEDS_PER_GPU)
    sp = system({})
    g = pack(sp, [bases[s] for s in SEEDS])
    np.savez_compressed(os.path.join(HERE, 'c2_bench_batch.npz'), seeds=np.array(SEEDS), **g)
    sp = system(LAT_REG)
    g = pack(sp, [bases[0]])
    np.savez_compressed(os.path.join(HERE, 'c2_bench_single_regularised.npz'), **g)
    print('written c2_bench_batch.npz, c2_bench_single_regularised.npz')


if __name__ == '__main__':
    main()
