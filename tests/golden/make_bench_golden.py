"""Golden vectors for the EXACT configuration bench.py times: full-size C2 (n=32, k=4, 500 slices, m=8,
(T,s)=(5,3)), restart seeds {0, 31, 63} of the 64 per GPU, and one control set of the same problem with dwdt + forbidden-level
regularisers (what a plain Grape() call runs: n_seeds=1, AUTO = latency mode).

Outputs come from the CPU oracle (oracle/grape_oracle.py, the restatement of core/tensorflow_state.py:204-242,323-356 and
core/run_session.py:47-69); they are stored so that the GPU-box tests cost no CPU seconds.  Run in the build container:

    python tests/golden/make_bench_golden.py
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


def main():
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
