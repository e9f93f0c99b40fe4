"""Wall-clock attribution for one trajectory: iteration time with one kernel group left out (QOC_DEBUG_SKIP).
Needs a library built with the hook compiled in: `QOC_DEBUG_BUILD=1 python -c "import __graft_entry__ as g; g.build(force=True)"` (the product
library does not read QOC_DEBUG_SKIP since round 4)."""
import os, subprocess, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
code = r'''
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd')]
import bench
from quantum_optimal_control.core import hip_engine
c, Hs, U0, V, W, dt = bench.build_problem()
path, variant = int(sys.argv[1]), int(sys.argv[2])
e = hip_engine.HipEngine(Hs, U0, V, W, c['maxA'], dt, c['total_time'], bench.SLICES, 5, 3, reg_coeffs={}, n_seeds=1, path=path, variant=variant)
e.set_base(bench.seed_bases(0, 1))
p = e.adam_params(max_iterations=10**9, poll_every=10**9, conv_target=-1.0, min_grad=-1.0)
e.iterate(p, 30); e.sync()
t0 = time.perf_counter(); e.iterate(p, 300); e.sync()
print('%.1f' % ((time.perf_counter() - t0) / 300 * 1e6))
'''
path, variant = sys.argv[1], sys.argv[2]
names = {0: 'nothing skipped', 1: 'k_controls', 2: 'exponentials (+ chain products)', 4: 'forward', 8: 'k_loss', 16: 'backward', 32: 'k_finish', 63: 'everything (empty loop)', 64: 'NOTHING, forward launched twice', 192: 'NOTHING, forward twice + backward-before-loss extra'}
base = None
for m in (0, 1, 2, 4, 8, 16, 32):
    r = subprocess.run([sys.executable, '-c', code, path, variant], env=dict(os.environ, QOC_DEBUG_SKIP=str(m)), capture_output=True, text=True)
    us = float(r.stdout.strip().splitlines()[-1])
    base = us if base is None else base
    print('skip %-34s: %7.1f us per iteration (%+7.1f)' % (names[m], us, us - base), flush=True)
