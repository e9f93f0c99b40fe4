"""Direct state-transfer route on padded problems (n = 40 / 48 / 56 levels in N = 64) x 64 control sets: ms per iteration (A/B of the active-column chain:
QOC_EXPERIMENTAL=1 QOC_DPP_ACTIVE_COLUMNS=0 runs the 16-columns-per-wave chain on the packed image)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/tools', ROOT+'/quantum-optimal-control_amd']
from bench_configs import run
from tests.golden import cases
for n in (40, 48, 56):
    for seeds in (64,):
        run('state transfer n=%d x%d' % (n, seeds), cases.case_c3(n=n), seeds, 5, path=4, chunks=1)
