#!/usr/bin/env python
"""Row g2 of the round-5 verdict: ONE control set (and small batches) of the sizes the reference is used at -- qubit, qutrit, two / three transmons --
on the workgroup-resident path (QOC_PATH_SMALL, csrc/qoc_small_kernel.h) against the MFMA path's latency mode / batch kernels (path = 2: what AUTO took
before round 6).  us per iteration of the whole batch, wall clock around qoc_iterate(iters) + qoc_sync (the small path runs the loop inside one launch).
  python tools/small_n_latency.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases

quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'


def both(name, c, seeds, groups=(0,), old=True, rows=(0,)):
    for g in groups:
        for r in rows:
            try:
                bench_configs.run('%s [small, groups=%s%s]' % (name, g or 'auto', ', rows=%d' % r if r else ''), c, seeds, 200, path=5, chunks=g, variant=r)
            except Exception as err:
                print('%s [small, groups=%s rows=%s]: not taken (%s)' % (name, g, r, str(err)[:120]))
    if old:                     # AUTO as it was before round 6 (latency mode of the MFMA path / batch kernels / GEMM route)
        os.environ['QOC_EXPERIMENTAL'] = '1'; os.environ['QOC_SMALL_AUTO'] = '0'
        bench_configs.run('%s [AUTO of round 5]' % name, c, seeds, 50, path=0)
        del os.environ['QOC_SMALL_AUTO']; del os.environ['QOC_EXPERIMENTAL']


both('C1 qubit', cases.case_c1(), 1, rows=(0, 16, 32))
both('C1 qubit x64', cases.case_c1(), 64, rows=(0, 16, 32))
for n in (3, 4, 6, 8):
    c = cases.case_c2(n=n, k=4, steps=500, m=min(8, n), taylor=(5, 3), seed=2)
    both('n=%d x 500 slices' % n, c, 1, groups=(0,) if quick else (0, 8, 16, 32), rows=(0,) if n > 4 else (16, 32))
c = cases.case_c2(n=9, k=4, steps=300, m=4, taylor=(5, 3), seed=2)
c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [8, 5]}
both('two qutrits n=9 x 300 + dwdt + forbidden', c, 1, groups=(0,) if quick else (0, 5, 10, 19))
c = cases.case_c2(n=9, k=4, steps=500, m=9, taylor=(5, 3), seed=2)
both('n=9 x 500 slices', c, 1)
if not quick:
    c = cases.case_c2(n=12, k=4, steps=250, m=8, taylor=(5, 3), seed=2)
    both('n=12 x 250 slices', c, 1)
    c = cases.case_c2(n=8, k=4, steps=100, m=8, taylor=(5, 3), seed=2)
    both('n=8 x 100 slices', c, 1, groups=(0, 7))
    both('n=8 x 100 slices x16', c, 16)
    c = cases.case_c2(n=4, k=2, steps=200, m=4, taylor=(5, 3), seed=2)
    for s in (1, 4, 16, 64):
        both('n=4 x 200 slices x%d' % s, c, s)
    both('small_auto n=4 x 40', cases.case_small_auto(), 1)
    both('dressed forbidden n=6 x 30', cases.case_dressed(), 1)
    both('state transfer n=5 x 30', cases.case_state_small(), 1)
