#!/usr/bin/env python
"""Unitary gates of 33 <= n <= 64 levels x 64 control sets (k = 4, 500 slices, m = 8, (T, s) = (5, 3)): k_mfma_expm_rows runs the block steps over the
ACTIVE inner 4-row strips ceil(n / 4) of the matrices padded to 48 / 64.  QOC_EXPERIMENTAL=1 QOC_ROWS_QA_FULL=1: the padded problem in full (A/B).
padded_sizes_nt34.py <seeds>: another number of control sets (1: the latency mode, whose slice kernel takes the active strips too)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for n in (33, 36, 40, 44, 48, 49, 52, 56, 60, 64):
    nt = (n + 15) // 16
    bench_configs.run('n=%d x%d (active strips %d of %d)' % (n, seeds, (n + 3) // 4, 4 * nt), cases.case_c2(n=n, k=4, steps=500, m=8, taylor=(5, 3), seed=2), seeds, 5)
