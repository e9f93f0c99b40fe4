#!/usr/bin/env python
"""C3 (state transfer, n = 64, k = 6, 1000 slices, forbidden levels + dwdt) for 1 / 64 / 256 control sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'quantum-optimal-control_amd'), os.path.join(ROOT, 'tools')]
import bench_configs
from tests.golden import cases
c = cases.case_c3()
for seeds in (1, 64, 256):
    bench_configs.run('C3 x%d' % seeds, c, seeds, 5)
