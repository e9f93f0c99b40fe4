#!/usr/bin/env python
"""C3 (state transfer n=64 k=6 steps=1000 m=1 T=10, dwdt + two forbidden levels) at 64 and 256 control sets: ms per iteration through the C ABI.
Usage: c3_batches.py [seeds ...]   (default 64 256)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
from bench_configs import run  # noqa: E402
from tests.golden import cases  # noqa: E402

if __name__ == '__main__':
    for seeds in ([int(a) for a in sys.argv[1:]] or [64, 256]):
        run('C3 state transfer x%d seeds' % seeds, cases.case_c3(), seeds, 5)
