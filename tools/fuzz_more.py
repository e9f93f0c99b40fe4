#!/usr/bin/env python
"""The randomised differential test of tests/test_hip_fuzz.py over MORE seeds (128 .. 699: 572 further problems on every engine path that takes them,
each against the CPU checker) -- run on the GPU box after kernel changes; the test suite itself keeps the first 128 seeds.  Round 4, final kernels: 0 failures."""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/quantum-optimal-control_amd']
import pytest
from tests import test_hip_fuzz as tf
bad = 0
for seed in range(128, 700):
    try:
        tf.test_random_problem_all_paths(seed)
    except Exception as exc:
        bad += 1
        print('FAIL seed', seed, str(exc)[:300], flush=True)
print('done, failures:', bad)
