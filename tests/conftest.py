import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'quantum-optimal-control_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


def pytest_sessionstart(session):
    """The native libraries are git-ignored build products: build them once if a fresh checkout has none."""
    lib = os.path.join(PKG, 'lib', 'libqoc_hip.so')
    if not os.path.exists(lib):
        try:
            import __graft_entry__
            __graft_entry__.build()
        except Exception as exc:                      # leave the failure to the tests that need the library
            print('conftest: could not build libqoc_hip.so: %r' % (exc,))
