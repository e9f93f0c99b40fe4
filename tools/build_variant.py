#!/usr/bin/env python
"""A/B builds of libqoc_hip.so: recompile ONE translation unit with extra flags, link it with the product objects of the others.
    python tools/build_variant.py <name> <unit> [-DFLAG=..] ...     ->  quantum-optimal-control_amd/lib_<name>/libqoc_hip.so
Run on the GPU box with QOC_HIP_LIBRARY=<that path> (hip_engine.LIB_PATH); tools/ab_bench.sh and friends take such paths.  lib_*/ is git-ignored."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
csrc, objdir = os.path.join(g.PKG, 'csrc'), os.path.join(g.PKG, 'build')
out = os.path.join(g.PKG, 'lib_' + name)
os.makedirs(out, exist_ok=True)
obj = os.path.join(out, unit + '.o')
subprocess.check_call(['/opt/rocm/bin/hipcc'] + g.HIPFLAGS + g.UNIT_FLAGS.get(unit, []) + flags + ['-fPIC', '-c', os.path.join(csrc, unit + '.hip'), '-o', obj])
objs = [obj if u == unit else os.path.join(objdir, u + '.o') for u in g.ENGINE_UNITS]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', os.path.join(out, 'libqoc_hip.so')])
print(os.path.relpath(os.path.join(out, 'libqoc_hip.so'), ROOT))
