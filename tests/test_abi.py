"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/qoc.h declares,
and refuses to run without a HIP device (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'qoc.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(qoc_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from quantum_optimal_control.core import hip_engine
    lib = hip_engine.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'libqoc_hip.so does not export %s' % name
    # and the Python binding declares a prototype for each of them
    assert sorted(hip_engine.EXPORTED_SYMBOLS) == declared
    assert b'gfx950' in lib.qoc_version()


def test_config_struct_layout_matches_header():
    from quantum_optimal_control.core import hip_engine
    # 8 int32 + 2 double + 6 int32 + 6 double + 7 int32 + variant + plan_seeds + time_shards + time_rank + 3 reserved int32, natural alignment
    assert ctypes.sizeof(hip_engine.QocConfig) == 8 * 4 + 2 * 8 + 6 * 4 + 6 * 8 + 14 * 4
    assert ctypes.sizeof(hip_engine.QocAdamParams) == 4 * 8 + 2 * 4


def test_no_cpu_fallback_without_gpu():
    from quantum_optimal_control.core import hip_engine
    if hip_engine.device_count() > 0:
        pytest.skip('a HIP device is visible')
    n, k, m, steps = 2, 1, 2, 4
    Hs = np.zeros((k + 1, n, n), dtype=np.complex128)
    with pytest.raises(hip_engine.QocError, match='no HIP device'):
        hip_engine.HipEngine(Hs, np.eye(n), np.eye(n)[:, :m], np.eye(n)[:, :m], [1.0], 0.1, 0.4, steps, 3, 0,
                             reg_coeffs={})


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'quantum-optimal-control_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no CPU fallback', ''), '%s mentions the oracle' % f


def test_grape_argument_errors_match_reference():
    from quantum_optimal_control.main_grape.grape import Grape
    H0 = np.zeros((2, 2)); Hops = [np.eye(2)]
    with pytest.raises(ValueError, match='file_name, is not specified'):
        Grape(H0, Hops, ['x'], np.eye(2), 1.0, 4, [0, 1])                       # save=True default
    with pytest.raises(ValueError, match='data_path, is not specified'):
        Grape(H0, Hops, ['x'], np.eye(2), 1.0, 4, [0, 1], file_name='a')
    with pytest.raises(KeyError):
        Grape(H0, Hops, ['x'], np.eye(2), 1.0, 4, [0, 1], save=False, freq_unit='THz')
    with pytest.raises(ValueError, match='Initial guess has strength > max_amp'):
        Grape(H0 + np.eye(2), Hops, ['x'], np.eye(2), 1.0, 4, [0, 1], save=False, maxA=[0.1],
              initial_guess=np.ones((1, 4)), reg_coeffs={})


def test_plan_limits_header_is_the_single_source_of_autos_numbers():
    """csrc/qoc_plan_limits.h holds every measured number of AUTO's dispatch table as `#define QOC_PLAN_<NAME> <integer>`: the engine compiles them, tests/test_auto_plan.py
    parses them.  Here (no GPU): the header parses, the engine sources contain no second copy of the latency-mode limits, and the limits are ordered as the table assumes."""
    import re
    csrc = os.path.join(ROOT, 'quantum-optimal-control_amd', 'csrc')
    lim = {}
    for line in open(os.path.join(csrc, 'qoc_plan_limits.h')):
        mt = re.match(r'#define\s+QOC_PLAN_(\w+)\s+(\d+)\b', line)
        if mt:
            lim[mt.group(1)] = int(mt.group(2))
    assert len(lim) >= 40
    assert lim['LAT_WORK_SRC'] <= lim['LAT_WORK'] and lim['LAT_WORK_PER_STRIP'] < lim['LAT_WORK_PER_STRIP_SRC']
    assert lim['NT4_MIN_SETS_K4'] <= lim['NT4_MIN_SETS'] and lim['ST_BIG_DPP'] <= lim['ST_BIG_DPP_SRC'] <= lim['ST_BIG_N64_SRC']
    assert lim['ST_DIRECT_DPP'] <= lim['ST_DIRECT_DPP_SRC'] <= lim['ST_DIRECT_N64'] <= lim['ST_DIRECT_N32']
    assert lim['CHUNK_ITEMS'] == 1024 and lim['CHUNKS_MAX'] <= lim['CHUNKS_MAX_NT2']
    assert lim['SMALL_MAX_MODEL_US'] <= lim['SMALL_MAX_MODEL_US_SRC'] and lim['SMALL_MAX_SETS'] >= 64
    engine = open(os.path.join(csrc, 'qoc_engine.hip')).read()
    assert '#include "qoc_plan_limits.h"' in engine
    assert 'QOC_LATENCY_MAX_WORK' not in engine                                  # the old private macros are gone
    for name in lim:
        used = any(('QOC_PLAN_' + name) in open(os.path.join(csrc, f)).read() for f in ('qoc_engine.hip', 'qoc_mfma_backward.hip', 'qoc_small.hip'))
        assert used, 'QOC_PLAN_%s is defined but no engine source uses it' % name


def test_no_dpp_read_hazard_in_the_built_device_code():
    """The fp64 products written as `v_fmac_f64_dpp` inline assembly (csrc/qoc_small_kernel.h, csrc/qoc_gemm_chain_dpp.h) need two wait states behind a VALU write of
    their DPP source; the compiler pads none for an asm statement and, under register pressure, places spill copies right in front of one (round 6: wrong gradients of
    one n = 8 build).  tools/dpp_hazard_scan.py disassembles the gfx950 code of every built object and must find no such pair."""
    import glob
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = sorted(glob.glob(os.path.join(root, 'quantum-optimal-control_amd', 'build', '*.o')))
    if not objs or not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('no built objects (python -c "import __graft_entry__ as g; g.build()") or no llvm-objdump')
    spec = importlib.util.spec_from_file_location('dpp_hazard_scan', os.path.join(root, 'tools', 'dpp_hazard_scan.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(objs) == 0


def test_dpp_hazard_scan_recognises_the_patterns_it_guards_against():
    """The scanner on hand-written disassembly: a spill copy right in front of a DPP read, the same behind s_nop 1, an EXEC write four / five slots before."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('dpp_hazard_scan', os.path.join(root, 'tools', 'dpp_hazard_scan.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fma = '\tv_fmac_f64_dpp v[10:11], v[30:31], v[40:41] row_newbcast:1 row_mask:0xf bank_mask:0xf // 0000: 0\n'
    head = '0000000000001000 <kernel_a>:\n'
    assert mod.scan(head + '\tv_mov_b64_e32 v[30:31], v[200:201]\n' + fma) == (1, [('kernel_a', 'v_mov_b64_e32', fma.split('//')[0].strip())])
    assert mod.scan(head + '\tv_accvgpr_read_b32 v31, a5\n\tv_add_f64 v[2:3], v[4:5], v[6:7]\n' + fma)[1] != []          # one instruction in between: still one wait state short
    assert mod.scan(head + '\tv_mov_b64_e32 v[30:31], v[200:201]\n\ts_nop 1\n' + fma) == (1, [])
    assert mod.scan(head + '\tv_mov_b64_e32 v[32:33], v[200:201]\n' + fma) == (1, [])                                      # another register
    four = '\tv_add_f64 v[2:3], v[4:5], v[6:7]\n' * 4
    assert mod.scan(head + '\ts_and_saveexec_b64 s[0:1], vcc\n' + four + fma)[1] != []
    assert mod.scan(head + '\ts_and_saveexec_b64 s[0:1], vcc\n\ts_nop 4\n' + fma) == (1, [])
    assert mod.scan(head + '\ts_mov_b64 exec, s[0:1]\n' + four + '\tv_add_f64 v[2:3], v[4:5], v[6:7]\n' + fma) == (1, [])
