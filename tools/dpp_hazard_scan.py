#!/usr/bin/env python
"""Checks the built device code for the one hazard hipcc cannot see: a DPP read inside an asm statement.

The double-precision products of the workgroup-resident path (csrc/qoc_small_kernel.h) and of the mat-vec chains (csrc/qoc_gemm_chain_dpp.h) are written as
`v_fmac_f64_dpp ... row_newbcast` inline assembly; gfx950 needs two wait states between a VALU write of a register and a DPP read of it, the compiler pads none for an
asm statement, and under register pressure it may place a copy of the broadcast operand (v_mov_b64 from a spill register) directly in front of one -- the lane then
multiplies a stale value (seen in round 6: wrong gradients of one n = 8 build).  The statements of the builds under register pressure open with `s_nop 1` for that reason (QOC_SMALL_DPP_PAD); this tool
proves the result on the objects: it unbundles the gfx950 code of build/*.o, disassembles it and reports every v_fmac_f64_dpp whose DPP source was written by one of the two preceding
instructions, or that follows a write of EXEC by fewer than five (s_nop n counts as n + 1).

    python tools/dpp_hazard_scan.py [objects ...]        exit status 1 when a hazard is found     (tests/test_abi.py runs it on the built objects)
"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def disassemble(obj, tmp):
    fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'dev.co')
    for f in (fat, co):
        if os.path.exists(f):
            os.remove(f)
    subprocess.run([LLVM + '/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat], check=True, stderr=subprocess.DEVNULL)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat, '--targets=' + TARGET, '--output=' + co], check=True,
                   stderr=subprocess.DEVNULL)
    return subprocess.run([LLVM + '/llvm-objdump', '-d', '--no-show-raw-insn', co], check=True, capture_output=True, text=True).stdout


def written(text):
    m = re.match(r'\S+\s+([^,\s]+)', text)
    if not m:
        return set()
    dst = m.group(1)
    mm = re.match(r'v\[(\d+):(\d+)\]$', dst)
    if mm:
        return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
    mm = re.match(r'v(\d+)$', dst)
    return {int(mm.group(1))} if mm else set()


def scan(dis):
    """-> (number of DPP FMAs, [(kernel, offending writer, the DPP instruction)])"""
    found, count, kernel, prev = [], 0, '?', []
    for ln in dis.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
        if m:
            kernel, prev = m.group(1), []
            continue
        t = ln.split('//')[0].strip()
        if not t or t.startswith(';'):
            continue
        op = t.split()[0]
        if op == 's_nop':
            prev = (prev + [('s_nop', set())] * (int(t.split()[1]) + 1))[-6:]
            continue
        if op == 'v_fmac_f64_dpp':
            count += 1
            m = re.match(r'v_fmac_f64_dpp\s+v\[\d+:\d+\],\s*-?v\[(\d+):(\d+)\]', t)
            src = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for wop, w in prev[-2:]:
                if w != 'exec' and w & src:
                    found.append((kernel, wop, t))
            for wop, w in prev[-5:]:                       # a write of EXEC needs five wait states before a DPP instruction
                if w == 'exec':
                    found.append((kernel, wop + ' (EXEC)', t))
        exec_write = op.startswith('s_') and (op.endswith('saveexec_b64') or re.match(r'\S+\s+exec\b', t) is not None)
        prev = (prev + [(op, 'exec' if exec_write else (written(t) if op.startswith('v_') else set()))])[-6:]
    return count, found


def main(objs):
    bad = 0
    with tempfile.TemporaryDirectory(prefix='qoc_dpp_') as tmp:
        for obj in objs:
            dis = disassemble(obj, tmp)
            if dis is None:
                continue
            n, found = scan(dis)
            if n:
                print('%-28s %6d DPP FMAs, %d hazards' % (os.path.basename(obj), n, len(found)))
            for kernel, wop, t in found[:8]:
                print('    %s: %s written by the instruction before\n        %s' % (kernel, wop, t))
            bad += len(found)
    return bad


if __name__ == '__main__':
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'quantum-optimal-control_amd', 'build', '*.o')))
    sys.exit(1 if main(objs) else 0)
