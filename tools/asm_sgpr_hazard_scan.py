"""Static scan for one gfx9 hazard the compiler does not resolve around INLINE ASM: a VALU instruction that writes an SGPR (v_readlane_b32,
v_readfirstlane_b32, v_cmp* with an SGPR destination ...) followed by a vector-memory instruction that READS that SGPR needs 5 wait states.  hipcc
inserts them for its own instructions; an `asm volatile("global_load_dwordx4 %0, %1, %2" :: "s"(base))` right behind the v_readlane that restores a
SPILLED base pointer gets none -- the load then goes to a stale address (round 6: a memory fault in head_tail.hip as soon as `fuse_bias` was given,
because that pointer had been spilled to a VGPR lane).  The asm statements of csrc/ carry their own `s_nop 4`; this scan checks that none is left
without.   python tools/asm_sgpr_hazard_scan.py [csrc/file.hip ...]      (needs hipcc; no GPU; exit status 1 if anything is reported)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'segdistill_amd', 'csrc')
SREG = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]')


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path):
    lines = [l.strip() for l in open(path).read().splitlines()]
    findings, kernel, in_asm = [], None, False
    recent = []          # (wait states ago, sgprs written by a VALU instruction)
    for ln, l in enumerate(lines, 1):
        if l.endswith(':') and l.startswith('_Z'):
            kernel, recent = l[:-1], []
            continue
        if l.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if l.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        op = l.split()[0]
        states = 1
        if op == 's_nop':
            states = int(l.split()[1]) + 1
        if in_asm and re.match(r'(global|buffer|flat)_(load|store)', op):
            need = sregs(l.split(None, 1)[1])
            for age, regs in recent:
                if age < 5 and regs & need:
                    findings.append((kernel, ln, l, sorted(regs & need), age))
        recent = [(a + states, r) for a, r in recent if a + states < 6]
        if op.startswith('v_') and not in_asm:
            dst = l.split(None, 1)[1].split(',')[0]
            w = sregs(dst)
            if w:
                recent.append((0, w))
    return findings


def main():
    files = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            src = open(f).read()
            if not re.search(r'asm volatile\("[^"]*(global_load|buffer_load)', src):
                continue
            s = os.path.join(tmp, os.path.basename(f) + '.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-S',
                            '--cuda-device-only', f, '-o', s], check=True, stderr=subprocess.DEVNULL)
            found = scan(s)
            print(f'{os.path.basename(f)}: {len(found)} asm vector-memory instruction(s) reading an SGPR a VALU instruction wrote < 5 wait states earlier')
            for k, ln, l, regs, age in found[:6]:
                print(f'   line {ln}: {l}   [s{regs}] written {age} wait state(s) earlier   ({str(k)[:60]})')
            bad += len(found)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
