"""Audit of the "pinned load" idiom (asm volatile global loads that hipcc does not count, waited for by an explicit s_waitcnt statement;
cdna_hip_programming.md section 5.7 item 1): between such a load and the wait that covers it the compiler believes the destination
registers already hold their data, so it may copy, spill (v_accvgpr_write), overwrite or use them as an address -- garbage, or a memory
access fault when a late-returning load lands on a register that meanwhile holds an address (round 3: ce_up_bwd_mc at factor 8, 298
registers).  This scan walks the gfx950 assembly of every kernel linearly and reports every COMPILER instruction (outside
;;#ASMSTART / ;;#ASMEND) that reads or writes a register with an asm load still pending.

    python tools/asm_pending_audit.py [csrc/file.hip ...]      (needs hipcc; no GPU)

Linear text order, not control flow: a loop whose wait sits at the bottom is scanned once top to bottom, which is the order that matters
(the back edge starts with nothing pending).  Exit status 1 if anything is reported."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'segdistill_amd', 'csrc')
REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def audit(asm_path):
    """`pending`: the wave's outstanding vector-memory operations in issue order -- asm loads with their destination registers, everything
    else (compiler loads / stores, asm stores) as anonymous slots so that a counted `s_waitcnt vmcnt(N)` retires the right ones (all but
    the N youngest)."""
    lines = open(asm_path).read().splitlines()
    findings, kernel, in_asm, pending = [], None, False, []
    VM = re.compile(r'(global|buffer|flat|scratch)_(load|store|atomic)')

    def retire(text):
        nonlocal pending
        m = re.search(r'vmcnt\((\d+)\)', text)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n and n < len(pending) else ([] if n == 0 else pending)

    for ln, raw in enumerate(lines, 1):
        st = raw.strip()
        l = st if st.startswith(';;#') else raw.split(';')[0].strip()
        m = re.match(r'^(_Z\S+):', raw)
        if m:
            kernel, pending, in_asm = m.group(1), [], False
            continue
        if st.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if st.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not l or l.endswith(':') or l.startswith('.'):
            continue
        if l.startswith('s_endpgm'):
            pending = []
            continue
        if l.startswith('s_waitcnt'):
            retire(l)
            continue
        if in_asm:
            if VM.match(l):
                is_load = '_load' in l and ' lds' not in l and '_load_lds' not in l      # LDS-DMA: the VGPR operand is an address, nothing lands in it
                pending.append((regs_of(l.split(None, 1)[1].split(',')[0]) if is_load else set(), ln))
            continue
        live = set().union(*[r for r, _ in pending]) if pending else set()
        if live:
            hit = regs_of(l) & live
            if hit:
                findings.append((kernel, ln, l, sorted(hit)[:4]))
        if VM.match(l):
            pending.append((set(), ln))
    return findings


def main():
    files = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            if not re.search(r'asm volatile\("[^"]*(global_load|buffer_load)', open(f).read()):
                continue
            s = os.path.join(tmp, os.path.basename(f) + '.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-S',
                            '--cuda-device-only', f, '-o', s], check=True, stderr=subprocess.DEVNULL)
            found = audit(s)
            per = {}
            for k, ln, l, hit in found:
                per.setdefault(k, []).append((ln, l, hit))
            print(f'{os.path.basename(f)}: {len(found)} compiler instruction(s) touching a register with an asm load pending, in {len(per)} kernel(s)')
            for k, items in per.items():
                print(f'   {k[:110]}: {len(items)}; first: line {items[0][0]}: {items[0][1][:70]}   {items[0][2]}')
            bad += len(found)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
