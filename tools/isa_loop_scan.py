"""Static scan of the gfx950 ISA of csrc/*.hip for innermost loops that keep few bytes in flight per lane (<= 16 B of global loads per
iteration): the "one request at a time" walks that run at the load latency instead of the HBM rate (DESIGN 3.13 fixed four of them:
ln_bwd, dw3x3_fwd, bias_grad, multi_colsum_partials).  Over-reports remainder loops; the list is a set of leads, not a verdict.

    python tools/isa_loop_scan.py            (needs hipcc; no GPU)
"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'segdistill_amd', 'csrc')
BR = re.compile(r's_cbranch_\w+\s+(\.LBB\d+_\d+)')


def scan(asm_path, tag):
    lines = open(asm_path).read().splitlines()
    start, out = None, []
    for idx, l in enumerate(lines):
        m = re.match(r'^(_ZN2sd\S*):', l)
        if m:
            start = (idx, m.group(1))
        if start and 's_endpgm' in l:
            body = [x.strip() for x in lines[start[0] + 1:idx]]
            labels = {m.group(1): i for i, b in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', b)] if m}
            for i, b in enumerate(body):
                m = BR.search(b)
                if not (m and m.group(1) in labels and labels[m.group(1)] < i):
                    continue
                lo, hi = labels[m.group(1)], i
                seg = body[lo:hi + 1]
                loads = [x for x in seg if x.startswith(('global_load', 'buffer_load'))]
                nbytes = sum(4 * int(re.search(r'dwordx(\d)', x).group(1)) if 'dwordx' in x else 4 for x in loads)
                nested = any(BR.search(x) and lo < labels.get(BR.search(x).group(1), -1) < hi for x in seg[:-1])
                if loads and nbytes <= 16 and not nested:
                    name = subprocess.run(['c++filt', start[1]], capture_output=True, text=True).stdout.strip() or start[1]
                    name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
                    out.append((tag, name[:70], len(loads), nbytes, len(seg)))
            start = None
    return out


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
            tag = os.path.basename(src)[:-4]
            asm = os.path.join(tmp, tag + '.s')
            r = subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-S',
                                '--cuda-device-only', src, '-o', asm], capture_output=True, text=True)
            if r.returncode == 0:
                rows += scan(asm, tag)
    seen = set()
    for r in rows:
        key = (r[0], re.sub(r'<.*', '', r[1]))
        if key in seen:
            continue
        seen.add(key)
        print('%-11s %-70s loads/iter %d  bytes/lane/iter %3d  loop length %d' % r)


if __name__ == '__main__':
    main()
