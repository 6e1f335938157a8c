"""Summarise a rocprofv3 --kernel-trace CSV of bench.py into a per-step kernel table.

usage: python tools/prof_summary.py <dir with *_kernel_trace.csv> [--marker cgd_up_fwd_partials] [--skip 5] [--top 40] [--out file]
Steady-state steps are delimited by successive dispatches of the marker kernel (one per KD step);
the first --skip steps (warm-up, MIOpen find) are dropped.  Prints per-step average time by kernel."""
import argparse
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+)', name)
    base = m.group(1) if m else name
    if 'ck::' in name or name.startswith('_ZN2ck'):
        kind = re.search(r'kernel_[a-z_0-9]+', name)
        base = 'ck::' + (kind.group(0) if kind else 'kernel')
    if 'elementwise_kernel' in name:
        f = re.search(r'native::([A-Za-z0-9_]+Functor[A-Za-z0-9_]*|[a-z_]+_kernel_cuda[^,>]*|[A-Za-z_]+Functor)', name)
        base = 'at::elementwise<' + (f.group(1) if f else '?') + '>'
    return base[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--marker', default='cgd_up_fwd_partials')
    ap.add_argument('--skip', type=int, default=5)
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--out', default=None)
    ap.add_argument('--gaps', type=int, default=0, help='also list the N idle-gap sites with the largest total (kernel before -> kernel after)')
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        sys.exit('no kernel_trace.csv under ' + a.dir)
    rows = []
    for f in files:
        with open(f, newline='') as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [s for s, e, n in rows if a.marker in n]
    if len(marks) <= a.skip + 1:
        sys.exit(f'only {len(marks)} marker dispatches')
    t0, t1 = marks[a.skip], marks[-1]
    nsteps = len(marks) - 1 - a.skip
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in rows:
        if t0 <= s < t1:
            k = short(n)
            agg[k][0] += e - s
            agg[k][1] += 1
            busy += e - s
    # union of the kernel intervals (at least one kernel resident) and the idle gaps between them
    union, cur_s, cur_e, gaps = 0, None, None, []
    sites, last_name = defaultdict(lambda: [0, 0]), None
    for s, e, n in rows:
        if not (t0 <= s < t1):
            continue
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
                gaps.append(s - cur_e)
                site = sites[(short(last_name)[:44], short(n)[:44])]
                site[0] += s - cur_e
                site[1] += 1
            cur_s, cur_e, last_name = s, e, n
        else:
            if e > cur_e:
                cur_e, last_name = e, n
    if cur_e is not None:
        union += cur_e - cur_s
    gaps.sort()
    gap_note = (f'idle gaps: {len(gaps) / nsteps:.0f}/step, total {sum(gaps) / nsteps / 1e6:.3f} ms/step, median {gaps[len(gaps) // 2] / 1e3:.1f} us'
                if gaps else 'no idle gaps')
    lines = [f'# rocprofv3 kernel trace, steady state: {nsteps} KD steps, wall {(t1 - t0) / nsteps / 1e6:.3f} ms/step, '
             f'GPU busy {busy / nsteps / 1e6:.3f} ms/step, {sum(v[1] for v in agg.values()) / nsteps:.0f} kernel launches/step',
             f'# at least one kernel resident {union / nsteps / 1e6:.3f} ms/step (sum of durations / that = {busy / max(union, 1):.2f}x overlap); {gap_note}',
             f'{"kernel":92s} {"ms/step":>9s} {"%busy":>7s} {"calls/step":>10s} {"avg us":>9s}']
    for k, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        lines.append(f'{k:92s} {ns / nsteps / 1e6:9.3f} {100 * ns / busy:7.2f} {c / nsteps:10.1f} {ns / c / 1e3:9.1f}')
    if a.gaps:
        lines.append(f'# idle-gap sites (kernel that ended last -> kernel that started next), by total idle time')
        for (ka, kb), (ns, c) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:a.gaps]:
            lines.append(f'#   {ka:44s} -> {kb:44s} {ns / nsteps / 1e3:8.1f} us/step  {c / nsteps:5.1f}/step  avg {ns / c / 1e3:7.1f} us')
    text = '\n'.join(lines)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        open(a.out, 'w').write(text + '\n')


if __name__ == '__main__':
    main()
