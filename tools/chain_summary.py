"""Per-QUEUE view of a rocprofv3 --kernel-trace of bench.py: the student step (the queue that runs adamw_multi) is the critical chain, the look-ahead
teacher forward runs on another queue and only fills its gaps -- so what shortens a step is what shortens the STUDENT queue's span.
usage: python tools/chain_summary.py <dir with *_kernel_trace.csv> [--skip 8] [--top 30] [--out file]"""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--marker', default='adamw_multi')
    ap.add_argument('--skip', type=int, default=8)
    ap.add_argument('--top', type=int, default=30)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        sys.exit('no kernel_trace.csv under ' + a.dir)
    rows = []
    for f in files:
        with open(f, newline='') as fh:
            for r in csv.DictReader(fh):
                q = r.get('Queue_Id') or r.get('Stream_Id') or '0'
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], q))
    rows.sort()
    marks = [(s, q) for s, e, n, q in rows if a.marker in n]
    if len(marks) <= a.skip + 1:
        sys.exit(f'only {len(marks)} marker dispatches')
    main_q = marks[-1][1]
    t0, t1 = marks[a.skip][0], marks[-1][0]
    nsteps = len(marks) - 1 - a.skip
    out = []
    per_q = defaultdict(lambda: defaultdict(lambda: [0, 0]))
    span = defaultdict(list)
    for s, e, n, q in rows:
        if t0 <= s < t1:
            k = short(n)
            per_q[q][k][0] += e - s
            per_q[q][k][1] += 1
            span[q].append((s, e))
    out.append(f'# {nsteps} steady-state steps, wall {(t1 - t0) / nsteps / 1e6:.3f} ms/step; queues: ' + ', '.join(
        f'{q}{" (student: runs " + a.marker + ")" if q == main_q else ""}: {sum(v[0] for v in per_q[q].values()) / nsteps / 1e6:.3f} ms busy, '
        f'{sum(v[1] for v in per_q[q].values()) / nsteps:.0f} launches' for q in sorted(per_q)))
    # how much of the other queues' kernel time lies inside the student queue's own busy intervals (true concurrency) is not derivable from
    # durations alone; what is: the student queue's busy sum vs the step's wall = its idle share
    for q in sorted(per_q, key=lambda z: z != main_q):
        tot = sum(v[0] for v in per_q[q].values())
        out.append(f'\n## queue {q}{" -- the student step" if q == main_q else ""}: {tot / nsteps / 1e6:.3f} ms/step of kernel time')
        out.append(f'{"kernel":92s} {"ms/step":>8s} {"%":>6s} {"calls":>7s} {"avg us":>8s}')
        for k, (ns, c) in sorted(per_q[q].items(), key=lambda kv: -kv[1][0])[:a.top]:
            out.append(f'{k:92s} {ns / nsteps / 1e6:8.3f} {100 * ns / tot:6.2f} {c / nsteps:7.1f} {ns / c / 1e3:8.1f}')
    text = '\n'.join(out)
    if a.out:
        open(a.out, 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main()
