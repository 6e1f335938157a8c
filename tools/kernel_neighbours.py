"""Who launches a kernel?  For every launch of kernels matching PATTERN in the last steady-state step of a rocprofv3 kernel trace of bench.py,
print the kernels that ran just before and after it on the same queue (stream order) -- the consumer names the call site.
    python tools/kernel_neighbours.py DIR PATTERN [--marker=adamw_multi] [--context=2]"""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
marker = next((a.split('=', 1)[1] for a in sys.argv[3:] if a.startswith('--marker=')), 'adamw_multi')
ctx = int(next((a.split('=', 1)[1] for a in sys.argv[3:] if a.startswith('--context=')), 2))
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
rows = rows[marks[-2]:marks[-1]] if len(marks) >= 2 else rows
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get('Queue_Id', '0')].append(r)


def short(r):
    n = r['Kernel_Name']
    return f"{n[:70]:70s} grid {r['Grid_Size_X']:>8s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us"


seen = collections.Counter()
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if pat in r['Kernel_Name']:
            key = tuple(short(x)[:70] for x in rs[max(0, i - ctx):i + ctx + 1])
            seen[(r['Grid_Size_X'], key)] += 1
for (grid, key), n in sorted(seen.items(), key=lambda kv: -kv[1])[:25]:
    print(f'--- {n} x  grid {grid}')
    for k in key:
        print('     ', k)
