"""Per-launch-shape breakdown of selected kernels in a rocprofv3 kernel trace of bench.py (steady-state steps):
    rocprofv3 --kernel-trace --output-format csv -d DIR -o step -- python3 bench.py --no-cpu-baseline --no-roofline
    python tools/step_kernel_shapes.py DIR token_gemm linear_wgrad_direct [--marker=cgd_up_bwd]   ('' = every kernel)
Prints, per (kernel template, grid, workgroup), calls per step, median duration and ms per step."""
import collections
import csv
import glob
import re
import sys

d, pats = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith('--')]
steps = int(next((a.split('=', 1)[1] for a in sys.argv[2:] if a.startswith('--steps=')), 18))
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steady state = the last `steps` periods of a kernel that runs once per step (default: the fused CGD backward); warm-up, MIOpen's find
# runs and the graph capture come before
marker = next((a.split('=', 1)[1] for a in sys.argv[2:] if a.startswith('--marker=')), 'cgd_up_bwd')
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
if len(marks) > steps:
    rows = rows[marks[-steps - 1]:marks[-1]]
else:
    rows = rows[len(rows) // 3:]
g = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if any(p in n for p in pats):
        m = re.search(r'(\w+)<([^>]*)>', n)
        key = ((m.group(1) + '<' + m.group(2) + '>') if m else n[:60], r['Grid_Size_X'], r['Grid_Size_Y'] + 'x' + r.get('Grid_Size_Z', '1'), r['Workgroup_Size_X'])
        g[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f'{k[0][:70]:70s} grid {k[1]:>8s}x{k[2]:<8s} wg {k[3]:>4s}  n/step {len(v) / steps:5.1f}  median {v[len(v) // 2] / 1e3:8.1f} us  ms/step {sum(v) / steps / 1e6:6.3f}  share {100 * sum(v) / tot:5.1f} %')
