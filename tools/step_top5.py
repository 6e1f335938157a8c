"""The five kernels that dominate the steady-state KD step, for bench.py's `roofline.step_top5` (a STORED summary, refreshed by
tools/refresh_profiles.sh): name, ms per step and launches per step from the per-step kernel table (tools/prof_summary.py), bound and fraction
of that bound from the per-kernel-family table (bench.py --kernel-rooflines) where a family with that kernel was measured.

    python tools/step_top5.py train_step_kernels.txt [bench_kernels.json] > step_top5.json
"""
import json
import re
import sys


def main():
    table = open(sys.argv[1]).read().splitlines()
    fam = []
    if len(sys.argv) > 2:
        try:
            fam = [e for e in json.load(open(sys.argv[2])) if 'kernels' in e]
        except Exception:  # noqa: BLE001
            fam = []
    rows = []
    for ln in table:
        m = re.match(r'^(\S.*?)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)$', ln)
        if m and not ln.startswith('#') and not ln.startswith('kernel'):
            rows.append((m.group(1).strip(), float(m.group(2)), float(m.group(4))))
    # library GEMMs (Cijk_*) are many differently-named kernels: one row
    lib = [r for r in rows if r[0].startswith('Cijk_') or r[0].startswith('Custom_Cijk')]
    rows = [r for r in rows if r not in lib]
    if lib:
        rows.append(('hipBLASLt Cijk_* (all library GEMMs)', round(sum(r[1] for r in lib), 3), round(sum(r[2] for r in lib), 1)))
    rows.sort(key=lambda r: -r[1])
    out = []
    for name, ms, calls in rows[:5]:
        e = {'name': name[:60], 'ms_per_step': round(ms, 3), 'calls': calls}
        short = name.replace('sd::', '')
        hits = [f for f in fam if short and short.split('<')[0] in f.get('kernels', '')]
        if hits:
            e['bound'] = hits[0]['bound']
            e['frac_range'] = [round(min(h['frac'] for h in hits), 3), round(max(h['frac'] for h in hits), 3)]
        out.append(e)
    # stamped with what it was measured ON: bench.py quotes the table only while the source tree still has this fingerprint (a
    # stored table silently went stale when a kernel changed without a refresh); `commit` is informative (no .git on the GPU box -> null)
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import source_fingerprint
    try:
        commit = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:  # noqa: BLE001
        commit = None
    print(json.dumps({'fingerprint': source_fingerprint(), 'commit': commit, 'kernels': out}))


if __name__ == '__main__':
    main()
