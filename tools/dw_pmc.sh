#!/bin/bash
# HBM traffic and VALU counters of the depthwise 3x3 kernels at one shape (default: the teacher's stage-1 map of BASELINE config 2), each
# counter set in its own rocprofv3 pass.  usage (GPU box): bash tools/dw_pmc.sh [side,C] [f32|bf16] > gpurun_out/pmc_dw3x3.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SHAPE=${1:-128,256}
DT=${2:-f32}
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/pdw_*
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pdw_$c -o p -- python3 $R/tools/dw_bench.py --shape $SHAPE --dtype $DT > /dev/null 2>&1 || exit 1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pdw_sq -o p -- python3 $R/tools/dw_bench.py --shape $SHAPE --dtype $DT > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d /tmp/pdw_kt -o p -- python3 $R/tools/dw_bench.py --shape $SHAPE --dtype $DT > /dev/null 2>&1 || exit 1
cd $R || exit 1
python3 tools/pmc_summary.py /tmp/pdw_FETCH_SIZE /tmp/pdw_WRITE_SIZE /tmp/pdw_sq --match dw3x3 > /tmp/pdw.json
python3 - <<PY
import json, glob, csv
d = json.load(open('/tmp/pdw.json'))
dur = {}
for f in glob.glob('/tmp/pdw_kt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'dw3x3' in n or 'Gelu' in n:
            dur.setdefault(n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in list(dur):
    v = sorted(dur[k])
    print(f'# median of {len(v):4d} launches {v[len(v) // 2]:7.1f} us (min {v[0]:.1f})  {k[:90]}')
    dur[k] = v[len(v) // 2]
print('# shape (side,C) = $SHAPE  dtype = $DT;  per launch: HBM read MB (FETCH_SIZE x 2 on gfx950, see MI355X_MICROARCH.md), write MB, VALU instructions per wave, median duration,')
print('# VALU busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz)')
for k, v in d.items():
    g = lambda c: v.get(c, {}).get('mean', float('nan'))
    us = dur.get(k, float('nan'))
    print(f"{k:<48} {us:7.1f} us  read {g('FETCH_SIZE') * 2 * 1024 / 1e6:7.1f} MB  write {g('WRITE_SIZE') * 1024 / 1e6:7.1f} MB  "
          f"VALU/wave {g('SQ_INSTS_VALU') / g('SQ_WAVES'):7.0f}  VALU busy {g('SQ_INSTS_VALU') * 4 / (1024 * us * 2400):5.2f}")
PY
