#!/bin/bash
# PMC passes over the kernels one command launches: SQ issue / wait / MFMA counters, LDS counters, then FETCH_SIZE and WRITE_SIZE in passes of their
# own, then a kernel trace.   tools/pmc_run.sh OUTDIR MATCH TAG -- python3 script.py args...   (the program itself after --: no env / bash -c hops)
# Relative script / file arguments after `--` are resolved against the directory this is called from (the passes run in /tmp); a failed pass aborts
# with a non-zero status instead of leaving stale /tmp/pmc_*_TAG directories to be summarised.
set -eu
R=$(pwd)
OUT=$1; MATCH=$2; TAG=$3; shift 4
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
export TMPDIR=/tmp
ARGS=()
for a in "$@"; do
  if [[ $a != /* && -e $R/$a ]]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi
done
set -- "${ARGS[@]}"
rm -rf /tmp/pmc_a_$TAG /tmp/pmc_b_$TAG /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG /tmp/pmc_t_$TAG
: > $OUT/log_$TAG.txt
trap 'echo "pmc_run.sh: a rocprofv3 pass failed -- see $OUT/log_$TAG.txt" >&2' ERR
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_a_$TAG -o p -- "$@" >> $OUT/log_$TAG.txt 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d /tmp/pmc_b_$TAG -o p -- "$@" >> $OUT/log_$TAG.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$TAG -o p -- "$@" >> $OUT/log_$TAG.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$TAG -o p -- "$@" >> $OUT/log_$TAG.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_t_$TAG -o p -- "$@" >> $OUT/log_$TAG.txt 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc_a_$TAG /tmp/pmc_b_$TAG /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG --match "$MATCH" --out $OUT/pmc_$TAG.json > /dev/null
cp $(find /tmp/pmc_t_$TAG -name '*kernel_stats.csv' | head -1) $OUT/stats_$TAG.csv 2>/dev/null
python - <<PY
import json
d=json.load(open('$OUT/pmc_$TAG.json'))
for k,v in d.items():
    print(k[:90])
    print('   '+'  '.join(f"{c}={x['mean']:.4g}" for c,x in sorted(v.items())))
PY
grep "$MATCH" $OUT/stats_$TAG.csv | cut -c1-220
