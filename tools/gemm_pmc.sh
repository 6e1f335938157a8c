#!/bin/bash
# PMC passes over ONE GEMM kernel (tools/gemm_one.py): SQ issue / wait / MFMA counters, then FETCH_SIZE and WRITE_SIZE in passes of their own.
#   tools/gemm_pmc.sh OUTDIR [gemm_one.py arguments ...]
set -u
R=$(pwd)
OUT=$1; shift
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
export TMPDIR=/tmp
TAG=$(echo "$*" | tr -c 'A-Za-z0-9' '_')
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_a_$TAG -o p -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d /tmp/pmc_b_$TAG -o p -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$TAG -o p -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$TAG -o p -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_t_$TAG -o p -- python3 $R/tools/gemm_one.py "$@" > /dev/null 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc_a_$TAG /tmp/pmc_b_$TAG /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG --match token_gemm --out $OUT/pmc_$TAG.json > /dev/null
cp $(find /tmp/pmc_t_$TAG -name '*kernel_stats.csv' | head -1) $OUT/stats_$TAG.csv 2>/dev/null
python - <<PY
import json
d=json.load(open('$OUT/pmc_$TAG.json'))
for k,v in d.items():
    print(k[:90])
    print('   '+'  '.join(f"{c}={x['mean']:.3g}" for c,x in sorted(v.items())))
PY
grep token_gemm $OUT/stats_$TAG.csv | cut -c1-200
