#!/bin/bash
# Per-step kernel tables of the BASELINE configs other than the headline one: rocprofv3 kernel trace of bench.py
# (the python program directly after `--`), summarised by tools/prof_summary.py with the once-per-step optimizer launch as the step marker.
#   tools/profile_configs.sh OUTDIR [cfg ...]      cfg = file name under configs/kd without .py
set -u
R=$(pwd)
OUT=$1; shift
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( while true; do sleep 45; echo "[heartbeat] $(date +%T)" >> $OUT/progress.txt; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
CFGS=${*:-"cfg1_pspnet_r101_r18_cd cfg3_segformer_b2_b0_cgd_cd cfg4_pspnet_r18_swin_b_cgd_align cfg5_segformer_b4_b1_multistage_bf16"}
for c in $CFGS; do
  tag=${c%%_*}
  echo "[$tag] $(date +%T)" | tee -a $OUT/progress.txt
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o step -- python3 $R/bench.py --config $R/configs/kd/$c.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline --no-exact-f32 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err )
  python tools/prof_summary.py /tmp/prof_$tag --marker adamw_multi --skip 7 --top 60 --out $OUT/train_step_kernels_$tag.txt > /dev/null 2>> $OUT/progress.txt
  python tools/step_kernel_shapes.py /tmp/prof_$tag "" --marker=adamw_multi --steps=4 > $OUT/step_shapes_$tag.txt 2>> $OUT/progress.txt
  head -2 $OUT/train_step_kernels_$tag.txt | tee -a $OUT/progress.txt
  grep -c naive_conv $OUT/train_step_kernels_$tag.txt | sed "s/^/[$tag] naive_conv rows in the steady-state table: /" | tee -a $OUT/progress.txt
done
exit 0
