#!/bin/bash
# Regenerate the round's measurement artifacts on an MI355X box (run from the repo root; writes under gpurun_out/refresh/).
#   1 bench.json              the `python bench.py --kernel-rooflines` line (roofline + cpu_baseline + exact-f32 A/B) and bench_kernels.json,
#                             the per-kernel-family table (child processes started before bench.py touches the GPU)
#   2 bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command (+ the line it printed under the profiler)
#   3 train_step_kernels.txt  per-step kernel table of the steady-state steps (tools/prof_summary.py)
#   4 other_configs.txt       the other BASELINE configs, graph and hybrid mode
#   5 kernels_*.csv / pmc_*.json   rocprofv3 of tools/kernel_rooflines.py: kernel trace, then FETCH_SIZE / WRITE_SIZE / SQ_* in SEPARATE --pmc passes
#   6 gemm_bench.txt          token-major Linear products: library vs csrc/token_gemm.hip, device time
#   7 step_gap_probe.txt      unprofiled GPU start -> end of a step;   8 kernels_tails.txt / pmc_tails.json / *_tail_bench.txt / overlap_probe.txt (round 6)
# rocprofv3 is always given the python program directly after `--`.
set -u
R=$(pwd)
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
# heartbeat: a quiet step (MIOpen find for the ResNet / Swin configs takes minutes) must not look like a hang to the GPU-box watchdog
( while true; do sleep 45; echo "[heartbeat] $(date +%T)" >> $OUT/progress.txt; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
STEPS=${STEPS:-"1 2 3 4 5 6"}     # a full pass is ~17 GPU-minutes: STEPS="1 2 3 4" and STEPS="5 6" fit two calls
want() { [[ " $STEPS " == *" $1 "* ]]; }
echo "start $(date +%T) steps: $STEPS" | tee $OUT/progress.txt
if want 1; then
echo "[1] bench" | tee -a $OUT/progress.txt
python bench.py --kernel-rooflines --exact-f32-steps 20 --kernels-out $OUT/bench_kernels.json 2>$OUT/bench.err | tail -1 > $OUT/bench.json
fi
if want 2; then
echo "[2] bench under rocprof" | tee -a $OUT/progress.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python3 $R/bench.py --no-cpu-baseline --no-exact-f32 > /tmp/bench_prof.out 2>/dev/null )
tail -1 /tmp/bench_prof.out > $OUT/bench_under_rocprof.json
cp $(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
fi
if want 3; then
echo "[3] per-step table" | tee -a $OUT/progress.txt
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_step -o step -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-exact-f32 > /dev/null 2>&1 )
python tools/prof_summary.py /tmp/prof_step --skip 8 --top 90 --gaps 12 --out $OUT/train_step_kernels.txt > /dev/null
python tools/step_kernel_shapes.py /tmp/prof_step "" > $OUT/step_shapes.txt 2>/dev/null
python tools/step_top5.py $OUT/train_step_kernels.txt $OUT/bench_kernels.json > $OUT/step_top5.json 2>/dev/null
fi
if want 7; then
echo "[7] step gap probes" | tee -a $OUT/progress.txt
{ python tools/step_gap_probe.py 2>/dev/null | tail -1 | sed 's/^/cfg2 one rank:                      /'
  SEGDISTILL_FORCE_COLLECTIVES=1 SEGDISTILL_FORCE_SYNCBN=1 python tools/step_gap_probe.py 2>/dev/null | tail -1 | sed 's/^/cfg2 one rank, SyncBN + RCCL forced (3 segments): /'
  python tools/step_gap_probe.py --config configs/kd/cfg5_segformer_b4_b1_multistage_bf16.py 2>/dev/null | tail -1 | sed 's/^/cfg5 (bf16) one rank:               /'
} > $OUT/step_gap_probe.txt
fi
if want 4; then
echo "[4] other configs" | tee -a $OUT/progress.txt
: > $OUT/other_configs.txt
CONFIGS=${CONFIGS:-"cfg3_segformer_b2_b0_cgd_cd cfg5_segformer_b4_b1_multistage_bf16 cfg1_pspnet_r101_r18_cd cfg4_pspnet_r18_swin_b_cgd_align"}
for c in $CONFIGS; do
  for g in on hybrid; do
    timeout 900 python bench.py --config configs/kd/$c.py --steps 30 --warmup 5 --graph $g --no-cpu-baseline --no-roofline --no-exact-f32 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$c', 'graph=$g', d['config']['hip_graph'], d['value'], 'imgs/s', d['ms_per_step'], 'ms/step', d['dtype'], 'B=%d' % d['config']['per_gpu_batch'])" >> $OUT/other_configs.txt 2>&1
    echo "   $c $g done" | tee -a $OUT/progress.txt
  done
done
fi
if want 5; then
echo "[5] kernel rooflines: trace + PMC passes" | tee -a $OUT/progress.txt
for g in r1 tok aligntok bf16gemm wgradmulti r2 ce align sra optim dw ln upsum resize gemm pred ifvd pix pixup at wgrad_bf16 ppm wattn; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kr_$g -o k -- python3 $R/tools/kernel_rooflines.py --only $g > $OUT/kernels_$g.txt 2>/dev/null )
  cp $(find /tmp/kr_$g -name '*kernel_stats.csv' | head -1) $OUT/kernels_${g}_stats.csv 2>/dev/null
done
for g in r1 tok; do
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$g -o p -- python3 $R/tools/kernel_rooflines.py --only $g > /dev/null 2>&1 )
  ( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$g -o p -- python3 $R/tools/kernel_rooflines.py --only $g > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pf_$g /tmp/pw_$g --out $OUT/pmc_traffic_$g.json > /dev/null
done
for g in r2 ce; do
  ( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pv_$g -o p -- python3 $R/tools/kernel_rooflines.py --only $g > /dev/null 2>&1 )
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pvf_$g -o p -- python3 $R/tools/kernel_rooflines.py --only $g > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pv_$g /tmp/pvf_$g --out $OUT/pmc_valu_$g.json > /dev/null
done
fi
if want 8; then
echo "[8] the frozen teacher's tail kernels: trace + PMC passes" | tee -a $OUT/progress.txt
g=tails
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kr_$g -o k -- python3 $R/tools/kernel_rooflines.py --only $g > $OUT/kernels_$g.txt 2>/dev/null )
cp $(find /tmp/kr_$g -name '*kernel_stats.csv' | head -1) $OUT/kernels_${g}_stats.csv 2>/dev/null
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$g -o p -- python3 $R/tools/kernel_rooflines.py --only $g > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$g -o p -- python3 $R/tools/kernel_rooflines.py --only $g > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pv_$g -o p -- python3 $R/tools/kernel_rooflines.py --only $g > /dev/null 2>&1 )
python tools/pmc_summary.py /tmp/pf_$g /tmp/pw_$g /tmp/pv_$g --match "tail_x3" --out $OUT/pmc_$g.json > /dev/null
python tools/mixffn_tail_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/mixffn_tail_bench.txt
python tools/mixffn_tail_bench.py --dtype bf16 2>/dev/null | grep -v amdgpu.ids >> $OUT/mixffn_tail_bench.txt
python tools/head_tail_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/head_tail_bench.txt
python tools/overlap_probe.py 2>/dev/null | grep "full step" > $OUT/overlap_probe.txt
python tools/overlap_probe.py --config configs/kd/cfg5_segformer_b4_b1_multistage_bf16.py 2>/dev/null | grep "full step" >> $OUT/overlap_probe.txt
fi
if want 6; then
echo "[6] gemm bench" | tee -a $OUT/progress.txt
python tools/gemm_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/gemm_bench.txt
python tools/wgrad_splitk_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/wgrad_splitk_bench.txt
python tools/wgrad_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/wgrad_bench.txt
python tools/gemm_nosplit_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/gemm_nosplit_probe_product.txt
[ -f segdistill_amd/lib_ab/libsegdistill_hip.so ] && SEGDISTILL_LIB=$R/segdistill_amd/lib_ab/libsegdistill_hip.so python tools/gemm_nosplit_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/gemm_nosplit_probe_diag.txt
python tools/pred_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/pred_bench.txt
tools/gemm_pmc.sh $OUT --mode pl > $OUT/gemm_pmc_planes.txt 2>&1
fi
ls -la $OUT | tee -a $OUT/progress.txt
[ -f $OUT/bench.json ] && cut -c1-600 $OUT/bench.json
[ -f $OUT/other_configs.txt ] && cat $OUT/other_configs.txt
exit 0
