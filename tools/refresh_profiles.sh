#!/bin/bash
# Regenerate the round's measurement artifacts on an MI355X box (run from the repo root; writes under gpurun_out/refresh/).
#   1 bench.json              the default `python bench.py` line (roofline + cpu_baseline)
#   2 bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command (+ the line it printed under the profiler)
#   3 train_step_kernels.txt  per-step kernel table of the steady-state steps (tools/prof_summary.py)
#   4 other_configs.txt       the other BASELINE configs, graph and hybrid mode
#   5 microbench.txt          per-kernel micro-benchmarks (tools/kbench_step_kernels.py, tools/sra_bench.py)
set -u
R=$(pwd)
OUT=$R/gpurun_out/refresh
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python3 $R/bench.py --no-cpu-baseline > /tmp/bench_prof.out 2>/dev/null )
tail -1 /tmp/bench_prof.out > $OUT/bench_under_rocprof.json
cp $(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
# per-step table: the same steps without the roofline legs (they launch the marker kernel too)
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_step -o step -- python3 $R/bench.py --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
python tools/prof_summary.py /tmp/prof_step --skip 8 --top 70 --out $OUT/train_step_kernels.txt > /dev/null
: > $OUT/other_configs.txt
CONFIGS=${CONFIGS:-"cfg3_segformer_b2_b0_cgd_cd cfg5_segformer_b4_b1_multistage_bf16 cfg1_pspnet_r101_r18_cd cfg4_pspnet_r18_swin_b_cgd_align"}
for c in $CONFIGS; do
  for g in on hybrid; do
    timeout 900 python bench.py --config configs/kd/$c.py --steps 10 --warmup 4 --graph $g --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$c', 'graph=$g', d['config']['hip_graph'], d['value'], 'imgs/s', d['ms_per_step'], 'ms/step', d['dtype'], 'B=%d' % d['config']['per_gpu_batch'])" >> $OUT/other_configs.txt 2>&1
  done
done
( python tools/kbench_step_kernels.py; python tools/sra_bench.py; python tools/sra_bench.py bf16 ) 2>/dev/null | grep -v amdgpu.ids > $OUT/microbench.txt
ls -la $OUT
cat $OUT/bench.json | cut -c1-1500
cat $OUT/other_configs.txt
