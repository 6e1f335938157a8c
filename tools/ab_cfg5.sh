#!/bin/bash
# Same-box A/B of config 5's switches next to configs 2 and 3 on the same tree (30 steps after 5 warm-ups):
#   bash tools/ab_cfg5.sh [VAR=value ...] > gpurun_out/ab_cfg5.txt
run() { cfg=$1; shift; env "$@" python bench.py --config configs/kd/$cfg.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-f32 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-42s %-34s %8.1f imgs/s %8.3f ms/step %s' % ('$cfg', '$*', d['value'], d['ms_per_step'], d['config']['hip_graph']))"; }
run cfg2_segformer_b2_b0_cgd DEFAULT=1
run cfg5_segformer_b4_b1_multistage_bf16 DEFAULT=1
for sw in "$@"; do run cfg5_segformer_b4_b1_multistage_bf16 $sw; done
run cfg5_segformer_b4_b1_multistage_bf16 DEFAULT=1
run cfg3_segformer_b2_b0_cgd_cd DEFAULT=1
run cfg2_segformer_b2_b0_cgd DEFAULT=1
