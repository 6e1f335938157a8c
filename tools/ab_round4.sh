#!/bin/bash
# Same-box A/B of round 4's step-level changes (30 steps after 5 warm-ups, graph replay, no roofline / CPU legs):
#   bash tools/ab_round4.sh > gpurun_out/ab_round4.txt
# (the PSPNet / Swin configs spend minutes in MIOpen's find: a heartbeat keeps the GPU-box watchdog quiet)
( while true; do sleep 60; echo "[heartbeat] $(date +%T)" >&2; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
run() { cfg=$1; shift; env "$@" python bench.py --config configs/kd/$cfg.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-f32 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-42s %-34s %8.1f imgs/s %8.3f ms/step' % ('$cfg', '$*', d['value'], d['ms_per_step']))"; }
run cfg2_segformer_b2_b0_cgd DEFAULT=1
run cfg2_segformer_b2_b0_cgd DEFAULT=1
run cfg3_segformer_b2_b0_cgd_cd DEFAULT=1
run cfg3_segformer_b2_b0_cgd_cd SEGDISTILL_FUSE_PAIRS=0
run cfg5_segformer_b4_b1_multistage_bf16 DEFAULT=1
run cfg4_pspnet_r18_swin_b_cgd_align DEFAULT=1
run cfg4_pspnet_r18_swin_b_cgd_align SEGDISTILL_PPM_POOL=0
run cfg1_pspnet_r101_r18_cd DEFAULT=1
run cfg1_pspnet_r101_r18_cd SEGDISTILL_PPM_POOL=0
