#!/bin/bash
# Same-box A/B of config 4's step-level switches (30 steps after 5 warm-ups; MIOpen's find runs for minutes: heartbeat for the GPU-box watchdog):
#   bash tools/ab_cfg4.sh > gpurun_out/ab_cfg4.txt
( while true; do sleep 60; echo "[heartbeat] $(date +%T)" >&2; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
run() { cfg=$1; shift; env "$@" python bench.py --config configs/kd/$cfg.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-f32 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-42s %-34s %8.1f imgs/s %8.3f ms/step' % ('$cfg', '$*', d['value'], d['ms_per_step']))"; }
run cfg4_pspnet_r18_swin_b_cgd_align DEFAULT=1
for sw in "$@"; do run cfg4_pspnet_r18_swin_b_cgd_align $sw; done
run cfg4_pspnet_r18_swin_b_cgd_align DEFAULT=1
