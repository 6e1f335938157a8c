#!/bin/bash
# Same-box A/B of one config's switches (30 steps after 5 warm-ups; default first and last):
#   bash tools/ab_cfg.sh cfg2_segformer_b2_b0_cgd VAR=value [VAR=value ...] > gpurun_out/ab_cfg2_x.txt
( while true; do sleep 60; echo "[heartbeat] $(date +%T)" >&2; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cfg=$1; shift
run() { env "$@" python bench.py --config configs/kd/$cfg.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-f32 --no-deterministic-child $BENCH_FLAGS 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-42s %-38s %8.1f imgs/s %8.3f ms/step %s' % ('$cfg', '$*', d['value'], d['ms_per_step'], d['config']['hip_graph']), d.get('value_min'), d.get('value_max'), 'peak_mem_gb', d['config'].get('peak_mem_gb'))"; }
run DEFAULT=1
for sw in "$@"; do run $sw; done
run DEFAULT=1
