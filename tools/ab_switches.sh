#!/bin/bash
# Same-box A/B of the step-level switches on the headline workload (config 2, graph replay, 30 steps after 5 warm-ups, no roofline / CPU legs).
#   bash tools/ab_switches.sh > gpurun_out/ab_switches.txt
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-f32 2>/dev/null | tail -1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-58s %8.1f imgs/s %8.3f ms/step' % ('$*', d['value'], d['ms_per_step']))"; }
run DEFAULT=1
run SEGDISTILL_HIP_ADAMW=0
run SEGDISTILL_SPLITK_WGRAD=0
run SEGDISTILL_SPLIT_BF16=0
run SEGDISTILL_TOKEN_GEMM=0
run SEGDISTILL_PRED_PLANES=0
run SEGDISTILL_SPLITK_WGRAD=0      # round 3's remaining step-level switches off
run SEGDISTILL_FUSE_PAIRS=0          # (config 3 only) the two criteria on linear_pred as two passes
run SEGDISTILL_SPLIT_BF16=0 SEGDISTILL_TOKEN_GEMM=0 SEGDISTILL_HIP_ADAMW=0
run DEFAULT=1
