#!/bin/bash
# Which kernels make `bench.py --deterministic` slow?  Kernel trace of a short deterministic run -> per-step table (gpurun_out/det_step_kernels.txt).
set -u
R=$(pwd)
OUT=$R/gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
CFG=${1:-configs/kd/cfg2_segformer_b2_b0_cgd.py}
TAG=${2:-det}
rm -rf /tmp/prof_$TAG
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o step -- python3 $R/bench.py --config $R/$CFG --deterministic --steps 6 --warmup 4 --repeats 1 \
    --no-cpu-baseline --no-roofline --no-exact-f32 --no-deterministic-child > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err )
python tools/prof_summary.py /tmp/prof_$TAG --skip 5 --top 40 --gaps 6 --out $OUT/${TAG}_step_kernels.txt > /dev/null
head -45 $OUT/${TAG}_step_kernels.txt
