#!/bin/bash
# Copy what tools/refresh_profiles.sh (and tools/profile_configs.sh) left under gpurun_out/ into profiles/ under the round's prefix:
#   tools/publish_profiles.sh r03
# profiles/ is what is tracked and judged; gpurun_out/ is scratch.  traffic_rNN.json (the R1 kernels' PMC traffic) is the file bench.py reads
# for roofline.traffic.
set -u
P=${1:?round prefix, e.g. r03}
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/gpurun_out/refresh
DST=$R/profiles
n=0
put() { [ -s "$1" ] && cp "$1" "$2" && n=$((n + 1)); }
for f in bench.json bench_kernels.json bench_kernel_stats.csv bench_under_rocprof.json train_step_kernels.txt step_shapes.txt step_top5.json \
         gemm_bench.txt wgrad_splitk_bench.txt pred_bench.txt; do
  put $SRC/$f $DST/${P}_$f
done
put $SRC/other_configs.txt $DST/${P}_other_configs_throughput.txt
put $SRC/gemm_pmc_planes.txt $DST/${P}_pmc_gemm_planes.txt
for f in $SRC/kernels_*.txt $SRC/kernels_*_stats.csv $SRC/pmc_*.json; do
  [ -e "$f" ] && put $f $DST/${P}_$(basename $f)
done
put $SRC/pmc_traffic_r1.json $DST/traffic_${P}.json
put $R/gpurun_out/ab_switches.txt $DST/${P}_ab_switches.txt
for f in $R/gpurun_out/configs/train_step_kernels_cfg*.txt; do
  [ -e "$f" ] && put $f $DST/${P}_$(basename $f)
done
echo "published $n files under $DST/${P}_*"
