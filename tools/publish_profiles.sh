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
         gemm_bench.txt wgrad_splitk_bench.txt pred_bench.txt wgrad_bench.txt gemm_nosplit_probe_product.txt gemm_nosplit_probe_diag.txt step_gap_probe.txt; do
  put $SRC/$f $DST/${P}_$f
done
put $SRC/other_configs.txt $DST/${P}_other_configs_throughput.txt
put $SRC/gemm_pmc_planes.txt $DST/${P}_pmc_gemm_planes.txt
for f in $SRC/kernels_*.txt $SRC/kernels_*_stats.csv $SRC/pmc_*.json; do
  [ -e "$f" ] && put $f $DST/${P}_$(basename $f)
done
# traffic_rNN.json = the raw counter dump PLUS the summary keys bench.py's roofline.traffic reads (cgd_kl_r1_fwd_bwd_bytes ...), with the
# gfx950 counter corrections applied by bench.traffic_from_counters (round 3 copied the raw dump alone and the bench line said null)
if [ -s $SRC/pmc_traffic_r1.json ]; then
  ( cd $R && python - "$SRC/pmc_traffic_r1.json" "$DST/traffic_${P}.json" "$P" <<'PY'
import json, sys
from bench import traffic_from_counters
d = traffic_from_counters(json.load(open(sys.argv[1])))
d = {'note': f'{sys.argv[3]}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python3 tools/kernel_rooflines.py --only r1` '
             '(config-2 operand shape [8,150,512,512] fp32; tools/refresh_profiles.sh); counter unit 1024 B; FETCH_SIZE doubled per MI355X_MICROARCH.md '
             '(gfx950 tallies the 128-B requests of 16-B/lane streaming reads at 64 B); WRITE_SIZE exact for 16-B/lane streaming stores.', **d}
json.dump(d, open(sys.argv[2], 'w'), indent=1)
PY
  ) && n=$((n + 1))
fi
for f in align_tok_bench.txt pix_up_bench.txt bf16_gemm_bench.txt mixffn_tail_bench.txt head_tail_bench.txt overlap_probe.txt; do
  put $SRC/$f $DST/${P}_$f
done
for f in $R/gpurun_out/configs/train_step_kernels_cfg*.txt $R/gpurun_out/configs/step_shapes_cfg*.txt \
         $SRC/prof5/train_step_kernels_cfg*.txt $SRC/prof5/step_shapes_cfg*.txt; do
  [ -e "$f" ] && put $f $DST/${P}_$(basename $f)
done
echo "published $n files under $DST/${P}_*"
