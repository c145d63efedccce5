#!/bin/bash
# One bench line per BASELINE.json config (bench.py --config N: the label comes from the table, not from a guess) and
# for SURVEY 8(d)'s secondary inputs, into gpurun_out/<round>/variants/ (copy to profiles/<round>/variants/ afterwards):
#   bash tools/profile_variants.sh r03
R=${1:-r04}
PART=${2:-all}      # "a": configs[1] variants + configs[3]; "b1": configs[4]; "b2": configs[2], transports; "b": both; "all"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R/variants
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name"; timeout -k 10 900 python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "$name FAILED"; python3 - "$OUT/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   %.2f G k-mers/s  %.3f ms/step  %s | %s frac %.3f" % (d["value"] / 1e9, d["ms_per_step"], d["config"]["baseline_config"],
          d["roofline"]["kernel"], d["roofline"]["frac"]))
except Exception as e:
    print("   no line:", e)
PY
}
if [ "$PART" = a ] || [ "$PART" = all ]; then
run configs1_4x50Mbp --config 1 --steps 10 --warmup 2
run configs1_independent --config 1 --steps 10 --warmup 2 --no-cpu-baseline --independent
run configs1_masked --config 1 --steps 10 --warmup 2 --no-cpu-baseline --masked
KR_LANES=1 run configs1_one_sort_lane --config 1 --steps 10 --warmup 2 --no-cpu-baseline
KR_LANES=2 run configs1_two_sort_lanes --config 1 --steps 10 --warmup 2 --no-cpu-baseline
KR_LANES=4 run configs1_four_sort_lanes --config 1 --steps 10 --warmup 2 --no-cpu-baseline
KR_ISECT_KERNEL=1 run configs1_chunk_intersect_kernel --config 1 --steps 10 --warmup 2 --no-cpu-baseline
KR_ISECT_KERNEL=2 run configs1_pipelined_64bit_heads --config 1 --steps 10 --warmup 2 --no-cpu-baseline
KR_ISECT_FMT=2 run configs1_32bit_heads_64bit_state --config 1 --steps 10 --warmup 2 --no-cpu-baseline
KR_ISECT_SPLIT=1 run configs1_late_genomes_probe --config 1 --steps 10 --warmup 2 --no-cpu-baseline
run configs3_per_gpu_load_4x100Mbp --config 3 --steps 10 --warmup 2 --no-cpu-baseline
run configs3_all_32x100Mbp_one_gpu --config 3 --per-gpu 32 --steps 3 --warmup 1 --no-cpu-baseline
fi
if [ "$PART" = b ] || [ "$PART" = b1 ] || [ "$PART" = all ]; then
run configs4_2x3Gbp_28_1_2 --config 4
KR_LANES=1 run configs4_one_lane --config 4 --no-cpu-baseline
KR_SLICE_ROUTE=0 run configs4_round3_slice_route --config 4 --no-cpu-baseline
fi
if [ "$PART" = b ] || [ "$PART" = b2 ] || [ "$PART" = all ]; then
run configs2_8x500Mbp_32_60_32 --config 2
run configs2_8x500Mbp_32_60_32_mu0.001 --config 2 --mu 0.001 --records 24 --snp-every 20000 --no-cpu-baseline
run rccl_world1_selftest --config 1 --steps 5 --warmup 2 --no-cpu-baseline --force-comm
for N in 2 4; do
  # (no launcher: bench.py starts its own N ranks; they share this box's GPU over the file transport -- the N > 1 flow of
  # the library, not a scaling measurement)
  run selflaunch_n${N}_file_transport --gpus $N --transport dir --steps 5 --warmup 1 --no-cpu-baseline
done
fi
