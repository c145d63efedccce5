#!/bin/bash
# One bench line per BASELINE.json config (and SURVEY 8(d)'s secondary inputs), truthful labels, into
# gpurun_out/<round>/variants/ (copy to profiles/<round>/variants/ afterwards):
#   bash tools/profile_variants.sh r02
R=${1:-r02}
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
run configs1_4x50Mbp --steps 10 --warmup 2 --no-cpu-baseline
run configs1_independent --steps 10 --warmup 2 --no-cpu-baseline --independent
run configs1_masked --steps 10 --warmup 2 --no-cpu-baseline --masked
run configs3_per_gpu_load_4x100Mbp --steps 10 --warmup 2 --no-cpu-baseline --length 100000000
run configs3_all_32x100Mbp_one_gpu --steps 3 --warmup 1 --no-cpu-baseline --length 100000000 --per-gpu 32
run configs4_2x3Gbp_28_1_2 --steps 3 --warmup 1 --no-cpu-baseline --length 3000000000 --per-gpu 2 --ldr 28 1 2
run rccl_world1_selftest --steps 5 --warmup 2 --no-cpu-baseline --force-comm
echo "== configs[2] 8 x 500 Mbp 32/60/32 (wide path)"
timeout -k 10 900 python3 tools/c3_check.py 8 500 > "$OUT/configs2_8x500Mbp_32_60_32.log" 2>&1; tail -4 "$OUT/configs2_8x500Mbp_32_60_32.log"
for N in 2 4; do
  echo "== N=$N ranks sharing the GPU (file transport rehearsal)"
  rm -rf /tmp/krisp_reh_$N; mkdir -p /tmp/krisp_reh_$N
  pids=""
  for r in $(seq 0 $((N-1))); do
    RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N KRISP_COMM_FILE=/tmp/krisp_reh_$N/c timeout -k 10 600 python3 bench.py --gpus $N --steps 5 --warmup 1 --transport dir --length 10000000 > "$OUT/rehearsal_n${N}_rank$r.json" 2> "$OUT/rehearsal_n${N}_rank$r.err" &
    pids="$pids $!"
  done
  for p in $pids; do wait $p || echo "rank failed"; done
  tail -c 600 "$OUT/rehearsal_n${N}_rank0.json" | head -c 600; echo
done
