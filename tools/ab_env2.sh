#!/bin/bash
# tools/ab_env2.sh "VAR=a VAR2=b" "VAR=c" ...   -- A/B of ENVIRONMENT settings with the product library inside one gpurun call,
# interleaved over 3 rounds; extra bench.py arguments in $AB_ARGS
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
for round in 1 2 3; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $AB_ARGS > "$ROOT/gpurun_out/abenv_$i.$round.json" 2>/dev/null || { echo "bench with [$v] failed"; continue; }
    python3 - "$v" "$round" "$ROOT/gpurun_out/abenv_$i.$round.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms_per_step_calibration"]
print("[%s]" % sys.argv[1], "round", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "G/s %.2f" % (d["value"] / 1e9),
      " ".join(f"{k}={v:.3f}" for k, v in st.items() if v > 0.1))
PY
  done
done
