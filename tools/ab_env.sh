#!/bin/bash
# tools/ab_env.sh VAR v1 v2 ... : the bench under values of one environment switch, interleaved, inside ONE gpurun call
VAR=$1; shift
for round in 1 2; do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/abenv_$v.json 2>/dev/null || { echo "bench failed for $VAR=$v"; continue; }
    python3 - "$VAR=$v" gpurun_out/abenv_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms_per_step_calibration"]
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "G/s %.2f" % (d["value"] / 1e9), d["roofline"]["kernel"], "frac %.3f" % d["roofline"]["frac"],
      " ".join(f"{k}={v:.3f}" for k, v in st.items() if v > 0.1))
PY
  done
done
