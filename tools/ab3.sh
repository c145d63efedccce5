#!/bin/bash
# A/B inside one gpurun call: "name[:ENV=VAL[,ENV=VAL]]" ... ; name = a library under krisp_amd/variants/ or "lib" (the product library)
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
for round in $(seq 1 ${AB_ROUNDS:-3}); do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""
    [[ "$spec" == *:* ]] && envs=${spec#*:}
    (
      [[ "$v" != lib ]] && export KRISP_HIP_LIB="$ROOT/krisp_amd/variants/$v.so"
      IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [[ -n "$e" ]] && export "$e"; done
      tag=$(echo "$spec" | tr ':=,' '___')
      timeout -k 10 150 python3 bench.py --steps ${AB_STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > "$ROOT/gpurun_out/ab_$tag.$round.json" 2>"$ROOT/gpurun_out/ab_$tag.$round.err" || { echo "bench of $spec failed"; tail -3 "$ROOT/gpurun_out/ab_$tag.$round.err"; exit 0; }
      python3 - "$spec" "$round" "$ROOT/gpurun_out/ab_$tag.$round.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms_per_step_calibration"]
print(sys.argv[1], "round", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "G/s %.2f" % (d["value"] / 1e9),
      " ".join(f"{k}={v:.3f}" for k, v in st.items() if v > 0.05), flush=True)
PY
    )
  done
done
