#!/bin/bash
# A/B of build variants inside ONE gpurun call (boxes differ by up to 10 %, so variants must be
# compared on the same box, interleaved).  Build the variants first, here, with tools/ab_build.sh;
# then: gpurun -- 'bash tools/ab.sh base varA varB'.  A variant that fails the quick parity
# subset is not benchmarked.  Variants are selected with KRISP_HIP_LIB (krisp_amd/_native.py):
# the product library is never overwritten.
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
OK=""
for v in "$@"; do
  export KRISP_HIP_LIB="$ROOT/krisp_amd/variants/$v.so"
  if [ "$AB_NOTEST" = "1" ]; then OK="$OK $v"; continue; fi     # (variants that cannot change results: cache-policy bits)
  if timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu > "$ROOT/gpurun_out/ab_$v.test.log" 2>&1; then
    OK="$OK $v"
  else
    echo "variant $v FAILED the parity subset"; tail -5 "$ROOT/gpurun_out/ab_$v.test.log"
  fi
done
for round in 1 2 3; do
  for v in $OK; do
    export KRISP_HIP_LIB="$ROOT/krisp_amd/variants/$v.so"
    timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$ROOT/gpurun_out/ab_$v.$round.json" 2>/dev/null || { echo "bench of $v failed"; continue; }
    python3 - "$v" "$round" "$ROOT/gpurun_out/ab_$v.$round.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms_per_step_calibration"]
print(sys.argv[1], "round", sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "G/s %.2f" % (d["value"] / 1e9),
      " ".join(f"{k}={v:.3f}" for k, v in st.items() if v > 0.1))
PY
  done
done
unset KRISP_HIP_LIB
