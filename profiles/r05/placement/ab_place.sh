#!/bin/bash
# distribution of the bench value over processes (placement is per allocation): N runs per setting
for round in 1 2 3 4 5 6 7 8; do
  for cfg in "1 0" "8 0"; do
    set -- $cfg
    KR_PLACE_TRIES=$1 KR_PLACE_KEYS=$2 timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/abp.json 2>/dev/null || continue
    python3 - "tries=$1 keys=$2" gpurun_out/abp.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms_per_step_calibration"]
print(sys.argv[1], "G/s %.2f" % (d["value"] / 1e9), "scatter1=%.3f scatter2=%.3f localsort=%.3f intersect=%.3f" % (st["scatter1"], st["scatter2"], st["localsort"], st["intersect"]))
PY
  done
done
