#!/bin/bash
# the bench value of N fresh processes in a row on one box (placement classes: profiles/r05/README.md): value, ms per step,
# pass-1 / pass-2 stage times of every process.   bash tools/twenty_processes.sh [N=20]
N=${1:-20}
ROOT=$(pwd)
for i in $(seq 1 $N); do
  python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 > /tmp/tp.json 2>/dev/null || { echo "run $i failed"; continue; }
  python3 - $i <<'PY'
import json, sys
d = json.loads(open("/tmp/tp.json").read().strip().splitlines()[-1])
st = d["roofline"]["stage_ms_per_step_calibration"]
print("process %2s: %.2f G k-mers/s  %.3f ms/step  scatter1 %.3f  scatter2 %.3f  localsort %.3f  intersect %.3f" %
      (sys.argv[1], d["value"] / 1e9, d["ms_per_step"], st["scatter1"], st["scatter2"], st["localsort"], st["intersect"]), flush=True)
PY
done
