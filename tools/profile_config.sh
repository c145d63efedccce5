#!/bin/bash
# Counter-backed roofline of one BASELINE config (round 5):  bash tools/profile_config.sh 4 r05
#   three runs of `bench.py --config N` on the GPU box: rocprofv3 --kernel-trace --stats (time per kernel name), --pmc FETCH_SIZE,
#   --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) -> gpurun_out/<round>/configsN_*:
#     configsN_kernel_stats.csv   the --stats summary
#     traffic_cN.json             HBM bytes of every kernel of ONE step (profiles/make_traffic.py; FETCH x 2 + WRITE)
#     configsN_kernels.json       per kernel name: launches, ms (kernel alone: --lanes 1) and measured bytes per step, TB/s
#     configsN_bench.json         the bench line with roofline.step from those bytes
set -e
N=${1:-4}
R=${2:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# (--lanes 1: every kernel alone on the device -- with the library's lanes the launches of different genomes / slices overlap
# and a launch lasts longer while the step gets shorter; the step's own time comes from the plain bench line below)
echo "stats pass"; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c${N}_stats" -- python3 "$ROOT/bench.py" --config $N --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > "$OUT/c${N}_stats.log" 2>&1
echo "fetch pass"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/c${N}_fetch" -- python3 "$ROOT/bench.py" --config $N --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timers > "$OUT/c${N}_fetch.log" 2>&1
echo "write pass"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/c${N}_write" -- python3 "$ROOT/bench.py" --config $N --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timers > "$OUT/c${N}_write.log" 2>&1
cd "$ROOT"
find "$OUT/c${N}_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/configs${N}_kernel_stats.csv"
F=$(find "$OUT/c${N}_fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/c${N}_write" -name "*counter_collection.csv" | head -1)
python3 profiles/make_traffic.py "$F" "$W" "$OUT/traffic_c${N}.json" "$R" "--config $N"
python3 - "$OUT/configs${N}_kernel_stats.csv" "$OUT/traffic_c${N}.json" "$OUT/configs${N}_kernels.json" <<'PY'
import csv, json, sys
stats = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].split("(")[0].replace("void ", "").split("<")[0]
    s = stats.setdefault(name, [0, 0.0])
    s[0] += int(r["Calls"]); s[1] += float(r["TotalDurationNs"])
tj = json.load(open(sys.argv[2]))
steps_in_stats = 3          # --steps 2 --warmup 1 (no calibration steps: --no-stage-timers)
out = {}
for k, v in tj["_step"]["kernels"].items():
    if k not in stats:
        continue
    ms = stats[k][1] / 1e6 / steps_in_stats
    b = v["read_bytes"] + v["write_bytes"]
    out[k] = {"launches_per_step": v["launches"], "ms_per_step": round(ms, 3), "read_GB": round(v["read_bytes"] / 1e9, 3),
              "write_GB": round(v["write_bytes"] / 1e9, 3), "TBps": round(b / (ms * 1e-3) / 1e12, 3) if ms > 0 else None,
              "frac_of_8TBps": round(b / (ms * 1e-3) / 8e12, 4) if ms > 0 else None}
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"])), open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]:
    print(k, v)
PY
cp "$OUT/traffic_c${N}.json" "$ROOT/profiles/traffic_c${N}.json"
python3 bench.py --config $N > "$OUT/configs${N}_bench.json" 2> "$OUT/configs${N}_bench.err"
tail -1 "$OUT/configs${N}_bench.json" | head -c 400; echo
rm -rf "$OUT/c${N}_fetch" "$OUT/c${N}_write"
find "$OUT/c${N}_stats" -type f ! -name "*kernel_stats.csv" -delete
