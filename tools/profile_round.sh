#!/bin/bash
# Regenerates the measurement artefacts of one round on the GPU box (run through gpurun from the
# repo root):  bash tools/profile_round.sh r03
#   profiles/<round>/bench.json          the bench line (default workload = BASELINE configs[1])
#   profiles/<round>/kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
#   profiles/<round>/bench_one_lane.json, kernel_stats_one_lane.csv    the same pair with --lanes 1 (kernels alone)
#   profiles/traffic.json                HBM bytes per launch from the PMC passes (FETCH_SIZE, WRITE_SIZE)
# Everything is written under gpurun_out/<round>/ (merged back by gpurun); copy into profiles/ afterwards.
set -e
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cp "$OUT/bench.json" "$OUT/bench_first_process.json"      # (the box's first process: its placement class may differ from the last one's)
tail -1 "$OUT/bench.json" | head -c 1500; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timers > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timers > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
# the same pair with ONE sort lane: every kernel alone on the device (what roofline.alone of the default line reports;
# with the library's three lanes the launches of different genomes overlap, and under the profiler they overlap differently)
python3 bench.py --lanes 1 --no-cpu-baseline > "$OUT/bench_one_lane.json" 2>> "$OUT/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -- python3 "$ROOT/bench.py" --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/stats1.log" 2>&1
cd "$ROOT"
find "$OUT/stats1" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_one_lane.csv"
find "$OUT/stats1" -type f ! -name "*kernel_stats.csv" -delete
# one step as a timeline (kernel trace of the plain three-lane command)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-stage-timers > "$OUT/trace.log" 2>&1
cd "$ROOT"
T=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py "$T" > "$OUT/step_timeline_3lanes.txt" 2>> "$OUT/bench.err" || echo "no timeline"
rm -rf "$OUT/trace"
F=$(find "$OUT/pmc_fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/pmc_write" -name "*counter_collection.csv" | head -1)
python3 profiles/make_traffic.py "$F" "$W" "$OUT/traffic.json" "$R"
# the bench line once more, now with the PMC file of this very build beside it (roofline.traffic, roofline.step)
cp "$OUT/traffic.json" "$ROOT/profiles/traffic.json"
python3 bench.py > "$OUT/bench.json" 2>> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | head -c 600; echo
# the raw per-dispatch csv files are large: keep the summaries only
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
find "$OUT/stats" -type f ! -name "*kernel_stats.csv" -delete
ls -la "$OUT"
