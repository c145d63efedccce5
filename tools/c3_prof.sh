#!/bin/bash
# rocprofv3 kernel statistics of configs[2] (tools/c3_check.py) with and without dictionary slot tables
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/c3prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for s in ${C3_SLOTS:-1 0}; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s$s" -- python3 "$ROOT/tools/c3_check.py" 8 500 $s > "$OUT/run_s$s.log" 2>&1 || exit 1
  find "$OUT/s$s" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_s$s.csv"
  rm -rf "$OUT/s$s"
  grep -E "^run|slot bits" "$OUT/run_s$s.log"
  python3 - "$OUT/kernel_stats_s$s.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.1f} ms")
PY
done
