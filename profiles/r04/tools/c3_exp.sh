#!/bin/bash
# tools/c3_exp.sh VARIANT : rocprofv3 kernel statistics of configs[2] with a variant library (timing experiments)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/c3prof
mkdir -p "$OUT"
export TMPDIR=/tmp
export KRISP_HIP_LIB="$ROOT/krisp_amd/variants/$1.so"
export C3_NOCHECK=1
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/v$1" -- python3 "$ROOT/tools/c3_check.py" 8 500 1 > "$OUT/run_$1.log" 2>&1
find "$OUT/v$1" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$1.csv"
rm -rf "$OUT/v$1"
grep -E "^run|slot bits|Error|error" "$OUT/run_$1.log"
python3 - "$OUT/kernel_stats_$1.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.1f} ms")
PY
