#!/bin/bash
# SQ counter pass of the bench (one step, no stage timers): where the waves spend their cycles.
# bash tools/pmc_sq.sh r01  ->  gpurun_out/<round>/pmc_sq.csv + pmc_sq_summary.txt
set -e
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-stage-timers > "$OUT/pmc_sq.log" 2>&1
cd "$ROOT"
F=$(find "$OUT/pmc_sq" -name "*counter_collection.csv" | head -1)
python3 - "$F" "$OUT/pmc_sq_summary.txt" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["kernel                 waves_waiting  issuing  valu_active  lds_active   valu_insts/wave_cycle"]
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    wc = sum(c.get("SQ_WAVE_CYCLES", [0]))
    if wc <= 0 or not k.startswith("k_"):
        continue
    f = lambda n: sum(c.get(n, [0])) / wc
    lines.append(f"{k:22s} {f('SQ_WAIT_ANY'):12.2f} {f('SQ_ACTIVE_INST_ANY'):8.2f} {f('SQ_ACTIVE_INST_VALU'):12.2f} "
                 f"{f('SQ_ACTIVE_INST_LDS'):11.2f} {f('SQ_INSTS_VALU'):12.3f}")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf "$OUT/pmc_sq"
