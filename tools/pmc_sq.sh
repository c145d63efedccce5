#!/bin/bash
# SQ counters of one BASELINE config's kernels (round 6):  bash tools/pmc_sq.sh 2 r06
#   one rocprofv3 --pmc pass (kernel trace only: no other trace domain) of `bench.py --config N --steps 1 --warmup 0`;
#   per kernel name the sums of SQ_WAVE_CYCLES, SQ_WAIT_ANY (waves parked on s_waitcnt / a barrier), SQ_WAIT_INST_ANY (issue
#   stalls), SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU -> gpurun_out/<round>/sq_cN.json: where a kernel's
#   wave-cycles go (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
set -e
N=${1:-2}
R=${2:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/sq_c$N" -- python3 "$ROOT/bench.py" --config $N --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timers > "$OUT/sq_c$N.log" 2>&1
cd "$ROOT"
F=$(find "$OUT/sq_c$N" -name "*counter_collection.csv" | head -1)
python3 - "$F" "$OUT/sq_c$N.json" <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, v in acc.items():
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    out[k] = {c: v[c] for c in sorted(v)}
    out[k]["share"] = {c: round(v.get(c, 0.0) / wc, 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")}
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])), open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:8]:
    print(k, v["share"], "VALU insts", v.get("SQ_INSTS_VALU"), "VMEM_RD", v.get("SQ_INSTS_VMEM_RD"), "LDS", v.get("SQ_INSTS_LDS"))
PY
rm -rf "$OUT/sq_c$N"
