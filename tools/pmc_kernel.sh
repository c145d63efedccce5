#!/bin/bash
# counters of ONE kernel of the bench (one step): bash tools/pmc_kernel.sh k_intersect3 "SQ_WAVES SQ_INSTS_VALU ..." ["more counters" ...]
# one rocprofv3 pass per quoted group (a pass takes what the hardware can count at once)
K=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$K; mkdir -p "$OUT"; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-stage-timers ${BENCH_ARGS} > "$OUT/p$i.log" 2>&1 )
  F=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$K" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0] == sys.argv[2]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{sys.argv[2]} {k:28s} per launch {sum(v)/len(v):16.0f}   launches {len(v)}")
PY
  rm -rf "$OUT/p$i"
done
