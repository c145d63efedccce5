#!/bin/bash
# how many fast candidates does a pool of the placement search hold, and what does the step take?  (on the GPU box)
for tries in 8 8 8; do
  for r in 1 2 3; do
    KR_PLACE_TRIES=$tries KR_TRACE_ALLOC=1 python3 bench.py --no-cpu-baseline --steps 40 --warmup 3 --no-stage-timers > /tmp/pp.json 2> /tmp/pp.err
    python3 - $tries <<'PY'
import json, re, sys
ms = sorted(float(m) for m in re.findall(r"probe ([0-9.]+) ms", open("/tmp/pp.err").read()))
d = json.loads(open("/tmp/pp.json").read().strip().splitlines()[-1])
print(f"tries {sys.argv[1]:>2s}: pool {len(ms):3d}  best six {[round(x, 3) for x in ms[:6]]}  < 0.2 ms: {sum(x < 0.2 for x in ms)}  "
      f"-> {d['ms_per_step']:.3f} ms/step", flush=True)
PY
  done
done
