"""median / min of the A/B rounds tools/ab3.sh left under gpurun_out/ (ab_<spec>.<round>.json)"""
import glob, json, statistics, sys, collections
acc = collections.defaultdict(list)
st = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    name = f.split("/")[-1][3:].rsplit(".", 2)[0]
    acc[name].append(d["ms_per_step"])
    for k, v in d["roofline"]["stage_ms_per_step_calibration"].items():
        st[name][k].append(v)
for name, v in acc.items():
    print(f"{name:40s} n={len(v)} ms/step median {statistics.median(v):.3f} min {min(v):.3f}  " +
          " ".join(f"{k}={statistics.median(x):.3f}" for k, x in st[name].items() if statistics.median(x) > 0.03))
