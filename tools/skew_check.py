"""Sort time of a repeat-rich 50 Mbp genome (poly-A / (AT)n / (CAG)n tracts, an interspersed
repeat family, a satellite array) next to a clean random one: exercises the oversized-bucket
fallback (tile sort + merge rounds).  python tools/skew_check.py  (needs the GPU)"""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from krisp_amd import _native, synth
rng = np.random.default_rng(1)
n = 50_000_000
codes = rng.integers(0, 4, size=n, dtype=np.uint8)
# low complexity: 2% of the genome in poly-A / (AT)n / (CAG)n tracts of 100..1000 bp
pos = 0
tracts = 0
while tracts < 0.02 * n:
    p = int(rng.integers(0, n - 2000)); ln = int(rng.integers(100, 1000))
    unit = [np.array([0]), np.array([0, 3]), np.array([1, 0, 2])][int(rng.integers(0, 3))]
    codes[p:p + ln] = np.resize(unit, ln)
    tracts += ln
# interspersed repeat: one 300-bp element, 8000 copies at 10% divergence (~5% of the genome)
elem = rng.integers(0, 4, size=300, dtype=np.uint8)
for p in rng.integers(0, n - 300, size=8000):
    e = elem.copy()
    m = rng.random(300) < 0.10
    e[m] = (e[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
    codes[p:p + 300] = e
# one long satellite array: 171-bp monomer x 3000 copies, 2% divergence
mono = rng.integers(0, 4, size=171, dtype=np.uint8)
sat = np.tile(mono, 3000)
m = rng.random(len(sat)) < 0.02
sat[m] = (sat[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
codes[1_000_000:1_000_000 + len(sat)] = sat
text = synth.codes_to_text(codes, records=16)
clean = synth.codes_to_text(rng.integers(0, 4, size=n, dtype=np.uint8), records=16)
for name, t in (("clean", clean), ("repeat-rich", text)):
    with _native.Engine() as e:
        e.set_params(25, 1, 2, max_bases=len(t))
        e.upload(0, t)
        e.sort(0); e.count(0)
        t0 = time.perf_counter()
        e.sort(0); cnt = e.count(0)
        dt = time.perf_counter() - t0
        print(name, "keys", cnt, "sort ms", round(dt * 1e3, 2), "inversions", e.inversions(0), e.debug_info())
