"""The output side of long amplicons at scale: BASELINE configs[2]'s 8 x 500 Mbp,
32/60/32, with round 2's close relatives (mu = 0.001: > 10^6 surviving groups, > 10^7 member windows) -- how long it takes
to turn kr_wide_run's hits into the final text: fetch the hits, cut the windows on the device (kr_wide_fetch_windows),
render them in the library (kr_render_windows), against the general path on a sample of the groups.
    python tools/wide_render_check.py [length] [mu]            (on the GPU box; writes to stdout)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import _native, amplicon, synth  # noqa: E402

length = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
L, D, R = 32, 60, 32
k = L + D + R
t0 = time.time()
fam = synth.family(3, 4, 4, length, records=24, mu=mu, snp_every=20000)
labels = [nm for nm, _, _ in fam]
flags = [f for _, f, _ in fam]
print(f"8 x {length / 1e6:g} Mbp, mu = {mu:g}, {L}/{D}/{R}: generated in {time.time() - t0:.0f} s", flush=True)
with _native.Engine() as eng:
    eng.set_params_wide(L, D, R, max_bases=max(len(t) for _, _, t in fam))
    for i, (_, _, t) in enumerate(fam):
        eng.upload(i, t)
    del fam
    for rep in range(2):
        t1 = time.time()
        n = eng.wide_run(list(range(8)), flags, apply_filter=True)
        eng.sync()
        t2 = time.time()
        print(f"wide_run {rep}: {t2 - t1:.3f} s, {n:,} member windows", flush=True)
    t2 = time.time()
    hits = eng.wide_fetch(_native.WIDE_HITS)
    t3 = time.time()
    rows = eng.wide_windows(k)
    t4 = time.time()
wg = amplicon.WindowGroups(rows, hits["cand"], hits["genome"], labels, L, D, R)
ingroup = frozenset(lab for lab, f in zip(labels, flags) if f)
t5 = time.time()
csv, align = wg.render_text(ingroup, False)
t6 = time.time()
print(f"{len(wg):,} groups, {len(rows):,} windows: hits to the host {t3 - t2:.3f} s, windows cut on the device and copied "
      f"({rows.nbytes / 1e9:.2f} GB) {t4 - t3:.3f} s, render in the library (incl. the texts as Python strings) {t6 - t5:.3f} s "
      f"-> CSV {len(csv) / 1e6:.1f} MB, alignment {len(align) / 1e6:.1f} MB; total {t6 - t2:.3f} s")
# the general path on the first 20000 windows' groups (whole groups): the same text
cut = 20000
while cut < len(rows) and hits["cand"][cut] == hits["cand"][cut - 1]:
    cut += 1
sel = np.isin(hits["cand"], np.unique(hits["cand"][:cut]))
sub = amplicon.WindowGroups(rows[sel], hits["cand"][sel], hits["genome"][sel], labels, L, D, R)
t7 = time.time()
want = amplicon.render(sub.groups(), ingroup, False)
t8 = time.time()
got = sub.render_text(ingroup, False)
print(f"general path on {len(sub):,} groups: {t8 - t7:.3f} s ({(t8 - t7) / max(len(sub), 1) * len(wg):.1f} s extrapolated to all); "
      f"library == general path: {got == want}")
