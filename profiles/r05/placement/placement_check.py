"""Does the time of pass 1 depend on where its output buffer lands?  One process, the engine created anew
several times (every hipMalloc anew): stage times of sorting one 50 Mbp genome.  python tools/placement_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import _native, synth  # noqa: E402

fam = synth.family(2, 1, 0, 50_000_000, records=16, mu=0.01, snp_every=10000)
text = fam[0][2]
for rep in range(8):
    with _native.Engine() as eng:
        eng.set_params(25, 1, 2, max_bases=len(text))
        eng.upload(0, text)
        for _ in range(3):
            eng.sort(0)
        eng.sync()
        eng.stage_enable(True)
        eng.stage_reset()
        for _ in range(10):
            eng.sort(0)
        eng.sync()
        st = eng.stage_times()
        print(rep, " ".join(f"{k}={v[0] / v[1] * 1e3:.0f}us" for k, v in st.items() if v[1] and k in ("scatter1", "scatter2", "localsort", "hist2")), flush=True)
