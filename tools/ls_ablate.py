"""Where k_localsort's time goes: python tools/ls_ablate.py  (one 50 Mbp genome, 25/1/2)"""
import sys

sys.path.insert(0, ".")
from krisp_amd import _native, synth  # noqa: E402

fam = synth.family(2, 1, 0, 50_000_000, records=16, mu=0.01, snp_every=10000)
with _native.Engine() as eng:
    eng.set_params(25, 1, 2, max_bases=len(fam[0][2]))
    n = eng.add(0, fam[0][2])
    for mode, what in ((0, "full"), (64, "load+store only"), (128, "count+scan+place+copy (no ranking)"),
                       (256, "ranked, stored in placed order"), (0, "full")):
        ms = eng.lib.kr_debug_localsort(eng.ctx, 0, 10, mode)
        print(f"{what:40s} {ms:.3f} ms  {16 * n / ms / 1e6:.0f} GB/s", flush=True)
    assert eng.inversions(0) == 0
