"""Where k_intersect's time goes (4 x 50 Mbp, 25/1/2).  The ablation switches exist only in a -DKR_ABLATE build:
    bash tools/ab_build.sh ablate -DKR_ABLATE -DKR_EXPERIMENTS
    KRISP_HIP_LIB=$PWD/krisp_amd/variants/ablate.so python tools/isect_ablate.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from krisp_amd import _native, synth  # noqa: E402

fam = synth.family(2, 2, 2, 50_000_000, records=16, mu=0.01, snp_every=10000)
with _native.Engine() as eng:
    eng.set_params(25, 1, 2, max_bases=max(len(t) for _, _, t in fam))
    for i, (_, _, t) in enumerate(fam):
        eng.add(i, t)
    ids = np.arange(len(fam), dtype=np.int32)
    flags = np.array([1 if f else 0 for _, f, _ in fam], dtype=np.uint8)
    n = sum(eng.count(i) for i in range(len(fam)))
    for mode, what in ((0, "full"), (64, "keys streamed, no probes"), (128, "probes, no LDS update"),
                       (512, "anchor work only (no streaming)"),
                       (512 + 1024, "anchor only, no own masks"), (512 + 2048, "anchor only, no survivors pass"),
                       (512 + 4096, "anchor only, no sub-bin table"), (512 + 1024 + 2048 + 4096, "anchor only, none of the three"),
                       (0, "full")):
        ms = eng.lib.kr_debug_intersect(eng.ctx, _native._ptr(ids), len(ids), _native._ptr(flags), 10, mode)
        print(f"{what:32s} {ms:.3f} ms  {8 * n / ms / 1e6:.0f} GB/s", flush=True)
