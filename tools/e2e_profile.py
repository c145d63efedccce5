"""krisp_fasta.find_regions from plain FASTA files, packed (25/1/2) and long amplicons (32/60/32), once each -- to be run
under `rocprofv3 --kernel-trace --stats`: which kernels a whole command-line run spends its device time in (round 6: that is
how the one-workgroup line scan of the device's reader was found).
    python tools/e2e_profile.py [length, default 50e6] [genomes, default 4]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import synth  # noqa: E402
from krisp_amd import krisp_fasta as KF  # noqa: E402

length = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fam = synth.family(2, ng // 2, ng - ng // 2, length, records=16)
with tempfile.TemporaryDirectory() as td:
    paths = []
    for name, _, text in fam:
        p = os.path.join(td, name + ".fasta")
        synth.write_fasta(p, text)
        paths.append(p)
    for L, D, R in ((25, 1, 2), (32, 60, 32)):
        for rep in range(2):
            t0 = time.time()
            groups, stats = KF.find_regions(paths[:ng // 2], paths[ng // 2:], L, R, L + D + R)
            print(f"{L}/{D}/{R} run {rep}: {time.time() - t0:.3f} s, {len(groups)} groups, device part {stats.get('device_s', 0):.3f} s "
                  + ", ".join(f"{k} {v:.3f}" for k, v in stats.get("stage_s", {}).items()), flush=True)
