"""End-to-end wall time of the krisp_fasta flow on 4 x 50 Mbp synthetic FASTA files
(plain and gzip): host ingest vs device time.  python tools/e2e_check.py [length]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import synth
from krisp_amd import krisp_fasta as KF

length = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
fam = synth.family(2, 2, 2, length, records=16)
with tempfile.TemporaryDirectory() as td:
    paths = []
    for name, ing, text in fam:
        p = os.path.join(td, name + ".fasta")
        synth.write_fasta(p, text)
        paths.append(p)
    for rep in range(2):
        t0 = time.time()
        groups, stats = KF.find_regions(paths[:2], paths[2:], 25, 2, 28)
        print(f"plain fasta: total {time.time() - t0:.2f} s  read+check {stats['read_s']:.2f} s  "
              f"device(upload+sort+intersect+collect) {stats['device_s']:.2f} s  groups {len(groups)}")
