"""End-to-end with a LARGE result: 4 related genomes, all ingroup (no filter survivors to prune):
every conserved (left,right) group is reported.  python tools/e2e_big_output.py [length]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import amplicon, synth
from krisp_amd import krisp_fasta as KF

length = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
fam = synth.family(2, 4, 0, length, records=16)
with tempfile.TemporaryDirectory() as td:
    paths = []
    for name, ing, text in fam:
        p = os.path.join(td, name + ".fasta")
        synth.write_fasta(p, text)
        paths.append(p)
    t0 = time.time()
    groups, stats = KF.find_regions(paths, [], 25, 2, 28)
    t1 = time.time()
    csv, aln = amplicon.render(groups, None, dot=False)
    t2 = time.time()
    print(f"groups {len(groups):,}: find_regions {t1 - t0:.2f} s (read {stats['read_s']:.2f}, device {stats['device_s']:.2f}), "
          f"render {t2 - t1:.2f} s, csv {len(csv) / 1e6:.1f} MB, alignment {len(aln) / 1e6:.1f} MB")
