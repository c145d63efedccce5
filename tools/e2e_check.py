"""End-to-end krisp_fasta on 4 x 50 Mbp synthetic genomes as .fasta.gz (and plain .fasta) files, with
a stage table: read / inflate / parse (inside the library: kr_ingest_file, files concurrently),
IUPAC scan, upload + sort + intersect + collect (device), grouping, render.
    python tools/e2e_check.py [length] [ngenomes]            (on the GPU box; writes to stdout)"""
import gzip
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import amplicon, fasta, synth  # noqa: E402
from krisp_amd import krisp_fasta as KF  # noqa: E402

length = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fam = synth.family(2, ng // 2, ng - ng // 2, length, records=16)
with tempfile.TemporaryDirectory() as td:
    plain, gz = [], []
    for name, ing, text in fam:
        p = os.path.join(td, name + ".fasta")
        synth.write_fasta(p, text)
        plain.append(p)
        with open(p, "rb") as f, open(p + ".gz", "wb") as g:
            g.write(gzip.compress(f.read(), compresslevel=6))
        gz.append(p + ".gz")
    print(f"{ng} x {length / 1e6:g} Mbp; plain {os.path.getsize(plain[0]) / 1e6:.1f} MB, gz {os.path.getsize(gz[0]) / 1e6:.1f} MB per file")
    for kind, paths in (("fasta.gz", gz), ("fasta", plain)):
        for rep in range(2):
            fasta.LAST_TIMINGS.clear()
            t0 = time.time()
            groups, stats = KF.find_regions(paths[:ng // 2], paths[ng // 2:], 25, 2, 28)
            t1 = time.time()
            csv, align = amplicon.render(groups, [KF.simplename(p) for p in paths[:ng // 2]])
            t2 = time.time()
            tm = list(fasta.LAST_TIMINGS.values())
            mx = lambda key: max((t[key] for t in tm), default=0.0)  # noqa: E731
            print(f"{kind:9s} run {rep}: total {t2 - t0:.3f} s | ingest wall {stats['read_s']:.3f} s "
                  f"(slowest file: read {mx('read_s'):.3f} inflate {mx('inflate_s'):.3f} parse {mx('parse_s'):.3f}; "
                  f"libdeflate {any(t['libdeflate'] for t in tm)}) | upload+sort+intersect+collect+grouping "
                  f"{stats['device_s']:.3f} s | render {t2 - t1:.3f} s | {stats['kmers']:,} k-mers, {len(groups)} groups")
