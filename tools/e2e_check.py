"""End-to-end krisp_fasta on 4 x 50 Mbp synthetic genomes as .fasta.gz (gzip: one member; BGZF: members of 64 KiB,
inflated side by side), .fasta.bz2 (one stream; streams of 8 MB as pbzip2 writes them, side by side) and plain .fasta
files, with a stage table: read / inflate / parse (inside the library: kr_read_file, files concurrently), IUPAC scan,
upload + sort + intersect + collect (device), grouping, render.
    python tools/e2e_check.py [length] [ngenomes]            (on the GPU box; writes to stdout)"""
import bz2
import gzip
import struct
import zlib
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import amplicon, fasta, synth  # noqa: E402
from krisp_amd import krisp_fasta as KF  # noqa: E402



def bgzf(data, block=0xFF00):
    """the BGZF framing of bgzip (SAM spec 4.1)"""
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
                   + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


length = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fam = synth.family(2, ng // 2, ng - ng // 2, length, records=16)
with tempfile.TemporaryDirectory() as td:
    plain, gz, bg, bz, bzm = [], [], [], [], []
    for name, ing, text in fam:
        p = os.path.join(td, name + ".fasta")
        synth.write_fasta(p, text)
        plain.append(p)
        raw = open(p, "rb").read()
        for lst, sub, blob in ((gz, "gz1", lambda: gzip.compress(raw, compresslevel=6)), (bg, "bgzf", lambda: bgzf(raw)),
                               (bz, "bz1", lambda: bz2.compress(raw, 9)),
                               (bzm, "bzm", lambda: b"".join(bz2.compress(raw[i:i + (8 << 20)], 9) for i in range(0, len(raw), 8 << 20)))):
            os.makedirs(os.path.join(td, sub), exist_ok=True)
            q = os.path.join(td, sub, name + (".fasta.bz2" if sub.startswith("bz") else ".fasta.gz"))
            with open(q, "wb") as g:
                g.write(blob())
            lst.append(q)
    print(f"{ng} x {length / 1e6:g} Mbp; plain {os.path.getsize(plain[0]) / 1e6:.1f} MB, gz {os.path.getsize(gz[0]) / 1e6:.1f} MB, "
          f"bgzf {os.path.getsize(bg[0]) / 1e6:.1f} MB, bz2 {os.path.getsize(bz[0]) / 1e6:.1f} MB per file")
    # (round 6: a BGZF file is inflated on the device, kr_genome_upload_bgzf; "bgzf host" = KRISP_DEVICE_INFLATE=0, as before)
    for kind, paths in (("fasta.gz", gz), ("bgzf .gz", bg), ("bgzf host", bg), ("fasta.bz2", bz), ("bz2 x8MB", bzm), ("fasta", plain)):
        os.environ["KRISP_DEVICE_INFLATE"] = "0" if kind == "bgzf host" else "1"
        os.environ["KRISP_DEVICE_INFLATE_MIN"] = "0"          # (whatever the size: the default leaves files below 1 GB of text to the host)
        for rep in range(2):
            fasta.LAST_TIMINGS.clear()
            t0 = time.time()
            groups, stats = KF.find_regions(paths[:ng // 2], paths[ng // 2:], 25, 2, 28)
            t1 = time.time()
            csv, align = amplicon.render(groups, [KF.simplename(p) for p in paths[:ng // 2]])
            t2 = time.time()
            tm = list(fasta.LAST_TIMINGS.values())
            mx = lambda key: max((t.get(key, 0.0) for t in tm), default=0.0)  # noqa: E731
            print(f"{kind:9s} run {rep}: total {t2 - t0:.3f} s | ingest wall {stats['read_s']:.3f} s "
                  f"(slowest file: read {mx('read_s'):.3f} inflate {mx('inflate_s'):.3f} parse {mx('parse_s'):.3f}; "
                  f"libdeflate {any(t['libdeflate'] for t in tm)}"
                  + (f"; inflate kernels {mx('device_inflate_s'):.4f}" if any('device_inflate_s' in t for t in tm) else "")
                  + ") | upload+sort+intersect+collect+grouping "
                  f"{stats['device_s']:.3f} s | render {t2 - t1:.3f} s | {stats['kmers']:,} k-mers, {len(groups)} groups")
            if stats.get("stage_s") and stats["device_s"] > 0.15:
                print("          (device part: " + ", ".join(f"{k} {v:.3f}" for k, v in stats["stage_s"].items()) + ")")
