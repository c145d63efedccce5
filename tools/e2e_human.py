"""krisp_fasta end to end at human scale (BASELINE configs[4]'s genomes as FILES): two `gzip` FASTA files of `length`
bases each (1 in / 1 out, k = 31 as 28/1/2) -> read, inflate (one member per file, cut into chunks: csrc/h_pgzip.inc),
parse on the device, sort in key-space slices, intersect + filter, collect, render.  The genomes differ in `mu` of their
bases (default 2e-4: ~10^6 diagnostic groups; SURVEY 8(d)'s 0.01 would give 6.5e7 groups = gigabytes of text).
    python tools/e2e_human.py [length, default 3e9] [mu] [genomes, default 2]            (on the GPU box; writes to stdout)
With three genomes (2 in / 1 out) the sorted set no longer fits 288 GB: the streaming flow takes them in batches
(krisp_fasta._find_regions_streaming; round 5)."""
import os
import sys
import tempfile
import threading
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import amplicon, fasta, synth  # noqa: E402
from krisp_amd import krisp_fasta as KF  # noqa: E402

length = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-4
ngen = int(sys.argv[3]) if len(sys.argv) > 3 else 2
as_bgzf = len(sys.argv) > 4 and sys.argv[4] == "bgzf"       # (round 6: BGZF members of 65280 bytes, inflated on the device; "bgzf" as 4th argument)
t0 = time.time()
fam = synth.family(5, ngen - 1, 1, length, records=24, mu=mu, snp_every=100_000)
print(f"{ngen} x {length / 1e9:g} Gbp, mu = {mu:g}: generated in {time.time() - t0:.0f} s", flush=True)


class _BgzfWriter:
    """compressobj-like: the bytes handed over leave as BGZF members of 65280 bytes of text (bgzip's framing)"""

    def __init__(self, level):
        self.level, self.buf = level, bytearray()

    def _member(self, chunk):
        import struct
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        cd = co.compress(bytes(chunk)) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd
                + struct.pack("<II", zlib.crc32(bytes(chunk)) & 0xFFFFFFFF, len(chunk)))

    def compress(self, data):
        self.buf += data
        out = []
        while len(self.buf) >= 65280:
            out.append(self._member(self.buf[:65280]))
            del self.buf[:65280]
        return b"".join(out)

    def flush(self):
        out = (self._member(self.buf) if self.buf else b"") + self._member(b"")
        self.buf = bytearray()
        return out


def fasta_gz(path, text, width=80, level=1):
    """80-column FASTA of the records of `text` (records separated by newline), as one gzip member (or as BGZF)"""
    co = _BgzfWriter(level) if as_bgzf else zlib.compressobj(level, zlib.DEFLATED, 31)
    n = 0
    with open(path, "wb") as f:
        start = 0
        arr = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        ends = list(np.flatnonzero(arr == 10)) + [len(arr)]
        for i, e in enumerate(ends):
            rec = arr[start:e]
            start = e + 1
            if len(rec) == 0:
                continue
            f.write(co.compress(b">rec%d\n" % i))
            for a in range(0, len(rec), 64 * 1000 * width):
                part = rec[a:a + 64 * 1000 * width]
                whole = len(part) - len(part) % width
                rows = part[:whole].reshape(-1, width)
                body = np.concatenate([rows, np.full((rows.shape[0], 1), 10, dtype=np.uint8)], axis=1).tobytes()
                if whole < len(part):
                    body += part[whole:].tobytes() + b"\n"
                n += len(body)
                f.write(co.compress(body))
        f.write(co.flush())
    return n


with tempfile.TemporaryDirectory() as td:
    t1 = time.time()
    paths, sizes = [], [0] * ngen

    def one(i):
        name, ing, text = fam[i]
        p = os.path.join(td, name + ".fa.gz")
        paths.append((i, p))
        sizes[i] = fasta_gz(p, text)
    th = [threading.Thread(target=one, args=(i,)) for i in range(ngen)]
    [x.start() for x in th]
    [x.join() for x in th]
    paths = [p for _, p in sorted(paths)]
    del fam
    print(f"written as gzip -1 files in {time.time() - t1:.0f} s: {[round(os.path.getsize(p) / 1e9, 2) for p in paths]} GB "
          f"for {[round(s / 1e9, 2) for s in sizes]} GB of text", flush=True)
    for rep in range(2):
        fasta.LAST_TIMINGS.clear()
        t2 = time.time()
        groups, stats = KF.find_regions(paths[:ngen - 1], paths[ngen - 1:], 28, 2, 31)
        t3 = time.time()
        csv, align = amplicon.render(groups, [KF.simplename(p) for p in paths[:ngen - 1]])
        t4 = time.time()
        tm = list(fasta.LAST_TIMINGS.values())
        mx = lambda key: max((t.get(key, 0.0) for t in tm), default=0.0)  # noqa: E731
        print(f"run {rep}: total {t4 - t2:.2f} s | ingest wall {stats['read_s']:.2f} s (slowest file: read {mx('read_s'):.2f} inflate "
              f"{mx('inflate_s'):.2f}" + (f", inflate kernels {mx('device_inflate_s'):.3f}" if as_bgzf else "") + f") | upload + parse + sort + intersect + collect + grouping {stats['device_s']:.2f} s | render "
              f"{t4 - t3:.2f} s | {stats['kmers']:,} k-mers, {len(groups):,} groups, CSV {len(csv) / 1e6:.1f} MB, alignment {len(align) / 1e6:.1f} MB",
              flush=True)
        print("       device part: " + ", ".join(f"{k} {v:.3f}" for k, v in stats.get("stage_s", {}).items()), flush=True)
        if stats.get("streamed"):
            print(f"       streamed: batches of {stats['batch']} genome(s), {stats['batches']} batch(es), {stats['passes']} pass(es)", flush=True)
        del groups, csv, align
