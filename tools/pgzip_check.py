"""One gzip member on several host threads (csrc/h_pgzip.inc; SURVEY 8(f) rank 1, kstream.py:458-479): how long the library
takes to turn a `gzip genome.fa` file into text, with the member cut into chunks against one thread, for one large genome
and for the four 50 Mbp files of configs[1] read side by side as the command line reads them.
    python tools/pgzip_check.py [Mbp of the large genome, default 400] [gzip level, default 6]     (on the GPU box; writes to stdout)"""
import gzip
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import _native  # noqa: E402

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 400
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tmp = os.environ.get("TMPDIR", "/tmp")
rng = np.random.default_rng(31)


def fasta(nbases, records=16, width=80):
    per = nbases // records
    out = []
    for r in range(records):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=per)
        rows = seq[:per - per % width].reshape(-1, width)
        body = np.concatenate([rows, np.full((rows.shape[0], 1), 10, dtype=np.uint8)], axis=1).tobytes()
        out.append(b">record%d synthetic\n" % r + body)
    return b"".join(out)


def timed(paths, **env):
    os.environ.update({k: str(v) for k, v in env.items()})
    got = [None] * len(paths)

    def one(i):
        got[i] = _native.read_file(paths[i])
    t0 = time.time()
    th = [threading.Thread(target=one, args=(i,)) for i in range(len(paths))]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.time() - t0
    return dt, got


print(f"host cores {os.cpu_count()}, usable {len(os.sched_getaffinity(0))}", flush=True)
t0 = time.time()
text = fasta(mbp * 1_000_000)
big = os.path.join(tmp, "pgz_big.fa.gz")
with open(big, "wb") as f:
    f.write(gzip.compress(text, compresslevel=level))
print(f"{mbp} Mbp genome: {len(text) / 1e6:.0f} MB of text, {os.path.getsize(big) / 1e6:.0f} MB as .gz (level {level}), written in {time.time() - t0:.0f} s", flush=True)
for threads in (16, 32, 8, 4):
    dt, got = timed([big], KRISP_PGZIP=1, KRISP_INGEST_THREADS=threads)
    ok = got[0][0].tobytes() == text
    print(f"  chunks on {threads:2d} threads: {dt:.3f} s = {len(text) / dt / 1e9:.2f} GB/s of text (inflate alone "
          f"{got[0][2]['inflate_s']:.3f} s), same text: {ok}", flush=True)
    got = None
dt, got = timed([big], KRISP_PGZIP=0)
print(f"  one thread (libdeflate): {dt:.3f} s = {len(text) / dt / 1e9:.2f} GB/s of text, same text: {got[0][0].tobytes() == text}", flush=True)
got = None
small = []
texts = []
for g in range(4):
    t = fasta(50_000_000)
    p = os.path.join(tmp, f"pgz_small{g}.fa.gz")
    with open(p, "wb") as f:
        f.write(gzip.compress(t, compresslevel=6))
    small.append(p)
    texts.append(t)
for on in (1, 0, 1, 0):
    dt, got = timed(small, KRISP_PGZIP=on, KRISP_INGEST_THREADS=16)
    ok = all(g[0].tobytes() == t for g, t in zip(got, texts))
    print(f"4 x 50 Mbp files read side by side, chunks {'on ' if on else 'off'}: {dt:.3f} s, same text: {ok}", flush=True)
for p in [big] + small:
    os.unlink(p)
