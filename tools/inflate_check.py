"""kr_genome_upload_bgzf alone (round 6): a FASTA text of `mb` MB as a BGZF file (members of 65280 bytes, zlib level `level`)
-> the device inflate + parse, three times; prints the upload's wall time and the inflate kernels' (k_bgzf_inflate +
k_bgzf_crc) own.  Under `rocprofv3 --kernel-trace --stats` the two kernels show separately.
    python tools/inflate_check.py [mb, default 1024] [level, default 6]            (on the GPU box)"""
import os
import struct
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import _native  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(1)
n = mb << 20
t0 = time.time()
body = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n, dtype=np.uint8)]
body[80::81] = 10
text = b">chr1 synthetic\n" + body.tobytes()


def member(i):
    ch = text[i:i + 65280]
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cd = co.compress(ch) + co.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd
            + struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch)))


with ThreadPoolExecutor(16) as pool:
    raw = b"".join(pool.map(member, range(0, len(text), 65280)))
raw += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"
arr = np.frombuffer(raw, dtype=np.uint8)
print(f"{len(text) / 1e6:.0f} MB of text, {len(raw) / 1e6:.0f} MB as BGZF (level {level}), made in {time.time() - t0:.0f} s", flush=True)
with _native.Engine() as eng:
    eng.set_params(25, 1, 2, max_bases=len(text))
    for rep in range(3):
        t1 = time.time()
        got = eng.upload_bgzf(0, arr)
        t2 = time.time()
        assert got is not None, eng.last_bgzf
        print(f"run {rep}: {got[0]:,} bases, {got[5]} members; upload + inflate + parse {t2 - t1:.3f} s, inflate kernels {got[6] / 1e3:.1f} ms "
              f"= {len(text) / got[6] / 1e3:.1f} GB/s of text", flush=True)
    t1 = time.time()
    eng.upload_text(1, np.frombuffer(text, dtype=np.uint8), False)
    print(f"the text itself through kr_genome_upload_text: {time.time() - t1:.3f} s")
