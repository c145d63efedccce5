"""Seeded synthetic genomes for benchmarks and parity tests:
ancestor = iid uniform ACGT; each genome = ancestor with iid substitutions at
rate mu; every `snp_every` bases one planted site where all ingroup genomes get
base b1 and all outgroup genomes b2 != b1; `records` equal-length records.
Variants: `independent` (no common ancestor), `n_frac`/`lower_frac` (N runs of
1 kb, soft-masked stretches)."""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def ancestor(config, length):
    rng = np.random.Generator(np.random.PCG64(1000 + config))
    return rng.integers(0, 4, size=length, dtype=np.uint8)


def genome_codes(config, g, length, is_ingroup, mu=0.01, snp_every=10000, independent=False,
                 anc=None):
    """-> uint8 codes 0..3 of genome g."""
    rng = np.random.Generator(np.random.PCG64(2000 + 100 * config + g))
    if independent:
        return rng.integers(0, 4, size=length, dtype=np.uint8)
    if anc is None:
        anc = ancestor(config, length)
    codes = anc.copy()
    nmut = rng.binomial(length, mu)
    pos = rng.integers(0, length, size=nmut)
    codes[pos] = (codes[pos] + rng.integers(1, 4, size=nmut, dtype=np.uint8)) & 3
    # planted ingroup/outgroup SNPs (same sites and bases for every genome of the config)
    prng = np.random.Generator(np.random.PCG64(3000 + config))
    sites = np.arange(snp_every // 2, length, snp_every)
    b1 = prng.integers(0, 4, size=len(sites), dtype=np.uint8)
    b2 = (b1 + prng.integers(1, 4, size=len(sites), dtype=np.uint8)) & 3
    codes[sites] = b1 if is_ingroup else b2
    return codes


def codes_to_text(codes, records=16, n_frac=0.0, lower_frac=0.0, seed=0):
    """codes -> ASCII bases with '\\n' between `records` equal records (the layout
    kr_genome_upload takes)."""
    n = len(codes)
    rl = (n + records - 1) // records
    nparts = (n + rl - 1) // rl if n else 1
    out = np.empty(n + nparts - 1, dtype=np.uint8)
    p = 0
    for i in range(nparts):                    # one pass: the record's letters straight into place
        part = codes[i * rl:(i + 1) * rl]
        dst = out[p:p + len(part)]
        # A C G T = 65 67 71 84 from the codes 0 1 2 3 by byte arithmetic (a table look-up per
        # element is five times slower at 3 Gbp): 65 + 2 c, + 2 for c >= 2, + 11 for c = 3
        np.left_shift(part, 1, out=dst)
        dst += 65
        hi = part >> 1
        dst += hi << 1
        hi &= part
        dst += hi * np.uint8(11)
        p += len(part)
        if i + 1 < nparts:
            out[p] = 10
            p += 1
    if n_frac > 0 or lower_frac > 0:
        # N runs and soft-masked stretches at positions of the UNBROKEN text (as before: the same
        # seeded positions), mapped past the record separators
        rng = np.random.Generator(np.random.PCG64(4000 + seed))
        nrun = int(n * n_frac / 1000)
        for s0 in rng.integers(0, max(1, n - 1000), size=nrun):
            for a, b in _spans(int(s0), int(s0) + 1000, rl):
                out[a:b] = ord("N")
        nlow = int(n * lower_frac / 200)
        for s0 in rng.integers(0, max(1, n - 200), size=nlow):
            for a, b in _spans(int(s0), int(s0) + 200, rl):
                seg = out[a:b]
                seg[seg != ord("N")] |= 0x20
                seg[seg == ord("N")] = ord("n")
    return out


def _spans(a, b, rl):
    """[a, b) of the unbroken text -> the pieces of the text with one separator after every rl letters"""
    out = []
    while a < b:
        rec = a // rl
        e = min(b, (rec + 1) * rl)
        out.append((a + rec, e + rec))
        a = e
    return out


def family(config, n_in, n_out, length, records=16, mu=0.01, snp_every=10000, independent=False,
           n_frac=0.0, lower_frac=0.0, first=0):
    """-> list of (name, is_ingroup, uint8 text) for genomes first..first+n_in+n_out-1,
    the first n_in of the whole family being the ingroup."""
    anc = None if independent else ancestor(config, length)
    out = []
    for g in range(first, first + n_in + n_out):
        ing = g < first + n_in
        codes = genome_codes(config, g, length, ing, mu, snp_every, independent, anc)
        out.append((("in" if ing else "out") + str(g), ing,
                    codes_to_text(codes, records, n_frac, lower_frac, seed=100 * config + g)))
    return out


def write_fasta(path, text, width=80):
    recs = bytes(text).split(b"\n")
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">rec%d\n" % i)
            for j in range(0, len(r), width):
                f.write(r[j:j + width] + b"\n")
