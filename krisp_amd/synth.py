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
    text = _ACGT[codes]
    if n_frac > 0 or lower_frac > 0:
        rng = np.random.Generator(np.random.PCG64(4000 + seed))
        text = text.copy()
        nrun = int(len(text) * n_frac / 1000)
        for s in rng.integers(0, max(1, len(text) - 1000), size=nrun):
            text[s:s + 1000] = ord("N")
        nlow = int(len(text) * lower_frac / 200)
        for s in rng.integers(0, max(1, len(text) - 200), size=nlow):
            text[s:s + 200] |= 0x20
    n = len(text)
    rl = (n + records - 1) // records
    parts = [text[i:i + rl] for i in range(0, n, rl)]
    out = np.empty(n + len(parts) - 1, dtype=np.uint8)
    p = 0
    for i, part in enumerate(parts):
        out[p:p + len(part)] = part
        p += len(part)
        if i + 1 < len(parts):
            out[p] = 10
            p += 1
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
