"""A full-size oracle for ONE slice of the long-amplicon search (test infrastructure; VERDICT r5 item 4, last sentence).

The text oracle (oracle/krisp_oracle.py: per-genome sorted k-mer lines -> merge tree -> filter) is pure Python: a few Mbp.
BASELINE configs[2] is 8 x 500 Mbp.  What the reference computes splits by the first letters of the LEFT flank: a group is
one (left, right) pair, the merge tree keeps the pairs every genome holds (intersectAmplicons.py:232-310, shared.py:321-347),
the filter looks at one group at a time (filterAlignments.py:4-40) -- so the lines of the final file whose left flank starts
with `prefix` are the result of the same pipeline run on the windows whose left flank starts with `prefix` alone.

numpy does the bulk here -- select those windows in every genome (both strands: kstream.py:617-642, 679-694), pack their
flanks, keep the flank pairs that occur in every genome -- and hands the surviving windows, as the reference's
`left,diag,right` lines in its sort order, to the text oracle's merge tree and filter.  Inputs: texts as kr_genome_upload
takes them (records separated by newline), upper-case A C G T only (the synthetic families of the full-size tests).

tests/test_oracle.py checks this helper against the text oracle run on whole small genomes.
"""
import numpy as np

from oracle import krisp_oracle as O

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate(b"ACGT"):
    _CODE[_ch] = _i
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


def _pack(rows):
    """[n, w <= 32] letters -> uint64, first letter in the top bits: the numbers order as the texts do"""
    w = rows.shape[1]
    out = np.zeros(rows.shape[0], dtype=np.uint64)
    for j in range(w):
        out |= _CODE[rows[:, j]].astype(np.uint64) << np.uint64(2 * (31 - j))
    return out


def _find(text, pat):
    """positions i with text[i:i+len(pat)] == pat (len(pat) >= 4: the first four letters as one 32-bit compare per
    position, over the four byte phases of the text)"""
    n, m = len(text), len(pat)
    assert m >= 4
    w = np.uint32(int.from_bytes(bytes(pat[:4]), "little"))
    hits = []
    for o in range(4):
        cnt = (n - o) // 4
        if cnt > 0:
            hits.append(np.flatnonzero(text[o:o + 4 * cnt].view("<u4") == w).astype(np.int64) * 4 + o)
    p = np.sort(np.concatenate(hits)) if hits else np.empty(0, dtype=np.int64)
    for j in range(4, m):
        p = p[p + j < n]
        p = p[text[p + j] == pat[j]]
    return p


def _starts(text, k, prefix, strand):
    """window starts p (forward coordinates) whose window -- text[p:p+k], or its reverse complement for strand 1 -- begins
    with `prefix` and lies inside one record"""
    n, m = len(text), len(prefix)
    if n < k:
        return np.empty(0, dtype=np.int64)
    if strand == 0:
        p = _find(text, bytes(prefix))                                   # window[j] = text[p + j]
    else:
        rc = bytes(_COMP[np.frombuffer(bytes(prefix), dtype=np.uint8)][::-1])
        p = _find(text, rc) + m - k                                      # window[j] = comp(text[p + k - 1 - j])
    p = p[(p >= 0) & (p <= n - k)]
    bad = np.flatnonzero(text == 10)
    if len(bad):
        p = p[np.searchsorted(bad, p) == np.searchsorted(bad, p + k)]
    return p


def _windows(text, p, k, strand):
    rows = text[p[:, None] + np.arange(k)[None, :]]
    if strand:
        rows = _COMP[rows[:, ::-1]]
    return rows


def _flanks(text, p, k, L, R, strand):
    """the packed left and right flank of every window (first letter in the top bits), a gather per column"""
    lo = np.zeros(len(p), dtype=np.uint64)
    hi = np.zeros(len(p), dtype=np.uint64)
    comp = np.array([3, 2, 1, 0, 255], dtype=np.uint8)
    code = _CODE.copy()
    code[code == 255] = 4
    for j in range(L):
        c = code[text[p + j]] if strand == 0 else comp[code[text[p + k - 1 - j]]]
        lo |= c.astype(np.uint64) << np.uint64(2 * (31 - j))
    for j in range(R):
        c = code[text[p + k - R + j]] if strand == 0 else comp[code[text[p + R - 1 - j]]]
        hi |= c.astype(np.uint64) << np.uint64(2 * (31 - j))
    return lo, hi


def slice_lines(texts, labels, ingroup, L, D, R, prefix, do_filter=True):
    """-> the lines of the reference's final (filtered) merged file whose left flank starts with `prefix` (bytes, <= L
    letters), for flanks of at most 32 letters each"""
    assert 0 < L <= 32 and 0 < R <= 32 and 4 <= len(prefix) <= L
    k = L + D + R
    per = []
    for t in texts:
        t = np.frombuffer(bytes(t), dtype=np.uint8) if not isinstance(t, np.ndarray) else t
        t = np.ascontiguousarray(t)
        assert not np.isin(t, np.frombuffer(b"ACGT\n", dtype=np.uint8), invert=True).any(), "upper-case A C G T and newline only"
        ps, ls, rs, ss = [], [], [], []
        for strand in (0, 1):
            p = _starts(t, k, prefix, strand)
            lo, hi = _flanks(t, p, k, L, R, strand)
            ps.append(p), ls.append(lo), rs.append(hi), ss.append(np.full(len(p), strand, dtype=np.uint8))
        per.append((t, np.concatenate(ps), np.concatenate(ls), np.concatenate(rs), np.concatenate(ss)))
    # flank pairs every genome holds: first the left flanks (cheap), then the pairs among what is left
    common = None
    for _, _, lo, _, _ in per:
        u = np.unique(lo)
        common = u if common is None else np.intersect1d(common, u, assume_unique=True)
    # (a flank pair as ONE integer: rank of the left flank among the common ones x their number + rank of the right flank
    # among all right flanks seen -- 64-bit compares instead of compares of pairs)
    allr = np.unique(np.concatenate([hi[np.isin(lo, common)] for _, _, lo, hi, _ in per]))
    kept = []
    pairs = None
    for t, p, lo, hi, s in per:
        sel = np.isin(lo, common)
        p, lo, hi, s = p[sel], lo[sel], hi[sel], s[sel]
        code = np.searchsorted(common, lo).astype(np.int64) * np.int64(len(allr)) + np.searchsorted(allr, hi).astype(np.int64)
        kept.append((t, p, code, s))
        u = np.unique(code)
        pairs = u if pairs is None else np.intersect1d(pairs, u, assume_unique=True)
    files = []
    for (t, p, code, s), lab in zip(kept, labels):
        lines = []
        sel = np.isin(code, pairs)
        for strand in (0, 1):
            q = p[sel & (s == strand)]
            for row in _windows(t, q, k, strand):
                w = row.tobytes().decode()
                lines.append(f"{w[:L]},{w[L:L + D]},{w[L + D:]}")
        files.append((f"{lab}.{k}mers", O.gnu_sort(lines, [0, 2])))       # (krisp_fasta.py:16-43: sorted on the flanks)
    merged = O.merge_tree(files) if len(files) > 1 else [f"{ln},{labels[0]}" for ln in files[0][1]]
    if do_filter and D > 0:
        return O.filter_lines(merged, list(ingroup))
    return merged
