"""Packed key <-> text conversions (host side of the wire formats).

Key format: include/krisp_hip.h -- base j of left|right|diag in bits 63-2j..62-2j.
Text formats: sorted k-mer file lines "left,diag,right"
(kstream/kstream.py:832, 297) and merged-file lines
"left,diag,right,label[;label(count)]" (krisp_fasta/Amplicon.py:330-348).
"""
import numpy as np

_LUT_DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
_LUT_RNA = np.frombuffer(b"ACGU", dtype=np.uint8)
_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
_CODE[ord("U")] = 3


def effective_geometry(L, D, R):
    """kstream.py:824-830: split=[L,-R] with R == 0 takes the `size >= 0` branch
    and yields 'left,,rest' -- the diagnostic bases land in the THIRD column and the
    second is empty.  Downstream that is exactly geometry (L, 0, D)."""
    if R == 0 and D > 0:
        return L, 0, D
    return L, D, R


def keys_to_matrix(keys, L, D, R, rna=False):
    """keys (uint64, MSB aligned) -> uint8 matrix [n, k] of the string left|right|diag."""
    k = L + D + R
    lut = _LUT_RNA if rna else _LUT_DNA
    keys = np.asarray(keys, dtype=np.uint64)
    m = np.empty((len(keys), k), dtype=np.uint8)
    for j in range(k):
        m[:, j] = lut[((keys >> np.uint64(62 - 2 * j)) & np.uint64(3)).astype(np.intp)]
    return m


def keys_to_lines_bytes(keys, L, D, R, rna=False):
    """-> bytes of the reference's sorted k-mer file: 'left,diag,right\\n' per key."""
    k = L + D + R
    m = keys_to_matrix(keys, L, D, R, rna)
    out = np.empty((len(m), k + 3), dtype=np.uint8)
    out[:, :L] = m[:, :L]
    out[:, L] = ord(",")
    out[:, L + 1:L + 1 + D] = m[:, L + R:]
    out[:, L + 1 + D] = ord(",")
    out[:, L + 2 + D:L + 2 + D + R] = m[:, L:L + R]
    out[:, k + 2] = ord("\n")
    return out.tobytes()


def key_columns(key, L, D, R, rna=False):
    """one key -> (left, diag, right) str."""
    lut = "ACGU" if rna else "ACGT"
    s = "".join(lut[(int(key) >> (62 - 2 * j)) & 3] for j in range(L + D + R))
    return s[:L], s[L + R:], s[L:L + R]


def lines_to_keys(lines, L, D, R):
    """'left,diag,right' lines (bytes, one per entry, no newline) -> uint64 keys.
    Raises ValueError for letters outside ACGT(U)."""
    k = L + D + R
    n = len(lines)
    if n == 0:
        return np.empty(0, dtype=np.uint64)
    flat = np.frombuffer(b"".join(lines), dtype=np.uint8)
    if len(flat) != n * (k + 2):
        raise ValueError("k-mer lines do not have the expected left,diag,right widths")
    m = flat.reshape(n, k + 2)
    cols = list(range(0, L)) + list(range(L + 2 + D, L + 2 + D + R)) + list(range(L + 1, L + 1 + D))
    codes = _CODE[m[:, cols]]
    if codes.size and codes.max() > 3:
        raise ValueError("k-mer lines hold letters outside ACGT")
    keys = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        keys |= codes[:, j].astype(np.uint64) << np.uint64(62 - 2 * j)
    return keys


def prefix_mask(L, R):
    bits = 2 * (L + R)
    if bits == 0:
        return np.uint64(0)
    return np.uint64((~0 << (64 - bits)) & 0xFFFFFFFFFFFFFFFF)
