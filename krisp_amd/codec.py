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


class Fields(list):
    """the widths of a line's columns, in line order (empty ones included) + `.offsets`: where each column starts in the
    WINDOW.  kstream's split takes sizes off the front and off the end of the k-mer in turn (kstream.py:805-832) and writes
    the end parts in the order they were cut: with two or more of them the line's columns are not in window order."""

    def __init__(self, widths, offsets=None):
        super().__init__(widths)
        if offsets is None:
            offsets, at = [], 0
            for w in widths:
                offsets.append(at)
                at += w
        self.offsets = list(offsets)

    def in_window_order(self):
        live = [(o, w) for o, w in zip(self.offsets, self) if w > 0]
        return all(a[0] + a[1] == b[0] for a, b in zip(live, live[1:])) and (not live or live[0][0] == 0)


def field_offsets(fields):
    return fields.offsets if isinstance(fields, Fields) else Fields(fields).offsets


def keys_to_fields_bytes(keys, fields, rna=False):
    """keys whose bases are the WINDOW (engine geometry (k, 0, 0)) -> bytes of the output: the window's pieces `fields`
    (widths in line order, empty ones included; Fields.offsets: where each lies in the window) joined by ','."""
    k = sum(fields)
    m = keys_to_matrix(keys, k, 0, 0, rna)
    out = np.empty((len(m), k + len(fields)), dtype=np.uint8)
    dst = 0
    for w, src in zip(fields, field_offsets(fields)):
        out[:, dst:dst + w] = m[:, src:src + w]
        out[:, dst + w] = ord(",")
        dst += w + 1
    out[:, k + len(fields) - 1] = ord("\n")
    return out.tobytes()


def field_layout_ok(widths, order):
    """can a key hold the window's fields (widths, line order) in `order`?  kr_set_field_order moves each field with ONE
    shift -- at most one distinct left and one distinct right shift -- after an optional rotation of the window by a field
    boundary: every permutation of up to three fields qualifies (round 5; the rotation serves (last, middle, first) with
    outer fields of different widths).  Kept as the statement of that rule, checked against the library by the tests."""
    k = sum(widths)
    src0, at = [], 0
    for w in widths:
        src0.append(at)
        at += w
    dst, at = {}, 0
    for f in order:
        dst[f] = at
        at += widths[f]
    for r in [0] + src0[1:]:
        if r >= k > 0:
            continue
        src = [(s - r) % k if k else 0 for s in src0]
        if any(widths[f] and src[f] + widths[f] > k for f in range(len(widths))):
            continue
        left = {src[f] - dst[f] for f in range(len(widths)) if widths[f] and dst[f] < src[f]}
        right = {dst[f] - src[f] for f in range(len(widths)) if widths[f] and dst[f] > src[f]}
        if len(left) <= 1 and len(right) <= 1:
            return True
    return False


def merge_fields(fields, order):
    """the window's fields (widths, line order) in key order `order`, with neighbouring fields that stay neighbours and in
    order fused into blocks: (block widths in line order, their key order) -- what kr_set_field_order takes when the blocks
    are at most three --, or None.  Empty fields join the block in front of them."""
    off = field_offsets(fields)
    live = [f for f in order if fields[f] > 0]
    runs = []                                   # maximal runs of fields that follow each other in the WINDOW, inside `live`
    for f in live:
        if runs and off[runs[-1][-1]] + fields[runs[-1][-1]] == off[f]:
            runs[-1].append(f)
        else:
            runs.append([f])
    if len(runs) > 3:
        return None
    by_window = sorted(range(len(runs)), key=lambda i: off[runs[i][0]])   # blocks in window order (kr_set_field_order's widths)
    widths = [sum(fields[f] for f in runs[i]) for i in by_window]
    rank = {i: j for j, i in enumerate(by_window)}
    return widths, [rank[i] for i in range(len(runs))]


def key_pieces(fields, order):
    """the window's pieces in KEY order for kr_set_field_pieces: [(offset, width)] of the non-empty fields in `order`,
    neighbours in the window fused"""
    off = field_offsets(fields)
    out = []
    for f in order:
        if fields[f] <= 0:
            continue
        if out and out[-1][0] + out[-1][1] == off[f]:
            out[-1] = (out[-1][0], out[-1][1] + fields[f])
        else:
            out.append((off[f], fields[f]))
    return out


def keys_to_ordered_fields_bytes(keys, fields, order, rna=False):
    """keys that hold the window's fields in `order` (kr_set_field_order; the krisp_fasta layout is order 0 2 1) -> bytes
    of the sorted output: the fields in LINE order (widths `fields`, empty ones included) joined by ','."""
    k = sum(fields)
    m = keys_to_matrix(keys, k, 0, 0, rna)
    koff, at = {}, 0
    for f in order:
        koff[f] = at
        at += fields[f]
    out = np.empty((len(m), k + len(fields)), dtype=np.uint8)
    dst = 0
    for f, w in enumerate(fields):
        out[:, dst:dst + w] = m[:, koff[f]:koff[f] + w]
        out[:, dst + w] = ord(",")
        dst += w + 1
    out[:, k + len(fields) - 1] = ord("\n")
    return out.tobytes()


def order_string(s, fields, order):
    """the window string s with its fields (widths `fields`, line order) rearranged into `order`: what the sort compares"""
    cuts = [s[o:o + w] for o, w in zip(field_offsets(fields), fields)]
    return "".join(cuts[f] for f in order)


def pack_plain(strings):
    """ACGT strings (all of one length <= 32) -> uint64 keys, MSB aligned"""
    if not strings:
        return np.empty(0, dtype=np.uint64)
    k = len(strings[0])
    m = _CODE[np.frombuffer("".join(strings).encode(), dtype=np.uint8).reshape(len(strings), k)]
    keys = np.zeros(len(strings), dtype=np.uint64)
    for j in range(k):
        keys |= m[:, j].astype(np.uint64) << np.uint64(62 - 2 * j)
    return keys


def insertion_index_ordered(keys, t):
    """number of packed keys (pure ACGT strings in the key's field order) that sort before the string t, which holds at
    least one character outside ACGT, in C-locale byte order (upper case before lower case, IUPAC letters among ACGT)"""
    j = next(i for i, ch in enumerate(t) if ch not in "ACGT")
    prefix = 0
    for i in range(j):
        prefix |= "ACGT".index(t[i]) << (62 - 2 * i)
    c = sum(1 for b in "ACGT" if b < t[j])
    bound = prefix + (c << (62 - 2 * j))    # c == 4 carries into the prefix: everything under it is smaller
    if bound >= 1 << 64:
        return len(keys)
    return int(np.searchsorted(keys, np.uint64(bound), side="left"))


def merged_ordered_blocks(keys, specials, fields, order, rna=False, chunk=1 << 22):
    """Yield the sorted output as byte blocks: the packed keys decoded to lines with the k-mers the device alphabet cannot
    carry (`specials`: window strings, pre-split) spliced in where the sort puts them"""
    sp = sorted((order_string(s, fields, order), s) for s in specials)
    cuts = [(insertion_index_ordered(keys, t), s) for t, s in sp]
    pos = ci = 0
    n = len(keys)
    while pos < n or ci < len(cuts):
        end = min(n, pos + chunk)
        if ci < len(cuts):
            end = min(end, cuts[ci][0])
        if end > pos:
            yield keys_to_ordered_fields_bytes(keys[pos:end], fields, order, rna)
            pos = end
        while ci < len(cuts) and cuts[ci][0] <= pos:
            s = cuts[ci][1]
            parts = [s[o:o + w] for o, w in zip(field_offsets(fields), fields)]
            line = ",".join(parts) + "\n"
            if rna:
                line = line.replace("T", "U").replace("t", "u")
            yield line.encode("latin-1")
            ci += 1


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


# ----------------------------------------------------------------------------
# k-mers holding IUPAC ambiguity letters (kept by the reference, kstream.py:11-18) live on
# the host as strings; these helpers place them among the packed keys in file order
# ----------------------------------------------------------------------------
def split_window(w, L, D, R):
    """window string (left|diag|right as read from the genome) -> (left, diag, right)."""
    return w[:L], w[L:L + D], w[L + D:L + D + R]


def insertion_index(keys, left, diag, right):
    """Number of packed keys that sort before the k-mer (left, diag, right) in the reference's
    file order (left, right, diag; C-locale byte order, where IUPAC letters interleave with
    ACGT: A < B < C < D < G < H < K < M < R < S < T < V < W < Y)."""
    s = left + right + diag
    j = next(i for i, ch in enumerate(s) if ch not in "ACGT")
    prefix = 0
    for i in range(j):
        prefix |= "ACGT".index(s[i]) << (62 - 2 * i)
    c = sum(1 for b in "ACGT" if b < s[j].upper().replace("U", "T"))
    if s[j] in "Uu":                      # RNA text never reaches here: keys are DNA until output
        c = 3
    bound = prefix + (c << (62 - 2 * j))   # c == 4 carries into the prefix: everything under it is smaller
    if bound >= 1 << 64:
        return len(keys)
    return int(np.searchsorted(keys, np.uint64(bound), side="left"))


def merged_line_blocks(keys, specials, L, D, R, rna=False, chunk=1 << 22):
    """Yield the sorted k-mer file as byte blocks: packed keys decoded to text with the
    IUPAC k-mers (list of (left, diag, right) strings) spliced in at their places."""
    sp = sorted(specials, key=lambda t: (t[0], t[2], t[1]))
    cuts = [(insertion_index(keys, *t), t) for t in sp]
    pos = 0
    ci = 0
    n = len(keys)
    while pos < n or ci < len(cuts):
        end = min(n, pos + chunk)
        if ci < len(cuts):
            end = min(end, cuts[ci][0])
        if end > pos:
            yield keys_to_lines_bytes(keys[pos:end], L, D, R, rna)
            pos = end
        while ci < len(cuts) and cuts[ci][0] <= pos:
            l, d, r = cuts[ci][1]
            line = f"{l},{d},{r}\n"
            if rna:
                line = line.replace("T", "U").replace("t", "u")
            yield line.encode()
            ci += 1
