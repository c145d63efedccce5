"""FASTA / sequence-list ingest for the device path.

Mirrors the reference's reader semantics exactly (kstream/kstream.py:430-583):
  * .gz / .bz2 by extension (fileinput.hook_compressed, kstream.py:472-473);
  * FASTA mode iff the FIRST line contains '>' anywhere (kstream.py:510-537);
  * that first line is consumed by the detection when the input is a file or a
    one-shot iterator (kstream.py:450 rebinds the re-chained stream to an unused
    name) -- harmless for a FASTA header, a lost sequence otherwise;
  * lines are strip()ped and concatenated between headers, empty records dropped
    (kstream.py:556-583); non-FASTA input: one sequence per stripped line;
  * RNA iff the first record holding T/t/U/u holds U/u first (kstream.py:481-508).
The output is what kr_genome_upload takes: ASCII bases, '\\n' between records.
"""
import bz2
import gzip
import os

import numpy as np

# kstream.py:11-18 COMP_MAP keys
COMP_KEYS = frozenset("ATatGCgcRYryMKmkSWswBVbvDHdhNn")
_IUPAC = frozenset("RYMKSWBVDHrymkswbvdh")

_PLAIN = np.zeros(256, dtype=bool)
for _ch in b"ACGTNacgtn\n":
    _PLAIN[_ch] = True


def _read_raw_lines(filename):
    ext = os.path.splitext(filename)[1]
    if ext == ".gz":
        with gzip.open(filename, "rb") as f:
            return f.read().split(b"\n")
    if ext == ".bz2":
        with bz2.open(filename, "rb") as f:
            return f.read().split(b"\n")
    with open(filename, "rb") as f:
        return f.read().splitlines()       # text mode's universal newlines


def read_records(source):
    """-> list of bytes records, reference semantics (see module docstring)."""
    if isinstance(source, (str, os.PathLike)):
        lines = _read_raw_lines(os.fspath(source))
        if lines and lines[-1] == b"":
            lines.pop()                     # split() artefact, not a line
        one_shot = True
    else:
        one_shot = hasattr(source, "__next__")
        lines = [ln.encode() if isinstance(ln, str) else bytes(ln) for ln in source]
    fasta = bool(lines) and b">" in lines[0]
    if one_shot:
        lines = lines[1:]
    if not fasta:
        return [ln.strip() for ln in lines]
    recs, cur = [], []
    for ln in lines:
        ln = ln.strip()
        if ln.startswith(b">"):
            if cur:
                recs.append(b"".join(cur))
            cur = []
        elif ln:
            cur.append(ln)
    if cur:
        recs.append(b"".join(cur))
    return recs


def detect_rna(records):
    for s in records:
        if b"T" in s or b"t" in s:
            return False
        if b"U" in s or b"u" in s:
            return True
    return None


def to_bases(records, rna=False):
    """records -> uint8 array, '\\n' separated (U/u -> T/t for RNA, kstream.py:599)."""
    buf = b"\n".join(records)
    if rna:
        buf = buf.replace(b"U", b"T").replace(b"u", b"t")
    return np.frombuffer(buf, dtype=np.uint8)


class IupacWindowsUnsupported(NotImplementedError):
    """A surviving window holds an IUPAC ambiguity letter.  The reference keeps such
    k-mers (kstream.py:11-18); the 2-bit device path cannot represent them yet."""


def check_special(bases, k, omit_soft):
    """Resolve what the device cannot: characters outside ACGTN/acgtn.

    Raises KeyError(char) exactly when the reference's _get_complement would
    (kstream.py:658: a window that survives the soft-mask step holds a character
    outside COMP_MAP), IupacWindowsUnsupported when a surviving N-free window
    holds an IUPAC letter.  Cheap when the genome is plain (one vectorised pass).
    """
    special = np.flatnonzero(~_PLAIN[bases])
    if len(special) == 0:
        return
    text = bases.tobytes().decode("latin-1")
    seps = np.flatnonzero(bases == 10)
    iupac_hit = None
    done_until = -1
    for p in special:
        p = int(p)
        i = int(np.searchsorted(seps, p))
        rec_lo = int(seps[i - 1]) + 1 if i > 0 else 0
        rec_hi = int(seps[i]) if i < len(seps) else len(text)
        lo = max(rec_lo, p - k + 1, done_until + 1 - 0)
        hi = min(p, rec_hi - k)
        for s in range(max(rec_lo, p - k + 1), hi + 1):
            w = text[s:s + k]
            if omit_soft:
                if not w.isupper():
                    continue
            else:
                w = w.upper()
            for ch in reversed(w):
                if ch not in COMP_KEYS:
                    raise KeyError(ch)
            if "N" in w or "n" in w:
                continue
            if iupac_hit is None and any(ch in _IUPAC for ch in w):
                iupac_hit = (s, w)
    if iupac_hit is not None:
        raise IupacWindowsUnsupported(
            f"window {iupac_hit[1]!r} at offset {iupac_hit[0]} holds an IUPAC ambiguity letter; "
            "the reference keeps such k-mers, the 2-bit device path does not support them yet")
