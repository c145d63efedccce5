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
import time

import numpy as np

# kstream.py:11-18 COMP_MAP keys
COMP_KEYS = frozenset("ATatGCgcRYryMKmkSWswBVbvDHdhNn")
_IUPAC = frozenset("RYMKSWBVDHrymkswbvdh")

_PLAIN = np.zeros(256, dtype=bool)
for _ch in b"ACGTNacgtn\n":
    _PLAIN[_ch] = True


# texts the device's reader (kr_genome_upload_text) takes; KRISP_DEVICE_TEXT_MAX: tests force the host-parse fallback
DEVICE_TEXT_MAX = int(os.environ.get("KRISP_DEVICE_TEXT_MAX", (1 << 32) - 64))

# per file: the stage times of the last kr_ingest_file (read / inflate / parse), for the stage tables
LAST_TIMINGS = {}


def _read_raw_lines(filename):
    ext = os.path.splitext(filename)[1]
    if ext in (".gz", ".bz2"):
        with (gzip.open if ext == ".gz" else bz2.open)(filename, "rb") as f:
            lines = f.read().split(b"\n")
        if lines and lines[-1] == b"":
            lines.pop()                     # split() artefact after the final newline, not a line
        return lines
    with open(filename, "rb") as f:
        return f.read().splitlines()       # text mode's universal newlines


def read_records(source):
    """-> list of bytes records, reference semantics (see module docstring)."""
    if isinstance(source, (str, os.PathLike)):
        lines = _read_raw_lines(os.fspath(source))
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


def load_bases(filename):
    """file -> (upload buffer, is_rna, n_special): the same result as
    to_bases(read_records(filename)) through the library's one-pass host parser."""
    from . import _native
    ext = os.path.splitext(filename)[1]
    # read + inflate + parse inside the library (kr_ingest_file), into pinned memory; None: a .bz2 file on a box
    # without libbz2
    got = _native.ingest_file(filename)
    if got is not None:
        bases, _nrec, nspecial, rna, _fasta, timings = got
        LAST_TIMINGS[os.fspath(filename)] = timings
        return bases, bool(rna), nspecial
    if ext == ".gz":
        with gzip.open(filename, "rb") as f:
            data, universal = f.read(), False
    elif ext == ".bz2":
        with bz2.open(filename, "rb") as f:
            data, universal = f.read(), False
    else:
        with open(filename, "rb") as f:
            data, universal = f.read(), True
    bases, _nrec, nspecial, rna, _fasta = _native.fasta_to_bases(data, universal, one_shot=True)
    return bases, bool(rna), nspecial


class BgzfFile:
    """a BGZF file as read_text hands it to ingest_on_device: the bytes as they lie on disk (`raw`) -- the device inflates
    them, a lane per member (kr_genome_upload_bgzf) --, len() = the bytes of text the members' trailers promise"""

    def __init__(self, filename, raw, text_bytes):
        self.filename, self.raw, self.text_bytes = filename, raw, text_bytes

    def __len__(self):
        return self.text_bytes

    def inflate(self):
        """the text through the host path (read_text with the device inflate off)"""
        return _read_text_host(self.filename)[0]


# KRISP_DEVICE_INFLATE=0: BGZF files through the host inflate as before round 6 (A/B, tests).  A lane decodes its member
# in ~40 ms whatever the file's size (profiles/r06/e2e_4x50Mbp.log: a 50 MB text is 800 lanes, a few of the GPU's 256 CUs --
# 37 ms against 22 on host threads), so the device takes the files whose members occupy it: texts of KRISP_DEVICE_INFLATE_MIN
# bytes and more (default 256 MB = 4 x 10^3 members: 37 ms + a third of the bytes to copy, against 31 ms of all 16 host
# threads and the whole text to copy -- and the other genomes' files inflate on those threads meanwhile; a 3 GB text: 0.1 s
# against 0.35 s)
def device_inflate_on():
    return os.environ.get("KRISP_DEVICE_INFLATE", "1") != "0"


def device_inflate_min():
    return int(os.environ.get("KRISP_DEVICE_INFLATE_MIN", 1 << 28))


def read_text(filename):
    """file -> (its text, universal_newlines): read + inflate only -- inside the library into pinned memory
    (kr_read_file) where it takes the file, else through Python's gzip / bz2.  The parse is left to the device
    (ingest_on_device).  Round 6: a `.gz` file that is BGZF all the way (bgzip: members that say how long they are) is only
    READ here -- BgzfFile --: the device inflates it (kr_genome_upload_bgzf)."""
    if os.path.splitext(filename)[1] == ".gz" and device_inflate_on():
        size = os.path.getsize(filename)
        if 28 <= size < (1 << 32) - 64:
            tb = _bgzf_text_bytes(filename, size)
            if tb is not None and device_inflate_min() <= tb < DEVICE_TEXT_MAX:
                t0 = time.time()
                raw = np.fromfile(filename, dtype=np.uint8)
                LAST_TIMINGS[os.fspath(filename)] = dict(read_s=time.time() - t0, inflate_s=0.0, parse_s=0.0, members=0,
                                                         libdeflate=False, device_inflate=True)
                return BgzfFile(os.fspath(filename), raw, tb), False
    return _read_text_host(filename)


def _read_text_host(filename):
    from . import _native
    ext = os.path.splitext(filename)[1]
    got = _native.read_file(filename)           # (None: a .bz2 file on a box without libbz2)
    if got is not None:
        text, universal, timings = got
        LAST_TIMINGS[os.fspath(filename)] = dict(timings, parse_s=0.0)
        return text, universal
    if ext == ".bz2":
        with bz2.open(filename, "rb") as f:
            return np.frombuffer(f.read(), dtype=np.uint8), False
    if ext == ".gz":
        with gzip.open(filename, "rb") as f:
            return np.frombuffer(f.read(), dtype=np.uint8), False
    with open(filename, "rb") as f:
        return np.frombuffer(f.read(), dtype=np.uint8), True


def _bgzf_text_bytes(path, size):
    total, at = 0, 0
    try:
        with open(path, "rb") as f:
            while at < size:
                f.seek(at)
                h = f.read(18)
                if len(h) < 18 or h[:4] != b"\x1f\x8b\x08\x04" or h[12:14] != b"BC":
                    return None
                bsize = int.from_bytes(h[16:18], "little") + 1
                if bsize < 26 or at + bsize > size:
                    return None
                f.seek(at + bsize - 4)
                total += int.from_bytes(f.read(4), "little")
                at += bsize
    except OSError:
        return None
    return total


def estimate_text_bytes(path):
    """an estimate from above of a sequence file's text in bytes without reading it (what kr_reserve plans with): the
    file's size; `.gz`: the ISIZE word of its last member plus as many 4 GiB as its compressed size asks for (a file of
    several members comes out too small: the caller then plans again with the real size; a BGZF file: the sum of its members'
    ISIZE words, exact);
    `.bz2`: five times its size"""
    path = os.fspath(path)
    size = os.path.getsize(path)
    if path.endswith(".gz"):
        if size < 18:
            return 0
        with open(path, "rb") as f:
            head = f.read(18)
            f.seek(size - 4)
            est = int.from_bytes(f.read(4), "little")
        if len(head) == 18 and head[3] & 4 and head[12:14] == b"BC":
            # BGZF: every member says how long it is (BSIZE) and ends with the length of its text (ISIZE): the sum over the
            # members is the text's exact length, for two small reads per 64 KiB member (the 5 x guess made
            # reservations 10-50 % too large).  A file that is not BGZF all the way falls back to the guess.
            exact = _bgzf_text_bytes(path, size)
            return exact if exact is not None else 5 * size
        while est < size:
            est += 1 << 32
        return est
    if path.endswith(".bz2"):
        return 5 * size
    return size


def ingest_on_device(eng, gid, text, universal, k, omit_soft):
    """text of a sequence file -> genome gid of `eng`, parsed on the device with the reference reader's semantics
    (kr_genome_upload_text; the host never sees the bases unless the genome holds characters outside ACGTNacgtn:
    then they come back for the side channel of ingest()).  Returns (bases on the device, is_rna, IUPAC k-mers)."""
    if isinstance(text, BgzfFile):
        got = eng.upload_bgzf(gid, text.raw, one_shot=True)
        if got is not None:
            n, _nrec, nspecial, rna, _fasta, members, us = got
            tm = LAST_TIMINGS.get(text.filename)
            if tm is not None:
                tm.update(members=members, device_inflate_s=us * 1e-6)
            special = scan_special(eng.fetch_bases(gid, n), k, omit_soft) if nspecial else []
            return n, bool(rna), special
        # (not BGZF all the way after all, or a member that does not inflate to its trailer: the host path, whose verdict
        # on the file -- Python's gzip errors included -- stands)
        text = text.inflate()
    if len(text) >= DEVICE_TEXT_MAX:
        # the device's reader indexes its text with 32 bits: a longer text (a genome of more than 4.29e9 bases: the packed
        # path takes them up to 2^33) is parsed by the library's host parser, same result (tests compare the two)
        from . import _native
        bases, _nrec, nspecial, rna, _fasta = _native.fasta_to_bases(text, universal, one_shot=True)
        eng.upload(gid, bases)
        special = scan_special(bases, k, omit_soft) if nspecial else []
        return len(bases), bool(rna), special
    n, _nrec, nspecial, rna, _fasta = eng.upload_text(gid, text, universal, one_shot=True)
    special = scan_special(eng.fetch_bases(gid, n), k, omit_soft) if nspecial else []
    return n, bool(rna), special


def ingest(source, k, omit_soft):
    """file name or iterable of sequence strings -> (upload buffer, is_rna, IUPAC k-mers).
    Files go through the library's one-pass parser; characters the device cannot carry are
    resolved here (scan_special) -- skipped entirely when the genome has none."""
    if isinstance(source, (str, os.PathLike)):
        bases, rna, nspecial = load_bases(os.fspath(source))
    else:
        records = read_records(source)
        rna = bool(detect_rna(records))
        bases = to_bases(records, rna)
        nspecial = 1
    special = scan_special(bases, k, omit_soft) if nspecial else []
    return bases, rna, special


def load_any(source):
    """file name or iterable of sequence strings -> (upload buffer, is_rna, number of characters
    outside ACGTNacgtn), nothing resolved"""
    if isinstance(source, (str, os.PathLike)):
        return load_bases(os.fspath(source))
    records = read_records(source)
    rna = bool(detect_rna(records))
    bases = to_bases(records, rna)
    return bases, rna, int(np.count_nonzero(~_PLAIN[bases]))


def detect_rna(records):
    for s in records:
        if b"T" in s or b"t" in s:
            return False
        if b"U" in s or b"u" in s:
            return True
    return None


def sniff_rna(filename, chunk=1 << 16):
    """detect_rna(read_records(filename)) without reading the file: the head of the (inflated) stream until the first record
    that holds T / t or U / u is complete or says 'T' (kstream.py:481-508 decides on that record: T anywhere in it -> DNA,
    else U -> RNA).  For the flows that must know every genome's alphabet before the first genome is on the device (the
    streaming flow: a mixed DNA / RNA run sets the filter's mode first).  Reference reader semantics as read_records:
    FASTA iff the first line holds '>', that line consumed."""
    ext = os.path.splitext(filename)[1]
    opener = gzip.open if ext == ".gz" else (bz2.open if ext == ".bz2" else open)
    universal = ext not in (".gz", ".bz2")
    with opener(filename, "rb") as f:
        tail, first, fasta, saw_u = b"", True, False, False
        while True:
            buf = f.read(chunk)
            data = tail + buf
            if universal:
                data = data.replace(b"\r\n", b"\n").replace(b"\r", b"\n") if not (buf and data.endswith(b"\r")) else data
            lines = data.split(b"\n")
            tail = lines.pop() if buf else b""
            if universal and buf and tail.endswith(b"\r"):
                pass                                    # (a lone CR at the chunk edge may be half of CRLF: stays in the tail)
            for ln in lines:
                if first:
                    fasta = b">" in ln
                    first = False
                    continue                            # (consumed by the detection, kstream.py:450)
                ln = ln.strip()
                header = fasta and ln.startswith(b">")
                if header or not fasta:
                    if saw_u:                           # the record that held U ended without a T
                        return True
                    if header:
                        continue
                if b"T" in ln or b"t" in ln:
                    return False
                if b"U" in ln or b"u" in ln:
                    if not fasta:
                        return True                     # (one line = one record)
                    saw_u = True
            if not buf:
                return True if saw_u else None


def to_bases(records, rna=False):
    """records -> uint8 array, '\\n' separated (U/u -> T/t for RNA, kstream.py:599)."""
    buf = b"\n".join(records)
    if rna:
        buf = buf.replace(b"U", b"T").replace(b"u", b"t")
    return np.frombuffer(buf, dtype=np.uint8)


# kstream.py:11-18 COMP_MAP as a translation table (IUPAC letters complement to IUPAC letters)
_COMP_TABLE = str.maketrans("ATatGCgcRYryMKmkSWswBVbvDHdhNn", "TAtaCGcgYRyrKMkmSWswVBvbHDhdNn")


def scan_special(bases, k, omit_soft):
    """Everything the 2-bit device alphabet cannot carry, resolved exactly on the host.

    The device handles A/C/G/T (either case) and drops windows holding N, lower case
    under --omit-soft, or anything else.  The reference differs for two rare kinds of
    character: (1) a character outside COMP_MAP makes _get_complement raise KeyError
    (kstream.py:658) for every window that survives the soft-mask step -- raised here,
    identically; (2) IUPAC ambiguity letters are KEPT (kstream.py:11-18): a surviving,
    N-free window holding one yields two k-mers (itself and its reverse complement with
    the letters complemented).  Those k-mers are returned (strings, after the soft-mask
    mapping) and join the device results in krisp_fasta / kstream.
    One vectorised pass when the genome is plain; Python only around special characters.
    """
    special = np.flatnonzero(~_PLAIN[bases])
    if len(special) == 0:
        return []
    # the library's scan (kr_scan_special: the C ABI's side channel) finds the windows; the host
    # only cuts the text.  Bytes >= 0x80 near a special character: Python's own case rules below.
    from . import _native
    starts_lib = _native.scan_special_starts(bases, k, omit_soft)
    if starts_lib is not None:
        text = bases.tobytes().decode("latin-1")
        out = []
        for s in starts_lib.tolist():
            w = text[s:s + k]
            if not omit_soft:
                w = w.upper()
            out.append(w)
            out.append(w[::-1].translate(_COMP_TABLE))
        return out
    text = bases.tobytes().decode("latin-1")
    seps = np.flatnonzero(bases == 10)
    starts = set()
    for p in special:
        p = int(p)
        i = int(np.searchsorted(seps, p))
        rec_lo = int(seps[i - 1]) + 1 if i > 0 else 0
        rec_hi = int(seps[i]) if i < len(seps) else len(text)
        for s in range(max(rec_lo, p - k + 1), min(p, rec_hi - k) + 1):
            starts.add(s)
    out = []
    for s in sorted(starts):                     # stream order, so the first KeyError is the reference's
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
        if any(ch in _IUPAC for ch in w):
            out.append(w)
            out.append(w[::-1].translate(_COMP_TABLE))
    return out


class IupacWindowsUnsupported(NotImplementedError):
    """Raised where k-mers holding IUPAC ambiguity letters cannot be carried (packed files)."""


def check_special(bases, k, omit_soft):
    """KeyError exactly as the reference raises it; returns the IUPAC k-mers (see scan_special)."""
    return scan_special(bases, k, omit_soft)
