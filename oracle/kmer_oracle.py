"""ctypes face of oracle/kmer_oracle.c -- test infrastructure, NOT product code.
See the header of kmer_oracle.c for scope, pinning and the reference map."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libkmer_oracle.so")

CAND = np.dtype([("prefix", "<u8"), ("in_mask", "<u8"), ("out_mask", "<u8")])
RECORD = np.dtype([("key", "<u8"), ("genome", "<u4"), ("count", "<u4")])
ERRORS = {-2: "IUPAC", -3: "ILLEGAL", -4: "CAP", -5: "PARAM"}


def build():
    src = os.path.join(_DIR, "kmer_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.kro_sorted_keys.restype = ctypes.c_int64
        _lib.kro_intersect.restype = ctypes.c_int64
        _lib.kro_collect.restype = ctypes.c_int64
    return _lib


class OracleError(Exception):
    pass


def join_records(records):
    """records (str or bytes) -> one ASCII buffer, '\\n' between records."""
    recs = [r.encode() if isinstance(r, str) else r for r in records]
    return b"\n".join(recs)


def sorted_keys(bases, L, D, R, omit=False):
    buf = np.frombuffer(bases, dtype=np.uint8)
    cap = 2 * max(len(buf), 1)
    out = np.empty(cap, dtype=np.uint64)
    n = lib().kro_sorted_keys(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(buf)),
                              L, D, R, 1 if omit else 0,
                              out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
    if n < 0:
        raise OracleError(ERRORS.get(n, str(n)))
    return out[:n].copy()


def sorted_keys_slice(bases, L, D, R, topbits, topval, omit=False, cap=None):
    """the sorted keys whose top `topbits` bits equal topval (kro_sorted_keys_slice): one key-space slice of a genome"""
    buf = np.frombuffer(bases, dtype=np.uint8)
    if cap is None:
        cap = max(1024, int(2 * len(buf) / (1 << topbits) * 1.5) + 65536)
    out = np.empty(cap, dtype=np.uint64)
    fn = lib().kro_sorted_keys_slice
    fn.restype = ctypes.c_int64
    n = fn(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(buf)), L, D, R, 1 if omit else 0,
           ctypes.c_int(topbits), ctypes.c_uint64(topval), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
    if n < 0:
        raise OracleError(ERRORS.get(n, str(n)))
    return out[:n].copy()


def _ptrs(key_arrays):
    n = len(key_arrays)
    arr = (ctypes.c_void_p * n)(*[k.ctypes.data for k in key_arrays])
    cnt = (ctypes.c_int64 * n)(*[len(k) for k in key_arrays])
    return arr, cnt


def intersect(key_arrays, is_ingroup, L, D, R, apply_filter=True):
    n = len(key_arrays)
    arr, cnt = _ptrs(key_arrays)
    cap = max(1, min(len(k) for k in key_arrays))
    out = np.empty(cap, dtype=CAND)
    flags = (ctypes.c_uint8 * n)(*[1 if f else 0 for f in is_ingroup])
    m = lib().kro_intersect(arr, cnt, n, flags, L, D, R, 1 if apply_filter else 0,
                            out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
    if m < 0:
        raise OracleError(ERRORS.get(m, str(m)))
    return out[:m].copy()


def collect(key_arrays, cands, L, D, R):
    n = len(key_arrays)
    arr, cnt = _ptrs(key_arrays)
    cands = np.ascontiguousarray(cands, dtype=CAND)
    cap = max(1, sum(len(k) for k in key_arrays))
    out = np.empty(cap, dtype=RECORD)
    m = lib().kro_collect(arr, cnt, n, cands.ctypes.data_as(ctypes.c_void_p),
                          ctypes.c_int64(len(cands)), L, D, R,
                          out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
    if m < 0:
        raise OracleError(ERRORS.get(m, str(m)))
    return out[:m].copy()


def key_to_columns(key, L, D, R):
    """MSB-aligned key -> (left, diag, right) strings."""
    s = "".join("ACGT"[(int(key) >> (62 - 2 * j)) & 3] for j in range(L + D + R))
    return s[:L], s[L + R:], s[L:L + R]


def keys_to_lines(keys, L, D, R):
    """sorted keys -> the reference's sorted k-mer file lines 'left,diag,right'."""
    out = []
    for key in keys:
        l, d, r = key_to_columns(key, L, D, R)
        out.append(f"{l},{d},{r}")
    return out


def records_to_lines(records, labels, L, D, R):
    """(key, genome, count) records -> canonicalised merged-file lines
    'left,diag,right,label[;label(count)]' (one line per distinct sequence)."""
    by_key = {}
    for rec in records:
        by_key.setdefault(int(rec["key"]), {})
        lab = labels[int(rec["genome"])]
        by_key[int(rec["key"])][lab] = by_key[int(rec["key"])].get(lab, 0) + int(rec["count"])
    lines = []
    for key, labs in by_key.items():
        l, d, r = key_to_columns(key, L, D, R)
        ls = ";".join(n if c == 1 else f"{n}({c})" for n, c in sorted(labs.items()))
        lines.append(f"{l},{d},{r},{ls}")
    return sorted(lines)
