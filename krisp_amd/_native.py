"""ctypes binding of libkrisp_hip.so (include/krisp_hip.h).

There is NO fallback: if the shared library is missing or a call fails this
module raises.  The host layer never computes k-mers, sorts or intersects on
the CPU.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# KRISP_HIP_LIB: another build of the same library (A/B variants of tools/ab.sh); the product file
# is never overwritten by a variant
LIB_PATH = os.environ.get("KRISP_HIP_LIB") or os.path.join(HERE, "libkrisp_hip.so")

CAND = np.dtype([("prefix", "<u8"), ("in_mask", "<u8"), ("out_mask", "<u8")])
RECORD = np.dtype([("key", "<u8"), ("genome", "<u4"), ("count", "<u4")])
WIDE_HIT = np.dtype([("cand", "<u4"), ("genome", "<u4"), ("pos", "<u4"), ("strand", "<u4")])
WIDE_DICT_LEFT, WIDE_DICT_RIGHT, WIDE_GROUPS, WIDE_HITS, WIDE_COUNTS, WIDE_SLOT_BITS, WIDE_NGROUPS, WIDE_BATCH_USED, WIDE_LOCATED, WIDE_KEYS_LISTED = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
WIDE_MAX_K = 1024
WIDE_MAX_FLANK = 256
COMM_ID_BYTES = 128

SOFT_MAP, SOFT_OMIT = 0, 1
OPT_SLICE_BASES, OPT_GENERIC_INTERSECT, OPT_ISECT_FORMAT, OPT_ABLATE, OPT_WIDE_SLOTS, OPT_WIDE_ORDERED, OPT_PLACE_TRIES = 1, 2, 3, 4, 5, 6, 7
OPT_ISECT_KERNEL = 8
OPT_LANES = 9
OPT_LAZY_ORDER = 10
ERR_KEY, ERR_HOST = -5, -6
STRANDS_BOTH, STRANDS_FORWARD, STRANDS_CANONICAL = 0, 1, 2
STAGES = ["pack", "hist8", "reduce8", "scatter1", "hist2", "scan2", "scatter2", "chunks", "localsort",
          "fallback", "intersect", "compact", "collect", "merge", "locate"]
# stage -> the kernel(s) it times (names as rocprofv3 prints them)
STAGE_KERNELS = {"pack": "k_pack", "hist8": "k_hist8 (wide path: k_hist8w, slices: k_hist8k)", "reduce8": "k_reduce8a+k_reduce8b",
                 "scatter1": "k_scatter1p (wide path: k_scatter1w, slices: k_scatter1kp)",
                 "hist2": "k_hist16 (or k_hist2)", "scan2": "k_hist16_off (or k_scan2)", "scatter2": "k_scatter2",
                 "chunks": "k_chunk_table", "localsort": "k_localsort2",
                 "fallback": "k_seg_merge", "intersect": "k_intersect3 (+ k_intersect for oversized items)",
                 "compact": "k_scan+k_publish_reset+k_gather_items",
                 "collect": "k_collect_count+k_tile_sums+k_scan+k_tile_apply+k_publish+k_collect_emit",
                 "merge": "k_cands_flag+k_scan+k_cands_compact", "locate": "k_wide_locate"}

# every symbol include/krisp_hip.h declares: (name, restype, argtypes)
_c = ctypes
_P = _c.c_void_p
SYMBOLS = [
    ("kr_create", _P, [_c.c_int, _c.c_size_t]),
    ("kr_destroy", None, [_P]),
    ("kr_last_error", _c.c_char_p, [_P]),
    ("kr_set_params", _c.c_int, [_P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_size_t]),
    ("kr_set_strands", _c.c_int, [_P, _c.c_int]),
    ("kr_genome_upload", _c.c_int, [_P, _c.c_int, _P, _c.c_size_t]),
    ("kr_genome_sort", _c.c_int, [_P, _c.c_int]),
    ("kr_genome_add", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t]),
    ("kr_genome_count", _c.c_int64, [_P, _c.c_int]),
    ("kr_genome_load_sorted", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t]),
    ("kr_genome_fetch_keys", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t]),
    ("kr_genome_keys_in_order", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t]),
    ("kr_set_allow", _c.c_int, [_P, _c.c_uint]),
    ("kr_set_field_order", _c.c_int, [_P, _P, _P]),
    ("kr_set_field_pieces", _c.c_int, [_P, _c.c_int, _P, _P]),
    ("kr_genome_free", _c.c_int, [_P, _c.c_int]),
    ("kr_intersect", _c.c_int64, [_P, _P, _c.c_int, _P, _c.c_int]),
    ("kr_cands_count", _c.c_int64, [_P]),
    ("kr_cands_fetch", _c.c_int64, [_P, _P, _c.c_size_t]),
    ("kr_cands_load", _c.c_int64, [_P, _P, _c.c_size_t]),
    ("kr_cands_merge", _c.c_int64, [_P, _P, _c.c_size_t, _c.c_int, _c.c_int]),
    ("kr_cands_probe", _c.c_int64, [_P, _P, _c.c_int, _P, _c.c_int]),
    ("kr_comm_unique_id", _c.c_int, [_P]),
    ("kr_comm_init", _c.c_int, [_P, _c.c_int, _c.c_int, _P]),
    ("kr_comm_init_dir", _c.c_int, [_P, _c.c_int, _c.c_int, _c.c_char_p]),
    ("kr_comm_destroy", _c.c_int, [_P]),
    ("kr_comm_rank", _c.c_int, [_P]),
    ("kr_comm_world", _c.c_int, [_P]),
    ("kr_comm_rccl_ranks", _c.c_int, [_P]),
    ("kr_comm_barrier", _c.c_int, [_P]),
    ("kr_comm_allreduce", _c.c_int, [_P, _P, _c.c_int, _c.c_int]),
    ("kr_comm_allgather", _c.c_int, [_P, _P, _c.c_size_t, _P]),
    ("kr_cands_reduce", _c.c_int64, [_P, _c.c_int]),
    ("kr_cands_bcast", _c.c_int64, [_P]),
    ("kr_records_gather", _c.c_int64, [_P]),
    ("kr_collect", _c.c_int64, [_P, _P, _c.c_int]),
    ("kr_fetch", _c.c_int64, [_P, _P, _c.c_size_t]),
    ("kr_set_params_wide", _c.c_int, [_P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_size_t]),
    ("kr_wide_run", _c.c_int64, [_P, _P, _c.c_int, _P, _c.c_int]),
    ("kr_wide_fetch", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t]),
    ("kr_wide_fetch_windows", _c.c_int64, [_P, _P, _c.c_size_t]),
    ("kr_render_windows", _c.c_int64, [_P, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _c.c_size_t, _P, _c.c_size_t, _P,
                                       _c.c_int, _c.c_int, _P, _P, _P, _P]),
    ("kr_fasta_to_bases", _c.c_int64, [_P, _c.c_size_t, _c.c_int, _c.c_int, _P, _c.c_size_t, _P]),
    ("kr_ingest_file", _c.c_int64, [_c.c_char_p, _P, _P]),
    ("kr_read_file", _c.c_int64, [_c.c_char_p, _P, _P]),
    ("kr_genome_upload_text", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t, _c.c_int, _c.c_int, _P]),
    ("kr_genome_upload_bgzf", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t, _c.c_int, _P]),
    ("kr_reserve", _c.c_int, [_P, _P, _c.c_int, _c.c_size_t, _c.c_int]),
    ("kr_genome_fetch_bases", _c.c_int64, [_P, _c.c_int, _P, _c.c_size_t]),
    ("kr_host_free", None, [_P]),
    ("kr_scan_special", _c.c_int64, [_P, _c.c_size_t, _c.c_int, _c.c_int, _P, _c.c_size_t, _P]),
    ("kr_set_option", _c.c_int, [_P, _c.c_int, _c.c_int64]),
    ("kr_sync", _c.c_int, [_P]),
    ("kr_timer_begin", _c.c_int, [_P]),
    ("kr_timer_end_ms", _c.c_double, [_P]),
    ("kr_stage_enable", _c.c_int, [_P, _c.c_int]),
    ("kr_stage_select", _c.c_int, [_P, _c.c_uint]),
    ("kr_stage_reset", _c.c_int, [_P]),
    ("kr_stage_ms", _c.c_double, [_P, _c.c_int]),
    ("kr_stage_launches", _c.c_int64, [_P, _c.c_int]),
    ("kr_debug_fetch", _c.c_int64, [_P, _c.c_int, _c.c_int, _P, _c.c_size_t]),
    ("kr_debug_inversions", _c.c_int64, [_P, _c.c_int]),
    ("kr_debug_localsort", _c.c_double, [_P, _c.c_int, _c.c_int, _c.c_int]),
    ("kr_debug_intersect", _c.c_double, [_P, _P, _c.c_int, _P, _c.c_int, _c.c_int]),
    ("kr_debug_copy_gbps", _c.c_double, [_P, _c.c_size_t, _c.c_int]),
    ("kr_debug_copy_which", _c.c_char_p, [_P]),
    ("kr_mem_info", _c.c_int, [_P, _P]),
    ("kr_debug_comm", _c.c_int, [_P, _P]),
    ("kr_debug_place", _c.c_int, [_P, _P]),
    ("kr_debug_comm_probe", _c.c_int, [_P, _c.c_size_t, _c.c_int, _P]),
    ("kr_debug_cands_selfexchange", _c.c_int64, [_P, _c.c_int]),
    ("kr_debug_info", _c.c_int, [_P, _P]),
    ("kr_render_records", _c.c_int64, [_P, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_size_t, _P, _c.c_size_t, _P,
                                       _c.c_int, _P, _P, _P, _P]),
    ("kr_text_free", None, [_P]),
    ("kr_debug_isect", _c.c_int, [_P, _P]),
    ("kr_debug_lazy", _c.c_int, [_P, _P]),
    ("kr_set_mixed_alphabets", _c.c_int, [_P, _c.c_int]),
    ("kr_build_experiments", _c.c_int, []),
    ("kr_comm_set_timeout", _c.c_int, [_P, _c.c_int]),
    ("kr_debug_comm_hang", _c.c_int, [_P]),
    ("kr_debug_budget_left", _c.c_int64, [_P]),
    ("kr_debug_budget_set", _c.c_int, [_P, _c.c_int64]),
]


class KrispHipError(RuntimeError):
    """a negative return code of the library with kr_last_error's text; `.code` = the KR_ERR_* value (-3 = capacity:
    a caller buffer, the context's HBM budget or the device's memory does not hold what was asked for)"""
    code = None


ERR_CAPACITY = -3


_lib = None


def load():
    """Load the HIP library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KrispHipError(
            f"{LIB_PATH} is missing: build it with `python -m krisp_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def fasta_to_bases(data, universal_newlines, one_shot=True):
    """bytes of a FASTA / sequence file -> (uint8 upload buffer, records, special chars, rna, fasta)
    through the library's one-pass host parser (GIL released: callers may use threads)."""
    lib = load()
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(len(src) + 1, dtype=np.uint8)
    stats = np.zeros(4, dtype=np.int64)
    n = lib.kr_fasta_to_bases(_ptr(src) if len(src) else None, len(src), 1 if universal_newlines else 0,
                              1 if one_shot else 0, _ptr(out), len(out), _ptr(stats))
    if n < 0:
        raise KrispHipError(f"kr_fasta_to_bases: [{n}]")
    return out[:n], int(stats[0]), int(stats[1]), stats[2] == 1, bool(stats[3])


class _HostBlock:
    """a buffer the library owns (kr_ingest_file: pinned host memory on a GPU box); freed with the last view"""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            if self.ptr:
                self.lib.kr_host_free(self.ptr)
                self.ptr = None
        except Exception:  # noqa: BLE001
            pass


def ingest_file(path):
    """file -> (uint8 upload buffer in library-owned, pinned memory; records; special chars; rna; fasta;
    timings dict) through kr_ingest_file: read + inflate + parse inside the library, GIL released.
    Returns None for files the library leaves to the host layer (.bz2)."""
    lib = load()
    out = ctypes.c_void_p(0)
    stats = np.zeros(8, dtype=np.int64)
    n = lib.kr_ingest_file(os.fsencode(path), ctypes.byref(out), _ptr(stats))
    if n == ERR_HOST:
        return None
    if n < 0:
        msg = lib.kr_last_error(None).decode()
        if "cannot open" in msg:
            raise FileNotFoundError(msg)
        raise KrispHipError(f"kr_ingest_file({path}): [{n}] {msg}")
    block = _HostBlock(lib, out.value)
    raw = (ctypes.c_uint8 * max(int(n), 1)).from_address(out.value)
    raw._owner = block                       # the view keeps the block alive
    arr = np.frombuffer(raw, dtype=np.uint8)[:n]
    timings = dict(read_s=stats[4] / 1e6, inflate_s=stats[5] / 1e6, parse_s=stats[6] / 1e6,
                   members=int(stats[7] & 0xFFFFFFFF), libdeflate=bool(stats[7] >> 32))
    return arr, int(stats[0]), int(stats[1]), stats[2] == 1, bool(stats[3]), timings


def read_file(path):
    """file -> (its text as a uint8 view of library-owned, pinned memory; universal_newlines; timings) through
    kr_read_file: read + inflate inside the library, GIL released; the parse is left to the device
    (Engine.upload_text).  None for files the library leaves to the host layer (.bz2)."""
    lib = load()
    out = ctypes.c_void_p(0)
    stats = np.zeros(8, dtype=np.int64)
    n = lib.kr_read_file(os.fsencode(path), ctypes.byref(out), _ptr(stats))
    if n == ERR_HOST:
        return None
    if n < 0:
        msg = lib.kr_last_error(None).decode()
        if "cannot open" in msg:
            raise FileNotFoundError(msg)
        raise KrispHipError(f"kr_read_file({path}): [{n}] {msg}")
    block = _HostBlock(lib, out.value)
    raw = (ctypes.c_uint8 * max(int(n), 1)).from_address(out.value)
    raw._owner = block
    arr = np.frombuffer(raw, dtype=np.uint8)[:n]
    timings = dict(read_s=stats[4] / 1e6, inflate_s=stats[5] / 1e6, copy_s=stats[6] / 1e6,
                   members=int(stats[7] & 0xFFFFFFFF), libdeflate=bool(stats[7] >> 32))
    return arr, bool(stats[3]), timings


def render_records(records, label_of, label_text, label_in, L, D, R, dot=False):
    """kr_render_records: (csv_text, alignment_text, number of groups), or None when the library leaves a group to
    the general path.  records: RECORD array ordered by (key, label id); label_of: genome id -> label id;
    label_text: the distinct labels in string order; label_in: per label 1 / 0, or None (no outgroup given)."""
    lib = load()
    recs = np.ascontiguousarray(records, dtype=RECORD)
    lof = np.ascontiguousarray(label_of, dtype=np.uint32)
    texts = [t.encode("utf-8", "surrogateescape") for t in label_text]     # (labels come from file names)
    arr = (_c.c_char_p * max(len(texts), 1))(*texts)
    lin = None if label_in is None else np.ascontiguousarray(label_in, dtype=np.uint8)
    csv, align = _c.c_void_p(), _c.c_void_p()
    ncsv, nalign = _c.c_size_t(), _c.c_size_t()
    rc = lib.kr_render_records(_ptr(recs), len(recs), L, D, R, _ptr(lof), len(lof), arr, len(texts),
                               None if lin is None else _ptr(lin), 1 if dot else 0, _c.byref(csv), _c.byref(ncsv),
                               _c.byref(align), _c.byref(nalign))
    if rc == ERR_HOST:
        return None
    if rc < 0:
        raise KrispHipError(f"kr_render_records: [{rc}]")
    try:
        return (_c.string_at(csv, ncsv.value).decode("utf-8", "surrogateescape"),
                _c.string_at(align, nalign.value).decode("utf-8", "surrogateescape"), int(rc))
    finally:
        lib.kr_text_free(csv)
        lib.kr_text_free(align)


def render_windows(rows, cand, genome, label_of, label_text, label_in, L, D, R, dot=False, rna=False):
    """kr_render_windows: (csv_text, alignment_text, number of groups) from the member windows of long amplicons (rows:
    uint8 [n, L+D+R] in line order, cand / genome per row as kr_wide_fetch(HITS) gives them), or None when the library
    leaves a group to the general path"""
    lib = load()
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    cand = np.ascontiguousarray(cand, dtype=np.uint32)
    gen = np.ascontiguousarray(genome, dtype=np.uint32)
    lof = np.ascontiguousarray(label_of, dtype=np.uint32)
    texts = [t.encode("utf-8", "surrogateescape") for t in label_text]
    arr = (_c.c_char_p * max(len(texts), 1))(*texts)
    lin = None if label_in is None else np.ascontiguousarray(label_in, dtype=np.uint8)
    csv, align = _c.c_void_p(), _c.c_void_p()
    ncsv, nalign = _c.c_size_t(), _c.c_size_t()
    rc = lib.kr_render_windows(_ptr(rows) if len(rows) else None, len(rows), L, D, R, _ptr(cand) if len(rows) else None,
                               _ptr(gen) if len(rows) else None, _ptr(lof), len(lof), arr, len(texts),
                               None if lin is None else _ptr(lin), 1 if dot else 0, 1 if rna else 0, _c.byref(csv),
                               _c.byref(ncsv), _c.byref(align), _c.byref(nalign))
    if rc == ERR_HOST:
        return None
    if rc < 0:
        raise KrispHipError(f"kr_render_windows: [{rc}]")
    try:
        return (_c.string_at(csv, ncsv.value).decode("utf-8", "surrogateescape"),
                _c.string_at(align, nalign.value).decode("utf-8", "surrogateescape"), int(rc))
    finally:
        lib.kr_text_free(csv)
        lib.kr_text_free(align)


def _quiet_rccl():
    """RCCL prints its version banner (NCCL_DEBUG=VERSION, what some boxes export) and its warnings on STDOUT -- where the
    command line writes its CSV and bench.py its one JSON line: errors only, unless the caller asked for more than the banner"""
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "ERROR"


def comm_unique_id():
    """the RCCL unique id (rank 0 makes it, every rank passes it to Engine.comm_init)"""
    _quiet_rccl()
    lib = load()
    buf = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
    rc = lib.kr_comm_unique_id(_ptr(buf))
    if rc < 0:
        raise KrispHipError(f"kr_comm_unique_id: [{rc}] " + lib.kr_last_error(None).decode())
    return buf.tobytes()


def scan_special_starts(bases, k, omit_soft):
    """Window starts (ascending) of the surviving, N-free windows that hold an IUPAC ambiguity letter
    (kr_scan_special).  Raises KeyError(char) exactly where the reference does; returns None when the
    library leaves the case to the host (bytes >= 0x80 near a special character)."""
    lib = load()
    buf = np.ascontiguousarray(bases, dtype=np.uint8)
    bad = ctypes.c_int(0)
    n = lib.kr_scan_special(_ptr(buf) if len(buf) else None, len(buf), k, 1 if omit_soft else 0, None, 0,
                            ctypes.byref(bad))
    if n == ERR_KEY:
        raise KeyError(chr(bad.value))
    if n == ERR_HOST:
        return None
    if n < 0:
        raise KrispHipError(f"kr_scan_special: [{n}]")
    out = np.empty(max(n, 1), dtype=np.uint64)
    lib.kr_scan_special(_ptr(buf) if len(buf) else None, len(buf), k, 1 if omit_soft else 0, _ptr(out), n,
                        ctypes.byref(bad))
    return out[:n]


class Engine:
    """One GPU context (one HIP stream) -- thin object face of the C ABI."""

    def __init__(self, device=0, hbm_budget=0):
        self.lib = load()
        self.ctx = self.lib.kr_create(device, hbm_budget)
        if not self.ctx:
            raise KrispHipError("kr_create failed: " + self.lib.kr_last_error(None).decode())
        self.params = None

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.kr_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            e = KrispHipError(f"{what}: [{rc}] " + self.lib.kr_last_error(self.ctx).decode())
            e.code = int(rc)
            raise e
        return rc

    # ---- configuration
    def upload_text(self, gid, text, universal_newlines, one_shot=True):
        """file text -> genome gid, parsed on the device with the reference reader's semantics (kr_genome_upload_text).
        Returns (bases, records, special characters, rna, fasta)."""
        t = np.ascontiguousarray(text, dtype=np.uint8)
        stats = np.zeros(4, dtype=np.int64)
        n = self._check(self.lib.kr_genome_upload_text(self.ctx, gid, _ptr(t), len(t), 1 if universal_newlines else 0,
                                                       1 if one_shot else 0, _ptr(stats)), "kr_genome_upload_text")
        return n, int(stats[0]), int(stats[1]), stats[2] == 1, bool(stats[3])

    def upload_bgzf(self, gid, raw, one_shot=True):
        """the bytes of a BGZF file -> genome gid: inflated on the device, a lane per member, every member checked against
        its CRC-32 and length, then parsed there (kr_genome_upload_bgzf).  Returns (bases, records, special characters, rna,
        fasta, members, microseconds of the inflate kernels), or None when the file is not BGZF all the way or a member
        does not inflate to its trailer: nothing is uploaded then, the caller reads the file as any other .gz."""
        t = np.ascontiguousarray(raw, dtype=np.uint8)
        stats = np.zeros(8, dtype=np.int64)
        n = self.lib.kr_genome_upload_bgzf(self.ctx, gid, _ptr(t), len(t), 1 if one_shot else 0, _ptr(stats))
        if n == ERR_HOST:
            self.last_bgzf = (int(stats[4]), int(stats[5]), int(stats[6]), self.lib.kr_last_error(self.ctx).decode())
            return None
        n = self._check(n, "kr_genome_upload_bgzf")
        return n, int(stats[0]), int(stats[1]), stats[2] == 1, bool(stats[3]), int(stats[4]), int(stats[7])

    def reserve(self, ids, n_bases, with_text=False):
        """the large device buffers of genomes `ids` of up to n_bases bases, ahead of their uploads (kr_reserve)"""
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self._check(self.lib.kr_reserve(self.ctx, _ptr(a), len(a), int(n_bases), 1 if with_text else 0), "kr_reserve")

    def fetch_bases(self, gid, n):
        out = np.empty(max(n, 1), dtype=np.uint8)
        m = self._check(self.lib.kr_genome_fetch_bases(self.ctx, gid, _ptr(out), n), "kr_genome_fetch_bases")
        return out[:m]

    def set_option(self, option, value):
        """result-neutral options (OPT_SLICE_BASES, OPT_GENERIC_INTERSECT, OPT_ISECT_FORMAT, OPT_ISECT_KERNEL, OPT_WIDE_SLOTS, OPT_LANES); before set_params (OPT_LANES: any time)"""
        self._check(self.lib.kr_set_option(self.ctx, option, int(value)), "kr_set_option")

    def set_params(self, L, D, R, omit_soft=False, max_bases=0):
        self._check(self.lib.kr_set_params(self.ctx, L, D, R, SOFT_OMIT if omit_soft else SOFT_MAP,
                                           max_bases), "kr_set_params")
        self.params = (L, D, R)

    def set_mixed_alphabets(self, on=True):
        """DNA and RNA genomes in one run: the filter in its mode 2 (kr_set_mixed_alphabets)"""
        self._check(self.lib.kr_set_mixed_alphabets(self.ctx, 1 if on else 0), "kr_set_mixed_alphabets")

    def set_strands(self, mode):
        """STRANDS_BOTH (complements), STRANDS_FORWARD, STRANDS_CANONICAL (kstream canonicals)"""
        self._check(self.lib.kr_set_strands(self.ctx, mode), "kr_set_strands")

    # ---- genomes
    def upload(self, gid, bases):
        buf = np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases
        self._check(self.lib.kr_genome_upload(self.ctx, gid, _ptr(buf) if len(buf) else None, len(buf)),
                    "kr_genome_upload")

    def sort(self, gid):
        self._check(self.lib.kr_genome_sort(self.ctx, gid), "kr_genome_sort")

    def add(self, gid, bases):
        self.upload(gid, bases)
        self.sort(gid)
        return self.count(gid)

    def load_sorted(self, gid, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        return self._check(self.lib.kr_genome_load_sorted(self.ctx, gid, _ptr(keys) if len(keys) else None,
                                                          len(keys)), "kr_genome_load_sorted")

    def count(self, gid):
        return self._check(self.lib.kr_genome_count(self.ctx, gid), "kr_genome_count")

    def keys(self, gid):
        n = self.count(gid)
        out = np.empty(max(n, 1), dtype=np.uint64)
        self._check(self.lib.kr_genome_fetch_keys(self.ctx, gid, _ptr(out), n), "kr_genome_fetch_keys")
        return out[:n]

    def set_allow(self, bases):
        """kstream --allow on the device alphabet: an iterable of the bases (of ACGT) a k-mer may hold"""
        mask = sum(1 << "ACGT".index(b) for b in set(bases))
        self._check(self.lib.kr_set_allow(self.ctx, mask), "kr_set_allow")

    def set_field_order(self, widths, order):
        """kstream --sort-cols as a key layout: the window's fields (widths in line order) held in `order`; after
        set_params(k, 0, 0).  Raises for the one order a layout cannot express (see include/krisp_hip.h)."""
        w = np.asarray(list(widths) + [0] * (3 - len(widths)), dtype=np.int32)
        o = np.asarray(order, dtype=np.int32)
        self._check(self.lib.kr_set_field_order(self.ctx, _ptr(w), _ptr(o)), "kr_set_field_order")

    def set_field_pieces(self, pieces):
        """the key = the window's pieces [(offset, width)] one after the other (kr_set_field_pieces); after set_params(k, 0, 0)"""
        o = np.asarray([p[0] for p in pieces], dtype=np.int32)
        w = np.asarray([p[1] for p in pieces], dtype=np.int32)
        self._check(self.lib.kr_set_field_pieces(self.ctx, len(pieces), _ptr(o), _ptr(w)), "kr_set_field_pieces")

    def keys_in_order(self, gid, n_bases):
        """keys of an uploaded genome in stream order (no sort)"""
        out = np.empty(max(2 * n_bases, 1), dtype=np.uint64)
        n = self._check(self.lib.kr_genome_keys_in_order(self.ctx, gid, _ptr(out), len(out)), "kr_genome_keys_in_order")
        return out[:n]

    def free(self, gid):
        self._check(self.lib.kr_genome_free(self.ctx, gid), "kr_genome_free")

    # ---- intersection
    MAX_GENOMES_PER_CALL = 32      # MAXG of kr_intersect

    def intersect(self, gids, is_ingroup, apply_filter=True):
        ids = np.asarray(gids, dtype=np.int32)
        flags = np.asarray([1 if f else 0 for f in is_ingroup], dtype=np.uint8)
        # (more than 32 genomes: the library intersects them in batches and keeps the running list on the device)
        return self._check(self.lib.kr_intersect(self.ctx, _ptr(ids), len(ids), _ptr(flags),
                                                 1 if apply_filter else 0), "kr_intersect")

    def cands(self):
        n = self._check(self.lib.kr_cands_count(self.ctx), "kr_cands_count")
        out = np.empty(max(n, 1), dtype=CAND)
        self._check(self.lib.kr_cands_fetch(self.ctx, _ptr(out), n), "kr_cands_fetch")
        return out[:n]

    def load_cands(self, cands):
        cands = np.ascontiguousarray(cands, dtype=CAND)
        return self._check(self.lib.kr_cands_load(self.ctx, _ptr(cands) if len(cands) else None, len(cands)),
                           "kr_cands_load")

    def merge_cands(self, other=None, apply_filter=False):
        if other is None:
            return self._check(self.lib.kr_cands_merge(self.ctx, None, 0, 0, 1 if apply_filter else 0),
                               "kr_cands_merge")
        other = np.ascontiguousarray(other, dtype=CAND)
        return self._check(self.lib.kr_cands_merge(self.ctx, _ptr(other) if len(other) else None,
                                                   len(other), 1, 1 if apply_filter else 0),
                           "kr_cands_merge")

    def probe_cands(self, gids, is_ingroup, apply_filter=True):
        """candidates := those every listed genome holds, masks OR-ed by side, filter (kr_cands_probe)"""
        ids = np.asarray(gids, dtype=np.int32)
        flags = np.asarray([1 if f else 0 for f in is_ingroup], dtype=np.uint8)
        return self._check(self.lib.kr_cands_probe(self.ctx, _ptr(ids), len(ids), _ptr(flags), 1 if apply_filter else 0),
                           "kr_cands_probe")

    def collect(self, gids, fetch=True):
        ids = np.asarray(gids, dtype=np.int32)
        n = self._check(self.lib.kr_collect(self.ctx, _ptr(ids), len(ids)), "kr_collect")
        return self.fetch_records(n) if fetch else n

    def fetch_records(self, n):
        out = np.empty(max(n, 1), dtype=RECORD)
        self._check(self.lib.kr_fetch(self.ctx, _ptr(out), n), "kr_fetch")
        return out[:n]

    # ---- multi-GPU exchange (one process / context per GPU; krisp_amd/distributed.py does the rendezvous)
    def comm_init(self, rank, world, comm_id):
        """RCCL communicator from the unique id rank 0 made (comm_unique_id)"""
        _quiet_rccl()
        buf = np.frombuffer(comm_id, dtype=np.uint8)
        assert len(buf) == COMM_ID_BYTES
        self._check(self.lib.kr_comm_init(self.ctx, rank, world, _ptr(buf)), "kr_comm_init")

    def comm_init_dir(self, rank, world, directory):
        """rehearsal transport: the same messages through files (ranks may share a GPU)"""
        self._check(self.lib.kr_comm_init_dir(self.ctx, rank, world, os.fsencode(directory)), "kr_comm_init_dir")

    def comm_rccl_ranks(self):
        """ranks of the RCCL communicator as ncclCommCount reports them; 0 without one (file transport, no communicator)"""
        return int(self.lib.kr_comm_rccl_ranks(self.ctx))

    def comm_barrier(self):
        self._check(self.lib.kr_comm_barrier(self.ctx), "kr_comm_barrier")

    def comm_allreduce(self, values, op="sum"):
        v = np.asarray(values, dtype=np.float64).copy()
        self._check(self.lib.kr_comm_allreduce(self.ctx, _ptr(v), len(v), 1 if op == "max" else 0), "kr_comm_allreduce")
        return v

    def comm_allgather(self, data):
        """bytes of every rank (any length) -> list of bytes by rank"""
        world = self.lib.kr_comm_world(self.ctx)
        if world == 1:
            return [bytes(data)]
        n = int(self.comm_allreduce([float(len(data))], "max")[0]) + 8
        mine = np.zeros(n, dtype=np.uint8)
        mine[:8] = np.frombuffer(np.uint64(len(data)).tobytes(), dtype=np.uint8)
        mine[8:8 + len(data)] = np.frombuffer(bytes(data), dtype=np.uint8)
        out = np.empty(n * world, dtype=np.uint8)
        self._check(self.lib.kr_comm_allgather(self.ctx, _ptr(mine), n, _ptr(out)), "kr_comm_allgather")
        res = []
        for r in range(world):
            ln = int(np.frombuffer(out[r * n:r * n + 8].tobytes(), dtype=np.uint64)[0])
            res.append(out[r * n + 8:r * n + 8 + ln].tobytes())
        return res

    def cands_reduce(self, apply_filter):
        """tree reduction of every rank's candidates onto rank 0 (the count there, 0 elsewhere)"""
        return self._check(self.lib.kr_cands_reduce(self.ctx, 1 if apply_filter else 0), "kr_cands_reduce")

    def cands_bcast(self):
        return self._check(self.lib.kr_cands_bcast(self.ctx), "kr_cands_bcast")

    def records_gather(self):
        return self._check(self.lib.kr_records_gather(self.ctx), "kr_records_gather")

    # ---- wide windows (k > 32 or D > 16)
    def set_params_wide(self, L, D, R, omit_soft=False, max_bases=0):
        self._check(self.lib.kr_set_params_wide(self.ctx, L, D, R, SOFT_OMIT if omit_soft else SOFT_MAP,
                                                max_bases), "kr_set_params_wide")
        self.params = (L, D, R)

    def wide_run(self, gids, is_ingroup, apply_filter=True):
        ids = np.asarray(gids, dtype=np.int32)
        flags = np.asarray([1 if f else 0 for f in is_ingroup], dtype=np.uint8)
        return self._check(self.lib.kr_wide_run(self.ctx, _ptr(ids), len(ids), _ptr(flags),
                                                1 if apply_filter else 0), "kr_wide_run")

    def wide_count(self, what):
        """number of entries kr_wide_fetch(what) would return"""
        return self._check(self.lib.kr_wide_fetch(self.ctx, what, None, 0), "kr_wide_fetch")

    def wide_fetch(self, what):
        n = self._check(self.lib.kr_wide_fetch(self.ctx, what, None, 0), "kr_wide_fetch")
        out = np.empty(max(n, 1), dtype=WIDE_HIT if what == WIDE_HITS else np.uint64)
        self._check(self.lib.kr_wide_fetch(self.ctx, what, _ptr(out), out.nbytes), "kr_wide_fetch")
        return out[:n]

    def wide_windows(self, k):
        """the member windows of the latest wide_run as text, cut on the device (kr_wide_fetch_windows): uint8 [nhits, k]"""
        n = self._check(self.lib.kr_wide_fetch_windows(self.ctx, None, 0), "kr_wide_fetch_windows")
        out = np.empty((max(n, 1), k), dtype=np.uint8)
        self._check(self.lib.kr_wide_fetch_windows(self.ctx, _ptr(out), out.nbytes), "kr_wide_fetch_windows")
        return out[:n]

    # ---- timing
    def sync(self):
        self._check(self.lib.kr_sync(self.ctx), "kr_sync")

    def timer_begin(self):
        self._check(self.lib.kr_timer_begin(self.ctx), "kr_timer_begin")

    def timer_end_ms(self):
        ms = self.lib.kr_timer_end_ms(self.ctx)
        if ms < 0:
            raise KrispHipError("kr_timer_end_ms failed")
        return ms

    def stage_enable(self, on=True):
        self.lib.kr_stage_enable(self.ctx, 1 if on else 0)

    def stage_select(self, names):
        """time only the named stages (event pairs around their launches)"""
        mask = 0
        for n in names:
            mask |= 1 << STAGES.index(n)
        self.lib.kr_stage_select(self.ctx, mask)

    def stage_reset(self):
        self.lib.kr_stage_reset(self.ctx)

    def stage_times(self):
        return {name: (self.lib.kr_stage_ms(self.ctx, i), self.lib.kr_stage_launches(self.ctx, i))
                for i, name in enumerate(STAGES)}

    # ---- introspection (stage-level parity tests)
    def debug_info(self):
        o = np.zeros(8, dtype=np.int64)
        self.lib.kr_debug_info(self.ctx, _ptr(o))
        return dict(b=int(o[0]), nbuckets=int(o[1]), T=int(o[2]), CAP=int(o[3]), nwg=int(o[4]),
                    overflow_segments=int(o[5]), fallback_launches=int(o[6]), nslices=int(o[7]))

    def debug_isect(self):
        o = np.zeros(8, dtype=np.int64)
        self.lib.kr_debug_isect(self.ctx, _ptr(o))
        return dict(chunk_kernel_items=int(o[0]), slices_redone=int(o[1]), threads=int(o[2]), buckets_per_item_log2=int(o[3]),
                    heads32=int(o[4]), sort_lanes=int(o[5]), splits=int(o[6]), probed=int(o[7]))

    def debug_lazy(self):
        """KR_OPT_LAZY_ORDER's counters (kr_debug_lazy): LDS sorts of whole slices the sorts left out, made later, collects
        that sorted the touched buckets only, the option's value"""
        o = np.zeros(8, dtype=np.int64)
        self.lib.kr_debug_lazy(self.ctx, _ptr(o))
        return dict(skipped=int(o[0]), ordered_later=int(o[1]), touch_collects=int(o[2]), on=int(o[3]),
                    anchor_in_bucket_order=int(o[4]), fuse_anchor=int(o[5]))

    def copy_gbps(self, nbytes=1 << 30, reps=10):
        v = self.lib.kr_debug_copy_gbps(self.ctx, nbytes, reps)
        if v < 0:
            raise KrispHipError("kr_debug_copy_gbps failed")
        return v

    def mem_info(self):
        """bytes this context may still allocate (device free memory / rest of its budget), device total, held, budget"""
        o = np.zeros(4, dtype=np.int64)
        self._check(self.lib.kr_mem_info(self.ctx, _ptr(o)), "kr_mem_info")
        return dict(avail=int(o[0]), total=int(o[1]), used=int(o[2]), budget=int(o[3]))

    def debug_comm(self):
        """what the multi-GPU exchange has cost so far (kr_debug_comm)"""
        o = np.zeros(8, dtype=np.int64)
        self._check(self.lib.kr_debug_comm(self.ctx, _ptr(o)), "kr_debug_comm")
        return dict(syncs=int(o[0]), p2p=int(o[1]), collectives=int(o[2]), exchange_us=int(o[3]), reduces=int(o[4]),
                    message_entries=int(o[5]))

    def debug_place(self):
        """the latest placement search of the pass-1 output buffers (kr_debug_place)"""
        o = np.zeros(8, dtype=np.float64)
        self._check(self.lib.kr_debug_place(self.ctx, _ptr(o)), "kr_debug_place")
        return dict(candidates=int(o[0]), taken=int(o[1]), fastest_ms=[round(float(x), 4) for x in o[2:6]],
                    median_ms=round(float(o[6]), 4), slowest_ms=round(float(o[7]), 4))

    def comm_set_timeout(self, seconds):
        """the exchange's deadline: the watchdog aborts the communicator `seconds` after an exchange call began (kr_comm_set_timeout)"""
        self._check(self.lib.kr_comm_set_timeout(self.ctx, int(seconds)), "kr_comm_set_timeout")

    def debug_comm_hang(self):
        """test aid: a receive without a sender; raises once the watchdog has aborted the communicator (kr_debug_comm_hang)"""
        return self._check(self.lib.kr_debug_comm_hang(self.ctx), "kr_debug_comm_hang")

    def cands_selfexchange(self, apply_filter=False):
        """one round of the tree with the rank itself as partner, over RCCL (kr_debug_cands_selfexchange)"""
        return self._check(self.lib.kr_debug_cands_selfexchange(self.ctx, 1 if apply_filter else 0), "kr_debug_cands_selfexchange")

    def comm_probe(self, nbytes=64 << 10, reps=50):
        """microseconds per blocking call of the exchange (kr_debug_comm_probe; RCCL communicators only)"""
        o = np.zeros(3, dtype=np.float64)
        self._check(self.lib.kr_debug_comm_probe(self.ctx, nbytes, reps, _ptr(o)), "kr_debug_comm_probe")
        return dict(allreduce_sync_us=round(float(o[0]), 2), sendrecv_self_sync_us=round(float(o[1]), 2),
                    bcast_sync_us=round(float(o[2]), 2), message_bytes=int(nbytes))

    def copy_which(self):
        """the copy form and grid the latest copy_gbps() found fastest"""
        return (self.lib.kr_debug_copy_which(self.ctx) or b"").decode()

    def inversions(self, gid):
        return self._check(self.lib.kr_debug_inversions(self.ctx, gid), "kr_debug_inversions")

    def debug_fetch(self, gid, what, n_max):
        dt = {0: np.uint64, 1: np.uint32, 2: np.uint32, 3: np.uint32, 4: np.uint64, 5: np.uint64}[what]
        out = np.empty(max(n_max, 1), dtype=dt)
        n = self._check(self.lib.kr_debug_fetch(self.ctx, gid, what, _ptr(out), out.nbytes), "kr_debug_fetch")
        return out[:n]
