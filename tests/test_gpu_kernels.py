"""GPU parity: libkrisp_hip.so (through the C ABI) against the packed-key oracle
(oracle/kmer_oracle.c) on the same seeded inputs.  Bit-exact (integer keys)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from krisp_amd import _native
    return _native


@pytest.fixture(scope="module")
def K():
    from oracle import kmer_oracle
    kmer_oracle.build()
    return kmer_oracle


def _rand_text(seed, n, alphabet=b"ACGT", records=3):
    rng = np.random.default_rng(seed)
    a = np.frombuffer(alphabet, dtype=np.uint8)
    t = a[rng.integers(0, len(a), size=n)]
    if records > 1 and n > records:
        for p in rng.integers(0, n, size=records - 1):
            t[p] = 10
    return t


def _check_sorted(N, K, text, L, D, R, omit=False, stages=False, slice_bases=None):
    want = K.sorted_keys(text.tobytes(), L, D, R, omit=omit)
    with N.Engine() as e:
        if slice_bases is not None:
            e.set_option(N.OPT_SLICE_BASES, slice_bases)
        e.set_params(L, D, R, omit_soft=omit, max_bases=len(text))
        e.upload(0, text)
        e.sort(0)
        info = e.debug_info()
        assert e.count(0) == len(want), info
        # (the merge fallback for oversized buckets reuses the pass-1 array as its scratch)
        if stages and info["nslices"] == 1 and e.debug_info()["fallback_launches"] == 0:
            b = info["b"]
            top = (want >> np.uint64(64 - b)).astype(np.int64)
            hist = np.bincount(top, minlength=1 << b).astype(np.uint32)
            h8 = np.bincount((want >> np.uint64(56)).astype(np.int64), minlength=256)
            base1 = np.concatenate([[0], np.cumsum(h8)]).astype(np.uint32)
            assert np.array_equal(e.debug_fetch(0, 2, 257), base1), "top-byte bucket bases"
            off = np.concatenate([[0], np.cumsum(hist)]).astype(np.uint32)
            assert np.array_equal(e.debug_fetch(0, 3, (1 << b) + 1), off), "bucket offsets"
            p1 = e.debug_fetch(0, 4, len(want) + 4)
            assert len(p1) == len(want)
            assert np.array_equal(np.sort(p1), want), "pass-1 output is a permutation of the keys"
            d1 = (p1 >> np.uint64(56)).astype(np.int64)
            assert np.all(np.diff(d1) >= 0), "pass-1 output partitioned by the top byte"
        got = e.keys(0)
        info = e.debug_info()
        assert np.array_equal(got, want), info
        return info


def test_sort_random_genome_with_stages(N, K):
    text = _rand_text(1, 300_000, b"ACGT" * 12 + b"acgtN", records=5)
    info = _check_sorted(N, K, text, 25, 1, 2, stages=True)
    assert info["overflow_segments"] == 0


def test_sort_omit_soft(N, K):
    text = _rand_text(2, 200_000, b"ACGT" * 12 + b"acgtNn", records=4)
    _check_sorted(N, K, text, 25, 1, 2, omit=True, stages=True)


@pytest.mark.parametrize("L,D,R", [(25, 1, 2), (28, 1, 2), (3, 1, 2), (15, 2, 15), (16, 0, 16),
                                   (0, 2, 3), (3, 2, 0), (1, 0, 0), (10, 16, 6), (12, 4, 12)])
def test_sort_geometries(N, K, L, D, R):
    text = _rand_text(10 + L + D + R, 50_000, b"ACGT" * 10 + b"aN", records=3)
    _check_sorted(N, K, text, L, D, R, stages=True)


@pytest.mark.parametrize("sb", [None, 1])
@pytest.mark.parametrize("L,D,R", [(16, 16, 0), (16, 8, 8), (16, 1, 0), (17, 0, 15), (20, 6, 6), (24, 4, 4), (31, 1, 0),
                                   (31, 0, 1), (32, 0, 0), (30, 1, 1), (19, 2, 3), (15, 1, 16), (15, 16, 1)])
def test_pass_1_generator_that_shifts_the_word_string_once(N, K, L, D, R, sb):
    """k_scatter1p<., ., 1> (p1_fast_keys: `left` of 16 bases or more, both strands): every width of the three fields
    around its limits, window lengths from 17 to 32, with and without key-space slices (its BIG form is their pass 0);
    L = 15 beside them takes the general generator.  Texts whose length is no multiple of a code word, with N runs,
    soft-masked stretches and several records."""
    text = _rand_text(4000 + 37 * L + 5 * D + R, 61_003 + L, b"ACGT" * 14 + b"acgtNn", records=6)
    for omit in (False, True):
        _check_sorted(N, K, text, L, D, R, omit=omit, stages=sb is None, slice_bases=sb)


@pytest.mark.parametrize("n,L,D,R", [
    (300_000, 25, 1, 2),      # fan-out 2^9: fine offsets straight from the codes (k_hist16)
    (300_000, 5, 1, 2),       # 2 L = 10 >= 9: still k_hist16, the top bits end inside `left`
    (300_000, 4, 2, 3),       # 2 L = 8 < 9: k_hist2 / k_scan2 on the pass-1 output
    (2_500_000, 6, 2, 4),     # fan-out 2^12, 2 L = 12: the boundary case
    (2_500_000, 5, 3, 4),     # ... and just below it
    (5_000_000, 7, 0, 7),     # fan-out 2^13 (odd number of bits: 7 bases hold them), L = 7
    (5_000_000, 6, 1, 6),     # ... L = 6 is too short
    (24_000_000, 12, 4, 12),  # fan-out 2^15 = one full partition of 32768 bins
])
def test_sort_fine_offsets_from_codes_or_keys(N, K, n, L, D, R):
    """both ways to the fine bucket offsets (k_hist16 when the top b key bits lie inside `left`,
    else k_hist2 + k_scan2), checked stage by stage: bucket offsets, pass-1 permutation, keys"""
    text = _rand_text(1000 + n % 97 + L, n, alphabet=b"ACGTACGTACGTACGTN", records=5)
    info = _check_sorted(N, K, text, L, D, R, stages=True)
    assert info["overflow_segments"] == 0


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("KR_MID_SEEDS", "12"))))
def test_sort_random_mid_sizes(N, K, seed):
    """genomes large enough for the second partition pass (fan-out 2^9 .. 2^13), random lengths
    (word and tile boundaries fall anywhere), geometries on both sides of the 2 L >= b rule,
    N runs, soft masks, many short records"""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.integers(210_000, 6_000_000))
    L = int(rng.integers(1, 33))
    R = int(rng.integers(0, 33 - L))
    D = int(rng.integers(0, min(16, 32 - L - R) + 1))
    alphabet = [b"ACGT", b"ACGTACGTACGTN", b"ACGTACGTACGTacgtn", b"AACCGT"][seed % 4]
    text = _rand_text(6000 + seed, n, alphabet=alphabet, records=int(rng.integers(1, 400)))
    _check_sorted(N, K, text, L, D, R, omit=bool(seed % 3 == 1), stages=True)


@pytest.mark.parametrize("n", [0, 1, 5, 27, 28, 29, 31, 32, 33, 63, 64, 65, 100, 4095, 4097, 8193])
def test_sort_tiny_inputs(N, K, n):
    text = _rand_text(100 + n, n, b"ACGT", records=1)
    _check_sorted(N, K, text, 25, 1, 2)


def test_sort_all_invalid(N, K):
    text = np.frombuffer(b"N" * 1000 + b"\n" + b"acgt" * 100, dtype=np.uint8)
    _check_sorted(N, K, text, 5, 1, 2, omit=True)


def test_sort_duplicates_use_bitonic_path(N, K):
    # few distinct keys, many copies: crowded sub-bins inside the LDS sort
    rng = np.random.default_rng(5)
    unit = _rand_text(6, 97, b"ACGT", records=1)
    text = np.concatenate([unit] * 40 + [_rand_text(7, 3000, b"ACGT", records=1)])
    _check_sorted(N, K, text, 25, 1, 2)


def test_sort_skewed_genome_overflow_fallback(N, K):
    # 60 kbp of poly-A + a tandem repeat: one fine bucket far above the LDS capacity
    rng = np.random.default_rng(8)
    parts = [_rand_text(9, 400_000, b"ACGT", records=4),
             np.frombuffer(b"A" * 60_000, dtype=np.uint8),
             _rand_text(11, 100_000, b"ACGT", records=1),
             np.frombuffer(b"ACACACACAC" * 3000, dtype=np.uint8)]
    text = np.concatenate(parts)
    info = _check_sorted(N, K, text, 25, 1, 2)
    assert info["overflow_segments"] >= 1 and info["fallback_launches"] > 0


@pytest.mark.parametrize("sb", [1, 2])
@pytest.mark.parametrize("n,L,D,R,omit", [(300_000, 25, 1, 2, False), (2_500_000, 6, 2, 4, False),
                                          (1_200_000, 28, 1, 2, True), (40_000, 3, 1, 2, False),
                                          (700_000, 12, 4, 12, False), (64, 25, 1, 2, False)])
def test_sort_in_key_space_slices_vs_oracle(N, K, sb, n, L, D, R, omit):
    """the large-genome path at test sizes: KR_OPT_SLICE_BASES forces 4^sb key-space slices (pass 0
    over all keys, then per slice a pass 1 from the pass-0 array, pass 2, LDS sort); the fetched
    keys equal the packed oracle's"""
    text = _rand_text(300 + n % 89 + sb, n, alphabet=b"ACGTACGTACGTACGTNacgt", records=7)
    info = _check_sorted(N, K, text, L, D, R, omit=omit, slice_bases=sb)
    assert info["nslices"] == 4 ** min(sb, L)


@pytest.mark.parametrize("route", [1, 0])
@pytest.mark.parametrize("sb,n,L,D,R", [(3, 900_000, 28, 1, 2), (4, 1_500_000, 25, 1, 2), (3, 400_000, 5, 1, 2), (2, 600_000, 16, 0, 16)])
def test_slice_pass_1_counted_from_the_codes_or_from_the_keys(N, K, route, sb, n, L, D, R, monkeypatch):
    """round 4: a slice's pass 1 is a k_scatter2 over its pass-0 buckets whose (slice, top byte) bins were counted from the
    codes once per genome (64 and 256 slices as configs[4] would take them; where the top 8 + 2 sb bits reach beyond
    `left` -- L = 5 at sb = 3 -- the slice counts its keys again as in round 3); KR_SLICE_ROUTE=0 keeps round 3's route
    everywhere: the same sorted keys either way, also for a genome with a satellite (one bin far above the 16-bit counters
    of k_hist16) and N runs"""
    monkeypatch.setenv("KR_SLICE_ROUTE", str(route))
    parts = [_rand_text(500 + sb, n, alphabet=b"ACGTACGTACGTACGTNacgt", records=5), np.frombuffer(b"\n" + b"ACGTTGCA" * 12_000, dtype=np.uint8),
             np.frombuffer(b"\n" + b"T" * 70_000, dtype=np.uint8)]
    info = _check_sorted(N, K, np.concatenate(parts), L, D, R, slice_bases=sb)
    assert info["nslices"] == 4 ** min(sb, L)


@pytest.mark.parametrize("strands", [1, 2])
def test_single_strand_modes_sort_alike_with_and_without_slices(N, strands):
    """forward-only and canonical streams (kstream's other strand options) keep round 3's slice route (a counting pass per
    slice) behind the pass 0 that now partitions by the slice alone: the same keys as one sort unit gives"""
    text = _rand_text(77, 800_000, alphabet=b"ACGTACGTACGTNacgt", records=6)
    got = []
    for sb in (0, 2, 3):
        with N.Engine() as e:
            e.set_option(N.OPT_SLICE_BASES, sb)
            e.set_params(20, 0, 0, max_bases=len(text))
            e.set_strands(strands)
            e.upload(0, text)
            e.sort(0)
            assert e.debug_info()["nslices"] == 4 ** sb
            got.append(e.keys(0).copy())
    assert len(got[0]) > 100_000 and np.all(got[0][1:] >= got[0][:-1])
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])


@pytest.mark.parametrize("sb,generic,fmt,kern", [(1, 0, 0, 0), (2, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (2, 1, 1, 0),
                                                 (0, 0, 0, 1), (1, 0, 1, 1), (0, 0, 0, 2), (1, 0, 1, 2), (0, 0, 2, 0)])
@pytest.mark.parametrize("L,D,R,length,n", [(25, 1, 2, 200_000, 4), (12, 4, 12, 100_000, 3), (8, 1, 4, 20_000, 5)])
def test_intersect_and_collect_under_every_option(N, K, sb, generic, fmt, kern, L, D, R, length, n):
    """the result-neutral options of kr_set_option (key-space slices, generic intersect sub-tiles,
    narrow per-prefix state, the chunk kernel instead of the pipelined one): candidates, masks and
    records equal the packed oracle's under each"""
    fam = _family(L + D + R + sb, n, length)
    flags = [f for _, f, _ in fam]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for _, _, t in fam]
    with N.Engine() as e:
        e.set_option(N.OPT_SLICE_BASES, sb)
        e.set_option(N.OPT_GENERIC_INTERSECT, generic)
        e.set_option(N.OPT_ISECT_FORMAT, fmt)
        e.set_option(N.OPT_ISECT_KERNEL, kern)
        e.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        assert e.debug_info()["nslices"] == 4 ** min(sb, L)
        for i, (_, _, t) in enumerate(fam):
            assert e.add(i, t) == len(want_keys[i])
            assert np.array_equal(e.keys(i), want_keys[i])
        for filt in (False, True):
            want = K.intersect(want_keys, flags, L, D, R, apply_filter=filt)
            assert e.intersect(list(range(n)), flags, apply_filter=filt) == len(want)
            got = e.cands()
            for f in ("prefix", "in_mask", "out_mask"):
                assert np.array_equal(got[f], want[f]), f
            recs = np.sort(e.collect(list(range(n))), order=["key", "genome"])
            wrec = np.sort(K.collect(want_keys, want, L, D, R), order=["key", "genome"])
            assert np.array_equal(recs, wrec)


def test_options_are_checked(N, monkeypatch):
    """kr_set_option: bad values, bad order and the ablation switch of a regular build are refused;
    KR_DBG in the environment reaches only its documented, result-neutral bit"""
    monkeypatch.setenv("KR_DBG", str(64 | 128 | 512))      # round-1 ablation bits: must be ignored
    with N.Engine() as e:
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_SLICE_BASES, 5)
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_ISECT_FORMAT, 3)
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_LANES, 9)
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_ABLATE, 64)
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_PLACE_TRIES, 0)
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_WIDE_SLOTS, 2)
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_WIDE_ORDERED, -1)
        with pytest.raises(N.KrispHipError):
            e.set_option(99, 0)
        text = _rand_text(5, 100_000, b"ACGT", records=2)
        e.set_params(25, 1, 2, max_bases=len(text))
        e.upload(0, text)
        with pytest.raises(N.KrispHipError):
            e.set_option(N.OPT_SLICE_BASES, 1)             # after an upload
        e.sort(0)
        assert e.inversions(0) == 0                          # (KR_DBG=64 used to skip the sort)


def test_hard_limits_fail_loudly(N):
    """the library's caps are errors with a message, never silent truncation: genomes of 2^33 bases (2^32 on the wide path),
    more keys in one sort unit than a buffer descriptor spans, geometries beyond the key and mask formats"""
    with N.Engine() as e:
        with pytest.raises(N.KrispHipError, match="2\\^33"):
            e.set_params(25, 1, 2, max_bases=(1 << 33) - 10)     # (round 6: the packed path takes genomes below 2^33 bases)
        with pytest.raises(N.KrispHipError, match="2\\^32"):
            e.set_params_wide(30, 40, 30, max_bases=(1 << 32) - 10)
        with pytest.raises(N.KrispHipError):
            e.set_params(20, 1, 12, max_bases=1000)              # k = 33
        with pytest.raises(N.KrispHipError):
            e.set_params(5, 17, 5, max_bases=1000)               # D > 16
        e.set_option(N.OPT_SLICE_BASES, 0)
        with pytest.raises(N.KrispHipError, match="buffer descriptor"):
            e.set_params(25, 1, 2, max_bases=300_000_000)        # 6e8 keys forced into one sort unit
    with N.Engine() as e:
        e.set_params(25, 1, 2, max_bases=1000)
        with pytest.raises(N.KrispHipError, match="exceeds max_bases"):
            e.upload(0, np.full(2000, 65, dtype=np.uint8))
        text = _rand_text(3, 600, b"ACGT", records=1)
        for i in range(33):
            e.add(i, text)
        ids = np.arange(33, dtype=np.int32)
        flags = np.ones(33, dtype=np.uint8)
        # (more than 32 genomes per call: batches inside the library since round 3, the running list stays on the device)
        rc = e.lib.kr_intersect(e.ctx, N._ptr(ids), 33, N._ptr(flags), 0)
        assert rc > 0 and e.intersect(list(range(33)), [True] * 33, apply_filter=False) == rc
    with N.Engine() as e:
        with pytest.raises(N.KrispHipError):
            e.set_params_wide(257, 10, 20, max_bases=1000)       # flank > KR_WIDE_MAX_FLANK (256: eight pieces of 32 bases)
        with pytest.raises(N.KrispHipError):
            e.set_params_wide(30, 1000, 30, max_bases=1000)      # amplicon > KR_WIDE_MAX_K (1024)
        e.set_params_wide(65, 10, 20, max_bases=1000)            # (what rounds 1-5 refused)
        e.set_params_wide(30, 250, 30, max_bases=1000)


def test_more_oversized_buckets_than_the_list_holds(N, K):
    """150 tandem repeats of distinct 28-base units, 3000 copies each: thousands of fine buckets of
    3000 equal keys.  More than the per-workgroup notes and the 4096-entry overflow list hold: the
    sort falls back to tile sort + merge rounds over the whole array, and is still exact"""
    rng = np.random.default_rng(77)
    parts = []
    for _ in range(150):
        unit = b"ACGT"[0:0] + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=28))
        parts.append(unit * 3000)
        parts.append(bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=200)))
    text = np.frombuffer(b"".join(parts), dtype=np.uint8)
    info = _check_sorted(N, K, text, 25, 1, 2)
    assert info["overflow_segments"] >= 1 and info["fallback_launches"] > 0


def test_key_space_slices_report(N):
    """KR_SLICE_BASES=n forces 4^n key-space slices (the large-genome path) through every test."""
    import os
    with N.Engine() as e:
        e.set_params(25, 1, 2, max_bases=1000)
        want = 4 ** int(os.environ.get("KR_SLICE_BASES", "0"))
        assert e.debug_info()["nslices"] == want
    if "KR_SLICE_BASES" not in os.environ:
        # the size policy: one sort unit up to 4.2e8 keys, then slices of <= 1.05e8 keys
        for bases, want in ((50_000_000, 1), (200_000_000, 1), (500_000_000, 16), (3_000_000_000, 64)):
            with N.Engine() as e:
                e.set_params(25, 1, 2, max_bases=bases)
                assert e.debug_info()["nslices"] == want, bases


def _family(seed, n, length, mu=0.01):
    from krisp_amd import synth
    return synth.family(seed, n // 2, n - n // 2, length, records=4, mu=mu, snp_every=2000)


@pytest.mark.parametrize("L,D,R,length,n", [(25, 1, 2, 200_000, 4), (8, 1, 4, 20_000, 5),
                                            (12, 4, 12, 100_000, 3), (14, 0, 14, 100_000, 4),
                                            (3, 1, 0, 5_000, 2), (2, 1, 1, 3_000, 4)])
def test_intersect_and_collect(N, K, L, D, R, length, n):
    fam = _family(L + D + R, n, length)
    flags = [f for _, f, _ in fam]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for _, _, t in fam]
    with N.Engine() as e:
        e.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            assert e.add(i, t) == len(want_keys[i])
        for filt in (False, True):
            want = K.intersect(want_keys, flags, L, D, R, apply_filter=filt)
            ncand = e.intersect(list(range(n)), flags, apply_filter=filt)
            got = e.cands()
            assert ncand == len(want)
            assert np.array_equal(got["prefix"], want["prefix"])
            assert np.array_equal(got["in_mask"], want["in_mask"])
            assert np.array_equal(got["out_mask"], want["out_mask"])
            recs = e.collect(list(range(n)))
            wrec = K.collect(want_keys, want, L, D, R)
            got_sorted = np.sort(recs, order=["key", "genome"])
            want_sorted = np.sort(wrec, order=["key", "genome"])
            assert np.array_equal(got_sorted, want_sorted)
            assert np.array_equal(recs, got_sorted), "kr_collect returns (key, genome position) order"
            # ... for any order of the genomes in the call: position in the call breaks the ties
            perm = list(range(n))[::-1]
            recs2 = e.collect(perm)
            pos = np.array([perm.index(g) for g in range(n)])
            o2 = np.lexsort((pos[recs2["genome"].astype(np.int64)], recs2["key"]))
            assert np.array_equal(o2, np.arange(len(recs2))) and np.array_equal(np.sort(recs2, order=["key", "genome"]), want_sorted)
        # deferred filter == fused filter; list (x) list merge == n-way intersect
        want_f = K.intersect(want_keys, flags, L, D, R, apply_filter=True)
        e.intersect(list(range(n)), flags, apply_filter=False)
        assert e.merge_cands(None, apply_filter=True) == len(want_f)
        assert np.array_equal(e.cands()["prefix"], want_f["prefix"])
        if n >= 4:
            half = n // 2
            e.intersect(list(range(half)), flags[:half], apply_filter=False)
            a = e.cands().copy()
            e.intersect(list(range(half, n)), flags[half:], apply_filter=False)
            assert e.merge_cands(a, apply_filter=True) == len(want_f)
            got = e.cands()
            assert np.array_equal(got["prefix"], want_f["prefix"])
            assert np.array_equal(got["in_mask"], want_f["in_mask"])
            assert np.array_equal(got["out_mask"], want_f["out_mask"])


def _check_intersect(N, K, texts, flags, L, D, R, env=None, kern=0, fmt=0):
    """n-way intersect + collect of `texts` against the packed oracle, unfiltered and filtered
    (kern: KR_OPT_ISECT_KERNEL -- 0 the pipelined kernels, 32-bit heads where the geometry allows; 2 64-bit heads)"""
    n = len(texts)
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for t in texts]
    with N.Engine() as e:
        e.set_option(N.OPT_ISECT_KERNEL, kern)         # (these tests are about the pipelined kernels, whatever the environment says)
        e.set_option(N.OPT_ISECT_FORMAT, fmt)
        e.set_option(N.OPT_GENERIC_INTERSECT, 0)
        e.set_params(L, D, R, max_bases=max(len(t) for t in texts))
        for i, t in enumerate(texts):
            assert e.add(i, t) == len(want_keys[i])
        for filt in (False, True):
            want = K.intersect(want_keys, flags, L, D, R, apply_filter=filt)
            assert e.intersect(list(range(n)), flags, apply_filter=filt) == len(want)
            got = e.cands()
            for f in ("prefix", "in_mask", "out_mask"):
                assert np.array_equal(got[f], want[f]), (f, filt)
            recs = e.collect(list(range(n)))
            wrec = np.sort(K.collect(want_keys, want, L, D, R), order=["key", "genome"])
            assert np.array_equal(recs, wrec)
        return e.debug_isect(), e.debug_info()


@pytest.fixture(autouse=True)
def _one_sort_unit_for_the_shape_tests(request, monkeypatch):
    """the tests that name the kernel shapes the host picks (threads, heads, items) are about ONE sort unit: KR_SLICE_BASES
    in the environment (the builder's runs of the whole suite under the result-neutral switches) does not reach them"""
    if request.node.name.startswith("test_pipelined_intersect"):
        monkeypatch.delenv("KR_SLICE_BASES", raising=False)


@pytest.mark.parametrize("length,threads,mlog", [(500_000, 320, 0), (600_000, 384, 0), (700_000, 448, 0), (800_000, 512, 0),
                                                 (450_000, 512, 1), (3_000, None, None)])
def test_pipelined_intersect_item_shapes(N, K, length, threads, mlog, monkeypatch):
    """k_intersect3 at every workgroup size the host picks (items of 4 T slots sized from the bucket
    statistics: 256 .. 512 threads, one or several buckets per item), anchor = the shortest genome"""
    from krisp_amd import synth
    monkeypatch.delenv("KR_SLICE_BASES", raising=False)       # (the shapes below are those of ONE sort unit)
    fam = synth.family(length % 97, 2, 1, length, records=3, mu=0.01, snp_every=1500)
    texts = [t for _, _, t in fam]
    texts[1] = texts[1][: len(texts[1]) - 37]            # the anchor (fewest keys) is not genome 0
    for kern in (0, 2):
        info, _ = _check_intersect(N, K, texts, [f for _, f, _ in fam], 20, 2, 5, kern=kern)
        assert (threads is None or info["threads"] == threads) and info["slices_redone"] == 0
        assert threads is None or info["heads32"] == (1 if kern == 0 else 0)     # (the tiny case: many buckets per item, 64-bit heads)
        if mlog is not None:
            assert info["buckets_per_item_log2"] == mlog


@pytest.mark.parametrize("L,D,R,n,length,heads32", [(20, 1, 6, 7, 120_000, 1), (11, 6, 10, 4, 120_000, 1), (9, 10, 9, 5, 120_000, 1),
                                                    (12, 16, 4, 3, 120_000, 0), (16, 0, 16, 6, 120_000, 0), (10, 1, 3, 32, 120_000, 0),
                                                    (25, 1, 2, 3, 1_300_000, 1), (16, 0, 16, 3, 2_000_000, 0), (15, 2, 15, 3, 900_000, 0),
                                                    (13, 2, 13, 3, 900_000, 1),
                                                    (20, 1, 6, 24, 120_000, 1), (20, 1, 6, 25, 120_000, 1)])     # (24 genomes: the last that fit the 32-bit slot word)
def test_pipelined_intersect_formats_and_genome_counts(N, K, L, D, R, n, length, heads32):
    """every per-prefix state format (D <= 4, <= 8, <= 16), 3 .. 32 genomes per call, both pipelined kernels (32-bit
    heads where the sub-bin field lies at most 32 bits above the prefix's lowest bit, else 64-bit heads; and 64-bit
    heads on request)"""
    from krisp_amd import synth
    fam = synth.family(L + n, (n + 1) // 2, n // 2, length, records=2, mu=0.004, snp_every=700)
    info, _ = _check_intersect(N, K, [t for _, _, t in fam], [f for _, f, _ in fam], L, D, R)
    assert info["threads"] >= 256 and info["heads32"] == heads32
    if heads32:
        info, _ = _check_intersect(N, K, [t for _, _, t in fam], [f for _, f, _ in fam], L, D, R, kern=2)
        assert info["heads32"] == 0
        if D == 1:      # (one column, <= 24 genomes: a 32-bit word per prefix; KR_OPT_ISECT_FORMAT = 2: the 64-bit state)
            info, _ = _check_intersect(N, K, [t for _, _, t in fam], [f for _, f, _ in fam], L, D, R, fmt=2)
            assert info["heads32"] == 1


def test_pipelined_intersect_runs_of_equal_prefixes(N, K):
    """genomes that hold every stretch twice, the copy with substitutions: runs of anchor keys with one
    (left,right) prefix -- equal keys and keys that differ in the diagnostic columns only"""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, size=150_000)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    texts = []
    for g in range(4):
        a = base.copy()
        m = rng.random(len(a)) < 0.004
        a[m] = (a[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        b2 = a.copy()
        m = rng.random(len(a)) < 0.03
        b2[m] = (b2[m] + rng.integers(1, 4, size=int(m.sum()))) % 4
        texts.append(np.concatenate([acgt[a], [10], acgt[b2], [10], acgt[a[:40_000]]]).astype(np.uint8))
    _check_intersect(N, K, texts, [1, 1, 0, 0], 8, 2, 3)
    _check_intersect(N, K, texts, [1, 0, 1, 0], 13, 1, 2)
    info, _ = _check_intersect(N, K, texts, [1, 0, 1, 0], 20, 1, 4)
    assert info["heads32"] == 1
    _check_intersect(N, K, texts, [1, 0, 1, 0], 20, 1, 4, kern=2)


def test_pipelined_intersect_oversized_items(N, K, monkeypatch):
    """skew: satellites put more keys into some items than a workgroup has slots -- those items go to the
    chunk kernel; with more of them than its list holds (KR_ISECT_OVF_CAP lowers it for the test) the
    whole slice is redone by chunks.  Same candidates and records either way."""
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = rng.integers(0, 4, size=400_000)
    unit = rng.integers(0, 4, size=37)
    texts = []
    for g in range(3):
        a = base.copy()
        m = rng.random(len(a)) < 0.005
        a[m] = (a[m] + 1) % 4
        sat = np.tile(unit, 4000 + 50 * g)
        ms = rng.random(len(sat)) < 0.02
        sat[ms] = (sat[ms] + 2) % 4
        texts.append(np.concatenate([acgt[a[:200_000]], acgt[sat], acgt[a[200_000:]]]).astype(np.uint8))
    info, _ = _check_intersect(N, K, texts, [1, 0, 0], 14, 1, 6)
    assert info["chunk_kernel_items"] > 0 and info["slices_redone"] == 0
    monkeypatch.setenv("KR_ISECT_OVF_CAP", "2")
    info, _ = _check_intersect(N, K, texts, [1, 0, 0], 14, 1, 6)
    assert info["slices_redone"] > 0


def test_more_genomes_than_one_intersect_call_takes(N, K):
    """40 genomes (kr_intersect takes 32): the cascade over batches equals the oracle's n-way
    intersection, on the one-key path and on the wide path"""
    from krisp_amd import amplicon, synth
    from krisp_amd import krisp_fasta as KF
    L, D, R = 9, 1, 4
    fam = synth.family(5, 25, 15, 30_000, records=2, mu=0.0005, snp_every=400)
    flags = [f for _, f, _ in fam]
    ids = list(range(len(fam)))
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for _, _, t in fam]
    want = K.intersect(want_keys, flags, L, D, R, apply_filter=True)
    assert len(want) > 5
    with N.Engine() as e:
        e.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            e.add(i, t)
        assert e.intersect(ids, flags, apply_filter=True) == len(want)
        got = e.cands()
        for f in ("prefix", "in_mask", "out_mask"):
            assert np.array_equal(got[f], want[f])
        recs = e.collect(ids)
        packed = amplicon.groups_from_records(recs, [nm for nm, _, _ in fam], L, D, R)
    with N.Engine() as e:
        e.set_params_wide(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            e.upload(i, t)
        assert e.wide_run(ids, flags, apply_filter=True) > 0
        hits = e.wide_fetch(N.WIDE_HITS)
    wide = KF._groups_from_hits(hits, [t for _, _, t in fam], [nm for nm, _, _ in fam], L, D, R)
    assert amplicon.merged_lines(wide) == amplicon.merged_lines(packed)


@pytest.mark.parametrize("geo", [(12, 30, 12), (30, 8, 30), (32, 20, 32), (40, 12, 40)])
def test_wide_right_flank_through_the_left_dictionary(N, geo, monkeypatch):
    """L == R: the rights present in every genome are the reverse complements of the lefts, so kr_wide_run
    builds one spectrum and numbers a right by that left (two look-ups per window start).  Against the same
    run with both spectra built (KR_WIDE_SHARE=0): the same hits; the right dictionary made on request (flanks
    of one key) equals the one that run builds."""
    from krisp_amd import amplicon
    from krisp_amd import krisp_fasta as KF
    L, D, R = geo
    fam = _family(91, 4, 300_000)
    flags = [f for _, f, _ in fam]
    ids = list(range(len(fam)))
    texts, names = [t for _, _, t in fam], [nm for nm, _, _ in fam]
    monkeypatch.delenv("KR_WIDE_ORDERED", raising=False)      # (the ordered generator never shares the spectrum)

    def run(share):
        monkeypatch.setenv("KR_WIDE_SHARE", "1" if share else "0")
        with N.Engine() as e:
            e.set_params_wide(L, D, R, max_bases=max(len(t) for t in texts))
            for i, t in enumerate(texts):
                e.upload(i, t)
            n = e.wide_run(ids, flags, apply_filter=False)
            bits = [int(x) for x in e.wide_fetch(N.WIDE_SLOT_BITS)]
            right = e.wide_fetch(N.WIDE_DICT_RIGHT) if L <= 32 else None
            return n, bits, right, amplicon.merged_lines(KF._groups_from_hits(e.wide_fetch(N.WIDE_HITS), texts, names, L, D, R)), \
                e.wide_count(N.WIDE_DICT_RIGHT), int(e.wide_fetch(N.WIDE_NGROUPS)[0])

    n1, bits1, right1, lines1, nr1, ng1 = run(True)
    n0, bits0, right0, lines0, nr0, ng0 = run(False)
    assert bits1[3] == 254 and bits0[3] != 254
    assert n1 == n0 > 0 and lines1 == lines0
    assert nr1 == nr0 and ng1 == ng0
    if right1 is not None:
        assert np.array_equal(right1, right0)


@pytest.mark.parametrize("geo", [(12, 30, 12), (32, 20, 32), (40, 12, 36), (20, 17, 9)])
def test_wide_spectra_with_the_packed_kernels_change_nothing(N, geo, monkeypatch):
    """With both strands in play kr_wide_run takes its flank spectra over every flank-long window with the packed
    path's kernels -- a superset of the flanks of the valid amplicon-long windows.  Against the window-by-window
    spectra (KR_WIDE_SPECTRUM=0): the same hits and groups, and the same k-mer record counts (amplicon-long windows,
    counted by k_count_windows) on genomes with N runs, soft masks and many short records."""
    from krisp_amd import amplicon
    from krisp_amd import krisp_fasta as KF
    L, D, R = geo
    fam = _family(93, 4, 200_000)
    rng = np.random.default_rng(7)
    texts = []
    for _, _, t in fam:
        t = t.copy()
        for p in rng.integers(0, len(t) - 50, size=40):          # N runs of 1 .. 40 bases, some next to record breaks
            t[p:p + int(rng.integers(1, 41))] = ord("N")
        for p in rng.integers(0, len(t), size=60):
            t[p] = 10
        texts.append(t)
    names = [nm for nm, _, _ in fam]
    flags = [f for _, f, _ in fam]
    ids = list(range(len(fam)))

    def run(packed):
        monkeypatch.setenv("KR_WIDE_SPECTRUM", "1" if packed else "0")
        with N.Engine() as e:
            e.set_params_wide(L, D, R, max_bases=max(len(t) for t in texts))
            for i, t in enumerate(texts):
                e.upload(i, t)
            n = e.wide_run(ids, flags, apply_filter=False)
            counts = [int(x) for x in e.wide_fetch(N.WIDE_COUNTS)]
            lines = amplicon.merged_lines(KF._groups_from_hits(e.wide_fetch(N.WIDE_HITS), texts, names, L, D, R))
            return n, counts, lines, int(e.wide_fetch(N.WIDE_NGROUPS)[0]), e.wide_count(N.WIDE_DICT_LEFT)

    n1, c1, l1, g1, d1 = run(True)
    n0, c0, l0, g0, d0 = run(False)
    assert n1 == n0 > 0 and l1 == l0 and g1 == g0
    assert c1 == c0 and all(c > 0 for c in c0)
    assert d1 >= d0


@pytest.mark.parametrize("geo,filt", [((12, 30, 12), True), ((32, 20, 32), False), ((40, 12, 36), True), ((20, 17, 9), True)])
def test_wide_members_as_a_list_or_per_window_start_change_nothing(N, geo, filt, monkeypatch):
    """Round 6: kr_wide_run's locate pass lists the member windows (16 bytes per member) when they are few, instead of a group
    number per window start streamed twice more (k_wide_locate / k_wide_count_list / k_wide_emit_list).  The list
    (default), the dense form (KR_WIDE_LOCLIST=0), a list too short for the members (KR_WIDE_LOCCAP: the run falls back
    to the dense form by itself) and the dense form without the per-genome key cache (KR_WIDE_CACHE=0: a buffer of
    its own) give the same hits; KR_WIDE_LOCATED says which form ran."""
    L, D, R = geo
    fam = _family(94, 4, 150_000)
    rng = np.random.default_rng(11)
    texts = []
    for _, _, t in fam:
        t = t.copy()
        t[60_000:60_400] = t[20_000:20_400]                          # a repeat: groups with several members per genome
        for p in rng.integers(0, len(t) - 50, size=10):
            t[p:p + int(rng.integers(1, 20))] = ord("N")
        texts.append(t)
    flags = [f for _, f, _ in fam]
    ids = list(range(len(fam)))

    def run(env):
        for k2 in ("KR_WIDE_LOCLIST", "KR_WIDE_LOCCAP", "KR_WIDE_CACHE"):
            monkeypatch.delenv(k2, raising=False)
        for k2, v in env.items():
            monkeypatch.setenv(k2, v)
        with N.Engine() as e:
            e.set_params_wide(L, D, R, max_bases=max(len(t) for t in texts))
            for i, t in enumerate(texts):
                e.upload(i, t)
            n = e.wide_run(ids, flags, apply_filter=filt)
            hits = e.wide_fetch(N.WIDE_HITS)
            hits = hits[np.lexsort((hits["strand"], hits["pos"], hits["genome"], hits["cand"]))]
            n2 = e.wide_run(ids, flags, apply_filter=filt)             # (again in the same context: the buffers are there)
            assert n2 == n
            return n, hits.tobytes(), int(e.wide_fetch(N.WIDE_LOCATED)[0]), int(e.wide_fetch(N.WIDE_NGROUPS)[0])

    n1, h1, loc1, g1 = run({})
    n0, h0, loc0, g0 = run({"KR_WIDE_LOCLIST": "0"})
    n2, h2, loc2, g2 = run({"KR_WIDE_LOCCAP": "7"})
    n3, h3, loc3, g3 = run({"KR_WIDE_LOCLIST": "0", "KR_WIDE_CACHE": "0"})
    n4, h4, loc4, g4 = run({"KR_WIDE_CACHE": "0"})
    assert n1 > 0 and loc1 >= n1 // 2 and loc0 == 0 and loc2 == 0 and loc3 == 0 and loc4 == loc1
    assert n1 == n0 == n2 == n3 == n4 and g1 == g0 == g2 == g3 == g4
    assert h1 == h0 == h2 == h3 == h4


@pytest.mark.parametrize("geo,filt,mu", [((32, 20, 32), True, 0.03), ((32, 60, 32), False, 0.03), ((20, 17, 20), True, 0.04),
                                         ((32, 20, 32), True, 0.002)])
def test_wide_composite_keys_as_lists_change_nothing(N, geo, filt, mu, monkeypatch):
    """Round 6: where few window starts have a composite key (both flanks in the dictionaries) kr_wide_run keeps the keys of a
    genome as a LIST -- 16 bytes per key, a segment per workgroup of k_hist8w; k_scatter1l and k_wide_locate_list walk it --
    instead of 16 bytes per window start written once and streamed twice (Geom.wlcnt).  Lists (forced through
    KR_WIDE_KEYFRAC, the share of window starts a segment is sized for: these genomes are too small and too close for
    the default rule), the dense form (KR_WIDE_KEYLIST=0), segments too short (the run falls back to the dense form by
    itself, from the genome that overflowed on), lists with the members per window start (KR_WIDE_LOCLIST=0: the keys
    are made again), lists under key-space slices and in batches: the same hits; KR_WIDE_KEYS_LISTED says what ran."""
    L, D, R = geo
    from krisp_amd import synth
    fam = synth.family(95, 2, 2, 300_000, records=4, mu=mu, snp_every=150)
    texts = []
    for _, _, t in fam:
        t = t.copy()
        t[70_000:70_300] = t[30_000:30_300]
        texts.append(t)
    flags = [f for _, f, _ in fam]
    ids = list(range(len(fam)))
    envs = ("KR_WIDE_KEYLIST", "KR_WIDE_KEYFRAC", "KR_WIDE_LOCLIST", "KR_SLICE_BASES", "KR_WIDE_BATCH", "KR_WIDE_LOCCAP")

    def run(env):
        for k2 in envs:
            monkeypatch.delenv(k2, raising=False)
        for k2, v in env.items():
            monkeypatch.setenv(k2, v)
        with N.Engine() as e:
            e.set_params_wide(L, D, R, max_bases=max(len(t) for t in texts))
            for i, t in enumerate(texts):
                e.upload(i, t)
            n = e.wide_run(ids, flags, apply_filter=filt)
            hits = e.wide_fetch(N.WIDE_HITS)
            hits = hits[np.lexsort((hits["strand"], hits["pos"], hits["genome"], hits["cand"]))]
            listed = int(e.wide_fetch(N.WIDE_KEYS_LISTED)[0])
            assert e.wide_run(ids, flags, apply_filter=filt) == n          # (again in the same context)
            assert int(e.wide_fetch(N.WIDE_KEYS_LISTED)[0]) == listed
            return n, hits.tobytes(), listed, int(e.wide_fetch(N.WIDE_NGROUPS)[0])

    dense = run({"KR_WIDE_KEYLIST": "0"})
    assert dense[0] > 0 and dense[2] == 0
    sparse = mu >= 0.03
    lists = run({"KR_WIDE_KEYFRAC": "0.12"})
    assert lists[2] > 0 or not sparse                  # (close relatives: many starts have a key, segments may overflow)
    forms = [lists,
             run({"KR_WIDE_KEYFRAC": "0.0000001"}),                           # every segment 256 entries: too short, dense after all
             run({"KR_WIDE_KEYFRAC": "0.12", "KR_WIDE_LOCLIST": "0"}),
             run({"KR_WIDE_KEYFRAC": "0.12", "KR_WIDE_LOCCAP": "9"}),
             run({"KR_WIDE_KEYFRAC": "0.12", "KR_SLICE_BASES": "1"}),
             run({"KR_WIDE_KEYFRAC": "0.12", "KR_WIDE_BATCH": "2"}),
             run({})]
    for f in forms:
        assert f[0] == dense[0] and f[3] == dense[3] and f[1] == dense[1]
    if sparse:
        assert forms[2][2] > 0 and forms[4][2] > 0 and forms[5][2] > 0


def _bgzf(data, block=65280, level=6, strategy=0, eof=True):
    """bytes -> a BGZF file as bgzip writes it: gzip members of `block` bytes of text with the BC extra field"""
    import struct
    import zlib
    out = []
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof else [])
    for ch in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        cd = co.compress(ch) + co.flush()
        bsize = len(cd) + 26
        assert bsize <= 65536
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + cd
                   + struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch)))
    return b"".join(out)


def _fasta_text(seed, n, alphabet=b"ACGT", width=80, nrec=3, lower=True, nruns=True):
    rng = np.random.default_rng(seed)
    a = np.frombuffer(alphabet, dtype=np.uint8)
    recs = []
    for r in range(nrec):
        s = a[rng.integers(0, len(a), size=n // nrec)].copy()
        if nruns:
            p = int(rng.integers(0, max(1, len(s) - 500)))
            s[p:p + 400] = ord("N")
        if lower:
            p = int(rng.integers(0, max(1, len(s) - 300)))
            s[p:p + 200] |= 0x20
        body = b"\n".join(bytes(s[i:i + width]) for i in range(0, len(s), width))
        recs.append(b">rec%d some words\n" % r + body + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("what", ["fasta_l6", "fasta_l1", "fasta_l9", "stored", "fixed", "tiny_blocks", "random_bytes", "runs", "iupac",
                                  "empty", "one_byte", "no_eof", "rna"])
def test_bgzf_inflated_on_the_device_equals_the_host_path(N, what):
    """Round 6: kr_genome_upload_bgzf -- every member of a BGZF file a lane of k_bgzf_inflate (stored,
    fixed and dynamic blocks; copies over the lane's own output; CRC-32 and ISIZE checked on the device), then the
    device's reader -- against kr_genome_upload_text over the text Python's gzip module gives (the reference's reader,
    kstream.py:458-479): the same bases, records, special characters and alphabet.  Compression levels 0 / 1 / 6 / 9, fixed
    Huffman codes only, members of 100 bytes, bytes of every value (the codes of a dynamic block at their widest; the
    parser then sees one long line of odd characters), runs (distance 1, length 258), IUPAC letters, an empty file, one
    byte, no EOF member, RNA."""
    import gzip
    import zlib
    kw = {}
    if what.startswith("fasta"):
        text, kw = _fasta_text(1, 400_000), dict(level=int(what[-1]))
    elif what == "stored":
        text, kw = _fasta_text(2, 200_000), dict(level=0, block=60000)
    elif what == "fixed":
        text, kw = _fasta_text(3, 200_000), dict(strategy=zlib.Z_FIXED)
    elif what == "tiny_blocks":
        text, kw = _fasta_text(4, 30_000), dict(block=100)
    elif what == "random_bytes":
        text = b">r\n" + bytes(np.random.default_rng(5).integers(0, 256, size=300_000, dtype=np.uint8))
        kw = dict(block=40000)
    elif what == "runs":
        text = b">r\n" + b"A" * 200_000 + b"\n" + b"ACGT" * 30_000 + b"\n>s\n" + b"N" * 70_000 + b"\n"
    elif what == "iupac":
        text = _fasta_text(6, 100_000, alphabet=b"ACGTACGTACGTRYKMSWN")
    elif what == "empty":
        text = b""
    elif what == "one_byte":
        text = b"A"
    elif what == "no_eof":
        text, kw = _fasta_text(7, 150_000), dict(eof=False)
    else:
        text = _fasta_text(8, 150_000).replace(b"T", b"U").replace(b"t", b"u")
    raw = _bgzf(text, **kw)
    assert gzip.decompress(raw) == text
    with N.Engine() as e:
        e.set_params(25, 1, 2, max_bases=max(len(text), 64))
        got = e.upload_bgzf(0, np.frombuffer(raw, dtype=np.uint8))
        assert got is not None, e.last_bgzf
        n, nrec, nspecial, rna, fasta, members, us = got
        bases = e.fetch_bases(0, n).copy()
        n2, nrec2, nspecial2, rna2, fasta2 = e.upload_text(1, np.frombuffer(text, dtype=np.uint8), False)
        assert (n, nrec, nspecial, rna, fasta) == (n2, nrec2, nspecial2, rna2, fasta2)
        assert np.array_equal(bases, e.fetch_bases(1, n2))
        assert members == len(raw) and members == 0 or members >= 1
        if n:
            e.sort(0)
            e.sort(1)
            assert e.count(0) == e.count(1)


@pytest.mark.parametrize("seed", range(int(os.environ.get("KR_BGZF_SEEDS", "24"))))
def test_bgzf_device_inflate_on_random_streams(N, seed):
    """every compression level, zlib's strategies (default, filtered, Huffman only: no copies; RLE: copies at distance one --
    the overlapping, byte-wise copy; fixed codes), member sizes from 64 bytes to the format's 65280, texts from sequence
    files to bytes of every value with repeats at all distances: what the device inflates is what Python's gzip inflates"""
    import gzip
    import zlib
    rng = np.random.default_rng(4400 + seed)
    kind = seed % 4
    n = int(rng.integers(1_000, 400_000))
    if kind == 0:
        text = _fasta_text(seed, n, nrec=int(rng.integers(1, 6)), width=int(rng.choice([60, 70, 80, 100])))
    elif kind == 1:                 # a few distinct lines repeated: long copies at many distances
        lines = [bytes(rng.integers(65, 85, size=int(rng.integers(5, 200)), dtype=np.uint8)) for _ in range(int(rng.integers(2, 30)))]
        text = b"\n".join(lines[int(i)] for i in rng.integers(0, len(lines), size=max(1, n // 60)))
    elif kind == 2:
        text = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    else:                           # runs of one letter of all lengths
        text = b"".join(bytes([int(rng.integers(65, 70))]) * int(rng.integers(1, 600)) for _ in range(max(1, n // 300)))
    level = int(rng.integers(0, 10))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    block = int(rng.choice([64, 500, 4096, 30000, 65280]))
    if level == 0 or kind == 2:
        block = min(block, 60000)   # (stored or incompressible data: a member must stay below 64 KB)
    raw = _bgzf(text, block=block, level=level, strategy=strategy, eof=bool(seed % 3))
    assert gzip.decompress(raw) == text
    with N.Engine() as e:
        e.set_params(25, 1, 2, max_bases=max(len(text), 64))
        got = e.upload_bgzf(0, np.frombuffer(raw, dtype=np.uint8))
        assert got is not None, (e.last_bgzf, level, strategy, block, kind)
        want = e.upload_text(1, np.frombuffer(text, dtype=np.uint8), False)
        assert got[:5] == want
        assert np.array_equal(e.fetch_bases(0, got[0]), e.fetch_bases(1, want[0]))


def test_bgzf_members_that_do_not_inflate_to_their_trailers_go_to_the_host(N):
    """A damaged BGZF file: the device says so for every kind of damage -- a flipped byte in a member's data (an invalid
    code, a distance before the member's start, a wrong length, a CRC mismatch at the latest), in its CRC, in its ISIZE,
    a member cut short, a header that is not BGZF, plain gzip -- and uploads nothing; 300 single-byte flips anywhere in
    the file never do anything else than that or (a flip in a header's time stamp or OS byte) inflate to the same text."""
    import gzip
    text = _fasta_text(11, 300_000)
    raw = bytearray(_bgzf(text, block=30000))
    plain = np.frombuffer(text, dtype=np.uint8)
    rng = np.random.default_rng(12)
    with N.Engine() as e:
        e.set_params(25, 1, 2, max_bases=len(text))
        ok = e.upload_bgzf(0, np.frombuffer(bytes(raw), dtype=np.uint8))
        assert ok is not None
        want = e.fetch_bases(0, ok[0]).copy()

        def attempt(buf):
            got = e.upload_bgzf(0, np.frombuffer(bytes(buf), dtype=np.uint8))
            if got is None:
                return None
            return e.fetch_bases(0, got[0]).copy()
        first_len = int.from_bytes(raw[16:18], "little") + 1
        for at in (first_len - 8, first_len - 4, 40, first_len + 18 + 5):        # CRC, ISIZE, data of member 0, data of member 1
            bad = bytearray(raw)
            bad[at] ^= 0x5A
            assert attempt(bad) is None, at
            assert "member" in e.last_bgzf[3]
        assert attempt(raw[:len(raw) - 40]) is None                               # cut short
        assert attempt(b"\x1f\x8b\x08\x00" + bytes(raw[4:])) is None             # no extra field: not BGZF
        assert attempt(gzip.compress(text)) is None                               # plain gzip
        same = 0
        for _ in range(300):
            bad = bytearray(raw)
            at = int(rng.integers(0, len(bad)))
            bad[at] ^= int(rng.integers(1, 256))
            got = attempt(bad)
            if got is not None:
                assert np.array_equal(got, want), at
                same += 1
        assert same < 60
        # the host path still reads the intact file: the same bases
        n2 = e.upload_text(1, plain, False)[0]
        assert np.array_equal(e.fetch_bases(1, n2), want)


def test_placement_tries_change_nothing_but_time(N, K):
    """KR_OPT_PLACE_TRIES: the pass-1 output buffer is chosen among several allocations (each timed under
    pass 1's write pattern); the result is the same as with a plain allocation"""
    fam = _family(55, 4, 40_000_000)            # (buffers of >= 256 MB: below that nothing is tried)
    flags = [f for _, f, _ in fam]
    ids = list(range(len(fam)))
    got = []
    for tries in (1, 3):
        with N.Engine() as e:
            e.set_option(N.OPT_PLACE_TRIES, tries)
            e.set_params(25, 1, 2, max_bases=max(len(t) for _, _, t in fam))
            for i, (_, _, t) in enumerate(fam):
                e.add(i, t)
            assert all(e.inversions(i) == 0 for i in ids)
            n = e.intersect(ids, flags, apply_filter=True)
            got.append((n, e.cands().tobytes(), e.collect(ids).tobytes()))
    assert got[0][0] > 0 and got[0] == got[1]


def test_stage_timers_and_medium_size(N, K):
    fam = _family(77, 4, 1_000_000)
    flags = [f for _, f, _ in fam]
    with N.Engine() as e:
        e.set_params(25, 1, 2, max_bases=max(len(t) for _, _, t in fam))
        e.stage_enable(True)
        for i, (_, _, t) in enumerate(fam):
            e.upload(i, t)
        e.timer_begin()
        for i in range(4):
            e.sort(i)
        ncand = e.intersect([0, 1, 2, 3], flags, apply_filter=True)
        ms = e.timer_end_ms()
        st = e.stage_times()
        total = sum(e.count(i) for i in range(4))
        print(f"\n4 x 1 Mbp: {total} k-mers in {ms:.3f} ms = {total / ms / 1e6:.2f} G k-mers/s; "
              f"cands {ncand}; stages {st}; info {e.debug_info()}")
        want_keys = [K.sorted_keys(t.tobytes(), 25, 1, 2) for _, _, t in fam]
        want = K.intersect(want_keys, flags, 25, 1, 2, apply_filter=True)
        assert ncand == len(want)
        assert np.array_equal(e.cands()["prefix"], want["prefix"])
        for i in range(4):
            assert np.array_equal(e.keys(i), want_keys[i])


def test_exchange_is_all_or_none_when_a_rank_runs_out_of_memory(N, K, tmp_path):
    """kr_cands_reduce / kr_cands_bcast / kr_records_gather: a rank whose allocation fails between a count and
    the list it announces (here: rank 0's HBM budget is too small for the list rank 1 sends) answers that count
    with "no", keeps its place in the tree and returns its error; no rank is left inside a send or a receive
    (over RCCL those have no timeout).  Three ranks as threads over the file transport."""
    import threading
    import time
    L, D, R = 12, 1, 3
    fam = _family(21, 6, 60_000, mu=0.002)
    world = 3
    res, errs = [None] * world, [None] * world

    def work(rank):
        try:
            with N.Engine() as e:
                e.comm_init_dir(rank, world, str(tmp_path / "comm"))
                e.set_params(L, D, R, max_bases=60_100)
                ids = [g for g in range(len(fam)) if g % world == rank]
                for g in ids:
                    e.upload(g, fam[g][2])
                    e.sort(g)
                e.intersect(ids, [fam[g][1] for g in ids], apply_filter=False)
                if rank == 0:
                    # rank 0 holds its genomes and scratch, but from now on not one candidate list more
                    assert e.lib.kr_debug_budget_set(e.ctx, 1) == 0 and e.lib.kr_debug_budget_left(e.ctx) == 1
                try:
                    e.cands_reduce(apply_filter=False)
                    res[rank] = "reduced"
                    e.cands_bcast()
                    res[rank] = "bcast"
                except N.KrispHipError as ex:
                    res[rank] = f"error: {ex}"
        except BaseException as ex:  # noqa: BLE001
            errs[rank] = ex

    ts = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    t0 = time.time()
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not any(t.is_alive() for t in ts), f"a rank hangs: {res}"
    assert errs == [None] * world, errs
    assert res[0].startswith("error:") and "budget" in res[0], res
    assert all(r.startswith("error:") for r in res), res            # (ranks 1 and 2: "rank 0 failed", from its answers)
    assert time.time() - t0 < 50


def test_fine_histogram_counts_beyond_sixteen_bits(N, K):
    """k_hist16 counts in 16 bits unless the top-byte counts of k_hist8 say that a bin of the workgroup's range could
    pass 65535; such a range counts in 32 bits, and what exceeds 16 bits leaves through the exception list.  A
    9 Mbp genome with a 150 kb poly-A run and a 90 kb dinucleotide run: the sort equals the oracle's."""
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    body = acgt[rng.integers(0, 4, size=9_000_000)]
    text = np.concatenate([body[:3_000_000], np.full(150_000, ord("A"), dtype=np.uint8), body[3_000_000:7_000_000],
                           np.tile(np.frombuffer(b"AC", dtype=np.uint8), 45_000), body[7_000_000:]]).astype(np.uint8)
    info = _check_sorted(N, K, text, 20, 1, 4)
    assert info["b"] > 8 and info["overflow_segments"] > 0


def test_device_reader_equals_the_host_parser(N, tmp_path):
    """kr_genome_upload_text (the reference reader on the device, csrc/k_text.inc) against kr_fasta_to_bases, the host
    parser pinned to the reference's reader by tests/test_host_glue.py: the reader vectors of the golden files, the
    test_data genomes, 600 adversarial random texts (headers, blank and indented lines, every newline flavour, '>'
    inside lines, RNA, non-FASTA inputs, first-line consumption, texts ending with and without a newline) and two
    shapes of large input (80-column lines; records of one long line) -- the same bytes, records, special-character
    count, RNA and FASTA verdicts."""
    import gzip
    import json
    import os
    import random
    golden = os.path.join(os.path.dirname(__file__), "golden")
    texts = []
    for c in json.load(open(os.path.join(golden, "kstream_cases.json"))):
        if c["file_text"] is not None:
            texts.append((c["file_text"].encode(), not c["fname"].endswith(".gz")))
    for fn in sorted(os.listdir(os.path.join(golden, "c1"))):
        texts.append((gzip.open(os.path.join(golden, "c1", fn), "rb").read(), False))
    rng = random.Random(5)
    for i in range(600):
        pieces = []
        for _ in range(rng.randint(0, 12)):
            kind = rng.random()
            if kind < 0.25:
                pieces.append(">" + "".join(rng.choice("abc >x") for _ in range(rng.randint(0, 6))))
            elif kind < 0.35:
                pieces.append(rng.choice(["", " ", "\t"]))
            else:
                pieces.append(rng.choice(["", " ", "  "]) +
                              "".join(rng.choice("ACGTacgtNnUuRY>x ") for _ in range(rng.randint(0, 30))) +
                              rng.choice(["", " ", "\t "]))
        nl = rng.choice(["\n", "\r\n", "\r", "\n"])
        text = (nl.join(pieces) + rng.choice(["", nl, nl + nl])).encode()
        texts.append((text, i % 2 == 0))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    nprng = np.random.default_rng(3)
    body = acgt[nprng.integers(0, 4, size=3_000_000)].tobytes()
    wrapped = b">chr1 some text\n" + b"\n".join(body[i:i + 80] for i in range(0, 1_500_000, 80)) + b"\n>chr2\n" + \
        b"\n".join(body[i:i + 60] for i in range(1_500_000, 3_000_000, 60)) + b"\n"
    texts.append((wrapped, False))
    texts.append((b">a\n" + body[:2_000_000] + b"\n>b\r\n" + body[2_000_000:].replace(b"T", b"U") + b"\n", True))
    with N.Engine() as e:
        e.set_params(25, 1, 2, max_bases=3_100_000)
        for i, (text, universal) in enumerate(texts):
            for one_shot in (True, False):
                want, wrec, wspecial, wrna, wfasta = N.fasta_to_bases(text, universal, one_shot)
                n, rec, special, rna, fa = e.upload_text(0, np.frombuffer(text, dtype=np.uint8), universal, one_shot)
                got = e.fetch_bases(0, n)
                assert n == len(want) and got.tobytes() == want.tobytes(), (i, one_shot, text[:200])
                assert (rec, special, rna, fa) == (wrec, wspecial, wrna, wfasta), (i, one_shot, text[:200])
        # ... and what was parsed there sorts like what was parsed here
        text, universal = texts[-2]
        want = N.fasta_to_bases(text, universal, True)[0]
        n = e.upload_text(1, np.frombuffer(text, dtype=np.uint8), universal)[0]
        e.sort(1)
        e.upload(2, want)
        e.sort(2)
        assert np.array_equal(e.keys(1), e.keys(2))


def test_sort_lanes_automatic_and_fixed(N, K):
    """KR_OPT_LANES: a context starts with one sort lane and takes three once it has sorted 16 genomes (consecutive
    sorts then overlap on the device, each lane with its own scratch); a fixed number on request, changed at any
    time.  The keys of 40 genomes sorted back to back -- nothing waits in between -- and the intersections over them
    equal the oracle's whatever lane took which genome."""
    from krisp_amd import synth
    L, D, R = 20, 1, 5
    fam = synth.family(77, 20, 20, 90_000, records=2, mu=0.004, snp_every=900)
    texts = [t for _, _, t in fam]
    flags = [f for _, f, _ in fam]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for t in texts]
    with N.Engine() as e:
        e.set_option(N.OPT_LANES, 0)
        e.set_option(N.OPT_SLICE_BASES, 0)             # (key-space slices keep one lane)
        e.set_params(L, D, R, max_bases=max(len(t) for t in texts))
        for i, t in enumerate(texts):
            e.upload(i, t)
        for i in range(len(texts)):
            e.sort(i)
            assert e.debug_isect()["sort_lanes"] == (1 if i < 15 else 3), i
        ids = list(range(8, 40))
        want = K.intersect([want_keys[i] for i in ids], [flags[i] for i in ids], L, D, R, apply_filter=False)
        assert e.intersect(ids, [flags[i] for i in ids], apply_filter=False) == len(want)
        assert np.array_equal(e.cands()["prefix"], want["prefix"])
        for i in range(len(texts)):
            assert np.array_equal(e.keys(i), want_keys[i]), i
        for lanes in (5, 2, 1, 8):
            e.set_option(N.OPT_LANES, lanes)
            for i in range(12):
                e.sort(i)
            assert e.debug_isect()["sort_lanes"] == lanes
            for i in (0, 5, 11):
                assert np.array_equal(e.keys(i), want_keys[i]), (lanes, i)


def test_automatic_lanes_stay_at_one_when_their_scratch_does_not_fit(N, K):
    """the 16th sort of a context would bring two more lanes; with no room left for their scratch (HBM budget) the
    context keeps sorting on one lane instead of failing"""
    from krisp_amd import synth
    L, D, R = 20, 1, 5
    fam = synth.family(78, 10, 10, 60_000, records=2, mu=0.004, snp_every=900)
    texts = [t for _, _, t in fam]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for t in texts]
    with N.Engine() as e:
        e.set_option(N.OPT_LANES, 0)
        e.set_option(N.OPT_SLICE_BASES, 0)
        e.set_params(L, D, R, max_bases=max(len(t) for t in texts))
        for i, t in enumerate(texts):
            e.upload(i, t)
        e.sort(0)
        assert e.count(0) == len(want_keys[0])
        assert e.lib.kr_debug_budget_set(e.ctx, 1 << 20) == 0          # (a lane's scratch is tens of megabytes)
        for i in range(len(texts)):
            e.sort(i)
        assert e.debug_isect()["sort_lanes"] == 1
        for i in range(len(texts)):
            assert np.array_equal(e.keys(i), want_keys[i]), i


def test_intersect_right_behind_sorts_that_need_the_merge_fallback(N, K):
    """kr_intersect does not wait for the sorts: it looks at what they left open after its own synchronisation.
    Genomes with satellites (buckets too large for the LDS sort) are sorted and intersected WITHOUT asking for their
    counts in between: the oversized buckets are repaired then and the intersection is redone -- same candidates and
    records as the oracle's."""
    rng = np.random.default_rng(23)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = rng.integers(0, 4, size=300_000)
    unit = rng.integers(0, 4, size=29)
    texts = []
    for g in range(3):
        a = base.copy()
        m = rng.random(len(a)) < 0.004
        a[m] = (a[m] + 1) % 4
        texts.append(np.concatenate([acgt[a[:100_000]], acgt[np.tile(unit, 3000 + 10 * g)], acgt[a[100_000:]]]).astype(np.uint8))
    L, D, R = 14, 1, 5
    flags = [1, 1, 0]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for t in texts]
    with N.Engine() as e:
        e.set_params(L, D, R, max_bases=max(len(t) for t in texts))
        for i, t in enumerate(texts):
            e.upload(i, t)
            e.sort(i)                                   # (no count, no keys: nothing makes the host look at the sort)
        for filt in (False, True):
            want = K.intersect(want_keys, flags, L, D, R, apply_filter=filt)
            assert e.intersect([0, 1, 2], flags, apply_filter=filt) == len(want)
            got = e.cands()
            for f in ("prefix", "in_mask", "out_mask"):
                assert np.array_equal(got[f], want[f]), (f, filt)
        assert e.debug_info()["overflow_segments"] > 0
        for i in range(3):
            assert np.array_equal(e.keys(i), want_keys[i])


@pytest.mark.parametrize("lanes", [2, 3, 4])
@pytest.mark.parametrize("n,perm", [(3, [0, 2, 1]), (4, [0, 1, 2, 3]), (5, [0, 2, 1, 3, 4]), (7, [3, 0, 1, 4, 2, 5, 6]),
                                    (8, [0, 1, 2, 3, 4, 5, 6, 7]), (4, [2, 3, 0, 1]), (6, [0, 1, 2, 3, 4, 5])])
def test_late_genomes_probe_the_candidates_of_the_early_ones(N, K, lanes, n, perm, monkeypatch):
    """round 4: with several sort lanes the genomes of the last round stay out of the big intersection -- it runs over
    the early ones beside the late sorts, filtered on partial masks -- and only look the surviving candidates up
    (k_cands_probe): candidates, masks and records equal the packed oracle's, whether the early group can prune (it holds
    both kinds of genome) or not (then every genome takes the big intersection), for every lane count, with the sorts
    still in flight when kr_intersect is called and again over finished sorts; KR_ISECT_SPLIT=0 gives the same"""
    L, D, R = 25, 1, 2
    fam = _family(60 + n, n, 300_000)
    fam = [fam[i] for i in perm]                    # (the family's first n // 2 genomes are the ingroup: `perm` places them)
    flags = [bool(f) for _, f, _ in fam]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for _, _, t in fam]
    want = K.intersect(want_keys, flags, L, D, R, apply_filter=True)
    wrec = np.sort(K.collect(want_keys, want, L, D, R), order=["key", "genome"])
    assert len(want) > 10
    for split in ("1", "0"):
        monkeypatch.setenv("KR_ISECT_SPLIT", split)
        with N.Engine() as e:
            e.set_option(N.OPT_LANES, lanes)
            e.set_option(N.OPT_SLICE_BASES, 0)          # (the probe is for one sort unit: pinned against KR_SLICE_BASES in the environment)
            e.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
            for i, (_, _, t) in enumerate(fam):
                e.upload(i, t)
            for rep in range(2):
                if rep == 0:
                    for i in range(n):
                        e.sort(i)                       # asynchronous: the intersection is called over sorts in flight
                assert e.intersect(list(range(n)), flags, apply_filter=True) == len(want)
                got = e.cands()
                for f in ("prefix", "in_mask", "out_mask"):
                    assert np.array_equal(got[f], want[f]), (f, split, rep)
                recs = e.collect(list(range(n)))
                assert np.array_equal(np.sort(recs, order=["key", "genome"]), wrec)
            late = (n - 1) % lanes + 1
            early = flags[:n - late]
            can_split = split == "1" and n - late >= 2 and any(early) and not all(early)
            info = e.debug_isect()
            assert info["splits"] == (2 if can_split else 0), (info, lanes, n)
            if can_split:
                assert info["probed"] >= len(want)
            # without the filter nothing is pruned: every genome takes the big intersection
            wnf = K.intersect(want_keys, flags, L, D, R, apply_filter=False)
            assert e.intersect(list(range(n)), flags, apply_filter=False) == len(wnf)
            assert np.array_equal(e.cands()["prefix"], wnf["prefix"]) and e.debug_isect()["splits"] == info["splits"]


def test_collect_in_chunks_of_candidates(N, K, monkeypatch):
    """kr_collect takes the candidates in chunks (a thread, a count and a place per (candidate, genome) pair: 2^28 pairs per
    chunk, so that neither the 32-bit pair index nor the scratch bounds a call -- ADVICE r4).  KR_COLLECT_PAIRS makes the
    chunks small: 1, 7 and 1000 pairs per chunk give the records of one chunk, in the same order, also when the record
    buffer has to grow in the middle, with and without the filter, with key-space slices"""
    fam = _family(31, 5, 120_000, mu=0.004)
    flags = [f for _, f, _ in fam]
    ids = list(range(len(fam)))
    for slices in (None, 1):
        got = {}
        for pairs in (None, "1000", "7", "1"):
            if pairs is None:
                monkeypatch.delenv("KR_COLLECT_PAIRS", raising=False)
            else:
                monkeypatch.setenv("KR_COLLECT_PAIRS", pairs)
            with N.Engine() as e:
                if slices is not None:
                    e.set_option(N.OPT_SLICE_BASES, slices)
                e.set_params(10, 2, 4, max_bases=max(len(t) for _, _, t in fam))
                for i, (_, _, t) in enumerate(fam):
                    e.add(i, t)
                res = []
                for filt in (True, False):
                    n = e.intersect(ids, flags, apply_filter=filt)
                    assert n > (5 if filt else 1000)
                    if pairs in ("1", "7") and not filt:
                        e.load_cands(e.cands()[:300].copy())         # (a chunk per candidate: keep the launch count sane)
                    res.append(e.collect(ids).tobytes())
                got[pairs] = res
        assert got[None] == got["1000"]
        assert got["7"] == got["1"] and got["7"][0] == got[None][0]


@pytest.mark.parametrize("slices", [None, 1, 2])
def test_probe_of_a_candidate_list_equals_the_intersection(N, K, slices):
    """kr_cands_probe (round 5): the candidates of the first m genomes looked up in the others -- presence in every genome,
    their diagnostic bases into the masks of their side, the filter -- are the candidates of the intersection over all of
    them, with and without the filter, for every split of 7 genomes (the later ones of ONE side only, too), with key-space
    slices; against the packed-key oracle."""
    L, D, R = 11, 2, 5
    fam = _family(77, 7, 90_000, mu=0.003)
    flags = [f for _, f, _ in fam]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R) for _, _, t in fam]
    ids = list(range(len(fam)))
    with N.Engine() as e:
        if slices is not None:
            e.set_option(N.OPT_SLICE_BASES, slices)
        e.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            e.add(i, t)
        for filt in (True, False):
            want = K.intersect(want_keys, flags, L, D, R, apply_filter=filt)
            assert len(want) > (3 if filt else 500)
            for order in (ids, [0, 6, 1, 2, 3, 4, 5], [3, 4, 5, 6, 0, 1, 2]):         # (the last: all outgroup first, ingroup by probe)
                for m in (1, 2, 4, 6):
                    head, tail = order[:m], order[m:]
                    e.intersect(head, [flags[g] for g in head], apply_filter=filt)
                    n = e.probe_cands(tail, [flags[g] for g in tail], apply_filter=filt)
                    got = e.cands()
                    assert n == len(want) == len(got), (filt, order, m)
                    for f in ("prefix", "in_mask", "out_mask"):
                        assert np.array_equal(got[f], want[f]), (filt, order, m, f)
        # a candidate list that came from somewhere else (kr_cands_load), genomes in two calls
        want = K.intersect(want_keys, flags, L, D, R, apply_filter=False)
        loaded = want.copy()
        loaded["in_mask"] = 0
        loaded["out_mask"] = 0
        e.load_cands(loaded)
        e.probe_cands(ids[:3], flags[:3], apply_filter=False)
        e.probe_cands(ids[3:], flags[3:], apply_filter=False)
        got = e.cands()
        for f in ("prefix", "in_mask", "out_mask"):
            assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_of_empty_and_uneven_candidate_lists(N, K, world, tmp_path):
    """the exchange's message is never shorter than its header: every rank without a single candidate (the agreed message
    size is 0 entries), one rank with none among ranks with some, lists of very different lengths -- ranks as threads
    over the file transport; what comes back is the intersection (empty, empty, the common part) with the masks OR-ed,
    on every rank after the broadcast, and the record gather of nothing is nothing"""
    import threading
    L, D, R = 9, 1, 3
    rng = np.random.default_rng(3)
    pool = np.sort(rng.choice(1 << 24, size=5000, replace=False).astype(np.uint64)) << np.uint64(64 - 2 * (L + R))

    def lists(case, rank):
        c = np.zeros(0, dtype=N.CAND)
        if case == "all_empty":
            return c
        if case == "one_empty" and rank == world - 1:
            return c
        take = pool[:: rank + 1] if case == "uneven" else pool[:200]
        c = np.zeros(len(take), dtype=N.CAND)
        c["prefix"] = take
        c["in_mask"] = 1 << rank
        c["out_mask"] = 16 << rank
        return c
    for case in ("all_empty", "one_empty", "uneven"):
        res, errs = [None] * world, [None] * world

        def work(rank):
            try:
                with N.Engine() as e:
                    e.comm_init_dir(rank, world, str(tmp_path / f"comm_{case}"))
                    e.set_params(L, D, R, max_bases=1000)
                    e.load_cands(lists(case, rank))
                    n = e.cands_reduce(apply_filter=False)
                    nb = e.cands_bcast()
                    got = e.cands().copy()
                    e.lib.kr_collect(e.ctx, None, 0)
                    tot = e.records_gather()
                    res[rank] = (n, nb, got, tot)
                    e.comm_barrier()
            except BaseException as ex:  # noqa: BLE001
                errs[rank] = ex
        ts = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
        assert not any(t.is_alive() for t in ts), (case, res)
        assert errs == [None] * world, (case, errs)
        want = pool[:0] if case != "uneven" else pool[:: int(np.lcm.reduce(np.arange(1, world + 1)))]
        for r in range(world):
            n, nb, got, tot = res[r]
            assert nb == len(want) and tot == 0 and (r != 0 or n == len(want)), (case, r, n, nb)
            assert np.array_equal(got["prefix"], want)
            if len(want):
                assert np.all(got["in_mask"] == (1 << world) - 1) and np.all(got["out_mask"] == 16 * ((1 << world) - 1))
