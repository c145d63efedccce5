"""Full-size runs checked through size-independent properties (no oracle reaches these
sizes in seconds): record count = 2 x valid windows, sortedness (adjacent inversions
counted on the device), idempotence of sort + intersect, every candidate re-found by the
collect in every genome, agreement between slicing configurations.

  * BASELINE configs[1] (4 x 50 Mbp, 25/1/2), configs[2] (8 x 500 Mbp, 32/60/32: the wide path) and
    configs[4] (2 x 3 Gbp, k = 31 as 28/1/2; 64 key-space slices, ~150 GB of HBM) all run in the
    regular `-m gpu` suite (KR_SKIP_BIG=1 leaves the two large ones out: ~2 minutes of host-side
    genome generation each).
  * BASELINE configs[3] (32 x 100 Mbp sharded over 8 GPUs): its whole genome set on ONE GPU through the same
    properties, and the same family sharded 4 per rank over WORLD 8 -- eight contexts sharing the test GPU, the
    exchange over the file transport -- through the very function bench.py --gpus 8 times
    (distributed.sharded_step), the result equal to the one-GPU run bit for bit.  What this cannot show is RCCL
    itself between eight devices (no such node for the builder: DESIGN.md "Multi-GPU").
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _family(config, n_in, n_out, length):
    from krisp_amd import synth
    return synth.family(config, n_in, n_out, length, records=16, mu=0.01, snp_every=10000)


def _valid_windows(text, k):
    seps = np.flatnonzero(text == 10)
    bounds = np.concatenate([[-1], seps, [len(text)]])
    lens = np.diff(bounds) - 1
    return int(np.maximum(lens - k + 1, 0).sum())


def _check_records(recs, c1, flags, ids, L, D, R):
    """every candidate is present in every genome, and ingroup / outgroup diagnostic bases differ"""
    n1 = len(c1)
    pm = np.uint64((~0 << (64 - 2 * (L + R))) & 0xFFFFFFFFFFFFFFFF)
    pre = recs["key"] & pm
    for i in ids:
        assert np.array_equal(np.unique(pre[recs["genome"] == i]), c1["prefix"])
    if D == 1:
        dshift = np.uint64(62 - 2 * (L + R))
        base = ((recs["key"] >> dshift) & np.uint64(3)).astype(np.int64)
        is_in = np.array(flags)[recs["genome"]]
        idx = np.searchsorted(c1["prefix"], pre)
        in_sets = np.zeros(n1, dtype=np.int64)
        out_sets = np.zeros(n1, dtype=np.int64)
        np.bitwise_or.at(in_sets, idx[is_in], 1 << base[is_in])
        np.bitwise_or.at(out_sets, idx[~is_in], 1 << base[~is_in])
        assert np.all((in_sets & out_sets) == 0)
        assert np.array_equal(in_sets, c1["in_mask"].astype(np.int64))
        assert np.array_equal(out_sets, c1["out_mask"].astype(np.int64))


def _run(fam, L, D, R, slice_bases=None, light=False):
    """light: for results too large for host-side set arithmetic (configs[4]: 6.5e7 candidates) --
    no second sort, and the record checks on a seeded sample of 20000 candidates"""
    from krisp_amd import _native
    eng = _native.Engine()
    if slice_bases is not None:
        eng.set_option(_native.OPT_SLICE_BASES, slice_bases)
    eng.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
    ids = list(range(len(fam)))
    flags = [f for _, f, _ in fam]
    for i, (_, _, t) in enumerate(fam):
        eng.upload(i, t)
    for i in ids:
        eng.sort(i)
    n1 = eng.intersect(ids, flags, apply_filter=True)
    c1 = eng.cands().copy()
    for i, (_, _, t) in enumerate(fam):
        assert eng.count(i) == 2 * _valid_windows(t, L + D + R)
        assert eng.inversions(i) == 0
    assert np.all(np.diff(c1["prefix"].astype(np.uint64)) > 0) if n1 > 1 else True
    if light:
        rng = np.random.default_rng(99)
        pick = np.sort(rng.choice(n1, size=min(n1, 20000), replace=False))
        sample = c1[pick].copy()
        eng.load_cands(sample)
        _check_records(eng.collect(ids), sample, flags, ids, L, D, R)
    else:
        # idempotence: sorting again from the resident bases and intersecting again changes nothing
        for i in ids:
            eng.sort(i)
        assert eng.intersect(ids, flags, apply_filter=True) == n1
        c2 = eng.cands()
        assert np.array_equal(c1, c2)
        _check_records(eng.collect(ids), c1, flags, ids, L, D, R)
    info = eng.debug_info()
    eng.close()
    return c1, info


def test_c2_full_size_properties():
    fam = _family(2, 2, 2, 50_000_000)
    c1, info = _run(fam, 25, 1, 2)
    assert info["nslices"] == 4 ** int(os.environ.get("KR_SLICE_BASES", "0")) and info["overflow_segments"] == 0
    assert len(c1) > 1000
    # the same workload through 4 key-space slices gives the same candidates
    c4, info4 = _run(fam, 25, 1, 2, slice_bases=1)
    assert info4["nslices"] == 4
    assert np.array_equal(c1, c4)


def test_c2_full_size_equals_the_oracle():
    """BASELINE configs[1] at the size the bench quotes (4 x 50 Mbp of SURVEY 8(d)'s generator, 25/1/2), bit for bit against
    the packed-key oracle (oracle/kmer_oracle.c: ~6 s with a thread per genome): every genome's sorted keys, the candidates
    with their masks, the records (the properties above are no longer the only check at this size)"""
    from concurrent.futures import ThreadPoolExecutor
    from krisp_amd import _native
    from oracle import kmer_oracle as K
    K.build()
    L, D, R = 25, 1, 2
    fam = _family(2, 2, 2, 50_000_000)
    flags = [f for _, f, _ in fam]
    with ThreadPoolExecutor(max_workers=len(fam)) as pool:          # (ctypes releases the GIL)
        want_keys = list(pool.map(lambda g: K.sorted_keys(g[2].tobytes(), L, D, R), fam))
    want = K.intersect(want_keys, flags, L, D, R, apply_filter=True)
    wrec = np.sort(K.collect(want_keys, want, L, D, R), order=["key", "genome"])
    with _native.Engine() as eng:
        eng.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        ids = list(range(len(fam)))
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
        for i in ids:
            eng.sort(i)
        for i in ids:
            got = eng.keys(i)
            assert len(got) == len(want_keys[i]) and np.array_equal(got, want_keys[i]), f"sorted keys of genome {i} differ"
            del got
        n = eng.intersect(ids, flags, apply_filter=True)
        got = eng.cands()
        assert n == len(want) > 1000
        for f in ("prefix", "in_mask", "out_mask"):
            assert np.array_equal(got[f], want[f]), f
        recs = np.sort(eng.collect(ids), order=["key", "genome"])
        assert np.array_equal(recs, wrec)
        # and without the filter: every conserved (left,right) pair of the four genomes
        want_all = K.intersect(want_keys, flags, L, D, R, apply_filter=False)
        assert eng.intersect(ids, flags, apply_filter=False) == len(want_all)
        got = eng.cands()
        for f in ("prefix", "in_mask", "out_mask"):
            assert np.array_equal(got[f], want_all[f]), f


BIG = pytest.mark.skipif(os.environ.get("KR_SKIP_BIG") == "1", reason="KR_SKIP_BIG=1")


@BIG
def test_a_genome_beyond_two_to_the_32_bases():
    """VERDICT r5 'lift the input limits' (a): a sequence file of 2^32 bases or more used to be refused.  Positions, word
    indexes and key counts of the packed path are 64 bits (kr_set_params: genomes below 2^33 bases), so one genome of
    4.4e9 bases (16 records, 8.8e9 k-mer records, 64 key-space slices) is sorted like any other.  Checked where 32-bit
    arithmetic would break it -- beyond base 2^32: a second, small genome is a mutated copy of the 3 Mbp that start at base
    4.33e9 of the large one (k = 31: a chance match of a 30-base prefix anywhere else has probability 1e-2 over the
    run), and the candidates, masks and records of the pair must be the packed oracle's for (small genome, that region) --
    plus the count / sortedness properties over all 8.8e9 keys."""
    import time
    from krisp_amd import _native, synth
    from oracle import kmer_oracle as K
    K.build()
    L, D, R = 28, 1, 2
    k = L + D + R
    n, records = 4_400_000_000, 16
    t0 = time.time()
    rng = np.random.Generator(np.random.PCG64(77))
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    big = synth.codes_to_text(codes, records=records)
    rl = (n + records - 1) // records
    a = 4_330_000_000
    assert a > (1 << 32) and a // rl == (a + 3_000_000) // rl          # (inside one record, beyond 2^32)
    region = codes[a:a + 3_000_000].copy()
    del codes
    small = region.copy()
    pos = rng.integers(0, len(small), size=3000)
    small[pos] = (small[pos] + rng.integers(1, 4, size=3000, dtype=np.uint8)) & 3
    small_t, region_t = synth.codes_to_text(small, records=1), synth.codes_to_text(region, records=1)
    t1 = time.time()
    want_keys = [K.sorted_keys(small_t.tobytes(), L, D, R), K.sorted_keys(region_t.tobytes(), L, D, R)]
    want = K.intersect(want_keys, [True, False], L, D, R, apply_filter=True)
    wrec = K.collect(want_keys, want, L, D, R)
    with _native.Engine() as eng:
        eng.set_params(L, D, R, max_bases=len(big))
        eng.upload(0, small_t)
        eng.upload(1, big)
        eng.sort(0)
        eng.sort(1)
        assert eng.debug_info()["nslices"] >= 64
        assert eng.count(1) == 2 * _valid_windows(big, k)
        assert eng.inversions(1) == 0
        nc = eng.intersect([0, 1], [True, False], apply_filter=True)
        got = eng.cands()
        assert nc == len(want) > 1000, (nc, len(want))
        for f in ("prefix", "in_mask", "out_mask"):
            assert np.array_equal(got[f], want[f]), f
        recs = np.sort(eng.collect([0, 1]), order=["key", "genome"])
        assert np.array_equal(recs, np.sort(wrec, order=["key", "genome"]))
        # ... and unfiltered: every 30-base pair the small genome shares with the large one lies in that region
        want_all = K.intersect(want_keys, [True, False], L, D, R, apply_filter=False)
        assert eng.intersect([0, 1], [True, False], apply_filter=False) == len(want_all)
        assert np.array_equal(eng.cands()["prefix"], want_all["prefix"])
    print(f"\n4.4 Gbp genome: generation {t1 - t0:.0f} s, device + checks {time.time() - t1:.0f} s; {nc} candidates")


@BIG
def test_c5_three_gbp_genomes():
    """BASELINE configs[4]: 2 x 3 Gbp (1 in / 1 out), k = 31 as 28/1/2: 1.2e10 k-mer records in 64
    key-space slices (pass 0 over all keys, then pass 1 / pass 2 / LDS sort per slice)"""
    import time
    t0 = time.time()
    fam = _family(5, 1, 1, 3_000_000_000)
    t1 = time.time()
    c1, info = _run(fam, 28, 1, 2, light=True)
    assert info["nslices"] == 64        # 6e9 keys per genome -> slices of <= 1.05e8 keys
    assert len(c1) > 10_000_000
    t2 = time.time()
    _check_last_slice_against_the_oracle(fam, 28, 1, 2, 64)
    print(f"\nC5: {len(c1)} candidates; generation {t1 - t0:.0f} s, device + checks {t2 - t1:.0f} s, "
          f"last slice of both genomes == oracle {time.time() - t2:.0f} s; {info}")


def _check_last_slice_against_the_oracle(fam, L, D, R, nslices):
    """an oracle check a human-scale genome can afford -- oracle/kmer_oracle.c restricted to ONE key-space
    slice (generate every key of the genome record by record on host threads, keep those of the slice, sort) against the
    device's sorted keys of that slice (kr_debug_fetch selector 5: the slice sorted last), bit for bit, for every genome.
    The slice is the last one (all-ones slice digits: TT.. lefts); 1 / nslices of the 6e9 keys each."""
    from concurrent.futures import ThreadPoolExecutor
    from krisp_amd import _native
    from oracle import kmer_oracle as K
    K.build()
    topbits = (nslices - 1).bit_length()
    with _native.Engine() as eng:
        eng.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        assert eng.debug_info()["nslices"] == nslices
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
            eng.sort(i)
            n = eng.count(i)
            got = eng.debug_fetch(i, 5, int(n / nslices * 1.5) + 65536)     # relative keys of the last slice
            got = (np.uint64(nslices - 1) << np.uint64(64 - topbits)) | (got >> np.uint64(topbits))
            recs = bytes(t).split(b"\n")
            with ThreadPoolExecutor(max_workers=min(16, len(recs))) as pool:          # (ctypes releases the GIL)
                parts = list(pool.map(lambda r: K.sorted_keys_slice(r, L, D, R, topbits, nslices - 1), recs))
            want = np.sort(np.concatenate(parts))
            assert len(got) == len(want) > n / nslices / 2, (i, len(got), len(want))
            assert np.array_equal(got, want), f"last slice of genome {i} differs from the oracle"
            eng.free(i)


@BIG
@pytest.mark.parametrize("mu,records,snp_every,min_groups", [
    pytest.param(0.01, 16, 10000, 1000, id="survey_8d_generator_mu0.01_16records_snp10kb"),
    pytest.param(0.001, 24, 20000, 100_000, id="close_relatives_mu0.001_24records_snp20kb")])
def test_c3_eight_half_gbp_genomes_long_amplicons(mu, records, snp_every, min_groups):
    """BASELINE configs[2]: 8 x 500 Mbp (4 in / 4 out), 32/60/32 amplicon search -- the wide path with
    key-space slices; checked through properties (every group holds every genome, groups ascend,
    one flank pair per group, a diagnostic column separates the groups).  Two inputs, named by their ids: SURVEY
    8(d)'s own generator (mu = 0.01, 16 records, a planted SNP per 10 kb: what `bench.py --config 2` times) and
    round 2's family of close relatives (mu = 0.001: ten times as many flanks survive the spectrum phase, > 10^5
    groups to cut and render).  Round 6: one left-flank slice of the result (1/1024 of the key space; 1/16384 for the close relatives) is compared with
    the oracle line for line."""
    import time
    from krisp_amd import _native, amplicon, synth
    from krisp_amd import krisp_fasta as KF
    L, D, R = 32, 60, 32
    t0 = time.time()
    fam = synth.family(3, 4, 4, 500_000_000, records=records, mu=mu, snp_every=snp_every)
    t1 = time.time()
    ids = list(range(len(fam)))
    flags = [f for _, f, _ in fam]
    with _native.Engine() as eng:
        eng.set_params_wide(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
        assert eng.debug_info()["nslices"] >= 4
        t2 = time.time()
        n = eng.wide_run(ids, flags, apply_filter=True)
        eng.sync()
        t3 = time.time()
        hits = eng.wide_fetch(_native.WIDE_HITS)
        rows = eng.wide_windows(L + D + R)
        sizes = [eng.wide_count(w) for w in (0, 1, 2)]
    assert n == len(hits) == len(rows) > 0
    labels = [nm for nm, _, _ in fam]
    names = set(labels)
    ingroup = {nm for nm, f, _ in fam if f}
    # over ALL groups, vectorised (close relatives leave 1.8e6 of them: an object per member window takes minutes):
    # every group holds every genome, one flank pair per group, no flank pair twice
    ug, first, inv = np.unique(hits["cand"], return_index=True, return_inverse=True)
    pairs = np.unique(inv.astype(np.int64) * 64 + hits["genome"].astype(np.int64))
    assert (np.bincount(pairs // 64, minlength=len(ug)) == len(fam)).all()
    flanks = np.ascontiguousarray(np.concatenate([rows[:, :L], rows[:, L + D:]], axis=1))
    assert (flanks == flanks[first[inv]]).all()
    keys = flanks[first].view([("f", f"S{L + R}")]).ravel()
    assert len(np.unique(keys)) == len(ug) > min_groups
    # the final text of all of them through the library, and a sample of whole groups through the general path: the
    # same bytes, every genome in every group, a diagnostic column that separates the groups, groups ascending
    wg = amplicon.WindowGroups(rows, hits["cand"], hits["genome"], labels, L, D, R)
    text = wg.render_text(ingroup, False)
    assert text is not None and text[0].count("\n") == len(ug) + 1
    pick = ug[np.random.default_rng(5).choice(len(ug), size=min(len(ug), 3000), replace=False)]
    sel = np.isin(hits["cand"], pick)
    sub = amplicon.WindowGroups(rows[sel], hits["cand"][sel], hits["genome"][sel], labels, L, D, R)
    groups = sub.groups()
    assert len(groups) == len(pick)
    got = []
    for g in groups:
        assert {lab for a in g for lab in a.labels} == names
        assert len({(a.left, a.right) for a in g}) == 1 and all(len(a.diag) == D for a in g)
        assert amplicon.ingroup_unique_columns(g, ingroup)
        got.append((g[0].left, g[0].right))
    assert got == sorted(got) and len(set(got)) == len(got)
    assert sub.render_text(ingroup, False) == tuple(amplicon.render(groups, ingroup, False))
    # round 6: ONE slice of the result against the oracle at full size -- every window of the eight
    # genomes whose left flank starts with these letters through tests/slice_oracle.py (numpy selection + the text oracle's
    # merge tree and filter): the same lines, labels and multiplicities included, no group missing, none too many
    from tests import slice_oracle
    prefix = b"GATCA" if mu >= 0.005 else b"GATCAGT"      # (close relatives: 3 x 10^5 groups before the filter under five letters -- minutes of pure-Python merge tree)
    t4 = time.time()
    want = slice_oracle.slice_lines([t for _, _, t in fam], labels, sorted(ingroup), L, D, R, prefix)
    insl = (rows[:, :len(prefix)] == np.frombuffer(prefix, dtype=np.uint8)).all(axis=1)
    part = amplicon.WindowGroups(rows[insl], hits["cand"][insl], hits["genome"][insl], labels, L, D, R)
    got_lines = amplicon.merged_lines(part.groups())
    assert sorted(got_lines) == sorted(want)
    assert len(want) >= (100 if mu < 0.005 else 0)
    groups = ug
    print(f"\nC3 slice {prefix.decode()}: {len(want)} lines of {len(set(ln.split(',')[0] + ln.split(',')[2] for ln in want))} groups "
          f"equal the oracle's ({time.time() - t4:.0f} s of host time)")
    print(f"C3 (mu={mu:g}, {records} records, SNP per {snp_every}): {len(groups)} groups, dictL {sizes[0]} dictR {sizes[1]} groups before the filter {sizes[2]}; "
          f"generation {t1 - t0:.0f} s, upload {t2 - t1:.0f} s, first wide run (with allocations) {t3 - t2:.1f} s")


def _groups_packed(fam, L, D, R, do_filter):
    from krisp_amd import _native, amplicon
    ids = list(range(len(fam)))
    with _native.Engine() as eng:
        eng.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
            eng.sort(i)
        n = eng.intersect(ids, [f for _, f, _ in fam], apply_filter=do_filter)
        recs = eng.collect(ids) if n else np.empty(0, dtype=_native.RECORD)
    return amplicon.groups_from_records(recs, [nm for nm, _, _ in fam], L, D, R)


def _groups_wide(fam, L, D, R, do_filter, slots=True, ordered=False):
    from krisp_amd import _native
    from krisp_amd import krisp_fasta as KF
    ids = list(range(len(fam)))
    with _native.Engine() as eng:
        eng.set_option(_native.OPT_WIDE_SLOTS, 1 if slots else 0)
        eng.set_option(_native.OPT_WIDE_ORDERED, 1 if ordered else 0)
        eng.set_params_wide(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
        n = eng.wide_run(ids, [f for _, f, _ in fam], apply_filter=do_filter)
        hits = eng.wide_fetch(_native.WIDE_HITS) if n else np.empty(0, dtype=_native.WIDE_HIT)
        info = dict(nl=eng.wide_count(_native.WIDE_DICT_LEFT), nr=eng.wide_count(_native.WIDE_DICT_RIGHT),
                    ng=int(eng.wide_fetch(_native.WIDE_NGROUPS)[0]), slot_bits=[int(x) for x in eng.wide_fetch(_native.WIDE_SLOT_BITS)])
    return KF._groups_from_hits(hits, [t for _, _, t in fam], [nm for nm, _, _ in fam], L, D, R), info


@pytest.mark.parametrize("geo,length,filt,sb", [((25, 1, 2), 4_000_000, True, None), ((12, 4, 12), 2_000_000, True, None),
                                                ((9, 16, 7), 1_000_000, True, None), ((14, 0, 14), 500_000, False, None),
                                                ((25, 1, 2), 3_000_000, True, 1), ((10, 6, 12), 1_000_000, True, 2)])
def test_wide_path_equals_the_packed_path_where_both_apply(geo, length, filt, sb, monkeypatch):
    """kr_wide_run (dictionary composite keys, three sorts) and the one-key path are different
    programs for the same definition: on geometries both can carry, at a size no text oracle
    reaches, they must produce the same groups, members, counts and order."""
    from krisp_amd import amplicon, synth
    fam = synth.family(11, 2, 2, length, records=7, mu=0.004, snp_every=3000, n_frac=0.001, lower_frac=0.01)
    if sb is not None:
        monkeypatch.setenv("KR_SLICE_BASES", str(sb))       # both paths sort in 4^sb key-space slices (environment default of KR_OPT_SLICE_BASES)
        monkeypatch.setenv("KR_WIDE_CACHE", "0")            # and the locate pass re-generates its keys
    a = _groups_packed(fam, *geo, filt)
    b, info = _groups_wide(fam, *geo, filt)
    la, lb = amplicon.merged_lines(a), amplicon.merged_lines(b)
    assert len(la) > 0
    assert la == lb
    assert info["ng"] >= len(b) and info["nl"] > 0 and info["nr"] > 0


def test_wide_run_at_scale_properties():
    """4 x 20 Mbp, 30/40/30: every group holds every genome, groups ascend, every hit's window
    re-read from the text carries its group's flanks; a second run with the dictionaries looked up
    through index + sorted keys instead of slot tables returns the same hits."""
    from krisp_amd import _native, amplicon, synth
    L, D, R = 30, 40, 30
    fam = synth.family(12, 2, 2, 20_000_000, records=16, mu=0.002, snp_every=5000)
    g1, info = _groups_wide(fam, L, D, R, True)
    g2, info2 = _groups_wide(fam, L, D, R, True, slots=False, ordered=True)      # (order-preserving ranks through index + sorted keys)
    g3, info3 = _groups_wide(fam, L, D, R, True, ordered=True)                   # (... through slot tables)
    assert info["slot_bits"][0] == 255 and info["slot_bits"][3] == 254      # (minimizer buckets for the 30-base lefts; the rights mirror them)
    assert info["slot_bits"][6] > 0 and info3["slot_bits"][0] > 0 and info3["slot_bits"][3] > 0 and not any(info2["slot_bits"])
    assert amplicon.merged_lines(g3) == amplicon.merged_lines(g1)
    l1 = amplicon.merged_lines(g1)
    assert l1 == amplicon.merged_lines(g2) and len(g1) > 100
    names = {nm for nm, _, _ in fam}
    ingroup = {nm for nm, f, _ in fam if f}
    pairs = []
    for g in g1:
        assert {lab for a in g for lab in a.labels} == names
        assert len({(a.left, a.right) for a in g}) == 1 and all(len(a.diag) == D for a in g)
        assert amplicon.ingroup_unique_columns(g, ingroup)
        pairs.append((g[0].left, g[0].right))
    assert pairs == sorted(pairs) and len(set(pairs)) == len(pairs)
    assert info["ng"] >= len(g1)


@BIG
def test_c4_thirty_two_genomes_on_one_gpu_and_sharded_over_world_eight(tmp_path, monkeypatch):
    """BASELINE configs[3]: 32 x 100 Mbp (16 in / 16 out), 25/1/2 -- (a) all on one GPU (64 GB of keys, one 32-way
    intersection) through the full-size properties; (b) sharded 4 per rank over world 8 with bench.py's own
    layout and step function, eight contexts on this GPU talking through the file transport: candidates and
    records equal (a)'s; (c) the calls of that step are the library's exchange entry points, in bench.py's order"""
    import inspect
    import threading
    import time
    import bench
    from krisp_amd import _native
    from krisp_amd import distributed as DD
    L, D, R = 25, 1, 2
    world, per = 8, 4
    t0 = time.time()
    shards = [bench.make_genomes(4, r, world, per, 100_000_000) for r in range(world)]      # (config# 4)
    t1 = time.time()
    fam = [(f"g{g}", ing, text) for sh in shards for g, ing, text in sh]
    assert [g for sh in shards for g, _, _ in sh] == list(range(world * per))
    # (a) one GPU, one context: the properties (filtered), then candidates + records without and with the filter
    # (unfiltered: ~3e4 groups conserved in all 32 genomes; filtered: the handful of them that hold a planted SNP)
    c1, info = _run(fam, L, D, R)
    assert info["overflow_segments"] == 0 and len(c1) >= 1
    one = {}
    with _native.Engine() as eng:
        eng.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
            eng.sort(i)
        flags = [f for _, f, _ in fam]
        for filt in (False, True):
            n = eng.intersect(list(range(len(fam))), flags, apply_filter=filt)
            one[filt] = (eng.cands().copy(), eng.collect(list(range(len(fam)))).copy())
            assert n == len(one[filt][0])
    assert np.array_equal(one[True][0], c1) and len(one[False][0]) > 5000
    t2 = time.time()
    # (b) world 8: a thread per rank (the library calls release the GIL), every rank its own context on this GPU
    monkeypatch.setenv("KR_PLACE_TRIES", "1")           # (eight contexts at once: no need to try 8 x 8 buffers)
    out, errs, calls = [None] * world, [None] * world, [None] * world
    costs = [[] for _ in range(world)]
    comm_dir = str(tmp_path / "comm")

    class Recorder:
        """the engine, with the names of the calls sharded_step makes on it"""
        def __init__(self, eng, log):
            self._e, self._log = eng, log
        def __getattr__(self, name):
            attr = getattr(self._e, name)
            if not callable(attr):
                return attr
            def call(*a, **k):
                self._log.append(name)
                return attr(*a, **k)
            return call

    def work(rank):
        try:
            with _native.Engine() as eng:
                eng.comm_init_dir(rank, world, comm_dir)
                eng.set_params(L, D, R, max_bases=max(len(t) for _, _, t in shards[rank]))
                ids = [g for g, _, _ in shards[rank]]
                for g, _, t in shards[rank]:
                    eng.upload(g, t)
                res = {}
                for filt in (False, True):
                    log = []
                    before = eng.debug_comm()
                    n, nrec = DD.sharded_step(Recorder(eng, log), ids, [f for _, f, _ in shards[rank]], world,
                                              apply_filter=filt)
                    after = eng.debug_comm()
                    costs[rank].append({k: after[k] - before[k] for k in ("syncs", "p2p", "collectives")})
                    calls[rank] = log
                    cands = eng.cands().copy()
                    total = eng.records_gather()
                    res[filt] = (n, cands, eng.fetch_records(total) if rank == 0 else None)
                    eng.comm_barrier()
                out[rank] = res
        except BaseException as e:  # noqa: BLE001
            errs[rank] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errs == [None] * world, errs
    for filt in (False, True):
        cand1, rec1 = one[filt]
        assert out[0][filt][0] == len(cand1)
        for r in range(world):                         # (the broadcast gave every rank the survivors)
            assert np.array_equal(out[r][filt][1], cand1), (filt, r)
        assert np.array_equal(np.sort(out[0][filt][2], order=["key", "genome"]), np.sort(rec1, order=["key", "genome"]))
    # (c) the same entry points in the same order on every rank, and bench.py times this very function
    want = ["sort"] * per + ["intersect", "cands_reduce", "cands_bcast", "collect"]
    assert all(c == want for c in calls), calls
    assert "D.sharded_step(eng, ids, flags, world" in inspect.getsource(bench.main)
    # (d) the exchange of one step is a bounded number of blocking calls (round 5): on rank 0 at world 8
    # one agreement before the tree (a second one only in the first exchange of a context, when buffers must grow), one
    # receive + one synchronisation per round of the tree, one agreement before the broadcast it is the root of -- 5 host
    # synchronisations from the second step on; a rank that only sends synchronises for the two agreements and the
    # broadcast's header
    first, second = costs[0]
    assert second["syncs"] <= 5 and first["syncs"] <= 6, costs[0]
    assert second["p2p"] == 3 + 7 and second["collectives"] == 2, costs[0]       # (file transport: the broadcast is 7 sends)
    for r in range(1, world):
        assert costs[r][1]["syncs"] <= 3 + (r & -r).bit_length() - 1, (r, costs[r])
    print(f"\nconfigs[3]: {len(one[False][0])} / {len(c1)} candidates without / with the filter, {len(one[False][1])} / "
          f"{len(one[True][1])} records; generation {t1 - t0:.0f} s, one GPU {t2 - t1:.0f} s, "
          f"world 8 {time.time() - t2:.0f} s")
