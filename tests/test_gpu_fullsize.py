"""Full-size runs checked through size-independent properties (no oracle reaches these
sizes in seconds): record count = 2 x valid windows, sortedness (adjacent inversions
counted on the device), idempotence of sort + intersect, every candidate re-found by the
collect in every genome, agreement between slicing configurations.

  * BASELINE configs[1] (4 x 50 Mbp, 25/1/2) runs in the regular `-m gpu` suite.
  * BASELINE configs[4] (2 x 3 Gbp, k = 31 as 28/1/2; 16 key-space slices, ~100 GB of HBM)
    is opt-in: KR_RUN_C5=1 (minutes of host-side genome generation).
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


def _run(fam, L, D, R, slice_bases=None):
    from krisp_amd import _native
    old = os.environ.get("KR_SLICE_BASES")
    if slice_bases is not None:
        os.environ["KR_SLICE_BASES"] = str(slice_bases)
    try:
        eng = _native.Engine()
        eng.set_params(L, D, R, max_bases=max(len(t) for _, _, t in fam))
    finally:
        if slice_bases is not None:
            if old is None:
                del os.environ["KR_SLICE_BASES"]
            else:
                os.environ["KR_SLICE_BASES"] = old
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
    # idempotence: sorting again from the resident bases and intersecting again changes nothing
    for i in ids:
        eng.sort(i)
    assert eng.intersect(ids, flags, apply_filter=True) == n1
    c2 = eng.cands()
    assert np.array_equal(c1, c2)
    assert np.all(np.diff(c1["prefix"].astype(np.uint64)) > 0) if n1 > 1 else True
    # every candidate is present in every genome, and ingroup / outgroup diagnostic bases differ
    recs = eng.collect(ids)
    pm = np.uint64((~0 << (64 - 2 * (L + R))) & 0xFFFFFFFFFFFFFFFF)
    pre = recs["key"] & pm
    for i in ids:
        assert np.array_equal(np.unique(pre[recs["genome"] == i]), c1["prefix"])
    if D == 1:
        dshift = np.uint64(62 - 2 * (L + R))
        base = ((recs["key"] >> dshift) & np.uint64(3)).astype(np.int64)
        is_in = np.array(flags)[recs["genome"]]
        order = np.argsort(pre, kind="stable")
        idx = np.searchsorted(c1["prefix"], pre)
        in_sets = np.zeros(n1, dtype=np.int64)
        out_sets = np.zeros(n1, dtype=np.int64)
        np.bitwise_or.at(in_sets, idx[is_in], 1 << base[is_in])
        np.bitwise_or.at(out_sets, idx[~is_in], 1 << base[~is_in])
        assert np.all((in_sets & out_sets) == 0)
        assert np.array_equal(in_sets, c1["in_mask"].astype(np.int64))
        assert np.array_equal(out_sets, c1["out_mask"].astype(np.int64))
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


@pytest.mark.skipif(os.environ.get("KR_RUN_C5") != "1", reason="opt-in: KR_RUN_C5=1 (2 x 3 Gbp, ~100 GB HBM)")
def test_c5_three_gbp_genomes():
    fam = _family(5, 1, 1, 3_000_000_000)
    c1, info = _run(fam, 28, 1, 2)
    assert info["nslices"] == 64        # 6e9 keys per genome -> slices of <= 1.05e8 keys
    print("C5:", len(c1), "candidates;", info)


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


def _groups_wide(fam, L, D, R, do_filter):
    from krisp_amd import _native
    from krisp_amd import krisp_fasta as KF
    ids = list(range(len(fam)))
    with _native.Engine() as eng:
        eng.set_params_wide(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
        n = eng.wide_run(ids, [f for _, f, _ in fam], apply_filter=do_filter)
        hits = eng.wide_fetch(_native.WIDE_HITS) if n else np.empty(0, dtype=_native.WIDE_HIT)
        info = dict(nl=len(eng.wide_fetch(_native.WIDE_DICT_LEFT)), nr=len(eng.wide_fetch(_native.WIDE_DICT_RIGHT)),
                    ng=len(eng.wide_fetch(_native.WIDE_GROUPS)))
    return KF._groups_from_hits(hits, [t for _, _, t in fam], [nm for nm, _, _ in fam], L, D, R), info


@pytest.mark.parametrize("geo,length,filt,sb", [((25, 1, 2), 4_000_000, True, None), ((12, 4, 12), 2_000_000, True, None),
                                                ((9, 16, 7), 1_000_000, True, None), ((14, 0, 14), 2_000_000, False, None),
                                                ((25, 1, 2), 3_000_000, True, 1), ((10, 6, 12), 1_000_000, True, 2)])
def test_wide_path_equals_the_packed_path_where_both_apply(geo, length, filt, sb, monkeypatch):
    """kr_wide_run (dictionary composite keys, three sorts) and the one-key path are different
    programs for the same definition: on geometries both can carry, at a size no text oracle
    reaches, they must produce the same groups, members, counts and order."""
    from krisp_amd import amplicon, synth
    fam = synth.family(11, 2, 2, length, records=7, mu=0.004, snp_every=3000, n_frac=0.001, lower_frac=0.01)
    if sb is not None:
        monkeypatch.setenv("KR_SLICE_BASES", str(sb))       # both paths sort in 4^sb key-space slices
        monkeypatch.setenv("KR_WIDE_CACHE", "0")            # and the locate pass re-generates its keys
    a = _groups_packed(fam, *geo, filt)
    b, info = _groups_wide(fam, *geo, filt)
    la, lb = amplicon.merged_lines(a), amplicon.merged_lines(b)
    assert len(la) > 0
    assert la == lb
    assert info["ng"] >= len(b) and info["nl"] > 0 and info["nr"] > 0


def test_wide_run_at_scale_properties():
    """4 x 20 Mbp, 30/40/30: every group holds every genome, groups ascend, every hit's window
    re-read from the text carries its group's flanks; a second run returns the same hits."""
    from krisp_amd import _native, amplicon, synth
    L, D, R = 30, 40, 30
    fam = synth.family(12, 2, 2, 20_000_000, records=16, mu=0.002, snp_every=5000)
    g1, info = _groups_wide(fam, L, D, R, True)
    g2, _ = _groups_wide(fam, L, D, R, True)
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
