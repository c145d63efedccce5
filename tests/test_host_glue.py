"""CPU tests of the host layer: ingest, codecs, rendering, option routing, the C ABI
surface.  No GPU: device results are stood in for by the packed-key oracle's records."""
import ctypes
import gzip
import json
import os
import re

import numpy as np
import pytest

from krisp_amd import _native, amplicon, codec, fasta
from krisp_amd import krisp_fasta as KF
from krisp_amd.kstream import kstream
from oracle import kmer_oracle as K
from oracle import krisp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
KS = json.load(open(os.path.join(GOLDEN, "kstream_cases.json")))
FC = json.load(open(os.path.join(GOLDEN, "fasta_cases.json")))


def _src(case, tmp_path):
    if case["file_text"] is None:
        return case["seqs"]
    p = str(tmp_path / case["fname"])
    opener = gzip.open if case["fname"].endswith(".gz") else open
    with opener(p, "wt") as f:
        f.write(case["file_text"])
    return p


@pytest.mark.parametrize("case", KS, ids=[c["name"] for c in KS])
def test_read_records_matches_reference_reader(case, tmp_path):
    src = _src(case, tmp_path)
    want = O.parse_records(O.read_lines(src), one_shot=True) if isinstance(src, str) \
        else O.parse_records(src, one_shot=False)
    got = [r.decode() for r in fasta.read_records(src)]
    assert got == want


def _safe_geometry(kwargs):
    try:
        return kstream(**kwargs).device_plan()
    except ValueError:
        return None


KS_MORE = json.load(open(os.path.join(GOLDEN, "kstream_cases_more.json")))


@pytest.mark.parametrize("case", KS_MORE, ids=lambda c: c["name"])
def test_kstream_host_chain_on_the_device_served_combinations(case, tmp_path):
    """strand mode x soft-mask rule x split x sort columns: the host chain (the device route's
    fallback) against vectors captured from the reference; the same cases run on the GPU in
    test_gpu_cli.py"""
    src = _src(case, tmp_path)
    ks = kstream(**case["kwargs"])
    assert ks.device_plan() is not None
    assert list(ks.host_lines(src)) == case["out"]


KS_ROUTES = json.load(open(os.path.join(GOLDEN, "kstream_cases_routes.json"))) + \
    json.load(open(os.path.join(GOLDEN, "kstream_cases_r6.json")))          # (round 6: split lists of any length)


@pytest.mark.parametrize("case", KS_ROUTES, ids=lambda c: c["name"])
def test_kstream_host_chain_on_the_round_4_routes(case, tmp_path):
    """round 5: every --sort-cols list over 2- and 3-field lines, --expand-iupac / --allow / kept
    lower case under --sort, several k with sort columns, k > 32 -- the host chain with its GNU-sort emulation
    (`_host_sorted`), which is what the random differential GPU tests compare the device routes with, against vectors
    the reference itself produced (real `sort -t, -kN,N`, kstream.py:83-119); the same vectors run on the GPU in
    test_gpu_cli.py.  Cases named host_* have no device plan on purpose: they pin the chain where it is the product."""
    src = _src(case, tmp_path)
    ks = kstream(**case["kwargs"])
    plan = ks.device_plan()
    if case["name"] in ("host_cols_beyond_fields",):
        assert plan is None and ks.plan_reason       # (the other host_* sets got a device plan in rounds 5 and 6, or decide at run time)
    if case["name"].startswith("msplit") or case["name"] == "host_three_splits":
        assert plan is not None, ks.plan_reason      # (round 6: every split list, every column order of up to eight pieces)
    assert list(ks.host_lines(src)) == case["out"]
    if "count" in case:
        assert len(case["out"]) == case["count"]


def test_route_vectors_cover_what_the_device_plans():
    """the vectors reach every kind of plan: each key layout, each strand mode, kept case, expansion, allow masks,
    several k, the wide path"""
    seen = set()
    for case in KS_ROUTES:
        ks = kstream(**case["kwargs"])
        plan = ks.device_plan()
        if plan is None:
            continue
        for sub in plan.get("multi") or [plan]:
            seen.add(("layout", sub["layout"]))
            seen.add(("strands", sub["strands"]))
            seen.add(("keepcase", sub["keepcase"]))
            seen.add(("expand", sub["expand"]))
            seen.add(("allow", sub["allow"] is not None))
            seen.add(("wide", bool(sub.get("wide"))))
            seen.add(("nfields", len(sub["fields"])))
        seen.add(("multi", bool(plan.get("multi"))))
    want = {("layout", x) for x in ("lrd", "ldr", "custom")} | {("strands", x) for x in (0, 1, 2)} | \
        {(f, b) for f in ("keepcase", "expand", "allow", "wide", "multi") for b in (False, True)} | \
        {("nfields", x) for x in (1, 2, 3)}
    assert want <= seen, want - seen


@pytest.mark.parametrize("case", [c for c in KS if "raises" in c or _safe_geometry(c["kwargs"]) is None],
                         ids=lambda c: c["name"])
def test_kstream_host_chain_matches_reference(case, tmp_path):
    src = _src(case, tmp_path)
    if "raises" in case:
        if case["raises"] == "ValueError":
            with pytest.raises(ValueError):
                kstream(**case["kwargs"])
            return
        ks = kstream(**case["kwargs"])
        if ks.device_plan() is not None:
            pytest.skip("accelerated combination: covered by the gpu tests")
        with pytest.raises(Exception) as ei:
            list(ks(src))
        assert type(ei.value).__name__ == case["raises"]
        return
    ks = kstream(**case["kwargs"])
    if case["use_write"]:
        out = str(tmp_path / "out.txt")
        n = ks.write(out, src)
        lines = open(out).read().split("\n")[:-1]
        assert n == case["count"]
    else:
        lines = list(ks(src))
    assert lines == case["out"]


def test_kstream_routes_only_the_krisp_fasta_combination_to_the_device():
    base = dict(kmers=28, complements=True, disallow="Nn", mapsoft=True, split=[25, -2], sort=True,
                sortcols=[0, 2])
    assert kstream(**base).device_geometry() == (25, 1, 2)
    assert kstream(**dict(base, split=[25, 0])).device_geometry() == (25, 0, 3)     # R = 0 quirk
    assert kstream(**dict(base, mapsoft=False, omitsoft=True)).device_geometry() == (25, 1, 2)
    for change in (dict(kmers=33, split=[30, -2]), dict(complements=False), dict(disallow="N"),
                   dict(mapsoft=False), dict(sort=False), dict(sortcols=None), dict(sortcols=[0]),
                   dict(allow="ACGT"), dict(expandiupac=True), dict(split=None), dict(kmers=[28, 29])):
        assert kstream(**dict(base, **change)).device_geometry() is None, change
    # the wider device route: strand modes, other splits and column orders
    plan = kstream(**base).device_plan()
    assert plan["layout"] == "lrd" and plan["strands"] == 0 and plan["geometry"] == (25, 1, 2)
    assert kstream(**dict(base, complements=False)).device_plan()["strands"] == 1
    assert kstream(**dict(base, complements=False, canonicals=True)).device_plan()["strands"] == 2
    for change in (dict(sortcols=None), dict(sortcols=[0]), dict(sortcols=[0, 1, 2]), dict(split=None, sortcols=None),
                   dict(split=[25], sortcols=None)):
        p = kstream(**dict(base, **change)).device_plan()
        assert p["layout"] == "ldr" and p["geometry"] == (28, 0, 0), change
    # round 4: every column order a key layout can hold, kept lower case, --expand-iupac, k > 32 (the wide path)
    for change, order in ((dict(sortcols=[1]), [1, 0, 2]), (dict(sortcols=[2]), [2, 0, 1]), (dict(sortcols=[1, 0]), [1, 0, 2]),
                          (dict(sortcols=[1, 2]), [1, 2, 0]), (dict(sortcols=[2, 0]), [2, 0, 1]),
                          (dict(split=[5, -3], sortcols=[0, 2]), [0, 2, 1]),            # a middle field of 20 bases
                          (dict(kmers=12, split=[5, -5], sortcols=[2, 1]), [2, 1, 0]),  # outer fields of one width
                          (dict(split=[25], sortcols=[1]), [1, 0])):
        p = kstream(**dict(base, **change)).device_plan()
        assert p["layout"] == "custom" and p["order"] == order and p["geometry"][1:] == (0, 0), change
    p = kstream(**dict(base, mapsoft=False)).device_plan()
    assert p["keepcase"] and p["layout"] == "lrd"
    assert kstream(**dict(base, expandiupac=True)).device_plan()["expand"]
    p = kstream(**dict(base, kmers=33, split=[30, -2])).device_plan()
    assert p["wide"] and p["geometry"] == (30, 1, 2)
    assert [q.get("wide", False) for q in kstream(**dict(base, kmers=[28, 33])).device_plan()["multi"]] == [False, True]
    # what stays on the host chain, each with its reason
    # round 5: any --disallow set (or none) and --allow of any letters: what they say about A C G T is the device's base
    # mask, the rest applies to the host's special windows
    for change, mask in ((dict(disallow="N"), None), (dict(disallow=None), None), (dict(disallow="RrNn"), None),
                         (dict(allow="ACGTR"), None), (dict(allow="ACGT-"), None), (dict(disallow="NnAT"), "CG"),
                         (dict(allow="ATRY"), "AT")):
        p = kstream(**dict(base, **change)).device_plan()
        assert p is not None and p["allow"] == mask, change
    # round 5: (last, middle, first) with outer fields of different widths -- a rotation of the window in front of the layout
    p = kstream(**dict(base, sortcols=[2, 1])).device_plan()
    assert p["layout"] == "custom" and p["order"] == [2, 1, 0]
    # round 6: bases both strands do not share -- two forward passes (split_strands), merged (sorted) or put together by
    # window start (stream order, one k or several)
    for change in (dict(allow="ACG"), dict(disallow="A"), dict(allow="ACG", sort=False), dict(disallow="A", sort=False)):
        p = kstream(**dict(base, **change)).device_plan()
        assert p is not None and p["split_strands"] and p["strands"] == 0, change
    assert all(q["split_strands"] for q in kstream(**dict(base, allow="ACG", sort=False, kmers=[27, 28])).device_plan()["multi"])
    for change, why in ((dict(kmers=40, split=[30, -2], complements=False), "outside the krisp_fasta combination"),
                        (dict(kmers=1100, split=[30, -2]), "k > 1024"), (dict(kmers=620, split=[270, -2]), "flanks outside"),
                        (dict(kmers=40, split=[30, -2], sort=False), "k > 32 without --sort")):
        ks = kstream(**dict(base, **change))
        assert ks.device_plan() is None and why in ks.plan_reason, (change, ks.plan_reason)
    # several k (one device sort per k, merged), --allow of plain bases (a base mask), stream order
    p = kstream(**dict(base, kmers=[28, 29])).device_plan()
    assert [q["k"] for q in p["multi"]] == [28, 29]
    p = kstream(**dict(base, allow="ACGT")).device_plan()
    assert p["allow"] is None and p["sorted"]               # (all four bases: no mask; the letters beyond them never reach the device)
    assert kstream(**dict(base, allow="ACGTN")).device_plan()["allow"] is None
    assert kstream(**dict(base, allow="AT", disallow=None)).device_plan()["allow"] == "AT"
    assert kstream(**dict(base, allow="AC", complements=False, disallow=None)).device_plan()["allow"] == "AC"
    assert kstream(**dict(base, allow="ACGTN", disallow=None)).device_plan() is not None   # (N windows survive: the host's specials)
    assert kstream(**dict(base, expandiupac=True, disallow=None)).device_plan()["expand"]
    # round 5: unsorted streams keep their lower case / expand IUPAC letters too (the host's k-mers placed by position)
    assert kstream(**dict(base, sort=False, mapsoft=False)).device_plan()["keepcase"]
    assert kstream(**dict(base, sort=False, expandiupac=True)).device_plan()["expand"]
    p = kstream(**dict(base, kmers=[28, 29], sort=False)).device_plan()          # several k in stream order: a pass per k
    assert p["sorted"] is False and [q["k"] for q in p["multi"]] == [28, 29]
    p = kstream(**dict(base, sort=False)).device_plan()
    assert p["sorted"] is False and p["geometry"] == (28, 0, 0) and p["fields"] == [25, 1, 2]


def test_ordered_field_codecs_place_host_kmers_where_the_sort_does():
    """codec.merged_ordered_blocks: packed keys in any field order decoded to lines, the k-mers the device alphabet cannot
    carry (IUPAC letters, lower case) spliced in -- against Python's sort of the same lines under the column list"""
    import random
    rng = random.Random(5)
    for it in range(60):
        fields = rng.choice([[3, 2, 4], [4, 0, 3], [2, 5], [6], [3, 3, 3], [0, 4, 2]])
        k = sum(fields)
        order = list(range(len(fields)))
        rng.shuffle(order)
        if not codec.field_layout_ok(fields, order):
            continue
        plain = ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(rng.randint(0, 40))]
        odd = ["".join(rng.choice("ACGTacgtRYn-") for _ in range(k)) for _ in range(rng.randint(0, 12))]
        odd = [s for s in odd if s.strip("ACGT")]
        keys = np.sort(codec.pack_plain([codec.order_string(s, fields, order) for s in plain]))

        def line(s):
            out, at = [], 0
            for w in fields:
                out.append(s[at:at + w])
                at += w
            return ",".join(out)
        want = sorted((line(s) for s in plain + odd), key=lambda ln: tuple(ln.split(",")[c] for c in order) + (ln,))
        got = b"".join(codec.merged_ordered_blocks(keys, odd, fields, order, chunk=7)).decode().split("\n")[:-1]
        assert got == want, (fields, order)


def test_codec_roundtrip_and_oracle_agreement():
    rng = np.random.default_rng(3)
    for (L, D, R) in [(25, 1, 2), (3, 1, 2), (16, 0, 16), (0, 2, 3), (10, 16, 6), (5, 0, 3)]:
        k = L + D + R
        keys = (rng.integers(0, 1 << 62, size=200, dtype=np.uint64) << np.uint64(2)) & codec_topmask(k)
        blob = codec.keys_to_lines_bytes(keys, L, D, R)
        lines = blob.split(b"\n")[:-1]
        assert [l.decode() for l in lines] == K.keys_to_lines(keys, L, D, R)
        assert np.array_equal(codec.lines_to_keys(lines, L, D, R), keys)
    assert codec.effective_geometry(3, 2, 0) == (3, 0, 2)
    assert codec.effective_geometry(3, 0, 0) == (3, 0, 0)


def codec_topmask(k):
    return np.uint64((~0 << (64 - 2 * k)) & 0xFFFFFFFFFFFFFFFF)


PACKABLE = [c for c in FC if c["L"] + c["D"] + c["R"] <= 32 and c["D"] <= 16 and "csv" in c]


def _paths(case, tmp_path):
    if case["name"].startswith("c1_"):
        return {fn: os.path.join(GOLDEN, "c1", fn) for fn in case["ingroup"] + case["outgroup"]}
    out = {}
    for fn, text in case["files"].items():
        p = tmp_path / fn
        p.write_text(text)
        out[fn] = str(p)
    return out


@pytest.mark.parametrize("case", PACKABLE, ids=[c["name"] for c in PACKABLE])
def test_render_from_records_matches_reference_text(case, tmp_path):
    """host glue only: (key, genome, count) records (here from the oracle) -> final text."""
    L, D, R = case["L"], case["D"], case["R"]
    k = L + D + R
    if "--amplicon" in case["main_args"]:
        k = int(case["main_args"][case["main_args"].index("--amplicon") + 1])
    Le, De, Re = codec.effective_geometry(L, k - L - R, R)
    paths = _paths(case, tmp_path)
    files = case["ingroup"] + case["outgroup"]
    keys = []
    for fn in files:
        recs = fasta.read_records(paths[fn])
        keys.append(K.sorted_keys(fasta.to_bases(recs).tobytes(), Le, De, Re, omit=case["omit_soft"]))
    labels = [KF.simplename(f) for f in files]
    ing = frozenset(KF.simplename(f) for f in case["ingroup"])
    flags = [lab in ing for lab in labels]
    do_filter = k > L + R
    if do_filter and De == 0:
        groups = []
    else:
        cands = K.intersect(keys, flags, Le, De, Re, apply_filter=do_filter)
        recs = K.collect(keys, cands, Le, De, Re)
        groups = amplicon.groups_from_records(recs, labels, Le, De, Re)
    ingroup = [KF.simplename(f) for f in case["ingroup"]] if case["outgroup"] else None
    csv, align = amplicon.render(groups, ingroup, dot=case["dot"])
    assert csv == case["csv"]
    assert align == case["align"]
    want = case["filtered_canon"] if "filtered_canon" in case else case["merged_canon"]
    assert sorted(amplicon.merged_lines(groups)) == want


def test_check_special_semantics():
    b = lambda s: np.frombuffer(s.encode(), dtype=np.uint8)          # noqa: E731
    fasta.check_special(b("ACGTNNacgt\nACGT"), 4, False)              # plain: nothing to do
    with pytest.raises(KeyError):
        fasta.check_special(b("ACGXAC"), 4, False)
    fasta.check_special(b("ACGxAC"), 4, True)                         # lower-case window dropped first
    with pytest.raises(KeyError):
        fasta.check_special(b("ACG-AC"), 4, True)                     # isupper() ignores '-'
    fasta.check_special(b("ACX\nGTAC"), 4, False)                     # no window spans the separator
    got = fasta.check_special(b("ACGTRACGT"), 4, False)               # IUPAC k-mers are kept, both strands
    assert sorted(got) == sorted(["CGTR", "YACG", "GTRA", "TYAC", "TRAC", "GTYA", "RACG", "CGTY"])
    assert fasta.check_special(b("ACNRNAC"), 3, False) == []          # every R window also holds N


def test_deduce_geometry_matches_reference_rules():
    p = KF.build_parser()
    for argv, want in [(["f", "--amplicon", "100", "--conserved", "30"], (30, 40, 30, 100)),
                       (["f", "--amplicon", "100", "--diagnostic", "41"], (29, 41, 29, 100)),
                       (["f", "--conserved-left", "25", "--conserved-right", "2", "--diagnostic", "1"], (25, 1, 2, 28)),
                       (["f", "-c", "30", "-d", "0"], (30, 0, 30, 60)),
                       (["f", "-a", "50", "--conserved-left", "10", "--conserved-right", "12"], (10, 28, 12, 50))]:
        a = KF.deduce_geometry(p.parse_args(argv), p)
        assert (a.conserved_left, a.diagnostic, a.conserved_right, a.amplicon) == want
        kw = {k: getattr(p.parse_args(argv), k) for k in
              ("conserved", "conserved_left", "conserved_right", "diagnostic", "amplicon")}
        assert O.deduce_ldr(**kw) == want
    with pytest.raises(SystemExit) as ei:
        KF.deduce_geometry(p.parse_args(["f", "-c", "5"]), p)
    assert ei.value.code == 1


def test_names():
    for fn in ["ingroup0.fasta.gz", "outX.v1.fna", "a.b.c.fa.bz2", "/x/y/in.fasta", "plain"]:
        assert KF.basename(fn) == O.basename(fn)
        assert KF.simplename(fn) == O.simplename(fn)


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    """No compute without a GPU: only load + symbol table + the failure mode of kr_create."""
    header = open(os.path.join(ROOT, "include", "krisp_hip.h")).read()
    declared = set(re.findall(r"\b(kr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = _native.load()
    bound = {name for name, _, _ in _native.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (kr_[a-z0-9_]+)", out))
    assert declared <= exported


def test_the_shipped_library_is_built_without_experiment_switches():
    """the ablation / experiment macros (KR_ABLATE, I3_ABL, P2_ABL, LS2_NORANK, KR_EXP_NOLOOKUP, KR_EXP_MZONLY: wrong
    results, timing only) cannot reach the product: the library says which of them it was compiled with (none), the product
    build passes no -D at all, and a translation unit that defines one without -DKR_EXPERIMENTS does not compile"""
    import inspect
    from krisp_amd import build as kb
    assert _native.load().kr_build_experiments() == 0
    assert "-D" not in inspect.getsource(kb.build)
    guard = open(os.path.join(ROOT, "krisp_amd", "csrc", "k_keys.inc")).read()
    for macro in ("KR_ABLATE", "I3_ABL", "P2_ABL", "LS2_NORANK", "KR_EXP_NOLOOKUP", "KR_EXP_MZONLY"):
        assert macro in guard.split("#error", 1)[0], macro
    # every `#if(n)def` / `#if` on one of them lies in the sources the guard names: no sixth switch crept in
    import glob
    pat = re.compile(r"#\s*if(?:n?def)?\s+.*\b([A-Z][A-Z0-9_]*(?:ABL|ABLATE|NORANK|NOLOOKUP|_EXP_[A-Z0-9_]*))\b")
    seen = set()
    for f in glob.glob(os.path.join(ROOT, "krisp_amd", "csrc", "*")):
        seen |= set(pat.findall(open(f).read()))
    assert seen <= {"KR_ABLATE", "I3_ABL", "P2_ABL", "LS2_NORANK", "KR_EXP_NOLOOKUP", "KR_EXP_MZONLY"}, seen


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libkrisp_hip.so")
    with pytest.raises(_native.KrispHipError):
        _native.load()
    with pytest.raises(_native.KrispHipError):
        _native.Engine()


def test_library_ingest_matches_the_python_reader(tmp_path):
    """kr_fasta_to_bases (one-pass host parser in the library) == to_bases(read_records(...)),
    the reader that is itself pinned to the reference (test above): headers, blank and indented
    lines, CR/LF flavours, gzip, RNA, non-FASTA inputs, first-line consumption."""
    import random

    def check(path):
        recs = fasta.read_records(path)
        rna = bool(fasta.detect_rna(recs))
        want = fasta.to_bases(recs, rna)
        got, r, nspecial = fasta.load_bases(path)
        assert got.tobytes() == want.tobytes(), path
        assert r == rna
        assert nspecial == int((~fasta._PLAIN[want]).sum())

    for c in KS:
        if c["file_text"] is not None:
            check(_src(c, tmp_path))
    for fn in os.listdir(os.path.join(GOLDEN, "c1")):
        check(os.path.join(GOLDEN, "c1", fn))
    rng = random.Random(5)
    for i in range(300):
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
        text = nl.join(pieces) + rng.choice(["", nl, nl + nl])
        for ext in (".fa", ".txt.gz"):
            p = str(tmp_path / f"r{i}{ext}")
            (gzip.open if ext.endswith(".gz") else open)(p, "wb").write(text.encode())
            check(p)


def _brute_wide_hits(texts, flags, L, D, R, omit, do_filter):
    """Pure-Python statement of what kr_wide_run returns (test-side, tiny inputs): member
    windows of the (left,right) groups present in every genome that pass the filter."""
    k = L + D + R
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    wins = []            # (genome, pos, strand, left, diag, right)
    for gi, t in enumerate(texts):
        s = bytes(t).decode()
        for pos in range(len(s) - k + 1):
            w = s[pos:pos + k]
            if "\n" in w:
                continue
            if omit and not w.isupper():
                continue
            w = w.upper()
            if not set(w) <= set("ACGT"):
                continue
            rc = "".join(comp[c] for c in reversed(w))
            for strand, x in ((0, w), (1, rc)):
                wins.append((gi, pos, strand, x[:L], x[L:L + D], x[L + D:]))
    by = {}
    for gi, pos, strand, l, d, r in wins:
        by.setdefault((l, r), []).append((gi, pos, strand, d))
    groups = sorted(p for p, m in by.items() if {x[0] for x in m} == set(range(len(texts))))
    hits = []
    for ci, p in enumerate(groups):
        m = by[p]
        if do_filter:
            keep = False
            for col in range(D):
                a = {x[3][col] for x in m if flags[x[0]]}
                b = {x[3][col] for x in m if not flags[x[0]]}
                keep = keep or not (a & b)
            if not keep:
                continue
        hits += [(ci, gi, pos, strand) for gi, pos, strand, _ in m]
    return np.array(hits, dtype=_native.WIDE_HIT) if hits else np.empty(0, dtype=_native.WIDE_HIT)


WIDE_RAND = [c for c in FC if c["name"].startswith("rand") and c["L"] + c["D"] + c["R"] > 32 and "csv" in c]


@pytest.mark.parametrize("case", WIDE_RAND, ids=[c["name"] for c in WIDE_RAND])
def test_render_from_wide_hits_matches_reference_text(case, tmp_path):
    """host glue of the wide path: hits (here from the brute-force statement above) -> groups
    -> the reference's final text, on the golden cases with amplicons longer than 32"""
    L, D, R = case["L"], case["D"], case["R"]
    paths = _paths(case, tmp_path)
    files = case["ingroup"] + case["outgroup"]
    texts = [fasta.to_bases(fasta.read_records(paths[fn])) for fn in files]
    labels = [KF.simplename(f) for f in files]
    ing = frozenset(KF.simplename(f) for f in case["ingroup"])
    flags = [lab in ing for lab in labels]
    hits = _brute_wide_hits(texts, flags, L, D, R, case["omit_soft"], D > 0)
    np.random.default_rng(0).shuffle(hits)                 # the device returns them unordered inside a group
    hits = hits[np.argsort(hits["cand"], kind="stable")]
    groups = KF._groups_from_hits(hits, texts, labels, L, D, R)
    ingroup = [KF.simplename(f) for f in case["ingroup"]] if case["outgroup"] else None
    csv, align = amplicon.render(groups, ingroup, dot=case["dot"])
    assert csv == case["csv"]
    assert align == case["align"]
    want = case["filtered_canon"] if "filtered_canon" in case else case["merged_canon"]
    assert sorted(amplicon.merged_lines(groups)) == want


def test_render_fast_path_equals_the_general_path():
    """amplicon.render shortcuts groups of one sequence when there is no outgroup (the bulk of a
    large result); the shortcut must print what the general functions print"""
    import random
    rng = random.Random(1)
    groups = []
    for _ in range(3000):
        L, D, R = rng.randint(0, 6), rng.randint(0, 4), rng.randint(0, 5)
        if L + D + R == 0:
            L = 1

        def mk(n):
            return "".join(rng.choice("ACGTN" if rng.random() < 0.1 else "ACGT") for _ in range(n))
        groups.append([amplicon.Amplicon(mk(L), mk(D), mk(R), [rng.choice("abc") for _ in range(rng.randint(1, 4))])])
    for dot in (False, True):
        csv, aln = amplicon.render(groups, None, dot)
        assert csv == "\n".join([amplicon.CSV_HEADER] + [amplicon.render_csv_row(g, None) for g in groups]) + "\n"
        assert aln == "".join(amplicon.render_alignment(g, None, dot) + "\n" for g in groups)


def test_no_kernel_of_the_library_uses_scratch_memory(tmp_path):
    """The library's kernels keep everything in registers and LDS: a private array indexed at run time or
    spilled registers show as `ScratchSize` in hipcc's resource remarks -- and have crept in unnoticed through an
    unrelated edit (spilled scalar registers in k_hist8w<2>, round 3).  Cross-compiles the translation unit (no GPU
    needed) and reads the remarks: no scratch, no spilled vector registers, and a bound on the scalar registers a
    kernel parks in vector lanes."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "krisp_amd", "csrc", "krisp_hip.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-I" + os.path.join(root, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "x.o"), src],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    name, bad, seen, spills = None, [], 0, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m:
            seen += 1
            if int(m.group(1)):
                bad.append((name, int(m.group(1))))
        m = re.search(r"SGPRs Spill: (\d+)", line)
        if m and int(m.group(1)):
            spills[name] = int(m.group(1))
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and int(m.group(1)):
            bad.append((name, "VGPR spill", int(m.group(1))))
    assert seen > 50, "no resource remarks: has the flag changed?"
    assert not bad, bad
    # scalar registers spilled to vector lanes cost no memory traffic, but hundreds of them (round 3: 200-280 in the wide
    # path's kernels, which took the dictionaries of both flanks by value) mean the kernel's arguments do not fit the
    # register file: SGPR_SPILL_BOUND is what the largest argument lists left are allowed
    # (round 6: k_wide_locate<true> -- Geom, the group dictionary and the member list's cursors -- spills 68 to vector lanes)
    SGPR_SPILL_BOUND = 72
    assert all(v <= SGPR_SPILL_BOUND for v in spills.values()), spills


def test_scan_special_through_the_abi_equals_the_python_scan():
    """kr_scan_special (the C ABI's side channel for windows the 2-bit alphabet cannot carry) against
    the pure-Python scan on random texts: same IUPAC k-mers, same KeyError character"""
    import random
    from krisp_amd import _native, fasta

    def py_scan(b, k, omit):
        orig = _native.scan_special_starts
        _native.scan_special_starts = lambda *a: None        # force the host loop
        try:
            return fasta.scan_special(b, k, omit)
        finally:
            _native.scan_special_starts = orig

    rng = random.Random(7)
    alph = "ACGTACGTACGTacgtNnRYMKSWBVDHrymkXx-.\n"
    seen_key = seen_iupac = 0
    for _ in range(1500):
        t = "".join(rng.choice(alph) for _ in range(rng.randint(0, 120))).encode()
        b = np.frombuffer(t, dtype=np.uint8)
        k, omit = rng.randint(1, 12), rng.random() < 0.5
        try:
            want = ("ok", py_scan(b, k, omit))
        except KeyError as e:
            want = ("key", e.args[0])
        try:
            got = ("ok", fasta.scan_special(b, k, omit))
        except KeyError as e:
            got = ("key", e.args[0])
        assert got == want, (t, k, omit)
        seen_key += want[0] == "key"
        seen_iupac += want[0] == "ok" and len(want[1]) > 0
    assert seen_key > 50 and seen_iupac > 50
    # bytes >= 0x80 are left to the host layer
    b = np.frombuffer("ACGR\xe9ACGT".encode("latin-1"), dtype=np.uint8)
    assert _native.scan_special_starts(b, 3, False) is None


def test_text_code_under_address_sanitizer(tmp_path):
    """the library's host-only text code (FASTA parser, side-channel scan) built alone with
    g++ -fsanitize=address,undefined and driven with random / adversarial inputs at exactly the
    buffer sizes the C ABI documents (GPU sanitizers do not run on this pool)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "asan_host")
    src = os.path.join(ROOT, "tests", "native", "asan_host.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", exe, src, "-lz", "-ldl"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ASAN_HOST_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_primer3_hook_with_a_stand_in_package(monkeypatch):
    """--primer3 (krisp_amd/primers.py): primer3-py is not in this image, so the hook is driven with a
    stand-in `primer3` module that records its calls: the template is the ingroup consensus, the target
    the diagnostic region, groups without a pair are left out, the CSV gets the reference's columns"""
    import sys
    import types
    from krisp_amd import primers
    calls = []

    def design_primers(seq_args, global_args):
        calls.append((seq_args, global_args))
        if seq_args["SEQUENCE_TEMPLATE"].startswith("TTTT"):
            return {"PRIMER_PAIR_NUM_RETURNED": 0}
        out = {"PRIMER_PAIR_NUM_RETURNED": 1, "PRIMER_LEFT_0": (0, 4), "PRIMER_RIGHT_0": (11, 4)}
        for t in primers.CSV_TAGS:
            out[t] = "ACGT" if t.endswith("SEQUENCE") else 1.5
        return out
    fake = types.ModuleType("primer3")
    fake.bindings = types.SimpleNamespace(design_primers=design_primers)
    monkeypatch.setitem(sys.modules, "primer3", fake)
    assert primers.available()
    A = amplicon.Amplicon
    groups = [[A("ACGTA", "C", "GGATC", ["inA"]), A("ACGTA", "T", "GGATC", ["outX"])],
              [A("TTTTA", "C", "GGATC", ["inA"]), A("TTTTA", "G", "GGATC", ["outX"])]]
    csv, align = primers.render(groups, ["inA"], primers.settings(primer_size=(18, 30)))
    assert len(calls) == 2 and calls[0][0] == {"SEQUENCE_TEMPLATE": "ACGTACGGATC", "SEQUENCE_TARGET": [5, 1]}
    assert calls[0][1]["PRIMER_OPT_SIZE"] == 24 and calls[0][1]["PRIMER_PRODUCT_SIZE_RANGE"] == [(80, 300)]
    rows = csv.split("\n")
    assert rows[0].startswith("left_seq,diag_seq,right_seq,pair_product_size,pair_penalty,left_sequence,right_sequence")
    assert len(rows) == 3 and rows[1].startswith("ACGTA,C,GGATC,1.5,1.5,ACGT,ACGT")      # the TTTT group has no pair
    assert "Forward" not in align.split("\n")[0] and "Primer statistics:" in align and "Pair statistics:" in align
    assert align.count(" : inA") == 1


def _random_records(rng, L, D, R, nl, ngroups, collide):
    from krisp_amd import _native
    k = L + D + R
    labels = [f"g{c}" for c in rng.permutation(nl)]
    if collide and nl > 1:
        labels[1] = labels[0]                       # two files whose names collapse to one label
    npre = 1 << min(2 * (L + R), 40)
    pres = np.sort(rng.choice(npre, size=min(ngroups, npre), replace=False)) if L + R > 0 else np.zeros(1, dtype=np.int64)
    recs = []
    for p in pres:
        nd = 1 if D == 0 else int(rng.integers(1, min(4, 4 ** D) + 1))
        for d in (np.sort(rng.choice(4 ** D, size=nd, replace=False)) if D > 0 else [0]):
            key = (int(p) << (64 - 2 * (L + R))) if L + R > 0 else 0
            if D > 0:
                key |= int(d) << (64 - 2 * k)
            for g in rng.choice(nl, size=int(rng.integers(1, nl + 1)), replace=False):
                recs.append((key, int(g), int(rng.integers(1, 4))))
    r = np.array(recs, dtype=_native.RECORD)
    return r[rng.permutation(len(r))], labels


def test_library_renderer_equals_the_general_path():
    """kr_render_records (one pass over the records, no object per Amplicon) against amplicon.render over
    groups_from_records -- the restatement of Amplicon.py:523-671 the golden cases pin -- on random groups: every
    geometry incl. L = 0 (the bracket quirk) and D = 0, multiplicities, label collisions, with and without an
    ingroup, both alignment forms; where the reference raises (no all-ingroup row for the consensus) the library
    declines and the general path raises as before"""
    from krisp_amd import amplicon
    rng = np.random.default_rng(3)
    same = declined = 0
    for it in range(400):
        L, D, R = int(rng.integers(0, 8)), int(rng.integers(0, 4)), int(rng.integers(0, 6))
        if L + D + R == 0:
            continue
        nl = int(rng.integers(1, 6))
        recs, labels = _random_records(rng, L, D, R, nl, int(rng.integers(1, 8)), collide=it % 5 == 0)
        ingroup = None if it % 3 == 0 else frozenset(labels[:max(1, nl // 2)])
        for dot in (False, True):
            rg = amplicon.RecordGroups(recs, labels, L, D, R)
            assert len(rg) == len(amplicon.groups_from_records(recs, labels, L, D, R))
            try:
                want = amplicon.render(amplicon.groups_from_records(recs, labels, L, D, R), ingroup, dot)
            except (ValueError, KeyError):
                assert rg.render_text(ingroup, dot) is None
                with pytest.raises((ValueError, KeyError)):
                    amplicon.render(rg, ingroup, dot)
                declined += 1
                continue
            assert amplicon.render(rg, ingroup, dot) == want, (L, D, R, labels, ingroup, dot)
            same += 1
    assert same > 300 and declined > 50


def test_library_renderer_at_scale():
    """2 x 10^5 groups of four records rendered by the library in a fraction of a second
    (the general path needs ~2 s for them), the same text"""
    import time
    from krisp_amd import _native, amplicon
    rng = np.random.default_rng(1)
    L, D, R = 25, 1, 2
    pre = np.unique(np.sort(rng.integers(0, 1 << 54, size=200_000, dtype=np.uint64)) << np.uint64(10))
    ng = len(pre)
    b1 = rng.integers(0, 4, size=ng).astype(np.uint64)
    b2 = (b1 + rng.integers(1, 4, size=ng).astype(np.uint64)) & np.uint64(3)
    recs = np.empty(ng * 4, dtype=_native.RECORD)
    recs["key"] = np.stack([pre | (b1 << np.uint64(8))] * 2 + [pre | (b2 << np.uint64(8))] * 2, axis=1).reshape(-1)
    recs["genome"] = np.tile(np.arange(4, dtype=np.uint32), ng)
    recs["count"] = 1
    recs = recs[np.lexsort((recs["genome"], recs["key"]))]
    labels, ingroup = ["inA", "inB", "outX", "outY"], frozenset(["inA", "inB"])
    t0 = time.time()
    fast = amplicon.render(amplicon.RecordGroups(recs, labels, L, D, R), ingroup)
    dt = time.time() - t0
    assert fast == amplicon.render(amplicon.groups_from_records(recs, labels, L, D, R), ingroup)
    assert dt < 1.0, dt


def test_bench_self_launch_ends_every_rank_when_one_fails(tmp_path):
    """bench.py --gpus 3 without a launcher on a box without a GPU: every rank fails at kr_create, the parent -- which
    never loads the library -- reports the first failure, ends the others and exits non-zero within seconds; with a
    launcher's variables present it does not launch again"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "KRISP_LAUNCHER")}
    env["HIP_VISIBLE_DEVICES"] = ""          # (also on a GPU box: no device for the ranks)
    env["ROCR_VISIBLE_DEVICES"] = ""
    env["TMPDIR"] = str(tmp_path)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--length", "100000", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "ending the other ranks" in r.stderr and "no HIP device" in r.stderr, r.stderr[-1500:]
    assert time.time() - t0 < 120
    assert [p for p in os.listdir(tmp_path) if p.startswith("krisp_bench_launch")] == []
    # a rank of somebody's launch (variables present) with the wrong --gpus: refused, not re-launched
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3"], env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "the launcher started 2 rank(s)" in r.stderr


def test_library_renderer_keeps_labels_that_are_not_ascii():
    """labels come from file names (shared.py:34-73): a name outside ASCII goes through kr_render_records as UTF-8 and
    comes back as the general path renders it (the library's text was decoded as ASCII)"""
    from krisp_amd import _native, amplicon
    L, D, R = 5, 1, 3
    labels = ["Phytophthora_ramorum_é", "outgroup_日本", "plain"]
    recs = []
    for i, pre in enumerate((3, 77, 1000)):
        key = pre << (64 - 2 * (L + R))
        for g, d in ((0, 1), (1, 2), (2, 2 if i else 3)):
            recs.append((key | (d << (64 - 2 * (L + D + R))), g, 1 + (g == 1)))
    recs = np.array(recs, dtype=_native.RECORD)
    for ingroup in (None, frozenset(labels[:1])):
        for dot in (False, True):
            rg = amplicon.RecordGroups(recs, labels, L, D, R)
            want = amplicon.render(amplicon.groups_from_records(recs, labels, L, D, R), ingroup, dot)
            assert rg.render_text(ingroup, dot) is not None
            assert amplicon.render(rg, ingroup, dot) == want
            assert "outgroup_日本(2)" in want[1]


def test_rendezvous_refuses_a_directory_that_is_not_private(tmp_path):
    """the .rv / .d directories hold the answers and messages a rank will trust: one that others can write to (made by
    somebody else under the guessable name, or with a lax mode) is refused"""
    from krisp_amd import distributed as D
    base = str(tmp_path / "comm")
    os.makedirs(base + ".rv", mode=0o777)
    os.chmod(base + ".rv", 0o777)
    with pytest.raises(PermissionError):
        D.rendezvous(base, 1, 2, timeout_s=1)
    os.chmod(base + ".rv", 0o700)
    assert D.private_dir(base + ".rv") == base + ".rv"
    assert D.rendezvous(base, 0, 1, b"x")[1] == b"x"


def _bgzf(data, block=3000):
    """the BGZF framing of bgzip (SAM spec 4.1): gzip members of <= 64 KiB with their length in a 'BC' extra subfield,
    closed by the empty end-of-file member"""
    import struct
    import zlib
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(comp) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                   + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def test_library_reads_bz2_and_block_gzip_as_python_does(tmp_path, monkeypatch):
    """kr_read_file / kr_ingest_file on .bz2 (libbz2 through dlopen: one stream, several streams -- side by side on host
    threads --, zero padding behind the last) and on BGZF files (members inflated side by side, with libdeflate and with
    zlib): the text Python's bz2 / gzip modules read (kstream.py:458-479 opens both through fileinput.hook_compressed);
    damaged files are refused, not half-read"""
    import bz2
    import gzip
    from krisp_amd import _native, fasta
    rng = np.random.default_rng(11)
    recs = ["".join(rng.choice(list("ACGTacgtNR"), size=int(n))) for n in (5000, 1, 70_000, 333)]
    text = "".join(f">r{i} x\n" + "\n".join(r[j:j + 60] for j in range(0, len(r), 60)) + "\n" for i, r in enumerate(recs)).encode()
    third = len(text) // 3
    cases = {
        "one.fa.bz2": bz2.compress(text),
        "multi.fa.bz2": bz2.compress(text[:third]) + bz2.compress(text[third:2 * third]) + bz2.compress(text[2 * third:]),
        "padded.fa.bz2": bz2.compress(text) + b"\0" * 7,
        "block.fa.gz": _bgzf(text),
        "plain.fa.gz": gzip.compress(text),
    }
    want_bases = fasta.to_bases(fasta.read_records(text.split(b"\n")[:-1]))
    for threads in ("4", "1"):
        monkeypatch.setenv("KRISP_INGEST_THREADS", threads)
        for nold in ("", "1"):
            if nold:
                monkeypatch.setenv("KRISP_NO_LIBDEFLATE", "1")      # (read when the library first inflates: only a fresh process sees it)
            for name, blob in cases.items():
                p = tmp_path / name
                p.write_bytes(blob)
                got = _native.read_file(str(p))
                assert got is not None, name
                arr, universal, timings = got
                assert arr.tobytes() == text and not universal, name
                if name == "multi.fa.bz2":
                    assert timings["members"] == 3
                if name == "block.fa.gz":
                    assert timings["members"] == len(text) // 3000 + 2
                bases, nrec, nspecial, rna, is_fasta, _t = _native.ingest_file(str(p))
                assert bases.tobytes() == want_bases.tobytes() and nrec == 4 and is_fasta and not rna and nspecial > 0
                # the host layer's reader takes the same route
                assert fasta.load_bases(str(p))[0].tobytes() == want_bases.tobytes()
    for name, blob in (("cut.fa.bz2", cases["one.fa.bz2"][:-9]), ("junk.fa.bz2", b"BZh9" + b"\x31\x41\x59\x26\x53\x59" + b"x" * 40),
                       ("cutblock.fa.gz", cases["block.fa.gz"][:-40]), ("nostream.fa.bz2", b"tail" + cases["one.fa.bz2"])):
        p = tmp_path / name
        p.write_bytes(blob)
        with pytest.raises(_native.KrispHipError):
            _native.read_file(str(p))
    # bytes behind a complete stream that start no stream: the reference reads .bz2 through fileinput.hook_compressed ->
    # bz2.open, which ignores them and whatever follows (round 4 raised on text tails and decoded streams
    # behind zero bytes).  The library gives what Python gives, on one thread and on several
    first = bz2.compress(text[:third])
    for threads in ("4", "1"):
        monkeypatch.setenv("KRISP_INGEST_THREADS", threads)
        for name, blob in (("tail.fa.bz2", cases["one.fa.bz2"] + b"tail"), ("tailnl.fa.bz2", cases["one.fa.bz2"] + b"XYZ\n"),
                           ("zeros_between.fa.bz2", first + b"\0\0" + bz2.compress(text[third:])),
                           ("fake_header.fa.bz2", first + b"BZh9 but no stream" + bz2.compress(text[third:])),
                           ("short_header.fa.bz2", cases["multi.fa.bz2"] + b"BZh")):
            p = tmp_path / name
            p.write_bytes(blob)
            try:
                with bz2.open(str(p), "rb") as f:
                    want = f.read()
            except EOFError:            # (a header that begins and is cut off by the end of the file: an error there and here)
                assert name == "short_header.fa.bz2"
                with pytest.raises(_native.KrispHipError):
                    _native.read_file(str(p))
                continue
            assert len(want) in (len(text), third), name
            got = _native.read_file(str(p))
            assert got is not None and got[0].tobytes() == want, (name, threads)


def test_text_size_estimates_for_the_memory_plan(tmp_path):
    """fasta.estimate_text_bytes: what kr_reserve plans with before a file is read -- the size of a plain file, ISIZE of
    a gzip member (plus the 4 GiB its compressed size asks for), five times a bz2 file; a file of several members says
    too little (the flow plans again)"""
    import bz2
    import gzip
    import struct
    from krisp_amd import fasta
    t = b">r\n" + b"ACGT" * 50_000 + b"\n"
    (tmp_path / "a.fa").write_bytes(t)
    (tmp_path / "a.fa.gz").write_bytes(gzip.compress(t))
    (tmp_path / "two.fa.gz").write_bytes(gzip.compress(t) + gzip.compress(t[:1000]))
    (tmp_path / "a.fa.bz2").write_bytes(bz2.compress(t))
    assert fasta.estimate_text_bytes(str(tmp_path / "a.fa")) == len(t)
    assert fasta.estimate_text_bytes(str(tmp_path / "a.fa.gz")) == len(t)
    assert fasta.estimate_text_bytes(str(tmp_path / "two.fa.gz")) == 1000
    assert fasta.estimate_text_bytes(str(tmp_path / "a.fa.bz2")) == 5 * os.path.getsize(tmp_path / "a.fa.bz2")
    # a member whose ISIZE word is smaller than the file: the text is at least 4 GiB longer
    blob = gzip.compress(t)
    fake = blob[:-4] + struct.pack("<I", 100) + b""
    (tmp_path / "big.fa.gz").write_bytes(fake)
    assert fasta.estimate_text_bytes(str(tmp_path / "big.fa.gz")) == 100 + (1 << 32)
    (tmp_path / "block.fa.gz").write_bytes(_bgzf(t))
    assert fasta.estimate_text_bytes(str(tmp_path / "block.fa.gz")) == len(t)          # (BGZF: the members' ISIZE words, exact)
    (tmp_path / "half.fa.gz").write_bytes(_bgzf(t)[:-30] + b"x" * 30)                   # (not BGZF all the way: the guess)
    assert fasta.estimate_text_bytes(str(tmp_path / "half.fa.gz")) == 5 * os.path.getsize(tmp_path / "half.fa.gz")


def test_bz2_blocks_decode_side_by_side(tmp_path, monkeypatch):
    """inflate_bz2_blocks (csrc/h_ingest.inc): the blocks of a bzip2 stream are found by their 48-bit mark on any bit, the
    file's structure is walked (stream headers, blocks back to back, end marks, combined checksums), every block is wrapped
    into a stream of its own and the blocks decode side by side -- the text bz2.open gives (kstream.py:458-479), for one
    stream of many blocks at several block sizes, several streams, an empty stream in front, long runs (one block: the
    sequential route); damaged and cut files are refused."""
    import bz2
    from krisp_amd import _native
    rng = np.random.default_rng(29)
    monkeypatch.setenv("KRISP_INGEST_THREADS", "4")
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=2_500_000)
    t = b">r\n" + b"\n".join(seq[i:i + 70].tobytes() for i in range(0, len(seq), 70)) + b"\n"
    runs = b"".join(bytes([65 + int(x)]) * int(k) for x, k in zip(rng.integers(0, 4, 9000), rng.integers(1, 600, 9000)))
    cases = [("l9", bz2.compress(t, 9), t, 1), ("l1", bz2.compress(t, 1), t, 1), ("l4", bz2.compress(t, 4), t, 1),
             ("three", bz2.compress(t[:400_000], 9) + bz2.compress(t[400_000:1_500_000], 2) + bz2.compress(t[1_500_000:], 9) + b"\0" * 9, t, 3),
             ("empty_first", bz2.compress(b"") + bz2.compress(t[:300_000], 1), t[:300_000], 2),
             ("runs", bz2.compress(runs, 9), runs, 1)]
    for name, blob, want, nstreams in cases:
        p = tmp_path / (name + ".fa.bz2")
        p.write_bytes(blob)
        for on in ("1", "0"):
            monkeypatch.setenv("KRISP_PBZ2", on)
            arr, universal, timings = _native.read_file(str(p))
            assert arr.tobytes() == want and timings["members"] == nstreams, (name, on)
    monkeypatch.setenv("KRISP_PBZ2", "1")
    blob = bytearray(cases[1][1])
    for at in (len(blob) // 2, len(blob) - 2, 3, 12):
        b2 = bytearray(blob)
        b2[at] ^= 0x10
        (tmp_path / "bad.fa.bz2").write_bytes(bytes(b2))
        with pytest.raises(_native.KrispHipError):
            _native.read_file(str(tmp_path / "bad.fa.bz2"))
    (tmp_path / "cut.fa.bz2").write_bytes(bytes(blob[:len(blob) // 2]))
    with pytest.raises(_native.KrispHipError):
        _native.read_file(str(tmp_path / "cut.fa.bz2"))


def test_one_gzip_member_on_several_threads(tmp_path, monkeypatch):
    """h_pgzip.inc: a large gzip member is cut into chunks, a block start is FOUND in each, the chunks decode side by side
    (16-bit symbols with markers for what they copy from the unknown 32 KB in front, zlib once no marker is left) and are
    stitched, checked against CRC-32 and ISIZE -- the text gzip.open gives (kstream.py:458-479), for DNA, repeats, noise,
    runs and prose at every kind of block (stored, fixed, dynamic), with header fields, small windows, several members and
    zero padding; chunks as small as they go, so that every file has dozens of joins.  Damaged and cut files are refused."""
    import struct
    import zlib
    from krisp_amd import _native
    rng = np.random.default_rng(23)
    monkeypatch.setenv("KRISP_PGZIP_MIN", "0")
    monkeypatch.setenv("KRISP_PGZIP_CHUNK", "65536")
    monkeypatch.setenv("KRISP_INGEST_THREADS", "4")

    def dna(n, width=70):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
        return b">r\n" + b"\n".join(seq[i:i + width].tobytes() for i in range(0, n, width)) + b"\n"

    def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, fields=False, wbits=15, memlevel=8):
        c = zlib.compressobj(level, zlib.DEFLATED, -wbits, memlevel, strategy)
        body = c.compress(data) + c.flush()
        hdr = b"\x1f\x8b\x08" + (b"\x0c" if fields else b"\0") + b"\0\0\0\0\0\xff"
        if fields:
            hdr += struct.pack("<H", 5) + b"hello" + b"file.fa\0"
        return hdr + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)

    texts = {
        "dna": dna(1_200_000),
        "repeats": dna(150_000) * 7,
        "noise": rng.integers(0, 256, size=500_000, dtype=np.uint8).tobytes(),
        "runs": b"".join(bytes([65 + int(x)]) * int(n) for x, n in zip(rng.integers(0, 4, 8000), rng.integers(1, 300, 8000))),
        "prose": (b"the quick brown fox jumps over the lazy dog; " * 40 + b"\x07\xf3") * 600,
    }
    cases = []
    for tn, t in texts.items():
        for level in (0, 1, 6, 9):
            cases.append((f"{tn}_l{level}", member(t, level), t))
        cases.append((f"{tn}_fixed", member(t, 6, zlib.Z_FIXED), t))
        cases.append((f"{tn}_huffman", member(t, 6, zlib.Z_HUFFMAN_ONLY), t))
        cases.append((f"{tn}_fields_w9", member(t, 6, fields=True, wbits=9, memlevel=1), t))
    t = texts["dna"]
    cases.append(("three", member(t[:400_000]) + member(t[400_000:900_000], 9) + member(t[900_000:], 1) + b"\0" * 5, t))
    cases.append(("many", b"".join(member(t[i:i + 5000]) for i in range(0, 300_000, 5000)), t[:300_000]))
    for name, blob, want in cases:
        p = tmp_path / (name + ".fa.gz")
        p.write_bytes(blob)
        for on, zl in (("1", "0"), ("1", "1"), ("0", "0")):      # our byte decoder behind the markers / zlib there / one thread
            monkeypatch.setenv("KRISP_PGZIP", on)
            monkeypatch.setenv("KRISP_PGZIP_ZLIB", zl)
            arr, universal, timings = _native.read_file(str(p))
            assert arr.tobytes() == want, (name, on, zl)
        assert timings["members"] == {"three": 3, "many": 60}.get(name, 1)
    monkeypatch.setenv("KRISP_PGZIP", "1")
    monkeypatch.setenv("KRISP_PGZIP_ZLIB", "0")
    blob = bytearray(member(t, 6))
    for at in (len(blob) // 2, len(blob) - 3, len(blob) - 7, 30):
        b2 = bytearray(blob)
        b2[at] ^= 0x5A
        (tmp_path / "bad.fa.gz").write_bytes(bytes(b2))
        with pytest.raises(_native.KrispHipError):
            _native.read_file(str(tmp_path / "bad.fa.gz"))
    (tmp_path / "cut.fa.gz").write_bytes(bytes(blob[:len(blob) * 2 // 3]))
    with pytest.raises(_native.KrispHipError):
        _native.read_file(str(tmp_path / "cut.fa.gz"))


def test_window_renderer_equals_the_general_path():
    """kr_render_windows (the member windows of long amplicons as text rows -> CSV + alignment bytes, groups ordered by
    (left, right), members by (diag, label), pieces rendered side by side) against amplicon.render over
    groups_from_windows -- the restatement of Amplicon.py:523-671 the long-amplicon goldens pin: random geometries with
    flanks beyond one key, interleaved rows, mirror group numbers, label collisions, RNA, both alignment forms, the
    cases where the reference raises (declined), and 4 x 10^4 groups through the threaded pieces"""
    from krisp_amd import amplicon
    rng = np.random.default_rng(17)
    same = declined = 0
    for it in range(260):
        big = it == 0
        L, D, R = int(rng.integers(1, 45)), int(rng.integers(0, 30)), int(rng.integers(1, 45))
        k = L + D + R
        ng = 40_000 if big else int(rng.integers(1, 9))
        nl = int(rng.integers(1, 6))
        labels = [f"g{c}" for c in rng.permutation(nl)]
        if it % 5 == 0 and nl > 1:
            labels[1] = labels[0]
        flanks = np.unique(rng.integers(0, 4, size=(ng, L + R)), axis=0)
        rows, cand, gen = [], [], []
        for gi, fl in enumerate(flanks):
            nd = 1 if D == 0 else int(rng.integers(1, 4))
            diags = rng.integers(0, 4, size=(nd, D))
            for d in diags:
                for g in rng.choice(nl, size=int(rng.integers(1, nl + 1)), replace=False):
                    for _rep in range(int(rng.integers(1, 3))):
                        rows.append(np.concatenate([fl[:L], d, fl[L:]]))
                        # (odd rounds: KR_WIDE_HITS order -- a group and its mirror, bit 31, share a region of ascending
                        # number and interleave inside it; even rounds: any order at all)
                        cand.append((gi // 2) | ((gi % 2) << 31) if it % 2 else gi | (0x80000000 if gi % 3 == 0 else 0))
                        gen.append(int(g))
        rows = np.frombuffer(b"ACGT", dtype=np.uint8)[np.array(rows)]
        if it % 2:
            region = np.array(cand, dtype=np.uint32) & np.uint32(0x7FFFFFFF)
            perm = np.lexsort((rng.random(len(rows)), region))
        else:
            perm = rng.permutation(len(rows))
        rows, cand, gen = rows[perm], np.array(cand, dtype=np.uint32)[perm], np.array(gen, dtype=np.uint32)[perm]
        ingroup = None if it % 3 == 0 else frozenset(labels[:max(1, nl // 2)])
        rna = it % 7 == 3
        for dot in (False, True):
            wg = amplicon.WindowGroups(rows, cand, gen, labels, L, D, R, rna=rna)
            general = amplicon.WindowGroups(rows, cand, gen, labels, L, D, R, rna=rna).groups()
            assert len(wg) == len(general)
            try:
                want = amplicon.render(general, ingroup, dot)
            except (ValueError, KeyError):
                assert wg.render_text(ingroup, dot) is None
                declined += 1
                continue
            assert amplicon.render(wg, ingroup, dot) == want, (L, D, R, labels, ingroup, dot)
            same += 1
    assert same > 250 and declined > 20
