"""The oracle (oracle/krisp_oracle.py) against golden vectors captured from the
reference (tests/golden/make_goldens.py) and the README known answers."""
import gzip
import hashlib
import json
import os

import pytest

from oracle import krisp_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KS = json.load(open(os.path.join(GOLDEN, "kstream_cases.json"))) + \
    json.load(open(os.path.join(GOLDEN, "kstream_cases_more.json"))) + \
    json.load(open(os.path.join(GOLDEN, "kstream_cases_routes.json"))) + \
    json.load(open(os.path.join(GOLDEN, "kstream_cases_r6.json")))
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden_cases import FC as _FC0, FC6, canon_equal       # noqa: E402
FC = _FC0 + FC6


@pytest.mark.parametrize("case", KS, ids=[c["name"] for c in KS])
def test_kstream_case(case, tmp_path):
    src = case["seqs"]
    if case["file_text"] is not None:
        src = str(tmp_path / case["fname"])
        opener = gzip.open if case["fname"].endswith(".gz") else open
        with opener(src, "wt") as f:
            f.write(case["file_text"])
    if "raises" in case:
        with pytest.raises(Exception) as ei:
            O.kstream_lines(src, **case["kwargs"])
        assert type(ei.value).__name__ == case["raises"]
        return
    out = O.kstream_lines(src, **case["kwargs"])
    assert out == case["out"]
    if "count" in case:
        assert len(out) == case["count"]


def _materialise(case, tmp_path):
    paths = {}
    if case["name"].startswith("c1_"):
        for fn in case["ingroup"] + case["outgroup"]:
            paths[fn] = os.path.join(GOLDEN, "c1", fn)
    else:
        for fn, text in case["files"].items():
            p = tmp_path / fn
            p.write_text(text)
            paths[fn] = str(p)
    return paths


def _amplicon(case):
    a = case["main_args"]
    if "--amplicon" in a:
        return int(a[a.index("--amplicon") + 1])
    return case["L"] + case["D"] + case["R"]


@pytest.mark.parametrize("case", FC, ids=[c["name"] for c in FC])
def test_fasta_case(case, tmp_path):
    paths = _materialise(case, tmp_path)
    ing = [paths[f] for f in case["ingroup"]]
    outg = [paths[f] for f in case["outgroup"]]
    res = O.run_krisp_fasta(ing, outg, case["L"], case["D"], case["R"],
                            amplicon=_amplicon(case) if "main_args" in case else None,
                            omit_soft=case["omit_soft"], dot=case["dot"]) \
        if "csv" in case else None
    if res is None:     # stages only (the reference's renderer crashes: IUPAC column)
        k = case["L"] + case["D"] + case["R"]
        sf = [(f"{O.basename(f)}.{k}mers",
               O.extract_sorted_kmers(paths[f], case["L"], case["R"], k, case["omit_soft"]))
              for f in case["ingroup"] + case["outgroup"]]
        merged = O.merge_tree(sf)
        res = {"sorted": dict(sf), "merged": merged,
               "filtered": O.filter_lines(merged, [O.simplename(f) for f in case["ingroup"]])}
    # stage 1: sorted k-mer files, byte for byte
    for fn, (tag, lines) in zip(case["ingroup"] + case["outgroup"], res["sorted"].items()):
        data = "".join(l + "\n" for l in lines).encode()
        info = case["sorted"][fn]
        assert len(lines) == info["lines"]
        assert hashlib.sha256(data).hexdigest() == info["sha256"], fn
    # stage 2/3: canonicalised (intra-group order depends on merge completion order)
    assert canon_equal(sorted(res["merged"]), case["merged_canon"])
    if "filtered_canon" in case:
        assert canon_equal(sorted(res["filtered"]), case["filtered_canon"])
    if "csv" in case:
        assert res["csv"] == case["csv"]
        assert res["align"] == case["align"]


def test_readme_known_answers():
    """README.md:121-124 (its row-2 typo 'K,A' is 'K,AC' in the program and in the
    README's own alignment, README.md:162-164), 157-166, 172-185, 251-256."""
    d = os.path.join(GOLDEN, "c1")
    ing = [f"{d}/ingroup{i}.fasta.gz" for i in (0, 1)]
    outg = [f"{d}/outgroup{i}.fasta.gz" for i in (0, 1, 2)]
    r = O.run_krisp_fasta(ing, outg, 25, 1, 2)
    assert r["csv"] == ("left_seq,diag_seq,right_seq\n"
                        "CGACAAGATACTCTCGCAGCTTGGT,M,AG\n"
                        "TGACGCAGATCATCCCGCGCTTACT,K,AC\n")
    assert r["align"] == (
        "CGACAAGATACTCTCGCAGCTTGGTCAG : ingroup0\n"
        "CGACAAGATACTCTCGCAGCTTGGTAAG : ingroup1\n"
        "CGACAAGATACTCTCGCAGCTTGGTGAG : outgroup0;outgroup1;outgroup2\n"
        "                        {#}\n\n"
        "TGACGCAGATCATCCCGCGCTTACTGAC : ingroup0\n"
        "TGACGCAGATCATCCCGCGCTTACTTAC : ingroup1\n"
        "TGACGCAGATCATCCCGCGCTTACTCAC : outgroup0;outgroup1;outgroup2\n"
        "                        {#}\n\n")
    r = O.run_krisp_fasta(ing, outg, 25, 1, 2, dot=True)
    assert r["align"].split("\n")[:3] == [
        "CGACAAGATACTCTCGCAGCTTGGTCAG : ingroup0",
        ".........................A.. : ingroup1",
        ".........................G.. : outgroup0;outgroup1;outgroup2"]
    r = O.run_krisp_fasta(ing + outg, [], 30, 0, 30)
    assert r["csv"] == ("left_seq,diag_seq,right_seq\n"
                        "ACGCACAAGGACAAGTGCCACTAAACCAGC,,CAGCCCTGACGCAGATCATCCCGCGCTTAC\n"
                        "AGTAAGCGCGGGATGATCTGCGTCAGGGCT,,GGCTGGTTTAGTGGCACTTGTCCTTGTGCG\n"
                        "CGCACAAGGACAAGTGCCACTAAACCAGCC,,AGCCCTGACGCAGATCATCCCGCGCTTACT\n"
                        "GTAAGCGCGGGATGATCTGCGTCAGGGCTG,,GCTGGTTTAGTGGCACTTGTCCTTGTGCGT\n")
    r = O.run_krisp_fasta(ing, outg, 30, 40, 30, amplicon=100, dot=True)
    assert r["align"].split("\n")[0] == (
        "ACGCACAAGGACAAGTGCCACTAAACCAGCCAGCCCTGACGCAGATCATCCCGCGCTTACTGACCAAGCTGCGAGAGTATCTTGTCGATGGGAACGATAG : ingroup0")


def test_deduce_ldr():
    assert O.deduce_ldr(amplicon=100, conserved=30) == (30, 40, 30, 100)
    assert O.deduce_ldr(amplicon=100, diagnostic=41) == (29, 41, 29, 100)
    assert O.deduce_ldr(conserved_left=25, conserved_right=2, diagnostic=1) == (25, 1, 2, 28)
    assert O.deduce_ldr(conserved=30, diagnostic=0) == (30, 0, 30, 60)
    assert O.deduce_ldr(amplicon=50, conserved_left=10, conserved_right=12) == (10, 28, 12, 50)
    assert O.deduce_ldr(conserved=5) is None
    assert O.deduce_ldr(amplicon=50, conserved_left=10) is None


@pytest.mark.parametrize("geo,prefix,do_filter", [((8, 10, 8), b"ACGA", True), ((12, 3, 5), b"GTTC", True), ((6, 20, 9), b"TTAGG", False),
                                                   ((32, 6, 32), b"CCAT", True)])
def test_slice_oracle_equals_the_text_oracle_on_whole_genomes(geo, prefix, do_filter, tmp_path):
    """tests/slice_oracle.py (numpy selection of one left-flank prefix + this oracle's merge tree and filter: what the
    full-size long-amplicon test compares the device with) against this oracle run on the whole genomes, restricted to
    the lines of that prefix."""
    from krisp_amd import synth
    from tests import slice_oracle
    L, D, R = geo
    k = L + D + R
    fam = synth.family(11, 2, 2, 50_000 if L < 32 else 100_000, records=3, mu=0.002, snp_every=400)
    files = []
    for name, _, text in fam:
        p = str(tmp_path / f"{name}.fa")
        synth.write_fasta(p, text)
        files.append(p)
    ingroup = [name for name, f, _ in fam if f]
    merged = O.merge_tree([(f"{O.basename(f)}.{k}mers", O.extract_sorted_kmers(f, L, R, k, False)) for f in files])
    want = O.filter_lines(merged, ingroup) if do_filter else merged
    want = [ln for ln in want if ln.startswith(prefix.decode())]
    got = slice_oracle.slice_lines([t for _, _, t in fam], [n for n, _, _ in fam], ingroup, L, D, R, prefix, do_filter)
    assert len(want) > 0
    assert sorted(got) == sorted(want)
