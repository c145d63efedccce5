"""The golden krisp_fasta cases (tests/golden/fasta_cases.json + the round-6 set fasta_cases_r6.json: mixed DNA / RNA runs,
flanks > 64, amplicons > 256) and the comparison of canonicalised line lists -- the long-amplicon cases keep their merged /
filtered files as {lines, sha256 of '\\n'.join(sorted lines)} instead of megabytes of text (make_goldens.py: _maybe_hashed)."""
import hashlib
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FC = json.load(open(os.path.join(GOLDEN, "fasta_cases.json")))
FC6 = json.load(open(os.path.join(GOLDEN, "fasta_cases_r6.json")))


def canon_equal(got_sorted_lines, want):
    """got (a sorted list of lines) against a golden `merged_canon` / `filtered_canon` entry"""
    if isinstance(want, dict):
        got = list(got_sorted_lines)
        return len(got) == want["lines"] and hashlib.sha256("\n".join(got).encode()).hexdigest() == want["sha256"]
    return list(got_sorted_lines) == want


def canon_lines(want):
    """the entry's lines, or None when only their hash was kept"""
    return None if isinstance(want, dict) else want
