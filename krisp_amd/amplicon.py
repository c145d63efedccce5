"""Host glue after the device path: survivor records -> labelled groups -> text.

Restates the reference's record / label / ordering / rendering semantics so the
final CSV and alignment text are byte-identical at --cores 1
(krisp_fasta/Amplicon.py:154-348 Amplicon, 351-692 ConservedEndAmplicons;
outputAlignments.py:26-162).  The diagnostic filter itself runs on the device
(kr_intersect / kr_cands_merge); nothing here decides which groups survive.
"""

import numpy as np

from . import codec

# Amplicon.py:10-12 -- sorted base tuple -> IUPAC letter, from Biopython's
# IUPACData.ambiguous_dna_values (standard table; 'X' and 'N' both map to ACGT
# and 'N', iterated later, wins; README.md:122-123 pins AC->M and GT->K).
_AMBIGUOUS = {"A": "A", "C": "C", "G": "G", "T": "T", "M": "AC", "R": "AG", "W": "AT",
              "S": "CG", "Y": "CT", "K": "GT", "V": "ACG", "H": "ACT", "D": "AGT",
              "B": "CGT", "X": "GATC", "N": "GATC"}
IUPAC_KEY = {tuple(sorted(v)): k for k, v in _AMBIGUOUS.items()}
IUPAC_KEY[("?",)] = "N"


class Amplicon:
    """One distinct sequence of a group + the sorted multiset of genome labels
    that carry it (Amplicon.py:154-210)."""
    __slots__ = ("left", "diag", "right", "labels")

    def __init__(self, left, diag, right, labels):
        self.left, self.diag, self.right = left, diag, right
        self.labels = sorted(labels)

    @property
    def sequence(self):
        return self.left + self.diag + self.right

    def label_string(self):
        """Amplicon.py:170-187: name or name(count), ';' joined, names sorted."""
        out, prev, run = [], None, 0
        for lab in self.labels:                 # (sorted: equal names are adjacent)
            if lab == prev:
                run += 1
                continue
            if prev is not None:
                out.append(prev if run == 1 else f"{prev}({run})")
            prev, run = lab, 1
        if prev is not None:
            out.append(prev if run == 1 else f"{prev}({run})")
        return ";".join(out)

    def line(self):
        """merged-file line, Amplicon.py:330-348."""
        return f"{self.left},{self.diag},{self.right},{self.label_string()}"

    def __str__(self):
        return f"{self.sequence} : {self.label_string()}"


def groups_from_records(records, labels, L, D, R, rna=False):
    """(key, genome, count) records of the survivors -> list of groups (lists of
    Amplicon), groups ascending by (left,right), Amplicons ascending by diag --
    the order every leaf file has (GNU sort's whole-line tie-break) and that
    ConservedEndAmplicons.add() preserves (Amplicon.py:448-481)."""
    if len(records) == 0:
        return []
    rec = np.sort(records, order=["key", "genome"])
    pm = codec.prefix_mask(L, R)
    keys = rec["key"]
    new_key = np.ones(len(rec), dtype=bool)
    new_key[1:] = keys[1:] != keys[:-1]
    pre = keys & pm
    new_group = np.ones(len(rec), dtype=bool)
    new_group[1:] = pre[1:] != pre[:-1]
    # decode every distinct key once, as one byte block of rows left|right|diag
    k = L + D + R
    text = codec.keys_to_matrix(keys[new_key], L, D, R, rna).tobytes().decode("ascii")
    groups, amp, row = [], None, 0
    for nk, ng, gi, cnt in zip(new_key.tolist(), new_group.tolist(), rec["genome"].tolist(),
                               rec["count"].tolist()):
        if nk:
            s = text[row * k:(row + 1) * k]
            row += 1
            amp = Amplicon(s[:L], s[L + R:], s[L:L + R], ())
            if ng:
                groups.append([])
            groups[-1].append(amp)
        amp.labels.extend([labels[gi]] * cnt)
    for g in groups:
        for a in g:
            a.labels.sort()
    return groups


def diagnostic_columns(group):
    """Amplicon.py:483-493: columns where the group's sequences differ."""
    return [i for i, col in enumerate(zip(*[a.diag for a in group])) if len(set(col)) > 1]


def ingroup_unique_columns(group, ingroup):
    """Amplicon.py:495-521 -- used here only to draw '#' in the bracket line."""
    if ingroup is None:
        return []
    ins, outs = [], []
    for a in group:
        for lab in a.labels:
            (ins if lab in ingroup else outs).append(a.diag)
    return [i for i in range(len(group[0].diag))
            if {d[i] for d in ins}.isdisjoint({d[i] for d in outs})]


def bracket_line(group, ingroup):
    """Amplicon.py:523-540."""
    start, dlen = len(group[0].left), len(group[0].diag)
    br = list(" " * (start - 1) + "{" + "-" * dlen + "}")
    for i in diagnostic_columns(group):
        br[start + i] = "*"
    for i in ingroup_unique_columns(group, ingroup):
        br[start + i] = "#"
    return "".join(br)


_PLAIN_BASES = frozenset("ACGTU")


def collapse_to_iupac(seqs):
    """Amplicon.py:42-66."""
    lens = [len(s) for s in seqs]
    width = max(lens)
    if len(set(lens)) != 1:
        return "-" * width
    first = seqs[0]
    if _PLAIN_BASES.issuperset(first) and all(s == first for s in seqs):
        return first              # one plain sequence: every column is its own consensus
    out = []
    for i in range(width):
        col = {s[i] for s in seqs}
        out.append("N" if col & {"*", "N", "?"} else IUPAC_KEY[tuple(sorted(col))])
    return "".join(out)


def render_csv_row(group, ingroup):
    """Amplicon.py:550-558, 663-671: consensus of the Amplicons whose labels are all
    ingroup (of all Amplicons when there is one, or no ingroup was given)."""
    if len(group) == 1 or ingroup is None:
        amps = group
    else:
        amps = [a for a in group if set(a.labels) <= ingroup]
    return ",".join(collapse_to_iupac([getattr(a, f) for a in amps]) for f in ("left", "diag", "right"))


def render_alignment(group, ingroup, dot):
    """Amplicon.py:598-661 without Primer3: rows stable-sorted by label LIST,
    ingroup-carrying rows first, then the bracket line (or dots)."""
    rows = sorted(group, key=lambda a: a.labels)
    if ingroup is not None:
        rows = [a for a in rows if set(a.labels) & ingroup] + \
               [a for a in rows if not (set(a.labels) & ingroup)]
    text = [str(a) for a in rows]
    if dot:
        top, width = text[0], len(group[0].sequence)
        text = [top] + ["".join("." if (i < width and c == top[i]) else c for i, c in enumerate(t))
                        for t in text[1:]]
    else:
        text.append(bracket_line(group, ingroup))
    text[-1] += "\n"
    return "\n".join(text)


CSV_HEADER = "left_seq,diag_seq,right_seq"      # outputAlignments.py:26-31


def render(groups, ingroup_labels, dot=False):
    """-> (csv_text, alignment_text) as render_output writes them at cores=1
    (outputAlignments.py:101-162): header, one row / block per group; each block is
    print()ed, hence the blank line after it."""
    ingroup = None if ingroup_labels is None else frozenset(ingroup_labels)
    csv = [CSV_HEADER]
    blocks = []
    for g in groups:
        blocks.append(render_alignment(g, ingroup, dot) + "\n")
        csv.append(render_csv_row(g, ingroup))
    return "\n".join(csv) + "\n", "".join(blocks)


def merged_lines(groups):
    """shared.py:419-438: one line per Amplicon."""
    return [a.line() for g in groups for a in g]
