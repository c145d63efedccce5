"""Host glue after the device path: survivor records -> labelled groups -> text.

Restates the reference's record / label / ordering / rendering semantics so the
final CSV and alignment text are byte-identical at --cores 1
(krisp_fasta/Amplicon.py:154-348 Amplicon, 351-692 ConservedEndAmplicons;
outputAlignments.py:26-162).  The diagnostic filter itself runs on the device
(kr_intersect / kr_cands_merge); nothing here decides which groups survive.
"""

import gc

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


_LABEL_STRINGS = {}


class Amplicon:
    """One distinct sequence of a group + the sorted multiset of genome labels
    that carry it (Amplicon.py:154-210)."""
    __slots__ = ("left", "diag", "right", "labels")

    def __init__(self, left, diag, right, labels):
        self.left, self.diag, self.right = left, diag, right
        self.labels = sorted(labels)

    @classmethod
    def presorted(cls, left, diag, right, labels):
        """labels already in sorted order (a list this object may keep)"""
        a = cls.__new__(cls)
        a.left, a.diag, a.right, a.labels = left, diag, right, labels
        return a

    @property
    def sequence(self):
        return self.left + self.diag + self.right

    def label_string(self):
        """Amplicon.py:170-187: name or name(count), ';' joined, names sorted."""
        key = tuple(self.labels)
        hit = _LABEL_STRINGS.get(key)
        if hit is not None:
            return hit
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
        text = ";".join(out)
        if len(_LABEL_STRINGS) < 100_000:       # few distinct label lists recur over millions of rows
            _LABEL_STRINGS[key] = text
        return text

    def line(self):
        """merged-file line, Amplicon.py:330-348."""
        return f"{self.left},{self.diag},{self.right},{self.label_string()}"

    def __str__(self):
        return f"{self.sequence} : {self.label_string()}"


def _ordered_records(records, labels):
    """the records in (key, rank of the genome's label) order: every Amplicon's label list is then born sorted.
    kr_collect returns (key, position of the genome in the call) order: callers that list the genomes by label
    get records that need no sort (one vectorised check instead of a lexsort)."""
    by_name = np.argsort(np.array(labels, dtype=object), kind="stable")
    label_rank = np.empty(len(labels), dtype=np.int64)
    label_rank[by_name] = np.arange(len(labels))
    rk = label_rank[records["genome"].astype(np.int64)]
    kk = records["key"]
    in_order = len(kk) < 2 or bool(np.all((kk[1:] > kk[:-1]) | ((kk[1:] == kk[:-1]) & (rk[1:] >= rk[:-1]))))
    return records if in_order else records[np.lexsort((rk, kk))]


class RecordGroups:
    """The surviving groups as the device left them -- (key, genome, count) records -- behind the interface of the
    list of groups groups_from_records() makes of them.  render() turns it into text in the library, in one pass
    over the records (kr_render_records: no object per Amplicon); anything else that walks the
    groups -- the Primer3 hook, the merged-file writer, tests -- gets the list, built on first use."""

    def __init__(self, records, labels, L, D, R):
        self.records = _ordered_records(records, labels) if len(records) else records
        self.labels, self.L, self.D, self.R = list(labels), L, D, R
        self._groups = None
        if len(self.records):
            pre = self.records["key"] & codec.prefix_mask(L, R)
            self._n = 1 + int(np.count_nonzero(pre[1:] != pre[:-1]))
        else:
            self._n = 0

    def groups(self):
        if self._groups is None:
            self._groups = groups_from_records(self.records, self.labels, self.L, self.D, self.R)
        return self._groups

    def __len__(self):
        return self._n

    def __iter__(self):
        return iter(self.groups())

    def __getitem__(self, i):
        return self.groups()[i]

    def render_text(self, ingroup_labels, dot=False):
        """-> (csv, alignment) through the library, or None when it leaves a group to the general path"""
        from . import _native
        distinct = sorted(set(self.labels))
        rank = {t: i for i, t in enumerate(distinct)}
        label_of = np.array([rank[t] for t in self.labels], dtype=np.uint32)
        label_in = None if ingroup_labels is None else np.array([1 if t in ingroup_labels else 0 for t in distinct],
                                                                dtype=np.uint8)
        out = _native.render_records(self.records, label_of, distinct, label_in, self.L, self.D, self.R, dot)
        return None if out is None else (out[0], out[1])


class WindowGroups:
    """The surviving groups of amplicons longer than one key as the device left them -- the member windows as text rows
    (kr_wide_fetch_windows), their group numbers and genomes -- behind the interface of the list of groups.  render()
    turns it into text in the library (kr_render_windows: groups ordered by (left, right), members by (diag, label),
    no object per Amplicon: configs[2] with close relatives yields > 10^5 groups); whoever walks the groups gets the
    list, built on first use."""

    def __init__(self, rows, cand, genome, labels, L, D, R, rna=False):
        self.rows = np.ascontiguousarray(rows, dtype=np.uint8)
        self.cand = np.ascontiguousarray(cand, dtype=np.uint32)
        self.genome = np.ascontiguousarray(genome, dtype=np.uint32)
        self.labels, self.L, self.D, self.R, self.rna = list(labels), L, D, R, bool(rna)
        self._groups = None
        self._n = None

    def groups(self):
        if self._groups is None:
            self._groups = groups_from_windows(self.rows, self.genome, self.labels, self.L, self.D, self.R)
            if self.rna:
                for g in self._groups:
                    for a in g:
                        a.left, a.diag, a.right = (x.replace("T", "U") for x in (a.left, a.diag, a.right))
        return self._groups

    def __len__(self):
        if self._n is None:             # (the number of distinct group numbers; the renderer reports it as well)
            self._n = int(len(np.unique(self.cand)))
        return self._n

    def __iter__(self):
        return iter(self.groups())

    def __getitem__(self, i):
        return self.groups()[i]

    def render_text(self, ingroup_labels, dot=False):
        """-> (csv, alignment) through the library, or None when it leaves a group to the general path"""
        from . import _native
        if self.rna:
            return None         # (the reference's renderer dies on the first column that holds U: the general path reproduces that)
        distinct = sorted(set(self.labels))
        rank = {t: i for i, t in enumerate(distinct)}
        label_of = np.array([rank[t] for t in self.labels], dtype=np.uint32)
        label_in = None if ingroup_labels is None else np.array([1 if t in ingroup_labels else 0 for t in distinct],
                                                                dtype=np.uint8)
        out = _native.render_windows(self.rows, self.cand, self.genome, label_of, distinct, label_in, self.L, self.D,
                                     self.R, dot, self.rna)
        if out is None:
            return None
        self._n = out[2]
        return out[0], out[1]


def groups_from_windows(rows, genome, labels, L, D, R):
    """member windows as text rows (uint8 [n, L+D+R], line order) + the genome of each -> groups of Amplicon in the
    reference's order: groups by (left, right), sequences by diag, labels sorted (the general path of long amplicons)"""
    if len(rows) == 0:
        return []
    k = L + D + R
    row = np.empty((len(rows), k + 4), dtype=np.uint8)
    row[:, 0:L] = rows[:, :L]
    row[:, L:L + R] = rows[:, L + D:]
    row[:, L + R:k] = rows[:, L:L + D]
    row[:, k:] = np.asarray(genome).astype(">u4").view(np.uint8).reshape(-1, 4)
    uniq, counts = np.unique(row, axis=0, return_counts=True)
    groups, last_cand, last_seq = [], None, None
    for r, cnt in zip(uniq, counts):
        cand = bytes(r[0:L + R])
        seq = bytes(r[0:k])
        gi = int.from_bytes(bytes(r[k:]), "big")
        if cand != last_cand:
            groups.append([])
            last_cand, last_seq = cand, None
        if seq != last_seq:
            txt = seq.decode("ascii")
            groups[-1].append(Amplicon(txt[:L], txt[L + R:], txt[L:L + R], []))
            last_seq = seq
        groups[-1][-1].labels.extend([labels[gi]] * int(cnt))
    for g in groups:
        for a in g:
            a.labels.sort()
    return groups


def groups_from_records(records, labels, L, D, R, rna=False):
    """(key, genome, count) records of the survivors -> list of groups (lists of
    Amplicon), groups ascending by (left,right), Amplicons ascending by diag --
    the order every leaf file has (GNU sort's whole-line tie-break) and that
    ConservedEndAmplicons.add() preserves (Amplicon.py:448-481)."""
    if len(records) == 0:
        return []
    records = _ordered_records(records, labels)
    keys, genome, count = records["key"], records["genome"], records["count"]
    pm = codec.prefix_mask(L, R)
    new_key = np.ones(len(keys), dtype=bool)
    new_key[1:] = keys[1:] != keys[:-1]
    pre = keys & pm
    new_group = np.ones(len(keys), dtype=bool)
    new_group[1:] = pre[1:] != pre[:-1]
    # decode every distinct key once, as one byte block of rows left|right|diag
    k = L + D + R
    first = np.flatnonzero(new_key)
    text = codec.keys_to_matrix(keys[first], L, D, R, rna).tobytes().decode("ascii")
    # one label per k-mer occurrence, in record order; [lo, hi) of every distinct key in that list
    ends = np.cumsum(count.astype(np.int64))
    if int(count.max()) == 1:
        occ = [labels[g] for g in genome.tolist()]
    else:
        occ = [labels[g] for g, c in zip(genome.tolist(), count.tolist()) for _ in range(c)]
    lo = np.concatenate([[0], ends[first[1:] - 1]]).tolist()
    hi = np.concatenate([ends[first[1:] - 1], [ends[-1]]]).tolist()
    groups = []
    gc_was_on = gc.isenabled()
    gc.disable()          # millions of small acyclic objects: generational collections only rescan them
    try:
        for row, (ng, a, b) in enumerate(zip(new_group[first].tolist(), lo, hi)):
            s = text[row * k:(row + 1) * k]
            if ng:
                groups.append([])
            groups[-1].append(Amplicon.presorted(s[:L], s[L + R:], s[L:L + R], occ[a:b]))
    finally:
        if gc_was_on:
            gc.enable()
    return groups


def groups_from_records_mixed(records, labels, L, D, R, rna_genomes):
    """groups_from_records for a run that mixes DNA and RNA genomes (kstream.py:481-508, 599: an RNA genome's k-mers are
    written with U): a record's text carries its genome's letter for code 3, so one key may be two sequences -- '..T..'
    with the DNA genomes' labels, '..U..' with the RNA genomes' -- as the reference's text files hold them.  The result
    sets of such runs are small (only (left,right) pairs free of T / U are in every genome): a plain loop."""
    if len(records) == 0:
        return []
    records = _ordered_records(records, labels)
    pm = codec.prefix_mask(L, R)
    k = L + D + R
    text = codec.keys_to_matrix(records["key"], L, D, R, False).tobytes().decode("ascii")
    groups, last_pre = [], None
    for row, (key, g, cnt) in enumerate(zip(records["key"].tolist(), records["genome"].tolist(), records["count"].tolist())):
        s = text[row * k:(row + 1) * k]
        if rna_genomes[g]:
            s = s.replace("T", "U")
        pre = key & int(pm)
        if pre != last_pre:
            groups.append([])
            last_pre = pre
        left, diag, right = s[:L], s[L + R:], s[L:L + R]
        for a in groups[-1]:
            if a.diag == diag and a.left == left and a.right == right:
                a.labels.extend([labels[g]] * cnt)
                break
        else:
            groups[-1].append(Amplicon(left, diag, right, [labels[g]] * cnt))
    for grp in groups:
        for a in grp:
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


# str.translate table: what is left is not a plain base.  'U' is NOT plain: the reference's consensus table is Biopython's
# DNA table (Amplicon.py:10-12), a column holding U has no entry there and its renderer dies on it (see render below)
_NOT_PLAIN = {ord(c): None for c in "ACGT"}


def collapse_to_iupac(seqs):
    """Amplicon.py:42-66."""
    first = seqs[0] if seqs else None
    if first is not None and seqs.count(first) == len(seqs) and not first.translate(_NOT_PLAIN):
        return first              # one plain sequence: every column is its own consensus
    lens = [len(s) for s in seqs]
    width = max(lens)
    if len(set(lens)) != 1:
        return "-" * width
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
# The reference renders in a worker process that buffers PRINT_BLOCK groups between writes (outputAlignments.py:66-99;
# krisp_fasta.py:284-290 passes print_block=1000).  A group whose consensus has no entry in its IUPAC table -- a column
# holding 'U' (every RNA run: the table is the DNA one), or a lone ambiguity letter -- raises KeyError there
# (Amplicon.py:65): the worker dies with a traceback, the blocks written so far stay, the rest is lost, the command exits 0.
# Reproduced as it is (cores = 1): same bytes in the files, the notice on stderr.
PRINT_BLOCK = 1000


class RendererStopped(KeyError):
    pass


def render(groups, ingroup_labels, dot=False):
    """-> (csv_text, alignment_text) as render_output writes them at cores=1
    (outputAlignments.py:101-162): header, one row / block per group; each block is
    print()ed, hence the blank line after it."""
    ingroup = None if ingroup_labels is None else frozenset(ingroup_labels)
    if isinstance(groups, (RecordGroups, WindowGroups)):
        text = groups.render_text(ingroup, dot)
        if text is not None:
            return text
        groups = groups.groups()
    csv = [CSV_HEADER]
    blocks = []
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        brackets = {}
        for gi_, g in enumerate(groups):
          try:
                if ingroup is None and len(g) == 1:
                    # the bulk of a large result (conserved regions, no outgroup): one sequence, no
                    # variable column -- the same text as the general path below, without its calls
                    a = g[0]
                    seq = a.left + a.diag + a.right
                    line = seq + " : " + a.label_string()
                    if dot:
                        blocks.append(line + "\n\n")
                    else:
                        shape = (len(a.left), len(a.diag))
                        br = brackets.get(shape)
                        if br is None:
                            br = brackets[shape] = " " * (shape[0] - 1) + "{" + "-" * shape[1] + "}"
                        blocks.append(line + "\n" + br + "\n\n")
                    if not seq.translate(_NOT_PLAIN):
                        csv.append(a.left + "," + a.diag + "," + a.right)
                    else:
                        csv.append(render_csv_row(g, ingroup))
                    continue
                blocks.append(render_alignment(g, ingroup, dot) + "\n")
                csv.append(render_csv_row(g, ingroup))
          except KeyError as e:
            import sys
            kept = gi_ // PRINT_BLOCK * PRINT_BLOCK
            print(f"krisp_fasta: the reference's renderer stops at group {gi_ + 1} (KeyError: {e.args[0]!r} has no IUPAC "
                  f"consensus letter, Amplicon.py:65); as there, the {kept} groups of the blocks already written stay and "
                  f"the rest is lost", file=sys.stderr)
            del csv[1 + kept:]
            del blocks[kept:]
            break
    finally:
        if gc_was_on:
            gc.enable()
    return "\n".join(csv) + "\n", "".join(blocks)


def merged_lines(groups):
    """shared.py:419-438: one line per Amplicon."""
    return [a.line() for g in groups for a in g]
