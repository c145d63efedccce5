"""CPU ORACLE -- test infrastructure, NOT product code.

A plain-Python restatement of the reference's (grunwaldlab/krisp @ 2024_10_08)
`krisp_fasta` hot path at the level the reference itself works: text lines.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; krisp_amd/ never does.

Parity status: PINNED.  Every function below is checked in tests/test_oracle.py
against golden vectors produced by running the reference in the build container
(tests/golden/make_goldens.py) and against the README known answers
(README.md:121-124, 157-166, 172-185, 251-256).

Reference map (file:line into /root/reference/src/krisp):
  read_lines / parse_records    kstream/kstream.py:458-479, 510-583
  detect_rna                    kstream/kstream.py:481-508, 585-615
  kstream_lines (filter chain)  kstream/kstream.py:203-235, 617-832
  gnu_sort                      kstream/kstream.py:83-119  (LC_ALL=C sort -t, -kN,N ...)
  labels_to_string / parse_line krisp_fasta/Amplicon.py:170-206, 298-348
  read_groups                   krisp_fasta/shared.py:350-398, 442-475 ; Amplicon.py:448-481
  intersect_pair                krisp_fasta/shared.py:210-347
  merge_tree                    krisp_fasta/intersectAmplicons.py:232-310
  unique_columns / filter       krisp_fasta/Amplicon.py:495-521 ; filterAlignments.py:4-40
  render_*                      krisp_fasta/Amplicon.py:42-66, 523-558, 598-671 ; outputAlignments.py:26-162
  deduce_ldr / run_krisp_fasta  krisp_fasta/krisp_fasta.py:126-298
"""
import bz2
import gzip
import itertools
from pathlib import Path

# kstream.py:11-18
COMPLEMENT = dict(zip("ATatGCgcRYryMKmkSWswBVbvDHdhNn",
                      "TAtaCGcgYRyrKMkmSWswVBvbHDhdNn"))
# kstream.py:21-42
IUPAC_EXPAND = {"R": "AG", "Y": "CT", "S": "GC", "W": "AT", "K": "GT", "M": "AC",
                "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}
IUPAC_EXPAND.update({k.lower(): v.lower() for k, v in list(IUPAC_EXPAND.items())})

# Amplicon.py:10-12 with Biopython's IUPACData.ambiguous_dna_values (standard
# IUPAC table; README.md:122-123 pins AC->M, GT->K).  'X' and 'N' share ACGT and
# the later key ('N') wins, as in the reference's dict comprehension.
_AMBIG = {"A": "A", "C": "C", "G": "G", "T": "T", "M": "AC", "R": "AG", "W": "AT",
          "S": "CG", "Y": "CT", "K": "GT", "V": "ACG", "H": "ACT", "D": "AGT",
          "B": "CGT", "X": "GATC", "N": "GATC"}
CONSENSUS = {tuple(sorted(v)): k for k, v in _AMBIG.items()}
CONSENSUS[("?",)] = "N"


# ----------------------------------------------------------------------------
# A1/A2: file -> records
# ----------------------------------------------------------------------------
def read_lines(filename):
    """kstream.py:458-479 -- fileinput + hook_compressed: .gz / .bz2 by extension."""
    ext = Path(filename).suffix
    if ext == ".gz":
        with gzip.open(filename, "rb") as f:
            return [ln.decode() for ln in f]
    if ext == ".bz2":
        with bz2.open(filename, "rb") as f:
            return [ln.decode() for ln in f]
    with open(filename, "r") as f:
        return list(f)


def parse_records(lines, one_shot=True):
    """kstream.py:430-456, 510-583 -- FASTA iff the FIRST line contains '>'
    anywhere.  kstream.py:450 binds _detect_FASTA's re-chained stream to an unused
    name, so when the input is a one-shot iterator (a file, a generator) the
    line consumed by the detection is LOST; a list/tuple is re-iterated whole."""
    lines = list(lines)
    fasta = bool(lines) and ">" in lines[0]
    if one_shot:
        lines = lines[1:]
    if fasta:
        recs, cur = [], ""
        for ln in lines:
            ln = ln.strip()
            if ln.startswith(">"):
                if cur:
                    recs.append(cur)
                cur = ""
            else:
                cur += ln
        if cur:
            recs.append(cur)
        return recs
    return [ln.strip() for ln in lines]


def detect_rna(records):
    """kstream.py:481-508 -- first record holding T/t => DNA, else U/u => RNA."""
    for s in records:
        if "T" in s or "t" in s:
            return False
        if "U" in s or "u" in s:
            return True
    return None


# ----------------------------------------------------------------------------
# A3-A9: the generator chain
# ----------------------------------------------------------------------------
def revcomp(s):
    """kstream.py:644-659 (KeyError for characters outside COMP_MAP)."""
    return "".join(COMPLEMENT[c] for c in reversed(s))


def split_columns(s, sizes):
    """kstream.py:805-832 (size >= 0 cuts from the left, < 0 from the right)."""
    left, right = [], []
    for z in sizes:
        if z >= 0:
            left.append(s[:z])
            s = s[z:]
        else:
            right.append(s[z:])
            s = s[:z]
    return ",".join(left + [s] + right)


def gnu_sort(lines, cols=None):
    """kstream.py:83-119: LC_ALL=C sort [-t, -k{c+1},{c+1} ...]; GNU sort (no -s)
    breaks key ties with a whole-line byte compare."""
    if cols is None:
        return sorted(lines)

    def key(ln):
        f = ln.split(",")
        return tuple(f[c] if c < len(f) else "" for c in cols) + (ln,)
    return sorted(lines, key=key)


def kstream_lines(sequences, kmers=None, complements=False, canonicals=False,
                  allow=None, disallow=None, omitsoft=False, mapsoft=False,
                  expandiupac=False, split=None, sort=False, sortmem=None,
                  sortcols=None, sortnp=1, parallel=1):
    """kstream.__init__/__call__/write (kstream.py:130-405) as one list-valued
    function.  `sequences` is a filename or an iterable of strings."""
    if omitsoft and mapsoft:
        raise ValueError("can't omit and map soft masked nucleotides")
    if complements and canonicals:
        raise ValueError("canonicals conflicts with complements")
    if isinstance(sequences, str):
        recs = parse_records(read_lines(sequences), one_shot=True)
    else:
        recs = parse_records(sequences, one_shot=hasattr(sequences, "__next__"))
    rna = detect_rna(recs)
    if rna:
        recs = [s.replace("U", "T").replace("u", "t") for s in recs]
    out = recs
    if kmers is not None:
        ks = [kmers] if isinstance(kmers, int) else list(kmers)
        out = [s[i:i + k] for s in out for k in ks for i in range(len(s) - k + 1)]
    if omitsoft:
        out = [s for s in out if s.isupper()]
    if mapsoft:
        out = [s.upper() for s in out]
    if complements:
        out = [x for s in out for x in (s, revcomp(s))]
    if allow is not None:
        ok = set(allow)
        out = [s for s in out if set(s) <= ok]
    if disallow is not None:
        bad = set(disallow)
        out = [s for s in out if not (set(s) & bad)]
    if expandiupac:
        exp = []
        for s in out:
            pos = [i for i, c in enumerate(s) if c in IUPAC_EXPAND]
            if not pos:
                exp.append(s)
                continue
            t = list(s)
            for combo in itertools.product(*(IUPAC_EXPAND[s[i]] for i in pos)):
                for i, c in zip(pos, combo):
                    t[i] = c
                exp.append("".join(t))
        out = exp
    if canonicals:
        out = [min(s, revcomp(s)) for s in out]
    if split is not None:
        sizes = [split] if isinstance(split, int) else list(split)
        out = [split_columns(s, sizes) for s in out]
    if sort:
        out = gnu_sort(out, sortcols)
    if rna:
        out = [s.replace("T", "U").replace("t", "u") for s in out]
        if sort:   # write() sorts the RNA text (kstream.py:291-322); same order
            out = gnu_sort(out, sortcols)
    return out


def extract_sorted_kmers(fasta, L, R, k, omit):
    """krisp_fasta.py:16-43."""
    kw = dict(kmers=k, disallow="Nn", complements=True, split=[L, -R], sort=True,
              sortcols=[0, 2])
    kw["omitsoft" if omit else "mapsoft"] = True
    return kstream_lines(fasta, **kw)


# ----------------------------------------------------------------------------
# I1-I5: records, groups, pairwise intersect, merge tree
# ----------------------------------------------------------------------------
def basename(filename):
    """shared.py:34-55."""
    parts = Path(filename).name.split(".")
    while parts[-1] in ("gz", "bz2", "fna", "fasta", "fa", "ffn", "frn"):
        parts.pop()
    return ".".join(parts)


def simplename(filename):
    """shared.py:58-73 -- truncated at the first dot."""
    return basename(filename).split(".")[0]


def labels_to_string(labels):
    """Amplicon.py:170-187."""
    out = []
    for name in sorted(set(labels)):
        c = labels.count(name)
        out.append(name if c == 1 else f"{name}({c})")
    return ";".join(out)


def parse_line(line, tag):
    """Amplicon.py:298-328 -> [left, diag, right, sorted labels]."""
    f = line.strip().split(",")
    if len(f) not in (3, 4):
        raise ValueError(f"Unrecognised string format : {line}")
    if len(f) == 3:
        labels = [tag]
    else:
        labels = []
        for item in f[3].split(";"):
            item = item.strip()
            if "(" in item:
                name, mult = item.split("(")
                labels += [name] * int(mult.strip(")"))
            else:
                labels.append(item)
    return [f[0], f[1], f[2], sorted(labels)]


def _group_add(group, amp):
    """Amplicon.py:448-475 -- linear search on the concatenated sequence."""
    seq = amp[0] + amp[1] + amp[2]
    for a in group:
        if a[0] + a[1] + a[2] == seq:
            a[3] = sorted(a[3] + amp[3])
            return
    group.append([amp[0], amp[1], amp[2], list(amp[3])])


def read_groups(lines, tag):
    """shared.py:350-398 + 442-475: lines -> amplicons (adjacent duplicates fused)
    -> groups of consecutive amplicons sharing (left, right)."""
    amps = []
    for ln in lines:
        a = parse_line(ln, tag)
        if amps and amps[-1][0] + amps[-1][1] + amps[-1][2] == a[0] + a[1] + a[2]:
            amps[-1][3] = sorted(amps[-1][3] + a[3])
        else:
            amps.append(a)
    groups = []
    for a in amps:
        if groups and (groups[-1][0][0], groups[-1][0][2]) == (a[0], a[2]):
            _group_add(groups[-1], a)
        else:
            groups.append([a])
    return groups


def _pair_of(group):
    return (group[0][0], group[0][2])


def _simplify(groups):
    """shared.py:210-240 on groups: adjacent groups with one primer pair fuse."""
    out = []
    for g in groups:
        if out and _pair_of(out[-1]) == _pair_of(g):
            for a in g:
                _group_add(out[-1], a)
        else:
            out.append([list(a[:3]) + [list(a[3])] for a in g])
    return out


def intersect_pair(g0, g1):
    """shared.py:321-347: simplify both, 2-way merge (ties: stream0 first,
    shared.py:311), drop singletons (shared.py:243-282), fuse the pairs."""
    g0, g1 = _simplify(g0), _simplify(g1)
    merged, i, j = [], 0, 0
    while i < len(g0) or j < len(g1):
        if i >= len(g0) or (j < len(g1) and _pair_of(g1[j]) < _pair_of(g0[i])):
            merged.append(g1[j])
            j += 1
        else:
            merged.append(g0[i])
            i += 1
    kept = []
    for idx, g in enumerate(merged):
        p = _pair_of(g)
        if (idx > 0 and _pair_of(merged[idx - 1]) == p) or \
           (idx + 1 < len(merged) and _pair_of(merged[idx + 1]) == p):
            kept.append(g)
    return _simplify(kept)


def groups_to_lines(groups):
    """shared.py:419-438 + Amplicon.py:330-348."""
    return [f"{a[0]},{a[1]},{a[2]},{labels_to_string(a[3])}" for g in groups for a in g]


def merge_tree(files):
    """intersectAmplicons.py:232-310.  `files` = list of (tag, lines).  Pairs are
    popped from the END of the list; next level = results (taken here in job
    order; the reference takes completion order) + the odd leftover."""
    files = [(simplename(t), list(l)) for t, l in files]
    n = 0
    while len(files) > 1:
        pairs = []
        while len(files) > 1:
            pairs.append((files.pop(), files.pop()))
        results = []
        for (t0, l0), (t1, l1) in pairs:
            out = groups_to_lines(intersect_pair(read_groups(l0, t0), read_groups(l1, t1)))
            results.append((f"tmp{n}", out))   # tmp names never become labels:
            n += 1                             # every line already carries its own
        files = results + files
    tag, lines = files[0]
    return lines if n else list(lines)


# ----------------------------------------------------------------------------
# F1: diagnostic filter
# ----------------------------------------------------------------------------
def unique_columns(group, ingroup):
    """Amplicon.py:495-521."""
    if ingroup is None:
        return []
    ins, outs = [], []
    for a in group:
        for lab in a[3]:
            (ins if lab in ingroup else outs).append(a[1])
    cols = []
    for i in range(len(group[0][1])):
        if {d[i] for d in ins}.isdisjoint({d[i] for d in outs}):
            cols.append(i)
    return cols


def filter_lines(lines, ingroup):
    """filterAlignments.py:4-40 (tag is irrelevant: lines carry labels)."""
    groups = read_groups(lines, "merged_file")
    if len(ingroup):
        groups = [g for g in groups if unique_columns(g, frozenset(ingroup))]
    return groups_to_lines(groups)


# ----------------------------------------------------------------------------
# R1: rendering
# ----------------------------------------------------------------------------
def collapse(seqs):
    """Amplicon.py:42-66."""
    lens = [len(s) for s in seqs]
    width = max(lens)           # ValueError on an empty list, as in the reference
    if len(set(lens)) != 1:
        return "-" * width
    out = []
    for i in range(width):
        col = {s[i] for s in seqs}
        if "*" in col or "N" in col or "?" in col:
            out.append("N")
        else:
            out.append(CONSENSUS[tuple(sorted(col))])
    return "".join(out)


def render_csv_row(group, ingroup):
    """Amplicon.py:550-558, 663-671."""
    if len(group) == 1 or ingroup is None:
        amps = group
    else:
        amps = [a for a in group if set(a[3]) <= set(ingroup)]
    return ",".join(collapse([a[c] for a in amps]) for c in (0, 1, 2))


def render_alignment(group, ingroup, dot):
    """Amplicon.py:523-540, 598-661 (without primer3)."""
    rows = sorted(group, key=lambda a: a[3])          # stable, keyed on label LIST
    text = [f"{a[0]}{a[1]}{a[2]} : {labels_to_string(a[3])}" for a in rows]
    if ingroup is not None:
        isin = [bool(set(a[3]) & set(ingroup)) for a in rows]
        text = [t for t, f in zip(text, isin) if f] + [t for t, f in zip(text, isin) if not f]
    if dot:
        top = text[0]
        width = len(group[0][0] + group[0][1] + group[0][2])
        res = [top]
        for t in text[1:]:
            t = list(t)
            for i in range(width):
                if t[i] == top[i]:
                    t[i] = "."
            res.append("".join(t))
        text = res
    else:
        start = len(group[0][0])
        dlen = len(group[0][1])
        br = list(" " * (start - 1) + "{" + "-" * dlen + "}")
        diags = [a[1] for a in group]
        for i, col in enumerate(zip(*diags)):
            if len(set(col)) > 1:
                br[start + i] = "*"
        for i in unique_columns(group, None if ingroup is None else frozenset(ingroup)):
            br[start + i] = "#"
        text.append("".join(br))
    text[-1] += "\n"
    return "\n".join(text)


PRINT_BLOCK = 1000          # krisp_fasta.py:288


def render_output(lines, ingroup, dot, tag="merged_file"):
    """outputAlignments.py:101-162 at cores=1 -> (csv_text, align_text).  `tag`
    labels lines that carry none (a single input genome: the result file is
    the moved k-mer file, tag = simplename("merged_file.txt"))."""
    groups = read_groups(lines, tag)
    csv = ["left_seq,diag_seq,right_seq"]
    aln = []
    for i, g in enumerate(groups):
        try:
            a = render_alignment(g, ingroup, dot)
            c = render_csv_row(g, ingroup)
        except KeyError:
            # outputAlignments.py:66-99: the worker process that renders buffers print_block groups between writes
            # (krisp_fasta.py:284-290: 1000); a consensus column without an entry in the IUPAC table (Amplicon.py:65 -- 'U',
            # a lone ambiguity letter) raises there, the worker dies, what it had not written yet is lost, main() exits 0
            kept = i // PRINT_BLOCK * PRINT_BLOCK
            del csv[1 + kept:]
            del aln[kept:]
            break
        aln.append(a)
        csv.append(c)
    return "\n".join(csv) + "\n", "".join(a + "\n" for a in aln)


# ----------------------------------------------------------------------------
# driver
# ----------------------------------------------------------------------------
def deduce_ldr(conserved=None, conserved_left=None, conserved_right=None,
               diagnostic=None, amplicon=None):
    """krisp_fasta.py:179-213 -> (L, D, R, amplicon) or None (=> exit 1)."""
    if amplicon is not None:
        if diagnostic is not None:
            conserved = (amplicon - diagnostic) // 2
            conserved_left = conserved_right = conserved
        elif conserved is not None:
            diagnostic = amplicon - 2 * conserved
            conserved_left = conserved_right = conserved
        elif conserved_left is not None and conserved_right is not None:
            diagnostic = amplicon - conserved_left - conserved_right
        else:
            return None
    elif diagnostic is not None:
        if conserved is not None:
            amplicon = diagnostic + 2 * conserved
            conserved_left = conserved_right = conserved
        elif conserved_left is not None and conserved_right is not None:
            amplicon = diagnostic + conserved_left + conserved_right
        else:
            return None
    else:
        return None
    return conserved_left, diagnostic, conserved_right, amplicon


def run_krisp_fasta(ingroup_files, outgroup_files, L, D, R, amplicon=None,
                    omit_soft=False, dot=False):
    """krisp_fasta.py:224-290 -> dict(sorted, merged, filtered, csv, align).
    Note: the k passed to kstream is `amplicon`, which differs from L+D+R when
    --amplicon/--diagnostic leave an odd remainder (krisp_fasta.py:182)."""
    k = amplicon if amplicon is not None else L + D + R
    files = list(ingroup_files) + list(outgroup_files)
    sorted_files = []
    for f in files:
        sorted_files.append((f"{basename(Path(f).name)}.{k}mers",
                             extract_sorted_kmers(f, L, R, k, omit_soft)))
    merged = merge_tree(sorted_files)
    result = merged
    filtered = None
    if k > L + R:
        filtered = filter_lines(merged, [simplename(f) for f in ingroup_files])
        result = filtered
    ingroup = [simplename(f) for f in ingroup_files] if outgroup_files else None
    csv, align = render_output(result, ingroup, dot)
    return {"sorted": dict((t, l) for t, l in sorted_files), "merged": merged,
            "filtered": filtered, "csv": csv, "align": align}
