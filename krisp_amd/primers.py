"""Primer3 hook of the renderer (`krisp_fasta --primer3`): SURVEY 8(f) rank 4.

The reference designs primers for every surviving group with the third-party primer3-py package
(Amplicon.py:103-151 run_primer3, :560-564 find_primers), drops the groups Primer3 finds no pair
for (outputAlignments.py:79-82), appends Primer3's figures to the CSV row
(outputAlignments.py:10-36, Amplicon.py:663-671) and annotates the alignment block
(Amplicon.py:631-656, :566-595).  primer3-py is not part of this image (no network), so this module
imports it lazily: `available()` says whether the hook can run, the command line exits with a
message when it cannot.  With the package present the calls and their parameters are the
reference's; the layout of the two statistics tables re-implements PrettyTable's border-less
format by hand (prettytable is not here either) and is NOT pinned by a golden vector.
"""
from statistics import mean

from . import amplicon

# the Primer3 output tags the reference puts into the CSV, in its column order (outputAlignments.py:10-22)
CSV_TAGS = [
    "PRIMER_PAIR_0_PRODUCT_SIZE", "PRIMER_PAIR_0_PENALTY",
    "PRIMER_LEFT_0_SEQUENCE", "PRIMER_RIGHT_0_SEQUENCE", "PRIMER_LEFT_0_PENALTY", "PRIMER_RIGHT_0_PENALTY",
    "PRIMER_LEFT_0_TM", "PRIMER_RIGHT_0_TM", "PRIMER_LEFT_0_GC_PERCENT", "PRIMER_RIGHT_0_GC_PERCENT",
    "PRIMER_LEFT_0_SELF_ANY_TH", "PRIMER_RIGHT_0_SELF_ANY_TH", "PRIMER_LEFT_0_SELF_END_TH", "PRIMER_RIGHT_0_SELF_END_TH",
    "PRIMER_LEFT_0_HAIRPIN_TH", "PRIMER_RIGHT_0_HAIRPIN_TH", "PRIMER_LEFT_0_END_STABILITY",
    "PRIMER_RIGHT_0_END_STABILITY", "PRIMER_PAIR_0_COMPL_ANY_TH", "PRIMER_PAIR_0_COMPL_END_TH",
]
CSV_COLUMNS = [t.replace("PRIMER_", "").replace("_0", "").lower() for t in CSV_TAGS]


def available():
    try:
        import primer3  # noqa: F401
        return True
    except ImportError:
        return False


def settings(tm=(53, 68), gc=(40, 70), amp_size=(80, 300), primer_size=(25, 35), max_sec_tm=40, gc_clamp=1,
             max_end_gc=4):
    """Primer3 global settings from the command line's figures (krisp_fasta.py:158-173 -> Amplicon.py:113-139)"""
    return {
        "PRIMER_TASK": "generic", "PRIMER_PICK_LEFT_PRIMER": 1, "PRIMER_PICK_RIGHT_PRIMER": 1,
        "PRIMER_LIBERAL_BASE": 1,
        "PRIMER_OPT_SIZE": mean(primer_size), "PRIMER_MIN_SIZE": primer_size[0], "PRIMER_MAX_SIZE": primer_size[1],
        "PRIMER_OPT_TM": mean(tm), "PRIMER_MIN_TM": tm[0], "PRIMER_MAX_TM": tm[1],
        "PRIMER_MIN_GC": gc[0], "PRIMER_MAX_GC": gc[1],
        "PRIMER_MAX_POLY_X": 4, "PRIMER_MAX_NS_ACCEPTED": 0, "PRIMER_THERMODYNAMIC_OLIGO_ALIGNMENT": 1,
        "PRIMER_MAX_SELF_ANY_TH": max_sec_tm, "PRIMER_MAX_SELF_END_TH": max_sec_tm,
        "PRIMER_PAIR_MAX_COMPL_ANY_TH": max_sec_tm, "PRIMER_PAIR_MAX_COMPL_END_TH": max_sec_tm,
        "PRIMER_MAX_HAIRPIN_TH": max_sec_tm,
        "PRIMER_PRODUCT_SIZE_RANGE": [amp_size], "PRIMER_GC_CLAMP": gc_clamp, "PRIMER_MAX_END_GC": max_end_gc,
    }


def design(group, ingroup, global_settings):
    """Primer3 on the ingroup consensus of one group, the diagnostic region as the target
    (Amplicon.py:560-564); None when Primer3 returns no pair"""
    import primer3
    amps = group if (len(group) == 1 or ingroup is None) else [a for a in group if set(a.labels) <= ingroup]
    template = "".join(amplicon.collapse_to_iupac([getattr(a, f) for a in amps]) for f in ("left", "diag", "right"))
    out = primer3.bindings.design_primers(
        {"SEQUENCE_TEMPLATE": template, "SEQUENCE_TARGET": [len(group[0].left), len(group[0].diag)]}, global_settings)
    return out if out.get("PRIMER_PAIR_NUM_RETURNED", 0) != 0 else None


def _table(header, rows):
    """left-aligned columns, one space of padding on either side, no border"""
    cells = [[str(round(x, 5)) if isinstance(x, float) else str(x) for x in r] for r in [header] + rows]
    width = [max(len(r[i]) for r in cells) for i in range(len(header))]
    return "\n".join(" " + "  ".join(c.ljust(w) for c, w in zip(r, width)).rstrip() for r in cells)


def stats_text(p3):
    """the two statistics tables under an alignment block (Amplicon.py:566-595)"""
    def part(prefix):
        return {k[len(prefix):]: v for k, v in p3.items() if prefix in k}
    left, right, pair = part("PRIMER_LEFT_0_"), part("PRIMER_RIGHT_0_"), part("PRIMER_PAIR_0_")
    title = lambda names: [n.title().replace("_", " ") for n in names]  # noqa: E731
    return ("\nPrimer statistics:\n" + _table(["Direction"] + title(left), [["Forward"] + list(left.values()),
                                                                          ["Reverse"] + list(right.values())])
            + "\n\nPair statistics:\n" + _table(title(pair), [list(pair.values())]))


def annotate(block_lines, p3, dot):
    """the Forward / Reverse marks under a block (Amplicon.py:631-650); block_lines end with the bracket
    line (or, with dots, the last sequence row)"""
    fwd, rev = p3["PRIMER_LEFT_0_SEQUENCE"], p3["PRIMER_RIGHT_0_SEQUENCE"]
    f0 = p3["PRIMER_LEFT_0"][0]
    r0 = p3["PRIMER_RIGHT_0"][0] - p3["PRIMER_RIGHT_0"][1]
    marks = (" " * f0 + "└" + "Forward".center(len(fwd) - 2, "─") + "┘"
             + " " * (r0 - f0 - len(fwd) + 1) + "└" + "Reverse".center(len(rev) - 2, "─") + "┘")
    if dot:
        return block_lines + [marks]
    last = block_lines[-1].ljust(len(marks))
    return block_lines[:-1] + ["".join(m if b == " " else b for b, m in zip(last, marks))]


def render(groups, ingroup_labels, global_settings, dot=False):
    """-> (csv_text, alignment_text) with Primer3: groups without a primer pair are left out"""
    ingroup = None if ingroup_labels is None else frozenset(ingroup_labels)
    csv = [amplicon.CSV_HEADER + "," + ",".join(CSV_COLUMNS)]
    blocks = []
    for g in groups:
        p3 = design(g, ingroup, global_settings)
        if p3 is None:
            continue
        lines = amplicon.render_alignment(g, ingroup, dot).rstrip("\n").split("\n")
        lines = annotate(lines, p3, dot)
        lines.append(stats_text(p3))
        blocks.append("\n".join(lines) + "\n\n")
        csv.append(amplicon.render_csv_row(g, ingroup) + "," + ",".join(str(p3[t]) for t in CSV_TAGS))
    return "\n".join(csv) + "\n", "".join(blocks)
