"""`krisp_fasta` -- same command line and stage functions as the reference
(krisp_fasta/krisp_fasta.py), with the k-mer generation / sort / multi-genome
intersection / diagnostic filter running on the MI355X through libkrisp_hip.so.

Seams kept:
  extractSortedKmers(fasta, primer_left, primer_right, ampl_len, output, sortmem,
                     parallel=1, verbose=True, omit=True)      krisp_fasta.py:16-66
  mergeFiles(files, output, parallel=1, workdir=None, verbose=True)
                                                               intersectAmplicons.py:232-310
  filterAlignments(kmerfile, output, ingroup)                  filterAlignments.py:31-40
  main()                                                       krisp_fasta.py:126-298
main() does not round-trip through text files: genomes are sorted once on the
device and intersected there (find_regions); the stage functions exist for
interoperability with the reference's intermediate files and for stage-level
parity tests.  There is no CPU implementation of the hot path behind any of them.
"""
import argparse
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

from . import amplicon, codec, fasta
from .kstream import kstream

_FASTA_EXT = ("gz", "bz2", "fna", "fasta", "fa", "ffn", "frn")


def basename(filename):
    """shared.py:34-55: file name without fasta / compression endings."""
    parts = Path(filename).name.split(".")
    while parts[-1] in _FASTA_EXT:
        parts.pop()
    return ".".join(parts)


def simplename(filename):
    """shared.py:58-73: ... truncated at the first dot -- the genome's label."""
    return basename(filename).split(".")[0]


def prettyTime(t):
    """shared.py:8-31."""
    if t < 60:
        return f"{t:.2f} seconds"
    minutes = int(t / 60)
    seconds = math.ceil(t - 60 * minutes)
    return (f"{minutes} minute{'s' if minutes > 1 else ''} and "
            f"{seconds} second{'s' if seconds > 1 else ''}")


class UnsupportedGeometry(NotImplementedError):
    pass


class MixedAlphabet(NotImplementedError):
    """DNA and RNA genomes in one run: the reference compares their k-mers as text ('T' never
    equals 'U', kstream.py:481-508, 599), which the 2-bit device alphabet cannot express."""


def _to_rna(groups):
    """kstream writes an RNA genome's k-mers back with U (kstream.py:297, 599)"""
    for g in groups:
        for a in g:
            a.left, a.diag, a.right = (x.replace("T", "U") for x in (a.left, a.diag, a.right))
    return groups


def _mixed_prefix_filter(eng, L, R):
    """DNA and RNA genomes in one run: the reference compares text, and 'T' never equals 'U' (kstream.py:481-508, 599 writes
    an RNA genome's k-mers with U; shared.py:321-347 merges on the (left,right) strings) -- a (left,right) pair that holds
    code 3 is in no DNA genome AND RNA genome alike.  The device matched them (one code for both letters): those candidates
    go.  Returns the number left (the list is replaced on the device when any went)."""
    cands = eng.cands()
    if not len(cands):
        return 0
    lr = L + R
    top = np.uint64((~0 << (64 - 2 * lr)) & 0xFFFFFFFFFFFFFFFF) if lr < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    p = cands["prefix"]
    threes = (p >> np.uint64(1)) & p & np.uint64(0x5555555555555555) & top
    keep = threes == 0
    if not keep.all():
        cands = cands[keep]
        eng.load_cands(cands)
    return len(cands)


def _mixed_finish(records, labels, geo, rna, ingroup_labels, do_filter):
    """... and the exact filter on the survivors of the device's mode-2 filter (k_intersect.inc: passes_filter -- a column
    whose ingroup and outgroup sets share nothing but code 3 passed there): here each genome's alphabet is known, the
    groups are text, and the reference's own predicate decides (Amplicon.py:495-521 via filterAlignments.py:4-40)"""
    Le, De, Re = geo
    groups = amplicon.groups_from_records_mixed(records, labels, Le, De, Re, rna)
    if do_filter and len(ingroup_labels):
        ing = frozenset(ingroup_labels)
        groups = [g for g in groups if amplicon.ingroup_unique_columns(g, ing)]
    return groups


def _check_geometry(L, D, R):
    """geometries of the packed path: the whole amplicon in one 64-bit key"""
    k = L + D + R
    if k > 32:
        raise UnsupportedGeometry(
            f"amplicon length {k} > 32: only the fused flow (find_regions / the command line) "
            "carries amplicons longer than one 64-bit key")
    if D > 16:
        raise UnsupportedGeometry(f"diagnostic length {D} > 16 exceeds the device mask format")


def _is_wide(L, D, R):
    return L + D + R > 32 or D > 16


def _check_wide(L, D, R):
    """geometries of the wide path (kr_wide_run): flanks of up to WIDE_MAX_FLANK bases (one key each, or -- longer than 32
    bases -- ranked piece by piece through up to eight keys), amplicons of up to WIDE_MAX_K"""
    from . import _native
    if not (1 <= L <= _native.WIDE_MAX_FLANK and 1 <= R <= _native.WIDE_MAX_FLANK and L + D + R <= _native.WIDE_MAX_K):
        raise UnsupportedGeometry(
            f"{L}/{D}/{R}: amplicons longer than 32 bases need 1 <= conserved-left, conserved-right "
            f"<= {_native.WIDE_MAX_FLANK} and a length <= {_native.WIDE_MAX_K}")


_COMP_U8 = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP_U8[_a] = _b


def _groups_from_hits(hits, texts, labels, L, D, R, rna_genomes=None):
    """kr_wide_run hits (group, genome, position, strand) -> groups of amplicon.Amplicon in the
    reference's order: groups by (left,right), sequences by diag; the window text is cut from
    the genome the host already holds (soft-mask mapped, kstream.py:622-642; reverse
    complemented for strand 1, kstream.py:644-659)."""
    if len(hits) == 0:
        return []
    k = L + D + R
    ar = np.arange(k, dtype=np.int64)
    rows = []
    for gi, text in enumerate(texts):
        sel = hits[hits["genome"] == gi]
        if len(sel) == 0:
            continue
        t = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        W = t[sel["pos"].astype(np.int64)[:, None] + ar] & np.uint8(0xDF)
        rc = sel["strand"] == 1
        W[rc] = _COMP_U8[W[rc][:, ::-1]]
        if rna_genomes is not None and rna_genomes[gi]:
            W[W == ord("T")] = ord("U")                 # (an RNA genome's k-mers are written with U: kstream.py:599)
        # rows order as the reference's merged file does: (left, right) groups, sequences by diag inside
        # (the device's group numbers need not ascend with (left, right): KR_OPT_WIDE_ORDERED)
        row = np.empty((len(sel), k + 4), dtype=np.uint8)
        row[:, 0:L] = W[:, :L]
        row[:, L:L + R] = W[:, L + D:]
        row[:, L + R:k] = W[:, L:L + D]
        row[:, k:] = np.full(len(sel), gi, dtype=">u4").view(np.uint8).reshape(-1, 4)
        rows.append(row)
    uniq, counts = np.unique(np.concatenate(rows), axis=0, return_counts=True)
    groups, last_cand, last_seq = [], None, None
    for row, cnt in zip(uniq, counts):
        cand = bytes(row[0:L + R])
        seq = bytes(row[0:k])
        gi = int.from_bytes(bytes(row[k:]), "big")
        if cand != last_cand:
            groups.append([])
            last_cand, last_seq = cand, None
        if seq != last_seq:
            txt = seq.decode("ascii")
            groups[-1].append(amplicon.Amplicon(txt[:L], txt[L + R:], txt[L:L + R], []))
            last_seq = seq
        groups[-1][-1].labels.extend([labels[gi]] * int(cnt))
    for g in groups:
        for a in g:
            a.labels.sort()
    return groups


# ----------------------------------------------------------------------------
# groups touched by k-mers that hold IUPAC ambiguity letters
# ----------------------------------------------------------------------------
_ACGT = frozenset("ACGT")


def _pure(s):
    return _ACGT.issuperset(s)


def _prefix_key(left, right):
    key = 0
    for i, ch in enumerate(left + right):
        key |= "ACGT".index(ch) << (62 - 2 * i)
    return key


def _special_groups(eng, ids, labels, specials, geo, ingroup, do_filter, rna_genomes=None):
    """The reference keeps k-mers with IUPAC letters (kstream.py:11-18); the device cannot pack
    them, so they arrive here as strings: specials[g] = [(left, diag, right), ...] of genome g.
    Every (left,right) group they touch is rebuilt exactly: its ACGT members are looked up on
    the device (kr_cands_load + kr_collect), the IUPAC members added, then the reference's
    rules applied -- present in every genome (intersectAmplicons.py:232-310), ingroup-unique
    column (Amplicon.py:495-521).  Rare by nature; everything else stays on the device.
    Returns (set of touched (left,right), list of surviving groups)."""
    from . import _native
    L, D, R = geo
    touched = {(l, r) for sp in specials for (l, d, r) in sp}
    if not touched:
        return touched, []
    pure = sorted({_prefix_key(l, r) for (l, r) in touched if _pure(l) and _pure(r)})
    recs = None
    if pure:
        cands = np.zeros(len(pure), dtype=_native.CAND)
        cands["prefix"] = np.array(pure, dtype=np.uint64)
        eng.load_cands(cands)
        recs = eng.collect(ids)
    return touched, _special_groups_from(recs, ids, labels, specials, geo, ingroup, do_filter, rna_genomes)


def _special_groups_from(recs, ids, labels, specials, geo, ingroup, do_filter, rna_genomes=None):
    """the groups IUPAC windows touch, from the device records of their ACGT members (`recs`: of one context, or
    gathered from every rank of a multi-GPU run) and the IUPAC members themselves (specials[g] of genome ids[g]).
    rna_genomes (a run that mixes DNA and RNA genomes): the text of an RNA genome's members carries U -- the reference
    compares text, so a group is a (left,right) pair every genome holds letter for letter"""
    L, D, R = geo
    members = {}            # (left,right) -> {(left,diag,right) -> {genome index -> count}}

    def text_of(gi, l, d, r):
        if rna_genomes is not None and rna_genomes[gi]:
            return l.replace("T", "U"), d.replace("T", "U"), r.replace("T", "U")
        return l, d, r
    if recs is not None:
        for rec in recs:
            gi = ids.index(int(rec["genome"]))
            l, d, r = text_of(gi, *codec.key_columns(rec["key"], L, D, R))
            members.setdefault((l, r), {}).setdefault((l, d, r), {})
            m = members[(l, r)][(l, d, r)]
            m[gi] = m.get(gi, 0) + int(rec["count"])
    for gi, sp in enumerate(specials):
        for (l, d, r) in sp:
            l, d, r = text_of(gi, l, d, r)
            m = members.setdefault((l, r), {}).setdefault((l, d, r), {})
            m[gi] = m.get(gi, 0) + 1
    groups = []
    for P in sorted(members):
        seqs = members[P]
        present = set()
        for m in seqs.values():
            present.update(m)
        if len(present) != len(ids):
            continue
        group = []
        for seq in sorted(seqs, key=lambda t: t[1]):
            labs = []
            for gi, cnt in seqs[seq].items():
                labs += [labels[gi]] * cnt
            group.append(amplicon.Amplicon(seq[0], seq[1], seq[2], labs))
        if do_filter and not amplicon.ingroup_unique_columns(group, ingroup):
            continue
        groups.append(group)
    return groups


def _wide_probe_members(eng, texts, gis, probes, probe_text, geo, pid):
    """the ACGT members of the touched (left,right) pairs with plain flanks in the genomes gis (texts[i] = genome gis[i]):
    one kr_wide_run per genome against a probe genome made of the pairs (left + A..A + right) returns every window of
    that genome whose flanks are one of them.  -> {(left,right): {(left,diag,right): {genome: count}}}"""
    from . import _native
    L, D, R = geo
    members = {}
    if not probes:
        return members
    wanted = set(probes)
    eng.upload(pid, probe_text)
    for gi, text in zip(gis, texts):
        nh = eng.wide_run([pid, gi], [True, True], apply_filter=False)
        if not nh:
            continue
        hits = eng.wide_fetch(_native.WIDE_HITS)
        hits = hits[hits["genome"] == 1].copy()
        hits["genome"] = 0
        for g in _groups_from_hits(hits, [text], ["x"], L, D, R):
            for a in g:
                if (a.left, a.right) in wanted:
                    m = members.setdefault((a.left, a.right), {}).setdefault((a.left, a.diag, a.right), {})
                    m[gi] = m.get(gi, 0) + len(a.labels)
    return members


def _special_groups_wide_from(members, n, labels, specials, touched, ingroup, do_filter, rna_genomes=None):
    """the groups IUPAC windows touch, from the ACGT members the device found (_wide_probe_members, of one context or
    merged over the ranks) and the IUPAC members themselves (specials[g] of genome g), under the reference's rules:
    present in every genome (intersectAmplicons.py:232-310), an ingroup-unique column (Amplicon.py:495-521)"""
    for gi, sp in enumerate(specials):
        for (l, d, r) in sp:
            m = members.setdefault((l, r), {}).setdefault((l, d, r), {})
            m[gi] = m.get(gi, 0) + 1
    if rna_genomes is not None:
        # (a run that mixes DNA and RNA genomes: every member as the text its genome's file holds -- U for an RNA genome --,
        # grouped again on that text; pairs that hold T / U then lack the genomes of the other alphabet)
        conv = {}
        for P, seqs in members.items():
            if P not in touched:
                continue
            for seq, m in seqs.items():
                for gi, cnt in m.items():
                    l, d, r = ((x.replace("T", "U") for x in seq) if rna_genomes[gi] else seq)
                    mm = conv.setdefault((l, r), {}).setdefault((l, d, r), {})
                    mm[gi] = mm.get(gi, 0) + cnt
        members = conv
        touched = set(conv)
    groups = []
    for P in sorted(members):
        if P not in touched:
            continue
        seqs = members[P]
        present = set()
        for m in seqs.values():
            present.update(m)
        if len(present) != n:
            continue
        group = []
        for seq in sorted(seqs, key=lambda t: t[1]):
            labs = []
            for gi, cnt in sorted(seqs[seq].items()):
                labs += [labels[gi]] * cnt
            group.append(amplicon.Amplicon(seq[0], seq[1], seq[2], sorted(labs)))
        if do_filter and not amplicon.ingroup_unique_columns(group, ingroup):
            continue
        groups.append(group)
    return groups


def _special_groups_wide(eng, texts, labels, specials, touched, probes, probe_text, geo, ingroup, do_filter, rna_genomes=None):
    """Wide-path twin of _special_groups.  A touched (left,right) pair with an IUPAC letter in a
    flank can only hold IUPAC k-mers (the host has them all); a pair with plain flanks also
    holds ACGT windows, which the device locates (_wide_probe_members)."""
    n = len(texts)
    members = _wide_probe_members(eng, texts, list(range(n)), probes, probe_text, geo, pid=n)
    return _special_groups_wide_from(members, n, labels, specials, touched, ingroup, do_filter, rna_genomes)


def _merge_groups(device_groups, touched, special_groups):
    """device groups minus the ones re-evaluated on the host, plus those; (left,right) byte order"""
    out = [g for g in device_groups if (g[0].left, g[0].right) not in touched] + special_groups
    out.sort(key=lambda g: (g[0].left, g[0].right))
    return out


RESERVE_MIN = 128 << 20      # bytes of file text from which the context's memory is made ahead of the uploads (kr_reserve)
STREAM_MIN = 1 << 30         # ... and from which the files are read one after the other, each on all host threads


def _find_regions_device_ingest(files, ingroup_files, L, R, k, geo, omit_soft, device, verbose, do_filter, quirk_all_fail,
                                workers, t0):
    """find_regions' packed path; a genome set the plan thought would fit and the device then could not hold (the plan
    is an estimate: candidate lists, slices of a skewed genome, another process on the device) is run again in batches"""
    from . import _native
    try:
        return _device_ingest_flow(files, ingroup_files, L, R, k, geo, omit_soft, device, verbose, do_filter, quirk_all_fail,
                                   workers, t0, None)
    except _native.KrispHipError as e:
        if e.code != _native.ERR_CAPACITY or len(files) < 2 or os.environ.get("KRISP_STREAM_BATCH"):
            raise
        if verbose:
            print(f"=> the genome set does not fit the device at once ({e}): in batches", file=sys.stderr)
    return _device_ingest_flow(files, ingroup_files, L, R, k, geo, omit_soft, device, verbose, do_filter, quirk_all_fail,
                               workers, t0, (len(files) + 1) // 2)


def _device_ingest_flow(files, ingroup_files, L, R, k, geo, omit_soft, device, verbose, do_filter, quirk_all_fail,
                        workers, t0, force_batch):
    """find_regions' packed path with the reader on the device: the host threads read and inflate, the main thread
    hands each text to the GPU (parse -> sort) as it arrives.  Same results as the host-parse flow (the device reader
    equals kr_fasta_to_bases byte for byte: tests/test_gpu_kernels.py)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import _native
    Le, De, Re = geo
    if len(files) == 1:
        labels = ["merged_file"]
    else:
        labels = [simplename(f) for f in files]
    ingroup_labels = frozenset(simplename(f) for f in ingroup_files)
    flags = [lab in ingroup_labels for lab in labels]
    try:
        est = max(fasta.estimate_text_bytes(f) for f in files)
    except OSError:
        est = 0
    ahead = int(os.environ.get("KRISP_RESERVE_MIN", RESERVE_MIN)) <= est < (1 << 32) - 128 and os.environ.get("KRISP_RESERVE") != "0"
    # files whose inflate takes every host thread by itself (one large gzip member, a bz2 stream of many blocks) are read
    # one after the other: the first genome is parsed and sorted on the device while the second still inflates
    # (plain text files use no inflate threads: they keep their readers side by side -- ADVICE r4)
    if ahead and est >= int(os.environ.get("KRISP_STREAM_MIN", STREAM_MIN)) and \
            any(str(f).lower().endswith((".gz", ".bz2", ".bgz")) for f in files):
        workers = 1
    budget = int(os.environ.get("KRISP_HBM_BUDGET", "0"))       # (bytes; tests and shared devices: kr_create's HBM budget)
    with ThreadPoolExecutor(max_workers=workers) as pool, _native.Engine(device=device, hbm_budget=budget) as eng:
        # a genome set the device cannot hold sorted at once goes through it in batches (the streaming flow above)
        batch = force_batch or (_plan_batch(eng, len(files), est) if len(files) > 1 else None)
        if batch is not None:
            t1 = time.time()
            eng.set_params(Le, De, Re, omit_soft=omit_soft, max_bases=max(est, 1))
            records, touched, sgroups, all_rna, stats = _find_regions_streaming(
                eng, pool, files, labels, flags, k, geo, omit_soft, verbose, do_filter, quirk_all_fail, batch, t0, max(est, 1))
            stats["device_s"] = time.time() - t1
            if quirk_all_fail:
                return [], stats
            if stats.get("mixed_rna"):
                groups = _mixed_finish(records, labels, (Le, De, Re), stats["mixed_rna"], ingroup_labels, do_filter)
                if touched:
                    groups = _merge_groups(groups, touched, sgroups)
                stats["candidates"] = len(groups)
                return groups, stats
            if not touched and not all_rna:
                return amplicon.RecordGroups(records, labels, Le, De, Re), stats
            groups = amplicon.groups_from_records(records, labels, Le, De, Re, rna=False)
            if touched:
                groups = _merge_groups(groups, touched, sgroups)
            return (_to_rna(groups) if all_rna else groups), stats
        futures = [pool.submit(fasta.read_text, f) for f in files]
        # large genomes: the context gets its memory (device memory another process has just given back takes the driver 15-40 ms per GB to hand over: seconds at 3 Gbp) while the
        # host threads read and inflate -- planned from the files' sizes, planned again below if they said too little
        planned = 0
        if ahead:
            # kr_reserve is an optimisation (include/krisp_hip.h), and the plan is an ESTIMATE from the files' sizes (5 x
            # the compressed bytes where real sequence text is 3.3-4.5 x): a reservation that does not fit says nothing
            # about the run itself -- the reserved buffers go back and the unplanned flow below sizes the context from the
            # texts once they are read
            planned = est
            try:
                eng.set_params(Le, De, Re, omit_soft=omit_soft, max_bases=planned)
                eng.reserve(list(range(len(files))), planned, with_text=True)
            except _native.KrispHipError as e:
                if verbose:
                    print(f"=> memory plan for {planned:,} bases per genome not taken ({e}): sizing from the texts instead", file=sys.stderr)
                for i in range(len(files)):
                    try:
                        eng.free(i)
                    except _native.KrispHipError:
                        pass
                planned = 0
        rna, specials = [], []
        first, t1 = [], None
        for i, fu in enumerate(futures):
            text, universal = fu.result()
            if t1 is None:
                read_s = time.time() - t0           # (until the first text is there: from then on the device has work)
                t1 = time.time()
            if planned and len(text) <= planned and not first:
                # the plan stands: this genome goes to the device now, the files behind it are still being read
                _n, r, sp = fasta.ingest_on_device(eng, i, text, universal, k, omit_soft)
                rna.append(r)
                specials.append([codec.split_window(w, Le, De, Re) for w in sp])
                eng.sort(i)
                del text
            else:
                first.append((text, universal))     # (no plan, or a file larger than planned: all sizes first)
        if first:
            done = len(files) - len(first)
            if done:
                # (a later file said too little about itself: everything again under a plan that holds them all -- the
                # texts already handed over are read once more)
                first = [fasta.read_text(f) for f in files[:done]] + first
                rna, specials = [], []
            true_max = max(max(len(t) for t, _ in first), 1)
            for i in (range(len(files)) if planned else ()):
                eng.free(i)
            eng.set_params(Le, De, Re, omit_soft=omit_soft, max_bases=true_max)
            for i, (text, universal) in enumerate(first):
                _n, r, sp = fasta.ingest_on_device(eng, i, text, universal, k, omit_soft)
                rna.append(r)
                specials.append([codec.split_window(w, Le, De, Re) for w in sp])
                eng.sort(i)
        t2 = time.time()
        del first
        t3 = time.time()
        mixed = any(rna) and not all(rna)
        if mixed:
            eng.set_mixed_alphabets(True)
        finish = _to_rna if all(rna) else (lambda groups: groups)
        ids = list(range(len(files)))
        ncand = eng.intersect(ids, flags, apply_filter=do_filter and not quirk_all_fail)
        if mixed and ncand:
            ncand = _mixed_prefix_filter(eng, Le, Re)
        counts = [eng.count(i) for i in ids]
        t4 = time.time()
        if verbose:
            for f, cnt in zip(files, counts):
                print(f"=> Extracted and sorted {cnt:,} {k}-kmers from {f}", file=sys.stderr)
        by_label = sorted(ids, key=lambda i: labels[i])
        records = eng.collect(by_label) if (ncand and not quirk_all_fail) else np.empty(0, dtype=_native.RECORD)
        touched, sgroups = set(), []
        if any(specials) and not quirk_all_fail:
            touched, sgroups = _special_groups(eng, ids, labels, specials, (Le, De, Re), ingroup_labels, do_filter,
                                               rna_genomes=rna if mixed else None)
        t5 = time.time()
    # (device_s ends where the context is gone: its buffers freed, the reader threads joined)
    stats = {"read_s": read_s, "device_s": time.time() - t1,
             "kmers": int(sum(counts)) + sum(len(sp) for sp in specials), "candidates": int(ncand),
             "stage_s": {"texts arrive, parse + sort launches": t2 - t1, "texts freed": t3 - t2, "sorts + intersect": t4 - t3,
                         "collect": t5 - t4, "context freed": time.time() - t5}}
    if quirk_all_fail:
        return [], stats
    if mixed:
        groups = _mixed_finish(records, labels, (Le, De, Re), rna, ingroup_labels, do_filter)
        if touched:
            groups = _merge_groups(groups, touched, sgroups)
        stats["candidates"] = len(groups)
        return groups, stats
    if not touched and not any(rna):
        return amplicon.RecordGroups(records, labels, Le, De, Re), stats
    groups = amplicon.groups_from_records(records, labels, Le, De, Re, rna=False)
    if touched:
        groups = _merge_groups(groups, touched, sgroups)
    return finish(groups), stats


# ----------------------------------------------------------------------------
# genome sets larger than one GPU's memory: the streaming flow (SURVEY section 7 "HBM sizing")
# ----------------------------------------------------------------------------
STREAM_EAGER_MAX = 4_000_000     # running candidates up to which a batch's records are collected at once (64 MB per genome)


def _plan_batch(eng, nfiles, est_bases):
    """How many genomes of ~est_bases the context can hold sorted at once beside its scratch (None: all of them).  Per
    genome: the bases and two 8-byte keys per base (+ 2 % slack for key-space slices); scratch: codes + pass-1 output
    per sort lane (three at most), the pass-0 array of two genome lanes when the genome is sorted in slices (> 2^28
    bases), the text buffer of the device reader, candidate buffers.  KRISP_STREAM_BATCH=n forces batches of n."""
    forced = os.environ.get("KRISP_STREAM_BATCH")
    if forced:
        return max(1, int(forced)) if int(forced) < nfiles else None
    if est_bases <= 0:
        return None
    per_genome = 17.4 * est_bases
    sliced = est_bases > (1 << 28)
    scratch = (2 * 16.2 * est_bases + 3 * 17.0 * est_bases / 16) if sliced else 3 * 17.0 * est_bases
    scratch += est_bases + (2 << 30)
    avail = eng.mem_info()["avail"]
    fit = int((avail - scratch) // per_genome)
    if fit >= nfiles:
        return None
    return max(1, fit)


def _find_regions_streaming(eng, pool, files, labels, flags, k, geo, omit_soft, verbose, do_filter, quirk_all_fail, batch, t0,
                            max_bases):
    """find_regions for a genome set that does not fit the GPU at once (the reference has no such limit: it sorts in
    external memory, kstream.py:108-119, and merges files pairwise, intersectAmplicons.py:232-310).  The genomes go
    through the device in batches of `batch`: sort the batch; the first batch is intersected (with the diagnostic filter:
    the predicate is monotone, so partial masks may prune -- in and out genomes are interleaved so that the first batch
    holds both), every later batch looks the running candidates up in its genomes (kr_cands_probe: presence, masks OR-ed,
    filter); collect the batch's records of the running candidates (a superset of the final ones) and free the batch.  The records of
    the final candidates are what remains of those; only when the running set was too large to collect from (no
    outgroup, no filter: every conserved pair is a candidate) or IUPAC windows touch groups whose ACGT members must be
    looked up, a second pass sorts every batch again and collects then.  Same records, same order, same text as the
    in-core flow (tests/test_gpu_cli.py)."""
    from . import _native
    Le, De, Re = geo
    n = len(files)
    ing = [i for i in range(n) if flags[i]]
    outg = [i for i in range(n) if not flags[i]]
    order = []
    for j in range(max(len(ing), len(outg))):
        order += ing[j:j + 1] + outg[j:j + 1]
    apply_f = do_filter and not quirk_all_fail
    rank_of = {g: r for r, g in enumerate(sorted(range(n), key=lambda i: labels[i]))}
    rna, specials, counts = [None] * n, [None] * n, [0] * n
    pmask = np.uint64((~0 << (64 - 2 * (Le + Re))) & 0xFFFFFFFFFFFFFFFF) if Le + Re < 32 else np.uint64(0xFFFFFFFFFFFFFFFF)
    stats = {"streamed": True, "batch": batch, "passes": 1, "batches": 0}
    read_s = [None]
    maxb = [max_bases]

    # the first batch must hold both sides when there are two: with one side only nothing prunes, and its candidate list
    # is every (left,right) pair of its genomes (more than 2^32 entries at 3 Gbp -- the library refuses such a
    # list, kr_intersect: KR_ERR_CAPACITY)
    first_min = 2 if (ing and outg and apply_f) else 1

    def batches(bsz):
        a = 0
        while a < n:
            m = max(bsz, first_min) if a == 0 else bsz
            yield order[a:a + m]
            a += m

    def load_batch(ids_b, futs, first_pass):
        """texts -> device, sorted; a batch that does not fit after all is given back whole (KrispHipError, code capacity)"""
        done = []
        texts = {g: futs[g].result() for g in ids_b}
        if read_s[0] is None:
            read_s[0] = time.time() - t0
        biggest = max(len(t) for t, _ in texts.values())
        if biggest > maxb[0]:
            # (the plan came from the files' sizes: a text longer than it said -- nothing is resident between batches)
            maxb[0] = biggest
            eng.set_params(Le, De, Re, omit_soft=omit_soft, max_bases=biggest)
        try:
            for g in ids_b:
                text, universal = texts.pop(g)
                done.append(g)          # (before the ingest: a genome that fails half way holds buffers, too -- ADVICE r5)
                _n, r, sp = fasta.ingest_on_device(eng, g, text, universal, k, omit_soft)
                del text
                if first_pass:
                    rna[g] = r
                    specials[g] = [codec.split_window(w, Le, De, Re) for w in sp]
                eng.sort(g)
            for g in ids_b:
                counts[g] = eng.count(g)
        except _native.KrispHipError:
            for g in done:
                try:
                    eng.free(g)
                except _native.KrispHipError:
                    pass
            raise

    def run_pass(bsz, first_pass, work):
        """every batch through load_batch + work(ids_b); the next batch's files are read while this one is on the device;
        a batch the device cannot hold is halved and tried again (the plan is an estimate)"""
        todo = list(batches(bsz))
        futs = {}
        while todo:
            ids_b = todo.pop(0)
            for g in ids_b + (todo[0] if todo else []):
                if g not in futs:
                    futs[g] = pool.submit(fasta.read_text, files[g])
            try:
                load_batch(ids_b, futs, first_pass)
            except _native.KrispHipError as e:
                if e.code != _native.ERR_CAPACITY or len(ids_b) == 1:
                    raise
                if first_pass and running[0] is None and len(ids_b) <= first_min:
                    err = _native.KrispHipError(
                        f"the device cannot hold one ingroup and one outgroup genome sorted at once ({e}); the first batch "
                        "needs both sides for the diagnostic filter to prune")
                    err.code = e.code
                    raise err from e
                half = (len(ids_b) + 1) // 2
                if first_pass and running[0] is None:
                    half = max(half, first_min)
                todo = [ids_b[:half], ids_b[half:]] + todo
                for g in ids_b:
                    futs[g] = pool.submit(fasta.read_text, files[g])
                stats["batch"] = min(stats["batch"], half)
                continue
            work(ids_b)
            for g in ids_b:
                eng.free(g)
                del futs[g]
            stats["batches"] += 1

    # every genome's alphabet before the first batch (fasta.sniff_rna reads the head of a file): a run that mixes DNA and
    # RNA genomes filters in mode 2 from the first batch on (kr_set_mixed_alphabets)
    kinds = [bool(fasta.sniff_rna(f)) for f in files]
    mixed = any(kinds) and not all(kinds)
    if mixed:
        eng.set_mixed_alphabets(True)
    running = [None]        # number of running candidates; the list itself stays on the device between the batches of pass 1
    eager = [True]          # (kr_genome_free does not touch it: no copy to the host and back per batch -- ADVICE r5)
    kept = []

    def pass1(ids_b):
        bflags = [flags[g] for g in ids_b]
        if running[0] is None:
            running[0] = eng.intersect(ids_b, bflags, apply_filter=apply_f)
        else:
            # (a later batch looks the running candidates up in its genomes -- kr_cands_probe: presence in every genome,
            # their diagnostic bases into the masks, the filter -- instead of being intersected whole: the candidate list
            # of a batch that holds one side only is every prefix of its genomes, 24 bytes each)
            running[0] = eng.probe_cands(ids_b, bflags, apply_filter=apply_f)
        if verbose:
            for g in ids_b:
                print(f"=> Extracted and sorted {counts[g]:,} {k}-kmers from {files[g]}", file=sys.stderr)
        if quirk_all_fail or not running[0]:
            return
        if eager[0] and running[0] <= STREAM_EAGER_MAX:
            kept.append(eng.collect(sorted(ids_b, key=lambda g: rank_of[g])))
        else:
            eager[0] = False
            kept.clear()

    run_pass(batch, True, pass1)
    if mixed and running[0]:
        running[0] = _mixed_prefix_filter(eng, Le, Re)
    final = eng.cands().copy() if running[0] else np.empty(0, dtype=_native.CAND)
    stats["mixed_rna"] = [bool(r) for r in rna] if mixed else None
    touched = {(l, r) for sp in specials for (l, d, r) in sp} if not quirk_all_fail else set()
    pure = sorted({_prefix_key(l, r) for (l, r) in touched if _pure(l) and _pure(r)})
    srecs = []
    if (len(final) and not eager[0] and not quirk_all_fail) or pure:
        stats["passes"] = 2
        tc = np.zeros(len(pure), dtype=_native.CAND)
        tc["prefix"] = np.array(pure, dtype=np.uint64)

        def pass2(ids_b):
            by = sorted(ids_b, key=lambda g: rank_of[g])
            if len(final) and not eager[0]:
                eng.load_cands(final)
                kept.append(eng.collect(by))
            if pure:
                eng.load_cands(tc)
                srecs.append(eng.collect(by))
        run_pass(max(1, stats["batch"]), False, pass2)
    if kept and len(final):
        records = np.concatenate(kept)
        records = records[np.isin(records["key"] & pmask, final["prefix"])]
        ranks = np.array([rank_of[g] for g in range(n)], dtype=np.int64)[records["genome"]]
        records = records[np.lexsort((ranks, records["key"]))]
    else:
        records = np.empty(0, dtype=_native.RECORD)
    sgroups = []
    if touched:
        recs = np.concatenate(srecs) if srecs else None
        sgroups = _special_groups_from(recs, list(range(n)), labels, specials, geo, frozenset(l for l, f in zip(labels, flags) if f),
                                       do_filter, rna_genomes=[bool(r) for r in rna] if mixed else None)
    stats.update(read_s=read_s[0] or 0.0, kmers=int(sum(counts)) + sum(len(sp) for sp in specials), candidates=int(len(final)))
    return records, touched, sgroups, all(rna), stats


def _find_regions_wide_device_ingest(files, ingroup_files, k, geo, omit_soft, device, verbose, do_filter, workers, t0):
    """find_regions for amplicons longer than one key with the PARSE on the device: the files are read (and inflated) on host
    threads, every text goes to the device as it is (fasta.ingest_on_device: kr_genome_upload_text, or kr_genome_upload_bgzf
    for a BGZF file), kr_wide_run follows.  The host sees a genome's bases again only where it needs them -- IUPAC windows
    (scan_special), mixed DNA / RNA runs, the groups IUPAC windows touch -- through kr_genome_fetch_bases.  Same results as
    the host-parse flow (KRISP_HOST_PARSE=1 keeps that one: A/B, tests).  -> (groups, stats), or None: take the host parse."""
    from concurrent.futures import ThreadPoolExecutor
    from . import _native
    Le, De, Re = geo
    with ThreadPoolExecutor(max_workers=workers) as pool:
        raw = list(pool.map(fasta.read_text, files))
    labels = ["merged_file"] if len(files) == 1 else [simplename(f) for f in files]
    ingroup_labels = frozenset(simplename(f) for f in ingroup_files)
    flags = [lab in ingroup_labels for lab in labels]
    stats = {"read_s": time.time() - t0}
    t1 = time.time()
    budget = int(os.environ.get("KRISP_HBM_BUDGET", "0"))
    maxb = max(max(len(t) for t, _ in raw), 64)
    with _native.Engine(device=device, hbm_budget=budget) as eng:
        eng.set_params_wide(Le, De, Re, omit_soft=omit_soft, max_bases=maxb)
        nb, rna, specials = [], [], []
        for i, (text, universal) in enumerate(raw):
            n, r, sp = fasta.ingest_on_device(eng, i, text, universal, k, omit_soft)
            nb.append(n)
            rna.append(bool(r))
            specials.append([codec.split_window(w, Le, De, Re) for w in sp])
            raw[i] = None
        mixed = any(rna) and not all(rna)
        touched = {(l, r) for sp in specials for (l, d, r) in sp}
        probes = sorted(p for p in touched if _pure(p[0]) and _pure(p[1]))
        probe_text = np.frombuffer("\n".join(l + "A" * De + r for l, r in probes).encode(), dtype=np.uint8)
        if len(probe_text) > maxb:
            return None
        texts = None

        def host_texts():
            nonlocal texts
            if texts is None:
                texts = [eng.fetch_bases(i, nb[i]).copy() for i in range(len(files))]
            return texts
        ids = list(range(len(files)))
        if mixed:
            eng.set_mixed_alphabets(True)
        nhits = eng.wide_run(ids, flags, apply_filter=do_filter)
        hits = eng.wide_fetch(_native.WIDE_HITS) if nhits else np.empty(0, dtype=_native.WIDE_HIT)
        ngroups = int(eng.wide_fetch(_native.WIDE_NGROUPS)[0])
        counts = eng.wide_fetch(_native.WIDE_COUNTS).tolist()
        stats["wide_batch"] = int(eng.wide_fetch(_native.WIDE_BATCH_USED)[0])
        if verbose:
            for f, cnt in zip(files, counts):
                print(f"=> Extracted and sorted {cnt:,} {k}-kmers from {f}", file=sys.stderr)
        finish = _to_rna if all(rna) else (lambda groups: groups)
        if mixed:
            groups = _groups_from_hits(hits, host_texts(), labels, Le, De, Re, rna_genomes=rna)
            groups = [g for g in groups if not (set(g[0].left + g[0].right) & set("TU"))]
            if do_filter and ingroup_labels:
                groups = [g for g in groups if amplicon.ingroup_unique_columns(g, ingroup_labels)]
            if touched:
                sgroups = _special_groups_wide(eng, host_texts(), labels, specials, touched, probes, probe_text,
                                               (Le, De, Re), ingroup_labels, do_filter, rna_genomes=rna)
                groups = _merge_groups(groups, touched | {(g[0].left, g[0].right) for g in sgroups}, sgroups)
            ngroups = len(groups)
            finish = lambda g: g        # noqa: E731
        elif touched:
            groups = _groups_from_hits(hits, host_texts(), labels, Le, De, Re)
            sgroups = _special_groups_wide(eng, host_texts(), labels, specials, touched, probes, probe_text,
                                           (Le, De, Re), ingroup_labels, do_filter)
            groups = _merge_groups(groups, touched, sgroups)
        else:
            rows = eng.wide_windows(k) if nhits else np.empty((0, k), dtype=np.uint8)
            groups = amplicon.WindowGroups(rows, hits["cand"], hits["genome"], labels, Le, De, Re, rna=all(rna))
            finish = lambda g: g        # noqa: E731  (the RNA letters are the renderer's)
    stats.update(device_s=time.time() - t1, kmers=int(sum(counts)) + sum(len(sp) for sp in specials), candidates=ngroups)
    return finish(groups), stats


# ----------------------------------------------------------------------------
# the fused device flow used by main()
# ----------------------------------------------------------------------------
def find_regions(ingroup_files, outgroup_files, L, R, amplicon_len, omit_soft=False,
                 device=0, verbose=False, keep_merged=False, wide=None):
    """FASTA files -> list of surviving groups (amplicon.Amplicon lists).

    = extractSortedKmers per file + mergeFiles + filterAlignments of the reference
    (krisp_fasta.py:237-272), on the GPU: one sort per genome, one n-way
    intersection with the diagnostic filter fused in, one collect.
    Returns (groups, stats)."""
    from . import _native
    k = amplicon_len
    D_nominal = k - L - R
    Le, De, Re = codec.effective_geometry(L, D_nominal, R)
    files = list(ingroup_files) + list(outgroup_files)
    # the reference filters whenever k > L + R (krisp_fasta.py:265); with R == 0 the
    # diagnostic column is empty (kstream.py:824-830) and every group fails the filter
    do_filter = k > L + R
    quirk_all_fail = do_filter and De == 0
    if wide is None:                 # (tests force the wide path on packable geometries)
        wide = _is_wide(Le, De, Re)
    if wide and not quirk_all_fail:
        _check_wide(Le, De, Re)
    elif not wide:
        _check_geometry(Le, De, Re)
    t0 = time.time()
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(len(files), os.cpu_count() or 1, 16))
    if not wide and os.environ.get("KRISP_HOST_PARSE") != "1":
        # files are read and inflated concurrently on the host (the library releases the GIL); the PARSE runs on the
        # device, genome by genome as the texts arrive (fasta.ingest_on_device), each followed at once by its sort
        return _find_regions_device_ingest(files, ingroup_files, L, R, k, (Le, De, Re), omit_soft, device, verbose,
                                           do_filter, quirk_all_fail, workers, t0)
    if wide and not quirk_all_fail and os.environ.get("KRISP_HOST_PARSE") != "1":
        # round 6: amplicons longer than one key take the device's reader, too (the host's parser was the larger half of such a
        # run from files: 1.8 of 2.4 s at 8 x 500 Mbp, tools/e2e_profile.py) -- None: a genome set whose IUPAC windows make a
        # probe text longer than its genomes (tiny inputs): the host parse below
        got = _find_regions_wide_device_ingest(files, ingroup_files, k, (Le, De, Re), omit_soft, device, verbose, do_filter,
                                               workers, t0)
        if got is not None:
            return got
    # ingest: files are read, inflated and parsed concurrently (the parser releases the GIL)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        loaded = list(pool.map(lambda f: fasta.ingest(f, k, omit_soft), files))
    texts = [b for b, _, _ in loaded]
    specials = [[codec.split_window(w, Le, De, Re) for w in sp] for _, _, sp in loaded]
    rna = [bool(r) for _, r, _ in loaded]
    mixed = any(rna) and not all(rna)
    finish = _to_rna if all(rna) else (lambda groups: groups)
    if len(files) == 1:
        # mergeFiles moves the lone k-mer file; its lines carry no label, so later stages
        # label them with the file they read: simplename('merged_file.txt') (shared.py:373)
        labels = ["merged_file"]
    else:
        labels = [simplename(f) for f in files]
    ingroup_labels = frozenset(simplename(f) for f in ingroup_files)
    flags = [lab in ingroup_labels for lab in labels]
    stats = {"read_s": time.time() - t0}
    t1 = time.time()
    if wide:
        if quirk_all_fail:
            stats.update(device_s=0.0, kmers=0, candidates=0)
            return [], stats
        # k-mers holding IUPAC letters (kept by the reference): the (left,right) groups they touch
        # are rebuilt on the host; those with plain flanks need their ACGT members from every
        # genome, found on the device through a probe "genome" of the touched flank pairs
        touched = {(l, r) for sp in specials for (l, d, r) in sp}
        probes = sorted(p for p in touched if _pure(p[0]) and _pure(p[1]))
        probe_text = np.frombuffer("\n".join(l + "A" * De + r for l, r in probes).encode(), dtype=np.uint8)
        budget = int(os.environ.get("KRISP_HBM_BUDGET", "0"))       # (bytes; tests and shared devices: kr_create's HBM budget)
        with _native.Engine(device=device, hbm_budget=budget) as eng:
            eng.set_params_wide(Le, De, Re, omit_soft=omit_soft,
                                max_bases=max(max(len(t) for t in texts), len(probe_text)))
            ids = list(range(len(files)))
            for i, t in enumerate(texts):
                eng.upload(i, t)
            if mixed:
                eng.set_mixed_alphabets(True)
            nhits = eng.wide_run(ids, flags, apply_filter=do_filter)
            hits = eng.wide_fetch(_native.WIDE_HITS) if nhits else np.empty(0, dtype=_native.WIDE_HIT)
            ngroups = int(eng.wide_fetch(_native.WIDE_NGROUPS)[0])
            counts = eng.wide_fetch(_native.WIDE_COUNTS).tolist()
            stats["wide_batch"] = int(eng.wide_fetch(_native.WIDE_BATCH_USED)[0])       # (0: every genome sorted at once)
            if verbose:
                for f, cnt in zip(files, counts):
                    print(f"=> Extracted and sorted {cnt:,} {k}-kmers from {f}", file=sys.stderr)
            if mixed:
                # (text decides: flank pairs that hold T / U are in no DNA genome and RNA genome alike, and the filter's
                # columns are compared letter by letter -- _mixed_finish's rule on member windows)
                groups = _groups_from_hits(hits, texts, labels, Le, De, Re, rna_genomes=rna)
                groups = [g for g in groups if not (set(g[0].left + g[0].right) & set("TU"))]
                if do_filter and ingroup_labels:
                    groups = [g for g in groups if amplicon.ingroup_unique_columns(g, ingroup_labels)]
                if touched:
                    sgroups = _special_groups_wide(eng, texts, labels, specials, touched, probes, probe_text,
                                                   (Le, De, Re), ingroup_labels, do_filter, rna_genomes=rna)
                    groups = _merge_groups(groups, touched | {(g[0].left, g[0].right) for g in sgroups}, sgroups)
                ngroups = len(groups)
                finish = lambda g: g        # noqa: E731
            elif touched:
                groups = _groups_from_hits(hits, texts, labels, Le, De, Re)
                sgroups = _special_groups_wide(eng, texts, labels, specials, touched, probes, probe_text,
                                               (Le, De, Re), ingroup_labels, do_filter)
                groups = _merge_groups(groups, touched, sgroups)
            else:
                # the member windows are cut on the device (kr_wide_fetch_windows) and rendered from their rows in the
                # library (amplicon.WindowGroups: no object per window; the list of groups only if someone walks it)
                rows = eng.wide_windows(k) if nhits else np.empty((0, k), dtype=np.uint8)
                groups = amplicon.WindowGroups(rows, hits["cand"], hits["genome"], labels, Le, De, Re, rna=all(rna))
                finish = lambda g: g        # noqa: E731  (the RNA letters are the renderer's)
        stats.update(device_s=time.time() - t1, kmers=int(sum(counts)) + sum(len(sp) for sp in specials),
                     candidates=ngroups)
        return finish(groups), stats
    with _native.Engine(device=device) as eng:
        eng.set_params(Le, De, Re, omit_soft=omit_soft, max_bases=max(len(t) for t in texts))
        counts = []
        for i, t in enumerate(texts):
            eng.upload(i, t)
            eng.sort(i)
        ids = list(range(len(files)))
        if mixed:
            eng.set_mixed_alphabets(True)
        ncand = eng.intersect(ids, flags, apply_filter=do_filter and not quirk_all_fail)
        if mixed and ncand:
            ncand = _mixed_prefix_filter(eng, Le, Re)
        counts = [eng.count(i) for i in ids]
        if verbose:
            for f, c in zip(files, counts):
                print(f"=> Extracted and sorted {c:,} {k}-kmers from {f}", file=sys.stderr)
        # (genomes listed by label: kr_collect's (key, position in the call) order is then the renderer's)
        by_label = sorted(ids, key=lambda i: labels[i])
        records = eng.collect(by_label) if (ncand and not quirk_all_fail) else np.empty(0, dtype=_native.RECORD)
        touched, sgroups = set(), []
        if any(specials) and not quirk_all_fail:
            touched, sgroups = _special_groups(eng, ids, labels, specials, (Le, De, Re), ingroup_labels,
                                               do_filter, rna_genomes=rna if mixed else None)
    stats.update(device_s=time.time() - t1, kmers=int(sum(counts)) + sum(len(sp) for sp in specials),
                 candidates=int(ncand))
    if quirk_all_fail:
        return [], stats
    if mixed:
        groups = _mixed_finish(records, labels, (Le, De, Re), rna, ingroup_labels, do_filter)
        if touched:
            groups = _merge_groups(groups, touched, sgroups)
        stats["candidates"] = len(groups)
        return groups, stats
    if not touched and not any(rna):
        # (plain records: the renderer works from them directly, the list of groups is built only if someone walks it)
        return amplicon.RecordGroups(records, labels, Le, De, Re), stats
    groups = amplicon.groups_from_records(records, labels, Le, De, Re, rna=False)
    if touched:
        groups = _merge_groups(groups, touched, sgroups)
    return finish(groups), stats


class PeerFailed(RuntimeError):
    """another rank of a multi-GPU run raised; this rank stops with it instead of waiting in a collective"""


def _distributed_rank(rank, world, device, connect, ingroup_files, outgroup_files, L, R, amplicon_len, omit_soft):
    """one rank of the multi-GPU flow (see find_regions_distributed); `connect(engine)` gives the
    engine its communicator.  Returns (groups, stats) on rank 0 and (None, stats) elsewhere."""
    from . import _native
    from . import distributed as D
    k = amplicon_len
    Le, De, Re = codec.effective_geometry(L, k - L - R, R)
    do_filter = k > L + R
    wide = Le + De + Re > 32 or De > 16            # amplicons longer than one key: kr_wide_run, collective inside
    if wide and not (do_filter and De == 0):
        _check_wide(Le, De, Re)
    elif not wide:
        _check_geometry(Le, De, Re)
    # interleave ingroup / outgroup files so that a round-robin shard holds both kinds
    ing, outg = list(ingroup_files), list(outgroup_files)
    order = []
    for i in range(max(len(ing), len(outg))):
        if i < len(ing):
            order.append(ing[i])
        if i < len(outg):
            order.append(outg[i])
    if len(order) < world:
        raise ValueError(f"{len(order)} genomes cannot be sharded over {world} GPUs")
    labels = [simplename(f) for f in order]
    # ingroup / outgroup by LABEL, as the reference classifies (Amplicon.py:495-521) and as
    # find_regions does: an outgroup file whose label equals an ingroup label counts as ingroup
    ingroup_labels = frozenset(simplename(f) for f in ing)
    mine = D.shard(list(range(len(order))), rank, world)

    with _native.Engine(device=device) as eng:
        connect(eng)

        def together(fn):
            """run fn() on this rank; every rank learns whether all succeeded before anyone goes on"""
            err, out = None, None
            try:
                out = fn()
            except BaseException as e:  # noqa: BLE001
                err = e
            failed = eng.comm_allreduce([1.0 if err is not None else 0.0], "max")[0] > 0
            if err is not None:
                raise err
            if failed:
                raise PeerFailed(f"rank {rank}: another rank failed")
            return out

        t0 = time.time()
        loaded = together(lambda: [fasta.ingest(order[g], k, omit_soft) for g in mine])
        kinds = eng.comm_allreduce([1.0 if any(sp for _, _, sp in loaded) else 0.0,
                                    1.0 if any(r for _, r, _ in loaded) else 0.0,
                                    float(max(len(b) for b, _, _ in loaded)),
                                    1.0 if any(not r for _, r, _ in loaded) else 0.0], "max")
        mixed = bool(kinds[1] and kinds[3])
        all_rna = bool(kinds[1]) and not mixed
        rna_all = None
        if mixed:
            # DNA and RNA genomes in one run (round 6; find_regions has the rules): every rank learns every genome's alphabet,
            # every context filters in mode 2, the candidates whose (left,right) pair holds T / U go, rank 0 decides the rest on text
            if kinds[0]:
                raise MixedAlphabet("some genomes are RNA (U) and some DNA (T), and windows hold IUPAC letters")
            import json
            rna_all = [False] * len(order)
            for blob in eng.comm_allgather(json.dumps({int(g): bool(r) for g, (_, r, _) in zip(mine, loaded)}).encode()):
                for g, r in json.loads(blob.decode()).items():
                    rna_all[int(g)] = bool(r)
            eng.set_mixed_alphabets(True)
        # windows with IUPAC letters (kept by the reference, kstream.py:11-18; the device alphabet cannot carry them):
        # every rank learns all of them -- they are rare --, the groups they touch are rebuilt on rank 0 from the
        # ranks' device look-ups of the ACGT members (below), as find_regions does on one GPU
        specials_all = None
        if kinds[0]:
            import json         # (plain data: genome number -> [left, diag, right] strings; nothing executable crosses ranks)
            mine_sp = {g: [codec.split_window(w, Le, De, Re) for w in sp] for g, (_, _, sp) in zip(mine, loaded)}
            specials_all = [[] for _ in order]
            for blob in eng.comm_allgather(json.dumps(mine_sp).encode()):
                for g, sp in json.loads(blob.decode()).items():
                    specials_all[int(g)] = [tuple(w) for w in sp]
        stats = {"read_s": time.time() - t0}
        t1 = time.time()
        quirk_all_fail = do_filter and De == 0
        filt = do_filter and not quirk_all_fail

        if wide:
            if quirk_all_fail:
                eng.comm_barrier()
                stats.update(device_s=0.0, kmers=0, candidates=0, records=0)
                return ([] if rank == 0 else None), stats

            def wide_part():
                # the spectra lists, the group list and the kept groups' masks are exchanged inside kr_wide_run
                # (the same calls in the same order on every rank); hits name genomes by their global number
                eng.set_params_wide(Le, De, Re, omit_soft=omit_soft, max_bases=int(kinds[2]))
                for g, (bases, _, _) in zip(mine, loaded):
                    eng.upload(g, bases)
                nh = eng.wide_run(mine, [labels[g] in ingroup_labels for g in mine], apply_filter=do_filter)
                return nh, int(sum(eng.wide_fetch(_native.WIDE_COUNTS).tolist())), int(eng.wide_fetch(_native.WIDE_NGROUPS)[0])

            nh, counts, ngroups = together(wide_part)
            hits = eng.wide_fetch(_native.WIDE_HITS) if (rank == 0 and nh) else np.empty(0, dtype=_native.WIDE_HIT)
            # windows with IUPAC letters (round 4; one GPU only before): every rank knows all of them (specials_all), so
            # every rank derives the same touched (left,right) pairs and the same probe genome, looks the pairs' ACGT
            # members up in ITS genomes -- a context of its own without a communicator: the look-ups are local, one
            # kr_wide_run per genome -- and rank 0 gets everybody's findings and rebuilds the touched groups
            wide_sp = None
            if specials_all is not None:
                import json
                touched_w = {(l, r) for sp in specials_all for (l, d, r) in sp}
                probes = sorted(p for p in touched_w if _pure(p[0]) and _pure(p[1]))
                probe_text = np.frombuffer("\n".join(l + "A" * De + r for l, r in probes).encode(), dtype=np.uint8)

                def probe_part():
                    if not probes:
                        return {}
                    with _native.Engine(device=device) as e2:
                        e2.set_params_wide(Le, De, Re, omit_soft=omit_soft, max_bases=max(int(kinds[2]), len(probe_text)))
                        for g, (bases, _, _) in zip(mine, loaded):
                            e2.upload(g, bases)
                        return _wide_probe_members(e2, [b for b, _, _ in loaded], list(mine), probes, probe_text,
                                                   (Le, De, Re), pid=len(order))

                local = together(probe_part)
                flat = [[list(P), list(seq), int(g), int(cnt)] for P, seqs in local.items() for seq, m in seqs.items()
                        for g, cnt in m.items()]
                members_w = {}
                for blob in eng.comm_allgather(json.dumps(flat).encode()):
                    for P, seq, g, cnt in json.loads(blob.decode()):
                        m = members_w.setdefault(tuple(P), {}).setdefault(tuple(seq), {})
                        m[g] = m.get(g, 0) + cnt
                wide_sp = (touched_w, members_w)
            eng.comm_barrier()
            stats.update(device_s=time.time() - t1, kmers=counts, candidates=ngroups, records=int(len(hits)))
            if rank != 0:
                return None, stats
            # rank 0 cuts the windows' text: its own genomes are in memory, the others' files are read here
            texts = [None] * len(order)
            for g, (bases, _, _) in zip(mine, loaded):
                texts[g] = bases
            for g in sorted(set(hits["genome"].tolist())):
                if texts[g] is None:
                    texts[g] = fasta.ingest(order[g], k, omit_soft)[0]
            wg = _groups_from_hits(hits, texts, labels, Le, De, Re, rna_genomes=rna_all)
            if mixed:
                wg = [g for g in wg if not (set(g[0].left + g[0].right) & set("TU"))]
                if do_filter and ingroup_labels:
                    wg = [g for g in wg if amplicon.ingroup_unique_columns(g, ingroup_labels)]
                stats["candidates"] = len(wg)
                return wg, stats
            if wide_sp is not None:
                touched_w, members_w = wide_sp
                sg = _special_groups_wide_from(members_w, len(order), labels, specials_all, touched_w, ingroup_labels, do_filter)
                wg = _merge_groups(wg, touched_w, sg)
            return (_to_rna(wg) if all_rna else wg), stats

        # Round 6: a rank whose shard does not fit its GPU sorted at once takes it in batches, as
        # find_regions does on one GPU (_find_regions_streaming): the first batch intersected, later batches probed, the
        # batch freed; the exchange of the candidate lists is the same; the records are then collected batch by batch in a
        # second pass over the shard and travel to rank 0 as bytes (kr_comm_allgather) instead of from the device buffer of
        # ONE kr_collect.  KRISP_STREAM_BATCH=n forces batches of n genomes per rank (tests).
        bases_of = {g: b for g, (b, _, _) in zip(mine, loaded)}
        flag_of = {g: labels[g] in ingroup_labels for g in mine}
        batched = [None]

        def shard_batches():
            B = batched[0]
            both = any(flag_of.values()) and not all(flag_of.values()) and filt
            at, first = 0, True
            while at < len(mine):
                m = max(B, 2) if (first and both) else B        # (the first batch holds both sides: nothing prunes else)
                yield mine[at:at + m]
                at, first = at + m, False

        def device_part():
            eng.set_params(Le, De, Re, omit_soft=omit_soft, max_bases=int(kinds[2]))
            batched[0] = _plan_batch(eng, len(mine), int(kinds[2])) if len(mine) > 1 else None
            if batched[0] is None:
                for g in mine:
                    eng.upload(g, bases_of[g])
                    eng.sort(g)
                eng.intersect(mine, [flag_of[g] for g in mine], apply_filter=filt)    # safe local pruning
                return sum(eng.count(g) for g in mine)
            total, running = 0, None
            for ids_b in shard_batches():
                for g in ids_b:
                    eng.upload(g, bases_of[g])
                    eng.sort(g)
                fl = [flag_of[g] for g in ids_b]
                running = eng.intersect(ids_b, fl, apply_filter=filt) if running is None else \
                    eng.probe_cands(ids_b, fl, apply_filter=filt)
                total += sum(eng.count(g) for g in ids_b)
                for g in ids_b:
                    eng.free(g)
            return total

        def collect_shard():
            """this rank's records of the candidates now on the device -> numpy (batched shards: a pass over the batches)"""
            cands = eng.cands().copy()
            parts = []
            for ids_b in shard_batches():
                for g in ids_b:
                    eng.upload(g, bases_of[g])
                    eng.sort(g)
                eng.load_cands(cands)
                if len(cands):
                    parts.append(eng.collect(ids_b))
                for g in ids_b:
                    eng.free(g)
            return np.concatenate(parts) if parts else np.empty(0, dtype=_native.RECORD)

        def gather_host_records(recs):
            """every rank's record array -> rank 0 (None elsewhere), as bytes through kr_comm_allgather"""
            got = eng.comm_allgather(np.ascontiguousarray(recs).tobytes())
            if rank != 0:
                return None
            parts = [np.frombuffer(b, dtype=_native.RECORD) for b in got if len(b)]
            return np.concatenate(parts) if parts else np.empty(0, dtype=_native.RECORD)

        counts = together(device_part)
        # (every rank learns whether ANY rank took its shard in batches: the records then travel as bytes on all of them)
        any_batched = eng.comm_allreduce([1.0 if batched[0] is not None else 0.0], "max")[0] > 0
        if any_batched and batched[0] is None:
            batched[0] = len(mine)
        ncand = eng.cands_reduce(apply_filter=filt)
        eng.cands_bcast()
        if mixed:
            together(lambda: _mixed_prefix_filter(eng, Le, Re))       # (the same list on every rank: the same cut)
        if any_batched:
            myrec = together(lambda: np.empty(0, dtype=_native.RECORD) if quirk_all_fail else collect_shard())
            nrec = len(myrec)
            allrec = gather_host_records(myrec) if not quirk_all_fail else None
        else:
            nrec = together(lambda: 0 if quirk_all_fail else eng.collect(mine, fetch=False))
            total = eng.records_gather() if not quirk_all_fail else 0
            allrec = eng.fetch_records(total) if rank == 0 and not quirk_all_fail else None
        sprec, touched = None, set()
        if specials_all is not None and not quirk_all_fail:
            # the ACGT members of every (left,right) group an IUPAC window touches: each rank looks them up in ITS
            # genomes (the touched prefixes as a candidate list), rank 0 gets all the records
            touched = {(l, r) for sp in specials_all for (l, d, r) in sp}
            pure = sorted({_prefix_key(l, r) for (l, r) in touched if _pure(l) and _pure(r)})

            def special_lookups():
                if not pure:
                    return 0
                cands = np.zeros(len(pure), dtype=_native.CAND)
                cands["prefix"] = np.array(pure, dtype=np.uint64)
                eng.load_cands(cands)
                return collect_shard() if any_batched else eng.collect(mine, fetch=False)

            sp_local = together(special_lookups)
            if pure and any_batched:
                sprec = gather_host_records(sp_local)
            elif pure:
                tot2 = eng.records_gather()
                sprec = eng.fetch_records(tot2) if rank == 0 else None
        eng.comm_barrier()
        stats.update(device_s=time.time() - t1, kmers=int(counts) + (sum(len(sp) for sp in specials_all) if specials_all else 0),
                     candidates=int(max(ncand, 0)), records=int(nrec))
    if rank != 0:
        return None, stats
    if quirk_all_fail:
        return [], stats
    if mixed:
        groups = _mixed_finish(allrec, labels, (Le, De, Re), rna_all, ingroup_labels, do_filter)
        stats["candidates"] = len(groups)
        return groups, stats
    finish = _to_rna if all_rna else (lambda groups: groups)
    if not touched and not all_rna:
        return amplicon.RecordGroups(allrec, labels, Le, De, Re), stats
    groups = amplicon.groups_from_records(allrec, labels, Le, De, Re)
    if touched:
        sgroups = _special_groups_from(sprec, list(range(len(order))), labels, specials_all, (Le, De, Re),
                                       ingroup_labels, do_filter)
        groups = _merge_groups(groups, touched, sgroups)
    return finish(groups), stats


def find_regions_distributed(ingroup_files, outgroup_files, L, R, amplicon_len, omit_soft=False,
                             transport="rccl", verbose=False):
    """find_regions over several GPUs: one process per GPU (started by torch.distributed.run,
    mpirun, srun ...: RANK / LOCAL_RANK / WORLD_SIZE in the environment; no PyTorch involved).
    Genomes are sharded round-robin over the ranks (ingroup and outgroup files interleaved so that
    every rank can prune with the monotone filter), each rank sorts and intersects its own, the
    candidate lists are tree-reduced between the GPUs (kr_cands_reduce: RCCL, device to device),
    the survivors broadcast, every rank collects its genomes' records and rank 0 gathers them.
    Returns (groups, stats) on rank 0 and (None, stats) elsewhere.  Amplicons longer than one key
    take kr_wide_run, which exchanges the flank spectra, the group list and the kept groups' masks
    between the ranks itself.  No IUPAC letters, no RNA: those need look-ups across all genomes
    and stay on one GPU.  A rank that fails (missing file, illegal character, out of memory) makes every
    rank raise before the next collective: nobody is left waiting."""
    from . import distributed as D
    rank, local_rank, world = D.env_rank_world()
    if transport == "dir":                 # rehearsal: the ranks share the visible GPU(s)
        local_rank %= max(1, int(os.environ.get("KRISP_VISIBLE_GPUS", "1")))
    return _distributed_rank(rank, world, local_rank, lambda eng: D.connect(eng, rank, world, transport=transport),
                             ingroup_files, outgroup_files, L, R, amplicon_len, omit_soft)


def find_regions_multi_device(ingroup_files, outgroup_files, L, R, amplicon_len, devices, omit_soft=False,
                              verbose=False):
    """The same flow inside ONE process: a thread per listed device (the library calls release the
    GIL), each with its own context and communicator -- the fallback for hosts where a launcher
    is not at hand.  Distinct devices talk over RCCL (the unique id is passed between
    the threads); a device listed twice makes the exchange go through files instead (RCCL refuses
    duplicate devices): that is how a one-GPU box rehearses it.  Returns (groups, stats)."""
    import tempfile
    import threading
    from . import _native
    world = len(devices)
    if world < 1:
        raise ValueError("find_regions_multi_device: no device listed")
    use_rccl = len(set(devices)) == world
    cid = _native.comm_unique_id() if use_rccl else None
    results, errors = [None] * world, [None] * world
    with tempfile.TemporaryDirectory(prefix="krisp_comm_") as td:
        def connect_for(rank):
            if use_rccl:
                return lambda eng: eng.comm_init(rank, world, cid)
            return lambda eng: eng.comm_init_dir(rank, world, td)

        def work(rank):
            try:
                results[rank] = _distributed_rank(rank, world, devices[rank], connect_for(rank), ingroup_files,
                                                  outgroup_files, L, R, amplicon_len, omit_soft)
            except BaseException as e:  # noqa: BLE001
                errors[rank] = e
        threads = [threading.Thread(target=work, args=(r,), name=f"krisp-gpu{devices[r]}") for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    real = [e for e in errors if e is not None and not isinstance(e, PeerFailed)]
    if real or any(errors):
        raise (real or [e for e in errors if e is not None])[0]
    groups, stats = results[0]
    stats["kmers"] = int(sum(r[1]["kmers"] for r in results))
    return groups, stats


# ----------------------------------------------------------------------------
# stage functions (reference signatures)
# ----------------------------------------------------------------------------
def _hit_windows(sel, text, k):
    """member windows of kr_wide_run hits of ONE genome as an (n, k) byte matrix: cut from the text
    the host holds, soft-mask mapped, reverse complemented for strand 1"""
    t = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    W = t[sel["pos"].astype(np.int64)[:, None] + np.arange(k, dtype=np.int64)] & np.uint8(0xDF)
    rc = sel["strand"] == 1
    W[rc] = _COMP_U8[W[rc][:, ::-1]]
    return W


def _extract_sorted_wide(fasta_file, L, R, k, output, omit, device, verbose):
    """extractSortedKmers for amplicons longer than one key (krisp_fasta.py:16-66 takes any k):
    the genome alone through kr_wide_run without a filter -- every window then belongs to a
    (left,right) group, the groups come in (left,right) order -- and the members of a group put in
    diag order here; lines 'left,diag,right' as kstream.py:832 writes them."""
    from . import _native
    Le, De, Re = codec.effective_geometry(L, k - L - R, R)
    _check_wide(Le, De, Re)
    bases, rna, special = fasta.ingest(fasta_file, k, omit)
    lut = np.arange(256, dtype=np.uint8)
    if rna:
        lut[ord("T")] = ord("U")
    with _native.Engine(device=device) as eng:
        eng.set_option(_native.OPT_WIDE_ORDERED, 1)       # (the chunks below rely on hits in (left, right) group order)
        eng.set_params_wide(Le, De, Re, omit_soft=omit, max_bases=max(len(bases), 1))
        eng.upload(0, bases)
        nhits = eng.wide_run([0], [True], apply_filter=False)
        hits = eng.wide_fetch(_native.WIDE_HITS) if nhits else np.empty(0, dtype=_native.WIDE_HIT)
    # IUPAC k-mers (kept by the reference, resolved on the host) join by a merge of sorted lines
    sp_lines = sorted(",".join(codec.split_window(w, Le, De, Re)).encode() for w in special)

    def key(ln):
        f = ln.split(b",")
        return (f[0], f[2], f[1])

    def chunks():
        """the hits cut at group boundaries (members of one group are ordered together), each chunk as
        its window matrix in (group, diag) order"""
        pos = 0
        while pos < len(hits):
            b = min(len(hits), pos + (1 << 20))
            while b < len(hits) and hits["cand"][b] == hits["cand"][b - 1]:
                b += 1
            sel = hits[pos:b]
            W = _hit_windows(sel, bases, k)
            rec = np.empty(len(sel), dtype=[("c", ">u4"), ("d", f"S{max(De, 1)}")])
            rec["c"] = sel["cand"]
            rec["d"] = W[:, Le:Le + De].copy().view(f"S{De}").ravel() if De else b""
            yield W[np.argsort(rec, order=["c", "d"], kind="stable")]
            pos = b

    n = 0
    with open(output, "wb") as f:
        if not sp_lines:
            for W in chunks():
                W = lut[W]
                out = np.empty((len(W), k + 3), dtype=np.uint8)
                out[:, :Le] = W[:, :Le]
                out[:, Le] = ord(",")
                out[:, Le + 1:Le + 1 + De] = W[:, Le:Le + De]
                out[:, Le + 1 + De] = ord(",")
                out[:, Le + 2 + De:k + 2] = W[:, Le + De:]
                out[:, k + 2] = ord("\n")
                f.write(out.tobytes())
                n += len(out)
        else:
            import heapq
            dev = []
            for W in chunks():
                for row in W:
                    s = bytes(row)
                    dev.append(s[:Le] + b"," + s[Le:Le + De] + b"," + s[Le + De:])
            tr = bytes.maketrans(b"T", b"U") if rna else None
            for ln in heapq.merge(dev, sp_lines, key=key):
                f.write((ln.translate(tr) if tr else ln) + b"\n")
                n += 1
    return n


def extractSortedKmers(fasta_file, primer_left, primer_right, ampl_len, output,
                       sortmem, parallel=1, verbose=True, omit=True, device=0):
    """Fasta file -> sorted 'left,diag,right' k-mer file (krisp_fasta.py:16-66)."""
    kw = dict(kmers=ampl_len, disallow="Nn", complements=True,
              split=[primer_left, -primer_right], sort=True, sortmem=sortmem,
              sortcols=[0, 2], sortnp=parallel, parallel=parallel, device=device)
    kw["omitsoft" if omit else "mapsoft"] = True
    ks = kstream(fasta_file, **kw)
    t0 = time.time()
    if verbose:
        print(f"Extracting {ampl_len}-mers from {fasta_file} and saving to {output}", file=sys.stderr)
    if ks.device_geometry() is not None:
        found = ks.write(output)
    else:
        # amplicons longer than one key: the wide path (raises UnsupportedGeometry beyond it)
        found = _extract_sorted_wide(fasta_file, primer_left, primer_right, ampl_len, output, omit, device, verbose)
    if verbose:
        print(f"=> Extracted and sorted {found:,} {ampl_len}-kmers from {fasta_file} in "
              f"{prettyTime(time.time() - t0)}", file=sys.stderr)


def sortedKmersSerial(files, outputs, ampl_len, primer_left, primer_right, verbose=True,
                      omit=True, device=0):
    for f, o in zip(files, outputs):
        extractSortedKmers(f, primer_left, primer_right, ampl_len, o, "80%", 1, verbose, omit, device)


def sortedKmersParallel(files, outputs, ampl_len, primer_left, primer_right, parallel=1,
                        verbose=True, omit=True, device=0):
    """krisp_fasta.py:86-123.  One GPU sorts a genome in milliseconds; the reference's
    process-per-genome fan-out has nothing left to hide, so this is the serial loop."""
    sortedKmersSerial(files, outputs, ampl_len, primer_left, primer_right, verbose, omit, device)


def _read_lines(path):
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    return lines


def mergeFiles(files, output, parallel=1, workdir=None, verbose=True, device=0):
    """Sorted 3-column k-mer files -> merged alignment file
    (intersectAmplicons.py:232-310): every (left,right) pair present in ALL files,
    one line 'left,diag,right,labels' per distinct sequence.  The keys of each file
    are packed on the host and adopted by the device as a sorted genome
    (kr_genome_load_sorted); the intersection and the collect run there."""
    from . import _native
    files = list(files)
    t0 = time.time()
    if len(files) == 1:
        os.replace(files[0], output)              # shutil.move(files[0], output)
        return
    contents = [_read_lines(f) for f in files]
    geo = None
    for lines in contents:
        if lines:
            f3 = lines[0].split(b",")
            if len(f3) != 3:
                raise ValueError("mergeFiles on the device takes the 3-column sorted k-mer files "
                                 "extractSortedKmers writes")
            geo = (len(f3[0]), len(f3[1]), len(f3[2]))
            break
    if geo is None:
        open(output, "w").close()
        return
    L, D, R = geo
    labels = [simplename(f) for f in files]
    if _is_wide(L, D, R):
        groups = _merge_files_wide(contents, labels, L, D, R, device)
        with open(output, "w") as f:
            for ln in amplicon.merged_lines(groups):
                f.write(ln + "\n")
        if verbose:
            print(f"=> Merged {len(files)} files -> {output} in {prettyTime(time.time() - t0)}", file=sys.stderr)
        return
    _check_geometry(L, D, R)
    keysets, specials = [], []
    for lines in contents:
        flat = b"".join(lines)
        if flat.translate(None, b"ACGT,"):       # some line holds an IUPAC letter: split them off
            sp = [ln for ln in lines if ln.translate(None, b"ACGT,")]
            lines = [ln for ln in lines if not ln.translate(None, b"ACGT,")]
            specials.append([tuple(x.decode() for x in ln.split(b",")) for ln in sp])
        else:
            specials.append([])
        keys = codec.lines_to_keys(lines, L, D, R)
        if len(keys) > 1 and not np.all(keys[1:] >= keys[:-1]):
            raise ValueError("k-mer file is not sorted by (left, right, diag)")
        keysets.append(keys)
    with _native.Engine(device=device) as eng:
        eng.set_params(L, D, R, max_bases=max(1, max(len(k) for k in keysets)))
        for i, keys in enumerate(keysets):
            eng.load_sorted(i, keys)
        ids = list(range(len(files)))
        ncand = eng.intersect(ids, [True] * len(ids), apply_filter=False)
        by_label = sorted(ids, key=lambda i: labels[i])
        records = eng.collect(by_label) if ncand else np.empty(0, dtype=_native.RECORD)
        touched, sgroups = set(), []
        if any(specials):
            touched, sgroups = _special_groups(eng, ids, labels, specials, (L, D, R), frozenset(), False)
    groups = amplicon.groups_from_records(records, labels, L, D, R)
    if touched:
        groups = _merge_groups(groups, touched, sgroups)
    with open(output, "w") as f:
        for ln in amplicon.merged_lines(groups):
            f.write(ln + "\n")
    if verbose:
        print(f"=> Merged {len(files)} files -> {output} in {prettyTime(time.time() - t0)}", file=sys.stderr)


def _merge_files_wide(contents, labels, L, D, R, device):
    """mergeFiles for k-mer files of amplicons longer than one key: every file becomes a "genome"
    whose records are its lines (one window each, forward strand only: the file already holds both
    strands), kr_wide_run finds the (left,right) pairs present in all files and where their members lie."""
    from . import _native
    _check_wide(L, D, R)
    k = L + D + R
    texts, rnas, specials = [], [], []
    for lines in contents:
        flat = b"".join(lines)
        rna = b"U" in flat and b"T" not in flat
        rnas.append(rna)
        plain = b"ACGU," if rna else b"ACGT,"
        if flat.translate(None, plain):
            # lines that hold IUPAC ambiguity letters (the reference keeps such k-mers: kstream.py:11-18, Amplicon.py:298-348
            # reads them like any line): they stay text -- the groups they touch are rebuilt on the host from them and from
            # the plain members the device finds for their (left,right) pairs, as find_regions does (round 6: refused before)
            sp = [ln for ln in lines if ln.translate(None, plain)]
            lines = [ln for ln in lines if not ln.translate(None, plain)]
            # (inside, an RNA file's lines are handled with T like its plain lines; they get their U back at the end)
            specials.append([tuple(x.decode().replace("U", "T") if rna else x.decode() for x in ln.split(b",")) for ln in sp])
        else:
            specials.append([])
        rec = [ln.replace(b",", b"") for ln in lines]
        if any(len(r) != k for r in rec):
            raise ValueError("k-mer file with lines of different lengths")
        t = b"\n".join(rec)
        texts.append(np.frombuffer(t.replace(b"U", b"T") if rna else t, dtype=np.uint8))
    ids = list(range(len(texts)))
    mixed = any(rnas) and not all(rnas)
    touched = {(l, r) for sp in specials for (l, d, r) in sp}
    probes = sorted(p for p in touched if _pure(p[0]) and _pure(p[1]))
    probe_text = np.frombuffer("\n".join(l + "A" * D + r for l, r in probes).encode(), dtype=np.uint8)
    with _native.Engine(device=device) as eng:
        eng.set_params_wide(L, D, R, omit_soft=False, max_bases=max(1, max(len(t) for t in texts), len(probe_text)))
        eng.set_strands(_native.STRANDS_FORWARD)
        for i, t in enumerate(texts):
            eng.upload(i, t)
        nhits = eng.wide_run(ids, [True] * len(ids), apply_filter=False)
        hits = eng.wide_fetch(_native.WIDE_HITS) if nhits else np.empty(0, dtype=_native.WIDE_HIT)
        # (an RNA genome's file holds U: its lines come back with U; a (left,right) pair that holds T / U is text no DNA file
        # and RNA file share -- shared.py:321-347 merges on the strings)
        groups = _groups_from_hits(hits, texts, labels, L, D, R, rna_genomes=rnas if any(rnas) else None)
        if mixed:
            groups = [g for g in groups if not (set(g[0].left + g[0].right) & set("TU"))]
        if touched:
            sgroups = _special_groups_wide(eng, texts, labels, specials, touched, probes, probe_text, (L, D, R), frozenset(), False,
                                           rna_genomes=rnas if any(rnas) else None)
            if any(rnas):       # (the device's groups of the touched pairs carry U where their files do)
                touched = touched | {(l.replace("T", "U"), r.replace("T", "U")) for l, r in touched}
            groups = _merge_groups(groups, touched, sgroups)
    return groups


def _parse_merged(path):
    """merged-file lines -> groups of amplicon.Amplicon (Amplicon.py:298-328,
    shared.py:350-398, 442-475)."""
    tag = simplename(path)
    groups = []
    for raw in _read_lines(path):
        f = raw.decode().strip().split(",")
        if len(f) not in (3, 4):
            raise ValueError(f"Unrecognised string format : {raw.decode()}")
        labels = [tag]
        if len(f) == 4:
            labels = []
            for item in f[3].split(";"):
                item = item.strip()
                if "(" in item:
                    name, mult = item.split("(")
                    labels += [name] * int(mult.strip(")"))
                else:
                    labels.append(item)
        amp = amplicon.Amplicon(f[0], f[1], f[2], labels)
        if groups and (groups[-1][0].left, groups[-1][0].right) == (amp.left, amp.right):
            for a in groups[-1]:
                if a.sequence == amp.sequence:
                    a.labels = sorted(a.labels + amp.labels)
                    break
            else:
                groups[-1].append(amp)
        else:
            groups.append([amp])
    return groups


_BASE_BIT = {"A": 0, "C": 1, "G": 2, "T": 3}


def filterAlignments(kmerfile, output, ingroup, device=0):
    """filterAlignments.py:31-40: keep the groups that have a diagnostic column whose
    ingroup and outgroup base sets are disjoint (Amplicon.py:495-521).  The host only
    re-encodes the text (one (in, out) base-set mask pair per group, include/krisp_hip.h
    kr_cand); the predicate itself is evaluated on the device (kr_cands_merge)."""
    from . import _native
    groups = _parse_merged(kmerfile)
    ingroup = frozenset(ingroup)
    keep = list(range(len(groups)))
    if len(ingroup) and groups:
        D = len(groups[0][0].diag)
        if D == 0:
            keep = []                               # diagnosticLength() == 0: no column can pass
        else:
            # the device's mask pair carries 16 columns: a group passes when ANY column separates the
            # groups, so longer diagnostic regions are filtered 16 columns at a time and the kept sets united
            host_keep, pure = [], []
            for gi, g in enumerate(groups):
                if not all(_pure(a.diag) for a in g):
                    # IUPAC letters in a diagnostic column: base sets are sets of letters
                    # (Amplicon.py:514-520); rare, evaluated here
                    if amplicon.ingroup_unique_columns(g, ingroup):
                        host_keep.append(gi)
                else:
                    pure.append(gi)
            kept = set()
            with _native.Engine(device=device) as eng:
                for c0 in range(0, D, 16):
                    Dc = min(16, D - c0)
                    rows = []
                    for gi in pure:
                        if gi in kept:
                            continue
                        im = om = 0
                        for a in groups[gi]:
                            m = 0
                            for c, ch in enumerate(a.diag[c0:c0 + Dc]):
                                m |= 1 << (4 * c + _BASE_BIT[ch])
                            for lab in set(a.labels):
                                if lab in ingroup:
                                    im |= m
                                else:
                                    om |= m
                        rows.append((gi, im, om))           # prefix = group index: sorted, unique
                    if not rows:
                        break
                    eng.set_params(1, Dc, 0, max_bases=64)
                    eng.load_cands(np.array(rows, dtype=_native.CAND))
                    eng.merge_cands(None, apply_filter=True)
                    kept.update(int(p) for p in eng.cands()["prefix"])
            keep = sorted(kept)
            keep = sorted(keep + host_keep)
    with open(output, "w") as f:
        for gi in keep:
            for a in groups[gi]:
                f.write(a.line() + "\n")


# ----------------------------------------------------------------------------
# command line (krisp_fasta.py:126-298)
# ----------------------------------------------------------------------------
def build_parser():
    p = argparse.ArgumentParser(description="Find diagnostic alignments for a set of fasta files",
                                prog="krisp", formatter_class=argparse.RawTextHelpFormatter,
                                epilog="Limits of the GPU path, none of which the reference has (it works on text of any length;\n"
                                       "kstream.py:617-642) -- each is refused with a message, never answered wrongly:\n"
                                       "  * a sequence file of 2^33 bases or more (8.6 Gbp: the sorted k-mers of one such genome are what one GPU's\n"
                                       "    memory holds at all); with amplicons longer than 32 bases, 2^32 bases;\n"
                                       "  * amplicons longer than 32 bases (or more than 16 diagnostic bases): conserved flanks of 1 .. 256\n"
                                       "    bases each, amplicons of at most 1024;\n"
                                       "  * DNA and RNA genomes in one run together with IUPAC ambiguity letters over several ranks.\n"
                                       "A genome set that does not fit the GPU's memory sorted at once goes through it in batches\n"
                                       "(same result; KRISP_STREAM_BATCH=n forces batches of n genomes).")
    p.add_argument("files", nargs="+", type=str, metavar="PATH", help="Fasta file to read. .gz, .bz2")
    p.add_argument("--outgroup", nargs="*", type=str, default=[], metavar="PATH",
                   help="Outgroup Fasta files. To be amplified, but not detected")
    p.add_argument("-c", "--conserved", type=int, metavar="INT",
                   help="Length of conserved regions on ends of amplicon")
    p.add_argument("--conserved-left", type=int, metavar="INT",
                   help="Length of conserved region on left of amplicon")
    p.add_argument("--conserved-right", type=int, metavar="INT",
                   help="Length of conserved region on right of amplicon")
    p.add_argument("-d", "--diagnostic", type=int, metavar="INT", help="Diagnostic region length for amplicon")
    p.add_argument("-a", "--amplicon", type=int, metavar="INT", help="Total amplicon length")
    p.add_argument("--omit-soft", action="store_true", help="Omit softmasked nucleotides")
    p.add_argument("--cores", type=int, default=1, metavar="INT",
                   help="Total number of processors to utilize. (default: %(default)s)")
    p.add_argument("--dot-alignment", action="store_true", help="Output as dot-based alignments")
    p.add_argument("-o", "--out_align", type=str, metavar="PATH",
                   help="Write results as human-readable alignments to a file. (default: do not write alignment output)")
    p.add_argument("-s", "--out_csv", type=str, metavar="PATH",
                   help="Write results to as a CSV (comma-separated value) file. (default: print to screen (stdout))")
    p.add_argument("-w", "--workdir", type=str, metavar="PATH", help="Work directory to place temporary files")
    p.add_argument("-p", "--primer3", action=argparse.BooleanOptionalAction,
                   help="Design primers with Primer3 for every region found (needs the primer3-py package)")
    p.add_argument("--tm", type=int, nargs=2, metavar="INT", default=[53, 68])
    p.add_argument("--gc", type=int, nargs=2, metavar="INT", default=[40, 70])
    p.add_argument("--amp_size", type=int, nargs=2, metavar="INT", default=[70, 150])
    p.add_argument("--primer_size", type=int, nargs=2, metavar="INT", default=[25, 35])
    p.add_argument("--max_sec_tm", type=int, default=40, metavar="INT")
    p.add_argument("--gc_clamp", type=int, default=1, metavar="INT")
    p.add_argument("--max_end_gc", type=int, default=4, metavar="INT")
    p.add_argument("--verbose", action="store_true", help="Print runtime information to sys.stderr")
    p.add_argument("--device", type=int, default=0, metavar="INT", help="GPU to run on (default: 0)")
    p.add_argument("--devices", type=str, default=None, metavar="LIST",
                   help="several GPUs from ONE process, e.g. 0,1,2,3: genomes sharded over them, candidate lists "
                        "tree-reduced over RCCL (or start one process per GPU with torch.distributed.run / mpirun)")
    return p


def deduce_geometry(args, parser):
    """krisp_fasta.py:179-213, same precedence; exits 1 with the same message."""
    def fail():
        print("ERROR: Could not deduce input parameters", file=sys.stderr)
        parser.print_help(sys.stderr)
        sys.exit(1)

    if args.amplicon is not None:
        if args.diagnostic is not None:
            args.conserved = (args.amplicon - args.diagnostic) // 2
            args.conserved_left = args.conserved_right = args.conserved
        elif args.conserved is not None:
            args.diagnostic = args.amplicon - 2 * args.conserved
            args.conserved_left = args.conserved_right = args.conserved
        elif args.conserved_left is not None and args.conserved_right is not None:
            args.diagnostic = args.amplicon - args.conserved_left - args.conserved_right
        else:
            fail()
    elif args.diagnostic is not None:
        if args.conserved is not None:
            args.amplicon = args.diagnostic + 2 * args.conserved
            args.conserved_left = args.conserved_right = args.conserved
        elif args.conserved_left is not None and args.conserved_right is not None:
            args.amplicon = args.diagnostic + args.conserved_left + args.conserved_right
        else:
            fail()
    else:
        fail()
    return args


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(sys.argv[1:] if argv is None else argv)
    args = deduce_geometry(args, parser)
    if args.primer3:
        from . import primers
        if not primers.available():
            print("ERROR: --primer3 needs the primer3-py package, which is not installed here "
                  "(the k-mer path itself does not use it)", file=sys.stderr)
            sys.exit(2)
    t0 = time.time()
    if args.verbose:
        print("Finding kmer-based diagnostic regions for:", file=sys.stderr)
        for i, f in enumerate(args.files):
            print(f"({i}) {f}", file=sys.stderr)
        print("With this as an outgroup:", file=sys.stderr)
        for i, f in enumerate(args.outgroup):
            print(f"({i}) {f}", file=sys.stderr)
        print(file=sys.stderr)
    devices = [int(x) for x in str(args.devices).split(",")] if getattr(args, "devices", None) else None
    from . import distributed as _dist
    launch_world = _dist.env_rank_world()[2]     # torchrun / mpirun / srun alike (one place decides: distributed.py)
    if devices and len(devices) > 1 and launch_world == 1:
        # one process, a thread per device
        groups, stats = find_regions_multi_device(args.files, args.outgroup, args.conserved_left,
                                                  args.conserved_right, args.amplicon, devices,
                                                  omit_soft=args.omit_soft, verbose=args.verbose)
    elif launch_world > 1:
        # one process per GPU (python -m torch.distributed.run ... -m krisp_amd.krisp_fasta ...)
        groups, stats = find_regions_distributed(args.files, args.outgroup, args.conserved_left,
                                                 args.conserved_right, args.amplicon, omit_soft=args.omit_soft,
                                                 transport=os.environ.get("KRISP_COMM_TRANSPORT", "rccl"),
                                                 verbose=args.verbose)
        if groups is None:
            return 0                     # rank 0 writes the output
    else:
        groups, stats = find_regions(args.files, args.outgroup, args.conserved_left, args.conserved_right,
                                     args.amplicon, omit_soft=args.omit_soft, device=args.device,
                                     verbose=args.verbose)
    if args.verbose:
        print("\nRendering output ... ", file=sys.stderr)
    ingroup = [simplename(f) for f in args.files] if len(args.outgroup) else None
    if args.primer3:
        from . import primers
        p3 = primers.settings(**{k: getattr(args, k) for k in ("tm", "gc", "amp_size", "primer_size", "max_sec_tm",
                                                              "gc_clamp", "max_end_gc")})
        csv_text, align_text = primers.render(groups, ingroup, p3, dot=args.dot_alignment)
    else:
        csv_text, align_text = amplicon.render(groups, ingroup, dot=args.dot_alignment)
    if args.out_csv is not None:
        with open(args.out_csv, "w") as f:
            f.write(csv_text)
    else:
        sys.stdout.write(csv_text)
    if args.out_align is not None:
        with open(args.out_align, "w") as f:
            f.write(align_text)
    if args.verbose:
        print(f"=> Found {len(groups):,} regions in {prettyTime(time.time() - t0)} "
              f"({stats['kmers']:,} k-mers, device {stats['device_s']:.3f} s)", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
