"""`kstream` -- the reference's k-mer generator surface, MI355X-backed.

Same constructor, iteration, call and write() contract as the reference class
(kstream/kstream.py:122-428) and the same `kstream` command line
(kstream.py:835-952).  k-mer streams of k <= 32 run on the GPU through
libkrisp_hip.so (`kstream.device_plan`): the option combination krisp_fasta uses
(krisp_fasta.py:21-43: kmers=k, complements, disallow="Nn", omitsoft|mapsoft,
split=[L,-R], sort with sortcols=[0,2]) and its neighbours -- forward strand only or
canonicals instead of complements, no split or a one-sided split, sort columns that
leave the fields in line order, several k (one device sort per k, the sorted streams
merged), --allow of plain bases (a base mask in the pack kernel), and no sort at all
(keys in stream order, kr_genome_keys_in_order).  Round 4 adds: every sort-column order of the
fields a key layout can hold (kr_set_field_order), --expand-iupac and streams that keep their lower
case (the device sorts the windows of plain upper-case ACGT; the few windows holding anything else
go through the reference's own chain on the host, window by window, and are merged into the sorted
stream where the sort puts them), and k > 32 for the krisp_fasta combination (the wide path).  No CPU
sort or k-mer generation stands in for that route when the library is missing: it raises.  What has
no device plan is served by the plain host generator chain below; `plan_reason` says why
(device_plan's docstring lists the cases).
"""
import argparse
import itertools
import sys


from . import codec, fasta

# kstream.py:11-18
COMP_MAP = dict(zip("ATatGCgcRYryMKmkSWswBVbvDHdhNn", "TAtaCGcgYRyrKMkmSWswVBvbHDhdNn"))
# kstream.py:21-42
IUPAC_BASE = {"R": "AG", "Y": "CT", "S": "GC", "W": "AT", "K": "GT", "M": "AC",
              "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}
IUPAC_BASE.update({k.lower(): v.lower() for k, v in list(IUPAC_BASE.items())})

_WRITE_CHUNK = 1 << 22      # keys decoded to text per chunk


def _revcomp(s):
    return "".join([COMP_MAP[c] for c in reversed(s)])   # KeyError as kstream.py:658


def _rewindable(sequences):
    """a one-shot iterator may have to be read twice (device attempt, then the host chain): keep
    its items; the reference's first-line quirk for one-shot inputs (kstream.py:450) is preserved
    by handing on an iterator again"""
    if isinstance(sequences, (str, bytes)) or hasattr(sequences, "__fspath__") or not hasattr(sequences, "__next__"):
        return sequences
    return _Replay(list(sequences))


class _Replay:
    """list-backed stand-in for a one-shot iterator: every iter() starts over, and it still
    looks one-shot (has __next__) to the reader's first-line rule"""

    def __init__(self, items):
        self.items = items
        self._it = iter(items)

    def __iter__(self):
        self._it = iter(self.items)
        return self

    def __next__(self):
        return next(self._it)


class kstream:
    def __init__(self, sequences=None, kmers=None, complements=False,
                 canonicals=False, allow=None, disallow=None, omitsoft=False,
                 mapsoft=False, expandiupac=False, split=None, sort=False,
                 sortmem=None, sortcols=None, sortnp=1, parallel=1, *, device=0):
        self.kmers = None
        if kmers is not None:
            self.kmers = [kmers] if isinstance(kmers, int) else list(kmers)
        if omitsoft is True and mapsoft is True:
            raise ValueError("can't omit and map soft masked nucleotides")
        if complements is True and canonicals is True:
            raise ValueError("canonicals conflicts with complements")
        self.omitsoft = omitsoft is True
        self.mapsoft = mapsoft is True
        self.complements = complements is True
        self.canonicals = canonicals is True
        self.allow = None if allow is None else set(allow)
        self.disallow = None if disallow is None else set(disallow)
        self.expandiupac = expandiupac is True
        self.split = None
        if split is not None:
            self.split = [split] if isinstance(split, int) else list(split)
        self.sort = sort
        self.sortnp, self.sortmem, self.sortcols = sortnp, sortmem, sortcols
        self.parallel = parallel
        self.sequences = sequences
        self.device = device

    # ------------------------------------------------------------------ device path
    def device_geometry(self):
        """(L, D, R) when this option set is THE krisp_fasta combination (krisp_fasta.py:21-43)."""
        plan = self.device_plan()
        if (plan is None or plan.get("multi") or plan.get("wide") or plan["strands"] != 0 or plan["layout"] != "lrd"
                or len(plan["fields"]) != 3 or not plan["sorted"] or plan["allow"] is not None
                or plan["keepcase"] or plan["expand"] or self.disallow != {"N", "n"} or self.allow is not None):
            return None
        return plan["geometry"]

    def _no_plan(self, why):
        self.plan_reason = why
        return None

    def device_plan(self):
        """How the GPU serves this option set, or None (-> the host generator chain; `plan_reason` then says why).

        On the device: k <= 32 (one or several: one sort per k, the sorted streams merged); both strands (complements),
        forward only, or canonicals; omitsoft, mapsoft or neither (lower case kept); any --disallow / --allow set, or none
        (what they say about A, C, G, T is a base mask on the device, the rest applies to the host's special windows);
        --expand-iupac; split None, [a], [a, -b] (a, b >= 0); sorted with ANY
        --sort-cols -- GNU sort falls back to the whole line, so a column list is a permutation of the fields followed
        by line order, and the key holds the fields in that order (kr_set_field_order; the krisp_fasta order (first,
        last, middle) keeps its own layout and kernels) -- or unsorted (stream order: the device's k-mers in window
        order, the host's special windows placed among them by position; several k record by record).  k > 32: the krisp_fasta
        combination through the wide path (flanks <= 256, k <= 1024).
        The device carries windows of plain ACGT (upper case only when lower case is kept or omitted); a window holding
        anything else that the chain would keep -- IUPAC letters, lower case under 'neither', other characters -- runs
        through the reference's chain on the host by itself and joins the sorted stream.
        Not on the device, with the reason in `plan_reason`:
          * a column order that cuts the window into more than eight pieces (any split list, any column list below that
            has a key layout: three blocks by shifts, kr_set_field_order; more, or pieces out of window order -- two
            split sizes counted from the end --, piece by piece, kr_set_field_pieces);
          * k > 32 outside the krisp_fasta combination or without --sort, flanks > 256, k > 1024."""
        self.plan_reason = None
        if self.kmers is None or len(self.kmers) < 1:
            return self._no_plan("no k given: the sequences pass through as they are")
        if len(self.kmers) > 1:
            # several k: one device plan (one sort) per k, the sorted streams merged by the same
            # comparator (unsorted, the windows of every record come k by k: host chain)
            plans = [self._plan_one(k) for k in self.kmers]
            if any(p is None for p in plans):
                return None
            # (unsorted, the windows of a record come k by k, kstream.py:631-642: one stream-order pass per k on the
            # device, put together record by record -- round 5)
            return dict(multi=plans, strands=plans[0]["strands"], layout="multi", fields=None, geometry=None,
                        sorted=self.sort is True)
        return self._plan_one(self.kmers[0])

    def _plan_one(self, k):
        if k < 1:
            return self._no_plan("k < 1")
        if self.sort not in (True, False):
            return self._no_plan("sort must be True or False")
        keepcase = not self.omitsoft and not self.mapsoft
        strands = 0 if self.complements else (2 if self.canonicals else 1)
        # --allow / --disallow (kstream.py:696-732).  The device carries windows of plain upper-case ACGT only, so of both
        # sets only what they say about A, C, G, T decides there: a base mask in the pack kernel (kr_set_allow).  Every
        # other character -- ambiguity letters, N when nothing drops it, '-', lower case that is kept -- makes its window a
        # "special" that runs through the reference's own chain on the host (_special_outputs), where both sets apply as
        # they are (round 5: any disallow set, none at all, --allow of any letters).
        # Both strands are emitted BEFORE the filters: the surviving bases must be closed under complement, or a window
        # and its reverse complement would have to be told apart on the device
        base_ok = set("ACGT")
        if self.allow is not None:
            base_ok &= self.allow
        if self.disallow is not None:
            base_ok -= self.disallow
        allow_bases = None if base_ok == set("ACGT") else "".join(sorted(base_ok))
        split_strands = False
        if strands == 0 and {COMP_MAP[b] for b in base_ok} != base_ok:
            # a window and its reverse complement are filtered each by itself (the complements are formed BEFORE the filters,
            # kstream.py:696-766): one strand may stay where the other goes.  Sorted streams (round 6): two forward-only
            # passes -- over the sequences and over their reverse complements -- merged; in stream order a window's two
            # k-mers come from the two passes by the position of their window (round 6, _device_keys)
            split_strands = True
        # fields of the output line (kstream.py:805-832)
        if self.split is None:
            fields = [k]
        else:
            # kstream.py:805-832: the sizes are taken in turn -- a size >= 0 cuts that many characters off the FRONT of what
            # is left, a negative one off its END --, the line is front parts + what is left + end parts, the end parts in
            # the order they were cut.  Any number of sizes (round 6: two at most before) whose parts come out in WINDOW
            # order: at most one of them negative (a second end part would stand behind the first in the line, in front
            # of it in the window: codec.Fields keeps every column's place in the window).  -0 is 0: an empty front part
            # (kstream.py:824-830).
            head, tail, lo, hi = [], [], 0, k
            for z in self.split:
                if (z >= 0 and z > hi - lo) or (z < 0 and -z > hi - lo):
                    return self._no_plan("split point outside the k-mer")
                if z >= 0:
                    head.append((lo, z))
                    lo += z
                else:
                    tail.append((hi + z, -z))
                    hi += z
            pieces = head + [(lo, hi - lo)] + tail                  # (offset in the window, width) of the line's columns
            fields = codec.Fields([w for _, w in pieces], [o for o, _ in pieces])
        common = dict(k=k, fields=fields, strands=strands, allow=allow_bases, keepcase=keepcase, expand=self.expandiupac,
                      split_strands=split_strands)
        if self.sort is False:
            if k > 32:
                return self._no_plan("k > 32 without --sort")
            # stream order: the window as it is, cut into its fields
            return dict(common, layout="ldr", order=list(range(len(fields))), geometry=(k, 0, 0), sorted=False)
        # effective order of the fields: listed columns, then line order
        cols = [] if self.sortcols is None else list(self.sortcols)
        if any((not isinstance(c, int)) or c < 0 or c >= len(fields) for c in cols):
            return self._no_plan("--sort-cols names a column the lines do not have")
        order = []
        for c in cols + list(range(len(fields))):
            if c not in order:
                order.append(c)
        live = [c for c in order if fields[c] > 0]             # empty fields do not order anything
        natural = [c for c in range(len(fields)) if fields[c] > 0]
        in_window_order = codec.Fields(fields, codec.field_offsets(fields)).in_window_order()
        krisp_order = in_window_order and len(fields) == 3 and live == [c for c in (0, 2, 1) if fields[c] > 0]
        if k > 32:
            # amplicons longer than one key: the wide path sorts the krisp_fasta combination (krisp_fasta.py:21-43)
            from . import _native
            if not (krisp_order and strands == 0 and allow_bases is None and not keepcase and not self.expandiupac
                    and self.disallow == {"N", "n"}):
                return self._no_plan("k > 32 outside the krisp_fasta combination (complements, disallow Nn, a soft-mask rule, split [L, -R], sort columns 0 2)")
            L, D, R = codec.effective_geometry(*fields)
            if not (1 <= L <= _native.WIDE_MAX_FLANK and 1 <= R <= _native.WIDE_MAX_FLANK and k <= _native.WIDE_MAX_K):
                return self._no_plan(f"k > 32 with flanks outside 1..{_native.WIDE_MAX_FLANK} or k > {_native.WIDE_MAX_K}")
            return dict(common, wide=True, layout="lrd", order=[0, 2, 1], geometry=tuple(fields), sorted=True)
        if krisp_order and fields[1] <= 16:
            layout, geometry = "lrd", (fields[0], fields[1], fields[2])
            order = [0, 2, 1]
        elif live == natural and in_window_order:
            layout, geometry = "ldr", (k, 0, 0)
            order = list(range(len(fields)))
        else:
            layout, geometry = "custom", (k, 0, 0)
            order = live + [c for c in range(len(fields)) if c not in live]
            merged = codec.merge_fields(fields, order)
            if merged is None or not codec.field_layout_ok(*merged):
                # (more than three blocks: the key as a list of pieces -- kr_set_field_pieces, round 6)
                if len(codec.key_pieces(fields, order)) > 8:
                    return self._no_plan("this column order cuts the window into more than eight pieces: no key layout holds it")
        # (--expand-iupac: windows holding N are dropped before the expansion -- the disallow / allow test above --, so an
        # expansion is the handful of combinations of a window's other ambiguity letters)
        return dict(common, layout=layout, order=order, geometry=geometry, sorted=True)

    def _char_tables(self, plan):
        """(plain, hard, soft, raiser) by byte value.  plain: what the device carries (A C G T; a c g t too under
        mapsoft).  hard: a character that drops its window BEFORE anything else looks at it -- the record boundary,
        lower case under omitsoft (the soft-mask step comes first).  soft: a character the filters drop -- after the
        soft-mask mapping it is in --disallow or outside --allow; with both strands emitted (the complements are formed
        BEFORE the filters) only when its complement is dropped too.  raiser: with both strands emitted, a character
        outside COMP_MAP makes the complement step raise KeyError (kstream.py:658) whatever the filters would have said
        about its window."""
        import numpy as np
        plain = np.zeros(256, dtype=bool)
        plain[list(b"ACGT")] = True
        if self.mapsoft:
            plain[list(b"acgt")] = True
        hard = np.zeros(256, dtype=bool)
        soft = np.zeros(256, dtype=bool)
        raiser = np.zeros(256, dtype=bool)
        hard[10] = True

        def bad(ch):
            return (self.disallow is not None and ch in self.disallow) or (self.allow is not None and ch not in self.allow)
        for c in range(256):
            ch = chr(c)
            if c == 10:
                continue
            if self.omitsoft and ch.islower():
                hard[c] = True
                continue
            m = ch.upper() if self.mapsoft else ch
            if len(m) != 1:
                continue
            if plan["strands"] == 0:
                if m in COMP_MAP:
                    soft[c] = bad(m) and bad(COMP_MAP[m])
                else:
                    raiser[c] = True
            else:
                soft[c] = bad(m)
        return plain, hard, soft, raiser

    def _has_specials(self, bases, plan):
        """does any character call for the host's special windows?  (in pieces: no second copy of a genome)"""
        plain, hard, soft, _ = self._char_tables(plan)
        either = plain | hard | soft
        step = 1 << 26
        return any(not either[bases[a:a + step]].all() for a in range(0, len(bases), step))

    def _special_outputs(self, bases, plan, by_window=False):
        """The k-mers of the windows the device does not carry but the chain may keep -- or raise on --, as the reference's
        own chain makes them (window by window, stream order: the first KeyError is the reference's): windows without a
        character that drops them at once, holding a raiser, or a character that is not plain and none the filters drop
        (_char_tables).  One vectorised pass finds them; Python only runs on those windows."""
        import numpy as np
        k = plan["k"]
        n = len(bases)
        if n < k:
            return []
        plain, hard, soft, raiser = self._char_tables(plan)
        isp = (~plain[bases]) & (~hard[bases]) & (~soft[bases])
        if not isp.any():
            return []

        def windows(flag):
            c = np.concatenate([[0], np.cumsum(flag, dtype=np.int64)])
            return c[k:] - c[:-k]
        nh, nsft, nsp, nr = windows(hard[bases]), windows(soft[bases]), windows(isp), windows(raiser[bases])
        starts = np.flatnonzero((nh == 0) & (((nsp > 0) & (nsft == 0)) | (nr > 0)))
        text = bases.tobytes().decode("latin-1")
        if by_window:
            # (unsorted streams: every window's k-mers with its start, to be put between the device's by position)
            return [(i, list(self._chain(iter([text[i:i + k]])))) for i in starts.tolist()]
        return list(self._chain(text[i:i + k] for i in starts.tolist()))

    def _device_window_starts(self, bases, plan):
        """starts of the windows the device emits k-mers for, ascending: k characters that are all plain (_char_tables) and,
        after the soft-mask mapping, bases the --allow / --disallow mask leaves"""
        import numpy as np
        k = plan["k"]
        plain = self._char_tables(plan)[0].copy()
        if plan["allow"] is not None:
            for ch in "ACGTacgt":
                if ch.upper() not in plan["allow"]:
                    plain[ord(ch)] = False
        if len(bases) < k:
            return np.zeros(0, dtype=np.int64)
        c = np.concatenate([[0], np.cumsum(~plain[bases], dtype=np.int64)])
        return np.flatnonzero(c[k:] - c[:-k] == 0)

    def _device_keys(self, sequences, plan, want_layout=False):
        """-> (sorted keys in the plan's field order, is_rna, host k-mers as window strings) or None when only the host
        chain reproduces the stream (plan_reason says why)."""
        import numpy as np
        from . import _native
        L, D, R = plan["geometry"]
        fields, order = plan["fields"], plan["order"]
        # (the krisp_fasta combination proper -- disallow "Nn", no --allow beyond plain bases and N -- has its own side
        # channel for IUPAC windows, kr_scan_special; every other option set finds its special windows from the tables)
        krisp_combo = (plan["strands"] == 0 and plan["layout"] == "lrd" and len(fields) == 3 and plan["sorted"]
                       and not plan["keepcase"] and not plan["expand"] and self.disallow == {"N", "n"}
                       and (self.allow is None or self.allow <= set("ACGTNacgtn")))
        allow = plan.get("allow")
        if krisp_combo:
            bases, rna, windows = fasta.ingest(sequences, plan["k"], self.omitsoft)    # (KeyError as the reference)
            # (--allow of plain bases drops every k-mer that holds an ambiguity letter)
            special = [] if self.allow is not None else list(windows)
        else:
            bases, rna, nspecial = fasta.load_any(sequences)
            special = []
            if nspecial or plan["keepcase"] or self._has_specials(bases, plan):
                # (unsorted: the host's k-mers are put between the device's by the position of their window, round 5)
                special = self._special_outputs(bases, plan, by_window=not plan["sorted"])
        with _native.Engine(device=self.device) as eng:
            # (lower case kept: the device takes the windows without any, as under omitsoft; the others are `special`)
            eng.set_params(L, D, R, omit_soft=self.omitsoft or plan["keepcase"], max_bases=len(bases))
            if plan["layout"] == "custom":
                merged = codec.merge_fields(fields, order)      # (neighbouring columns that stay neighbours are one block of the key)
                if merged is not None and codec.field_layout_ok(*merged):
                    mf, mo = merged
                    eng.set_field_order(mf + [0] * (3 - len(mf)), mo + list(range(len(mf), 3)))
                else:
                    eng.set_field_pieces(codec.key_pieces(fields, order))
            if plan["strands"]:
                eng.set_strands(plan["strands"])
            elif plan.get("split_strands"):
                eng.set_strands(_native.STRANDS_FORWARD)
            if allow is not None:
                eng.set_allow(allow)
            if plan.get("split_strands") and plan["sorted"]:
                # (the reverse complement of every record: a k-mer of it that the base mask lets through is the reverse
                # complement of a window whose own k-mer may have been dropped, and the other way round)
                comp = np.arange(256, dtype=np.uint8)
                for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
                    comp[a] = b
                eng.add(0, bases)
                eng.add(1, np.ascontiguousarray(comp[bases[::-1]]))
                keys = np.sort(np.concatenate([eng.keys(0), eng.keys(1)]), kind="stable")
            elif plan["sorted"]:
                eng.add(0, bases)
                keys = eng.keys(0).copy()
            elif plan.get("split_strands"):
                # stream order, bases the strands do not share (round 6): a window is followed by its reverse complement, each
                # filtered by itself (kstream.py:696-766).  Two forward-only passes in window order -- over the text: the
                # windows the base mask lets through; over its reverse complement: the reverse complements the mask lets
                # through, window q there = the window that starts at n - q - k here -- put together by window start
                comp = np.arange(256, dtype=np.uint8)
                for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
                    comp[a] = b
                nb, kk = len(bases), plan["k"]
                rcb = np.ascontiguousarray(comp[bases[::-1]])
                eng.upload(0, bases)
                kf = eng.keys_in_order(0, nb).copy()
                eng.upload(1, rcb)
                kr = eng.keys_in_order(1, nb).copy()[::-1]
                sf = self._device_window_starts(bases, plan)
                sr = (nb - self._device_window_starts(rcb, plan) - kk)[::-1]
                assert len(kf) == len(sf) and len(kr) == len(sr), (len(kf), len(sf), len(kr), len(sr))
                pos = np.concatenate([sf, sr])
                order = np.lexsort((np.concatenate([np.zeros(len(sf), dtype=np.int8), np.ones(len(sr), dtype=np.int8)]), pos))
                keys = np.concatenate([kf, kr])[order]
                split_starts, cnt = np.unique(pos, return_counts=True)
                split_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
            else:
                eng.upload(0, bases)
                keys = eng.keys_in_order(0, len(bases))
        if not plan["sorted"]:
            # stream order: (keys, rna, [(window start, its k-mers)], starts of the device's windows) -- _device_blocks interleaves
            if plan.get("split_strands"):
                # (0, 1 or 2 k-mers per window: their places in `keys`)
                if want_layout:
                    return keys, rna, special, split_starts, np.flatnonzero(bases == 10), split_off
                return keys, rna, special, split_starts, split_off
            if want_layout:
                return keys, rna, special, self._device_window_starts(bases, plan), np.flatnonzero(bases == 10)
            return keys, rna, special, (self._device_window_starts(bases, plan) if special else None)
        # host k-mers of plain ACGT (expansions of IUPAC letters) are keys like the device's
        if special and not krisp_combo:
            isplain = [not s.strip("ACGT") for s in special]
            pk = codec.pack_plain([codec.order_string(s, fields, order) for s, p in zip(special, isplain) if p])
            if len(pk):
                keys = np.sort(np.concatenate([keys, pk]), kind="stable")
            special = [s for s, p in zip(special, isplain) if not p]
        return keys, rna, special, None

    def _device_blocks(self, sequences, plan):
        """-> (iterator of byte blocks of the output, line count) or None"""
        if plan.get("multi") and not plan["sorted"]:
            # several k in stream order: per record, for each k in turn, that k's windows (kstream.py:631-642).  One stream-
            # order pass per k on the device; a record's share of each pass is cut out by the positions of its separators
            import numpy as np
            passes, total, rna, seps = [], 0, None, None
            for sub in plan["multi"]:
                got = self._device_keys(sequences, sub, want_layout=True)
                if got is None:
                    return None
                per = 2 if sub["strands"] == 0 else 1
                if len(got) == 6:           # (bases the strands do not share: 0, 1 or 2 k-mers per window)
                    keys, rna, special, starts, seps, off = got
                else:
                    keys, rna, special, starts, seps = got
                    assert len(keys) == per * len(starts), (len(keys), len(starts))
                    off = per * np.arange(len(starts) + 1, dtype=np.int64)
                passes.append((sub, keys, list(special), starts, off))
                total += int(len(keys)) + sum(len(km) for _, km in special)
            bounds = np.concatenate([[-1], seps, [np.iinfo(np.int64).max]]).astype(np.int64)

            def blocks():
                sp_at = [0] * len(passes)
                for r in range(len(bounds) - 1):
                    lo_pos, hi_pos = int(bounds[r]) + 1, int(bounds[r + 1])            # the record's characters [lo_pos, hi_pos)
                    for pi, (sub, keys, special, starts, off) in enumerate(passes):
                        a = int(np.searchsorted(starts, lo_pos))
                        b = int(np.searchsorted(starts, hi_pos))
                        while True:
                            # the record's special windows of this k, each put in by its position
                            nxt = special[sp_at[pi]] if sp_at[pi] < len(special) and special[sp_at[pi]][0] < hi_pos else None
                            upto = b if nxt is None else int(np.searchsorted(starts, nxt[0]))
                            for x in range(a, upto, _WRITE_CHUNK):
                                y = min(upto, x + _WRITE_CHUNK)
                                yield codec.keys_to_fields_bytes(keys[off[x]:off[y]], sub["fields"], rna)
                            a = upto
                            if nxt is None:
                                break
                            sp_at[pi] += 1
                            lines = [self._split_one(w) if self.split is not None else w for w in nxt[1]]
                            if rna:
                                lines = [w.replace("T", "U").replace("t", "u") for w in lines]
                            if lines:
                                yield ("\n".join(lines) + "\n").encode("latin-1")
            return blocks(), total
        if plan.get("multi"):
            # several k: every k sorted on the device, the streams merged under the one comparator
            # of `sort -t, -kN,N ...` (listed columns, then the whole line)
            import heapq
            streams, total = [], 0
            for sub in plan["multi"]:
                got = self._device_blocks(sequences, sub)
                if got is None:
                    return None
                streams.append(b"".join(got[0]).split(b"\n")[:-1])
                total += got[1]
            cols = self.sortcols

            def key(ln):
                if cols is None:
                    return (ln,)
                f = ln.split(b",")
                return tuple(f[c] if c < len(f) else b"" for c in cols) + (ln,)
            merged = heapq.merge(*streams, key=key)

            def blocks():
                buf = []
                for ln in merged:
                    buf.append(ln)
                    if len(buf) >= 1 << 16:
                        yield b"\n".join(buf) + b"\n"
                        buf = []
                if buf:
                    yield b"\n".join(buf) + b"\n"
            return blocks(), total
        if plan.get("wide"):
            # k > 32, the krisp_fasta combination: the wide path's sorted-file writer (krisp_fasta.extractSortedKmers)
            import os
            import tempfile
            from . import krisp_fasta as KF
            L, _, R = plan["fields"]
            fd, tmp = tempfile.mkstemp(prefix="kstream_wide_")
            os.close(fd)
            try:
                n = KF._extract_sorted_wide(sequences, L, R, plan["k"], tmp, self.omitsoft, self.device, False)
                with open(tmp, "rb") as f:
                    data = f.read()
            finally:
                os.unlink(tmp)
            return iter([data]), n
        got = self._device_keys(sequences, plan)
        if got is None:
            return None
        dev_off = None
        if len(got) == 5:               # (stream order with bases the strands do not share: 0, 1 or 2 k-mers per window)
            keys, rna, special, dev_starts, dev_off = got
        else:
            keys, rna, special, dev_starts = got
        if not plan["sorted"] and special:
            import numpy as np
            # the device's k-mers in stream order (per window: the window, then its reverse complement under --complements)
            # with the host's special windows' k-mers put in by the position of their window
            per = 2 if plan["strands"] == 0 else 1
            if dev_off is None:
                assert len(keys) == per * len(dev_starts), (len(keys), len(dev_starts))
                dev_off = per * np.arange(len(dev_starts) + 1, dtype=np.int64)
            fields = plan["fields"]
            width = plan["k"] + len(fields)                     # bytes of a device line, newline included

            def blocks():
                done = 0                                        # device windows written so far
                for start, kmers in special:
                    upto = int(np.searchsorted(dev_starts, start))
                    for a in range(done, upto, _WRITE_CHUNK):
                        b = min(upto, a + _WRITE_CHUNK)
                        yield codec.keys_to_fields_bytes(keys[dev_off[a]:dev_off[b]], fields, rna)
                    done = upto
                    lines = [self._split_one(x) if self.split is not None else x for x in kmers]
                    if rna:
                        lines = [x.replace("T", "U").replace("t", "u") for x in lines]
                    if lines:
                        yield ("\n".join(lines) + "\n").encode("latin-1")
                for a in range(done, len(dev_starts), _WRITE_CHUNK):
                    b = min(len(dev_starts), a + _WRITE_CHUNK)
                    yield codec.keys_to_fields_bytes(keys[dev_off[a]:dev_off[b]], fields, rna)
            return blocks(), int(len(keys)) + sum(len(km) for _, km in special)
        if (plan["layout"] == "lrd" and len(plan["fields"]) == 3 and not plan["keepcase"] and not plan["expand"] and plan["strands"] == 0
                and plan["sorted"] and self.disallow == {"N", "n"} and (self.allow is None or self.allow <= set("ACGTNacgtn"))):
            L, D, R = plan["geometry"]
            blocks = codec.merged_line_blocks(keys, [codec.split_window(w, L, D, R) for w in special], L, D, R, rna=rna,
                                              chunk=_WRITE_CHUNK)
        elif special or plan["layout"] != "ldr":
            blocks = codec.merged_ordered_blocks(keys, special, plan["fields"], plan["order"], rna=rna, chunk=_WRITE_CHUNK)
        else:
            blocks = (codec.keys_to_fields_bytes(keys[i:i + _WRITE_CHUNK], plan["fields"], rna)
                      for i in range(0, len(keys), _WRITE_CHUNK))
        return blocks, int(len(keys)) + len(special)

    # ------------------------------------------------------------------ host chain
    def _host_stream(self, sequences):
        records = [r.decode("latin-1") for r in fasta.read_records(sequences)]
        rna = None
        for s in records:                                    # kstream.py:481-508
            if "T" in s or "t" in s:
                rna = False
                break
            if "U" in s or "u" in s:
                rna = True
                break
        seqs = iter(records)
        if rna:
            seqs = (s.replace("U", "T").replace("u", "t") for s in seqs)
        if self.kmers is not None:
            ks = self.kmers
            seqs = (s[i:i + k] for s in seqs for k in ks for i in range(len(s) - k + 1))
        return self._split_all(self._chain(seqs)), rna

    def _chain(self, seqs):
        """the parsers behind the window cutter, in the reference's fixed order (kstream.py:203-235): soft mask,
        complements, allow, disallow, expand-iupac, canonicals -- everything but the split"""
        if self.omitsoft:
            seqs = (s for s in seqs if s.isupper())
        if self.mapsoft:
            seqs = (s.upper() for s in seqs)
        if self.complements:
            seqs = (x for s in seqs for x in (s, _revcomp(s)))
        if self.allow is not None:
            allow = self.allow
            seqs = (s for s in seqs if set(s) <= allow)
        if self.disallow is not None:
            bad = self.disallow
            seqs = (s for s in seqs if bad.isdisjoint(s))
        if self.expandiupac:
            seqs = self._expand(seqs)
        if self.canonicals:
            seqs = (min(s, _revcomp(s)) for s in seqs)
        return seqs

    def _split_all(self, seqs):
        if self.split is not None:
            seqs = (self._split_one(s) for s in seqs)
        return seqs

    @staticmethod
    def _expand(seqs):
        for s in seqs:
            pos = [i for i, c in enumerate(s) if c in IUPAC_BASE]
            if not pos:
                yield s
                continue
            t = list(s)
            for combo in itertools.product(*[IUPAC_BASE[s[i]] for i in pos]):
                for i, c in zip(pos, combo):
                    t[i] = c
                yield "".join(t)

    def _split_one(self, s):
        head, tail = [], []
        for z in self.split:
            if z >= 0:
                head.append(s[:z])
                s = s[z:]
            else:
                tail.append(s[z:])
                s = s[:z]
        return ",".join(head + [s] + tail)

    def _host_sorted(self, lines):
        """LC_ALL=C sort [-t, -kN,N ...] semantics (kstream.py:83-119): keys in byte
        order, then GNU sort's whole-line last-resort compare."""
        cols = self.sortcols
        if cols is None:
            return sorted(lines)

        def key(ln):
            f = ln.split(",")
            return tuple(f[c] if c < len(f) else "" for c in cols) + (ln,)
        return sorted(lines, key=key)

    # ------------------------------------------------------------------ public surface
    def __call__(self, sequences):
        plan = self.device_plan()
        if plan is not None:
            sequences = _rewindable(sequences)
            got = self._device_blocks(sequences, plan)
            if got is not None:
                for blob in got[0]:
                    yield from blob.decode("ascii").split("\n")[:-1]
                return
        yield from self.host_lines(sequences)

    def host_lines(self, sequences):
        """the plain host generator chain, whatever the option set (the fallback of the device
        route and the CPU tests' handle on it)"""
        seqs, rna = self._host_stream(sequences)
        if self.sort:
            seqs = self._host_sorted(list(seqs))
        if rna:
            seqs = (s.replace("T", "U").replace("t", "u") for s in seqs)
        yield from seqs

    def __iter__(self):
        return iter(self.__call__(self.sequences))

    def write(self, filename, sequences=None):
        """kstream.py:250-325: write (sorted) k-mers, return their number."""
        if sequences is None:
            sequences = self.sequences
        plan = self.device_plan()
        if plan is not None:
            sequences = _rewindable(sequences)
            got = self._device_blocks(sequences, plan)
            if got is not None:
                with open(filename, "wb") as f:
                    for blob in got[0]:
                        f.write(blob)
                return got[1]
        seqs, rna = self._host_stream(sequences)
        if rna:
            seqs = (s.replace("T", "U").replace("t", "u") for s in seqs)
        lines = list(seqs)
        if self.sort:
            lines = self._host_sorted(lines)
        with open(filename, "w") as f:
            for ln in lines:
                f.write(ln + "\n")
        return len(lines)


def parseArgs(sys_args):
    """kstream.py:835-922."""
    p = argparse.ArgumentParser(
        description="Read and parse kmers from fasta or kmer stream\nCompatible with gz, bz2, and stdin.",
        prog="kstream", formatter_class=argparse.RawTextHelpFormatter)
    p.add_argument("file", nargs="?", type=str, default="-",
                   help="Fasta file to read. .gz, .bz2, default stdin")
    p.add_argument("-k", "--kmers", type=int, nargs="+",
                   help="Convert sequences into kmers of given length(s).")
    g = p.add_mutually_exclusive_group()
    g.add_argument("--canonicals", action="store_true",
                   help="Print canonical sequences (alphabetically first)")
    g.add_argument("--complements", action="store_true", help="Add reverse complement to stream")
    p.add_argument("--disallow", type=str, help="Omit sequences containing dissallowed nucleotides")
    p.add_argument("--allow", type=str, help="Only accept sequences containing allowed nucleotides")
    p.add_argument("--expand-iupac", action="store_true",
                   help="Expand IUPAC nucleotide codes (including N's)")
    p.add_argument("--omit-softmask", action="store_true", help="Omit sequences containing soft masking")
    p.add_argument("--map-softmask", action="store_true", help="Unmask sequences containing soft masking")
    p.add_argument("--split", nargs="+", type=int, help="Split kmers into columns and delimit by ','")
    p.add_argument("-p", "--parallel", type=int, default=1, help="Number of processors to use. Default 1")
    p.add_argument("-s", "--sort", action="store_true", help="Sort resulting kmers")
    p.add_argument("--sort-np", type=int, default=1, help="Number of processores to use for sorting")
    p.add_argument("--sort-mem", type=str, help="Amount of memory to use, see linux sort mem usage")
    p.add_argument("--sort-cols", nargs="+", type=int, help="Sort based on these columns, 0-based indexing")
    p.add_argument("--output", help="Write output to file as opposed to terminal")
    p.add_argument("--device", type=int, default=0, help="GPU to use for the accelerated combination")
    p.add_argument("--version", action="version", version="%(prog)s 1.0")
    return p.parse_args(sys_args)


def main(argv=None):
    args = parseArgs(sys.argv[1:] if argv is None else argv)
    streamer = kstream(kmers=args.kmers, complements=args.complements, canonicals=args.canonicals,
                       allow=args.allow, disallow=args.disallow, omitsoft=args.omit_softmask,
                       mapsoft=args.map_softmask, expandiupac=args.expand_iupac, split=args.split,
                       parallel=args.parallel, sort=args.sort, sortnp=args.sort_np,
                       sortmem=args.sort_mem, sortcols=args.sort_cols, device=args.device)
    source = args.file
    if source == "-":
        source = iter(sys.stdin.buffer.read().splitlines())
    if args.output is not None:
        with open(args.output, "w") as fout:
            for seq in streamer(source):
                print(seq, file=fout)
    else:
        for seq in streamer(source):
            print(seq)


if __name__ == "__main__":
    main()
