"""`kstream` -- the reference's k-mer generator surface, MI355X-backed.

Same constructor, iteration, call and write() contract as the reference class
(kstream/kstream.py:122-428) and the same `kstream` command line
(kstream.py:835-952).  Sorted single-k streams run on the GPU through
libkrisp_hip.so (`kstream.device_plan`): the option combination krisp_fasta uses
(krisp_fasta.py:21-43: kmers=k, complements, disallow="Nn", omitsoft|mapsoft,
split=[L,-R], sort with sortcols=[0,2]) and its neighbours -- forward strand only or
canonicals instead of complements, no split or a one-sided split, sort columns that
leave the fields in line order.  No CPU sort or k-mer generation stands in for that
route when the library is missing: it raises.  Option sets outside it (allow,
expand-iupac, several k, unsorted streaming, kept lower case, other column orders) are
outside the accelerated hot path and are served by the plain host
generator chain below, as are inputs holding characters the 2-bit alphabet cannot carry
under the forward / canonical modes.
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
        if plan is None or plan["strands"] != 0 or plan["layout"] != "lrd" or len(plan["fields"]) != 3:
            return None
        return plan["geometry"]

    def device_plan(self):
        """How the GPU serves this option set, or None (-> the host generator chain).

        Accelerated: one k <= 32; both strands (complements), forward only, or canonicals; one of
        omitsoft / mapsoft; disallow == 'Nn'; sort=True; split None, [a], [a, -b] (a, b >= 0);
        sort columns that order the line's fields as (all fields in line order) or
        (first, last, middle) -- GNU sort falls back to the whole line, so any column list is a
        permutation of the fields followed by line order.  The window is packed as ONE key whose
        unsigned order is that field order:
          layout 'ldr'  the window as it is            -> engine geometry (k, 0, 0)
          layout 'lrd'  first | last | middle field    -> engine geometry (first, middle, last)
        Everything else the reference supports (allow, expand-iupac, several k, unsorted streaming,
        other column orders, keeping lower case) stays on the host chain."""
        if self.kmers is None or len(self.kmers) != 1:
            return None
        k = self.kmers[0]
        if not (1 <= k <= 32):
            return None
        if self.allow is not None or self.expandiupac or self.disallow != {"N", "n"}:
            return None
        if self.omitsoft == self.mapsoft or self.sort is not True:
            return None
        strands = 0 if self.complements else (2 if self.canonicals else 1)
        # fields of the output line (kstream.py:805-832)
        if self.split is None:
            fields = [k]
        else:
            if len(self.split) not in (1, 2):
                return None
            a = self.split[0]
            if a < 0 or a > k:
                return None
            if len(self.split) == 1:
                fields = [a, k - a]
            else:
                b = self.split[1]
                if b > 0 or a - b > k:
                    return None
                fields = [a, 0, k - a] if b == 0 else [a, k - a + b, -b]     # kstream.py:824-830
        # effective order of the fields: listed columns, then line order
        cols = [] if self.sortcols is None else list(self.sortcols)
        if any((not isinstance(c, int)) or c < 0 or c >= len(fields) for c in cols):
            return None
        order = []
        for c in cols + list(range(len(fields))):
            if c not in order:
                order.append(c)
        order = [c for c in order if fields[c] > 0]            # empty fields do not order anything
        natural = [c for c in range(len(fields)) if fields[c] > 0]
        if len(fields) == 3 and order == [c for c in (0, 2, 1) if fields[c] > 0]:
            layout, geometry = "lrd", (fields[0], fields[1], fields[2])
        elif order == natural:
            layout, geometry = "ldr", (k, 0, 0)
        else:
            return None
        if geometry[1] > 16:
            return None
        return dict(k=k, fields=fields, layout=layout, geometry=geometry, strands=strands)

    def _device_keys(self, sequences, plan):
        """-> (sorted keys, is_rna, IUPAC k-mers as field tuples) or None when the input holds
        characters the device alphabet cannot carry in a way only the host chain reproduces."""
        from . import _native
        L, D, R = plan["geometry"]
        krisp_combo = plan["strands"] == 0 and plan["layout"] == "lrd" and len(plan["fields"]) == 3
        if krisp_combo:
            bases, rna, windows = fasta.ingest(sequences, plan["k"], self.omitsoft)
            special = [codec.split_window(w, L, D, R) for w in windows]
        else:
            # forward / canonical strands, other layouts: anything outside ACGTNacgtn (IUPAC
            # letters are kept by the reference, other characters pass or raise depending on
            # the strand option) goes to the host chain as a whole
            bases, rna, nspecial = fasta.load_any(sequences)
            if nspecial:
                return None
            special = []
        with _native.Engine(device=self.device) as eng:
            eng.set_params(L, D, R, omit_soft=self.omitsoft, max_bases=len(bases))
            if plan["strands"]:
                eng.set_strands(plan["strands"])
            eng.add(0, bases)
            keys = eng.keys(0).copy()
        return keys, rna, special

    def _device_blocks(self, sequences, plan):
        """-> (iterator of byte blocks of the sorted output, line count) or None"""
        got = self._device_keys(sequences, plan)
        if got is None:
            return None
        keys, rna, special = got
        if plan["layout"] == "lrd" and len(plan["fields"]) == 3:
            blocks = codec.merged_line_blocks(keys, special, *plan["geometry"], rna=rna, chunk=_WRITE_CHUNK)
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
        if self.split is not None:
            seqs = (self._split_one(s) for s in seqs)
        return seqs, rna

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
