"""BASELINE configs[2]: 8 synthetic 500 Mbp genomes (4 in / 4 out), 32/60/32 amplicon search on one
MI355X -- the wide path with key-space slices.  python tools/c3_check.py [genomes] [Mbp] [slots 0|1]
Prints the run time and checks the result through properties (every group holds every genome,
groups ascend, flanks agree, a diagnostic column separates the groups)."""
import sys
import time


import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krisp_amd import _native, amplicon, synth  # noqa: E402
from krisp_amd import krisp_fasta as KF  # noqa: E402


def main():
    ng = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 500
    slots = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    L, D, R = 32, 60, 32
    t0 = time.time()
    fam = synth.family(3, ng // 2, ng - ng // 2, int(mbp * 1e6), records=24, mu=0.001, snp_every=20000)
    print(f"generated {ng} x {mbp} Mbp in {time.time() - t0:.1f} s", flush=True)
    ids = list(range(len(fam)))
    flags = [f for _, f, _ in fam]
    with _native.Engine() as eng:
        eng.set_option(_native.OPT_WIDE_SLOTS, slots)
        eng.set_params_wide(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        t0 = time.time()
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
        print(f"upload {time.time() - t0:.2f} s, slices {eng.debug_info()['nslices']}", flush=True)
        eng.stage_enable(True)
        for rep in range(2):
            eng.stage_reset()
            t0 = time.time()
            n = eng.wide_run(ids, flags, apply_filter=True)
            eng.sync()
            dt = time.time() - t0
            sizes = [eng.wide_count(w) for w in (0, 1, 2)]
            windows = 2 * sum(len(t) for _, _, t in fam)
            print(f"run {rep}: {dt:.3f} s, {windows / dt / 1e9:.2f} G windows/s, hits {n}, dictL {sizes[0]}, "
                  f"dictR {sizes[1]}, groups {sizes[2]}", flush=True)
            print("   ", {k: (round(v[0], 1), v[1]) for k, v in eng.stage_times().items() if v[1]}, flush=True)
            print("    (left,right) groups present in all genomes:", int(eng.wide_fetch(_native.WIDE_NGROUPS)[0]), flush=True)
            print("    slot bits (left 0..2, right 0..2, groups):", [int(x) for x in eng.wide_fetch(_native.WIDE_SLOT_BITS)], flush=True)
        hits = eng.wide_fetch(_native.WIDE_HITS)
    if os.environ.get("C3_NOCHECK") == "1":       # (timing experiments with variant libraries)
        return
    groups = KF._groups_from_hits(hits, [t for _, _, t in fam], [nm for nm, _, _ in fam], L, D, R)
    names = {nm for nm, _, _ in fam}
    ingroup = {nm for nm, f, _ in fam if f}
    pairs = []
    for g in groups:
        assert {lab for a in g for lab in a.labels} == names
        assert len({(a.left, a.right) for a in g}) == 1
        assert amplicon.ingroup_unique_columns(g, ingroup)
        pairs.append((g[0].left, g[0].right))
    assert pairs == sorted(pairs) and len(set(pairs)) == len(pairs)
    print(f"C3_OK groups {len(groups)} member lines {len(amplicon.merged_lines(groups))}", flush=True)


if __name__ == "__main__":
    main()
