"""Timing of the wide path (kr_wide_run) on a synthetic family: python tools/wide_check.py [Mbp] [L D R]"""
import sys
import time


sys.path.insert(0, ".")
from krisp_amd import _native, synth  # noqa: E402


def main():
    mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50
    L, D, R = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (30, 40, 30)
    fam = synth.family(2, 2, 2, int(mbp * 1e6), records=16, mu=0.002, snp_every=5000)
    ids = list(range(len(fam)))
    with _native.Engine() as eng:
        eng.set_params_wide(L, D, R, max_bases=max(len(t) for _, _, t in fam))
        for i, (_, _, t) in enumerate(fam):
            eng.upload(i, t)
        eng.stage_enable(True)
        for rep in range(3):
            eng.stage_reset()
            eng.sync()
            t0 = time.time()
            n = eng.wide_run(ids, [f for _, f, _ in fam], apply_filter=True)
            eng.sync()
            dt = time.time() - t0
            info = [eng.wide_count(w) for w in (0, 1, 2)]
            print(f"run {rep}: {dt * 1e3:.1f} ms, hits {n}, dictL {info[0]}, dictR {info[1]}, groups {info[2]}, "
                  f"{2 * sum(len(t) for _, _, t in fam) / dt / 1e9:.2f} G windows/s", flush=True)
            print("   ", {k: (round(v[0], 1), v[1]) for k, v in eng.stage_times().items() if v[1]}, flush=True)
        print(eng.debug_info())


if __name__ == "__main__":
    main()
