#!/usr/bin/env python3
"""bench.py -- k-mers/s sorted + intersected (BASELINE.json metric) on MI355X.

A "step" = one pass of the hot path over one batch of synthetic genomes that are
already resident in HBM as ASCII bases: per genome pack -> both-strand keys ->
MSD radix partition -> LDS sort, then the n-way intersection + diagnostic
filter.  At N = 1 the workload is BASELINE.json configs[1]: 4 synthetic 50 Mbp
genomes (2 in / 2 out), k = 28 as 25/1/2.  At N > 1 every rank gets its own 4
genomes of one 4N-genome family (weak scaling; SURVEY.md 8e, configs[3] style):
sort + local intersect per GPU, then ONE exchange -- a binary-tree reduction of
candidate lists over RCCL -- and the filter on rank 0.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).  After the W warm-up
steps come 3 untimed calibration steps with every kernel stage bracketed by HIP events (the stage
table, the dominant kernel); the K timed steps bracket the dominant kernel's launches only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MODEL_BYTES_PER_KMER = 136.5    # 0.5 + W + W + 2*W*P + W at W = 8, P = 7 (k = 28)

# ALGORITHMIC bytes per k-mer record of each kernel stage (DESIGN.md "kernels"):
# the stage's share of SURVEY 8(d)'s model -- bases in, key write, pass read+write, intersect read.
STAGE_BYTES = {
    "pack": 0.5 + 0.1875,           # 1 B/base in, 3 bits/base out; two records per base
    "hist8": 0.1875,                # codes + bad bits in
    "scatter1": 0.1875 + 8.0,       # codes in, one 8-byte key out
    "hist2": 8.0,                   # key in (k_hist2; the k_hist16 route reads the codes instead: 0.19 B x partitions)
    "scatter2": 16.0,               # key in, key out
    "localsort": 16.0,              # key in, key out
    "intersect": 8.0,               # every key of every genome read once
}


def make_genomes(config, rank, world, per_rank, length, independent=False, masked=False):
    from krisp_amd import synth
    anc = None if independent else synth.ancestor(config, length)
    out = []
    for g in range(rank * per_rank, (rank + 1) * per_rank):
        # every rank holds half ingroup / half outgroup genomes of the 4N-genome family, so the
        # diagnostic filter can already prune locally (it is monotone: DESIGN.md "Multi-GPU")
        ing = (g % per_rank) < per_rank // 2
        codes = synth.genome_codes(config, g, length, ing, mu=0.01, snp_every=10000,
                                   independent=independent, anc=anc)
        # --masked = SURVEY 8(d) secondary variant (ii): 0.1 % of the bases N in 1 kb runs, 5 % lower case
        out.append((g, ing, synth.codes_to_text(codes, records=16, n_frac=0.001 if masked else 0.0,
                                                lower_frac=0.05 if masked else 0.0, seed=100 * config + g)))
    return out


def cpu_baseline(config, L, D, R, length, full_length):
    """The packed-key C oracle (oracle/kmer_oracle.c) on a bounded sample of the same workload:
    same generator and parameters, genomes shortened so that the run takes roughly 10-20 s on
    this host (calibrated on 4 x 1 Mbp first).  One thread per genome for the sorts -- the
    reference's own parallelism, a process per genome (krisp_fasta.py:86-123) -- then the n-way
    intersection on one thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import kmer_oracle as K
    K.build()
    cores = max(1, min(4, os.cpu_count() or 1))

    def run(fam):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as pool:       # ctypes releases the GIL
            keys = list(pool.map(lambda g: K.sorted_keys(g[2].tobytes(), L, D, R), fam))
        K.intersect(keys, [f for _, f, _ in fam], L, D, R, apply_filter=True)
        return time.perf_counter() - t0, sum(len(k) for k in keys)

    if length <= 0:
        per_mbp = run(make_genomes(config, 0, 1, 4, 1_000_000))[0] / 4.0
        length = int(min(full_length, max(1_000_000, 15.0 / (4.0 * per_mbp) * 1e6)))
        length -= length % 1_000_000
    dt, n = run(make_genomes(config, 0, 1, 4, length))
    return {"value": n / dt, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": f"4 x {length / 1e6:g} Mbp genomes of the same generator, {L}/{D}/{R}, "
                      f"{n} k-mers in {dt:.1f} s (oracle/kmer_oracle.c: generate + LSD radix sort per genome on "
                      f"{cores} threads, then n-way intersect + filter on one)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--length", type=int, default=50_000_000, help="bases per genome (C2: 50 Mbp)")
    ap.add_argument("--per-gpu", type=int, default=4, help="genomes per GPU")
    ap.add_argument("--ldr", type=int, nargs=3, default=[25, 1, 2])
    ap.add_argument("--independent", action="store_true", help="independent random genomes")
    ap.add_argument("--masked", action="store_true", help="0.1 %% N in 1 kb runs + 5 %% lower case (soft-mask mapped)")
    ap.add_argument("--cpu-length", type=int, default=0,
                    help="bases per genome of the CPU baseline sample (0 = calibrate to ~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timers", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N > 1 flow with several ranks sharing the visible GPU(s)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl) even at world size 1 (plumbing self-test)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)

    from krisp_amd import _native
    dist = None
    device = None
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "gloo":            # rehearsal: ranks share the GPU(s), lists travel over gloo
            local_rank %= max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            device = torch.device("cuda", local_rank)
    from krisp_amd.distributed import tree_reduce_candidates

    L, D, R = args.ldr
    config = 2
    genomes = make_genomes(config, rank, world, args.per_gpu, args.length, args.independent, args.masked)
    eng = _native.Engine(device=local_rank)
    eng.set_params(L, D, R, omit_soft=False, max_bases=max(len(t) for _, _, t in genomes))
    ids = []
    for g, ing, text in genomes:
        eng.upload(g, text)            # inputs resident in HBM before the timed region
        ids.append(g)
    flags = [ing for _, ing, _ in genomes]
    del genomes

    def barrier():
        eng.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    def step():
        for g in ids:
            eng.sort(g)
        if world == 1:
            return eng.intersect(ids, flags, apply_filter=True)
        eng.intersect(ids, flags, apply_filter=True)     # safe local pruning (monotone predicate)
        return tree_reduce_candidates(eng, dist, rank, world, apply_filter=True, device=device)

    for _ in range(args.warmup):
        step()
    barrier()
    # calibration (untimed, after the warm-up): every stage bracketed by HIP events -> the stage
    # table and the dominant kernel.  Event pairs around all ~60 launches of a step cost ~4 % of it,
    # so the timed region below brackets the launches of the dominant kernel only.
    calib = {}
    dom_stage = None
    if not args.no_stage_timers:
        eng.stage_enable(True)
        eng.stage_reset()
        ncal = 3
        for _ in range(ncal):
            step()
        barrier()
        calib = {s: (v[0] / ncal, v[1] // ncal) for s, v in eng.stage_times().items() if v[1]}
        dom_stage = max((s for s in calib if s in STAGE_BYTES), key=lambda s: calib[s][0])
        eng.stage_select([dom_stage])
        eng.stage_reset()
        barrier()
    t0 = time.perf_counter()
    ncand = 0
    for _ in range(args.steps):
        ncand = step()
    barrier()
    dt = time.perf_counter() - t0
    stages = eng.stage_times() if not args.no_stage_timers else {}
    kmers_local = sum(eng.count(g) for g in ids)
    copy_gbps = eng.copy_gbps(1 << 30, 10) if rank == 0 else None      # measured streaming-copy peak of this box

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        k = torch.tensor([kmers_local], dtype=torch.int64, device=device if device is not None else "cpu")
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        kmers_total = int(k.item())
    else:
        kmers_total = kmers_local

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = kmers_total * args.steps / dt
        # dominant kernel stage of this rank (HIP events on the engine's stream)
        roof = None
        if stages:
            dom = dom_stage
            ms, launches = stages[dom]          # HIP events around its launches inside the timed region
            avg_ms = ms / launches
            # k-mers through one launch: per genome (and key-space slice) for the sort stages, all local
            # genomes (per slice) for the intersect.  Sliced genomes (> 4.2e8 keys) take one more pass:
            # their pass 0 writes all keys once (0.19 + 8 B), every slice's pass 1 reads and writes them
            nslices = eng.debug_info()["nslices"]
            stage_bytes = dict(STAGE_BYTES)
            if nslices > 1:
                stage_bytes["scatter1"] = 0.1875 + 8.0 + 16.0
                stage_bytes["hist8"] = 0.1875 + 8.0
            per_launch = kmers_local * args.steps / launches
            achieved = stage_bytes[dom] * kmers_local * args.steps / (ms * 1e-3) / 1e9
            # HBM bytes per launch of the same kernel from the PMC counters (rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE passes of this command, profiles/make_traffic.py), valid for
            # the default workload only; expressed like `achieved`: bytes per launch / launch time
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile) and args.length == 50_000_000 and args.per_gpu == 4 and [L, D, R] == [25, 1, 2]:
                try:
                    tb = json.load(open(tfile)).get(dom, {}).get("bytes_per_launch")
                    traffic = round(tb / (avg_ms * 1e-3) / 1e9, 1) if tb else None
                except Exception:  # noqa: BLE001
                    traffic = None
            roof = {"bound": "hbm", "kernel": _native.STAGE_KERNELS[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "copy_peak_measured": round(copy_gbps, 1), "frac_of_copy_peak": round(achieved / copy_gbps, 4),
                    "bytes_per_kmer": stage_bytes[dom], "kmers_per_launch": per_launch, "key_space_slices": nslices,
                    "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                    "pipeline_model_GBps": round(MODEL_BYTES_PER_KMER * value / world / 1e9, 1),
                    "pipeline_model_frac": round(MODEL_BYTES_PER_KMER * value / world / 1e9 / HBM_PEAK_GBPS, 4),
                    "stage_ms_per_step_calibration": {s: round(v[0], 4) for s, v in calib.items()}}
        out = {
            "metric": "k-mers/s sorted+intersected at k=28", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {args.per_gpu} synthetic {args.length / 1e6:g} Mbp random "
                                   f"genomes per GPU (half in / half out over the {args.per_gpu * world}-genome "
                                   f"family, mu=0.01, planted SNP / 10 kb"
                                   + (", independent genomes" if args.independent else "")
                                   + (", 0.1 % N runs + 5 % lower case" if args.masked else "")
                                   + f"), {L}/{D}/{R} spacer search",
                       "kmers_per_step": kmers_total, "candidates": int(ncand),
                       "parallelism": f"genome-sharded x{world}" + ("" if world == 1 else " + tree-reduce of candidates")},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(config, L, D, R, args.cpu_length, args.length)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
