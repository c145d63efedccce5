#!/usr/bin/env python3
"""bench.py -- k-mers/s sorted + intersected (BASELINE.json metric) on MI355X.

A "step" = one pass of the hot path over one batch of synthetic genomes that are already
resident in HBM as ASCII bases: per genome pack -> both-strand keys -> MSD radix partition into the
fine buckets all genomes share (round 6: the order INSIDE a bucket is made where a reader needs it --
KR_OPT_LAZY_ORDER; KR_LAZY_ORDER=0 ends every sort with the LDS sort as rounds 1-5 did), then the n-way
intersection + diagnostic filter and the collection of the candidate
records -- SURVEY 8(d): "from packed bases resident in HBM to filtered candidate records
resident in HBM".  At N = 1 the default workload is BASELINE.json configs[1]: 4 synthetic 50 Mbp
genomes (2 in / 2 out), k = 28 as 25/1/2.  At N > 1 every rank gets its own 4 genomes of one
4N-genome family (weak scaling; SURVEY.md 8e, configs[3] style): sort + local intersect per GPU,
then ONE exchange -- a binary-tree reduction of candidate lists between the GPUs (RCCL, inside
the library: csrc/h_comm.inc) -- the survivors broadcast and every rank collects its records.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N                   (no launcher: bench.py starts its own N ranks, self_launch below)

A launcher only starts the processes (RANK / LOCAL_RANK / WORLD_SIZE): nothing here imports
torch.  Every N runs the SAME per-GPU workload (configs[1]'s 4 x 50 Mbp per GPU, so the N = 1 line of
a 1 / 2 / 4 / 8 sweep is the plain bench line and the sweep is one weak-scaling series); configs[3]'s
4 x 100 Mbp per GPU is --config 3.  Barrier = stream sync + a reduction over all ranks (kr_comm_barrier); the step time is the
MAX over ranks (kr_comm_allreduce).  Rank 0 prints ONE JSON line (DESIGN.md "Measurement").
After the W warm-up steps come 3 untimed calibration steps with every kernel stage bracketed by
HIP events (the stage table, the dominant kernel); the K timed steps bracket the dominant kernel's
launches only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy there)


def model_bytes_per_kmer(k):
    """B = 0.5 + W + W + 2 W P + W with W = 8-byte keys, P = ceil(2 k / 8) LSD passes"""
    return 0.5 + 8 + 8 + 2 * 8 * -(-2 * k // 8) + 8


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

# the reference itself, timed in the build container (8 vCPU Xeon 2.1 GHz,
# GNU sort 8.32): NOT on the GPU box, carried here so that the line names what the Python did
REFERENCE_PYTHON = {"value": 1.04e5, "unit": "k-mers/s", "cores": 8,
                    "what": "reference krisp_fasta (--cores 8 equivalent) on 4 x 1 Mbp of the same generator: 77.1 s "
                            "for 8.00e6 k-mers (extract+sort 12.9 s, merge 57.7 s, filter 6.5 s)",
                    "where": "SURVEY.md section 6, measured in the build container (8 vCPU Intel Xeon 2.1 GHz), "
                             "not on the GPU box: the reference cannot travel there"}


# BASELINE.json `configs` by index (--config N): the label of a bench line comes from this table, never from a
# guess about the arguments.  `gen` = SURVEY 8(d)'s config# (ancestor seed 1000 + config#: C2..C5 = configs[1..4]);
# mu / records / snp_every are 8(d)'s generator (0.01 / 16 / 10 kb) for every config.
CONFIGS = {
    1: dict(per_gpu=4, length=50_000_000, ldr=[25, 1, 2], gen=2, gpus=1,
            what="4 synthetic 50 Mbp random genomes (2 in / 2 out), k=28 spacer search, 1 x MI355X"),
    2: dict(per_gpu=8, length=500_000_000, ldr=[32, 60, 32], gen=3, gpus=1,
            what="8 synthetic 500 Mbp genomes (4 in / 4 out), 32/60/32 amplicon search, 1 x MI355X"),
    3: dict(per_gpu=4, length=100_000_000, ldr=[25, 1, 2], gen=4, gpus=8,
            what="32 synthetic 100 Mbp genomes, 4 per GPU on 8 x MI355X, k=28, candidate exchange over RCCL"),
    4: dict(per_gpu=2, length=3_000_000_000, ldr=[28, 1, 2], gen=5, gpus=1,
            what="2 synthetic 3 Gbp human-scale genomes (1 in / 1 out), k=31, 1 x MI355X, key-space slices"),
}


def baseline_config_name(cfg, custom, world, independent, masked, mu, records, snp_every):
    """the label: BASELINE configs[N] only when every argument is that config's"""
    if custom:
        return "custom (not a BASELINE.json config)"
    name = f"BASELINE configs[{cfg}]"
    gen_8d = (mu, records, snp_every) == (0.01, 16, 10000)
    if independent or masked:
        return name + " geometry, SURVEY 8(d) secondary input"
    if not gen_8d:
        return name + f" geometry, generator mu={mu:g} / {records} records / SNP per {snp_every} (8(d): 0.01 / 16 / 10000)"
    if world != CONFIGS[cfg]["gpus"]:
        return name + (f" per-GPU load, weak-scaled to {world} GPUs" if world > 1 else " per-GPU load on one GPU")
    return name


def make_genomes(config, rank, world, per_rank, length, independent=False, masked=False, mu=0.01, records=16,
                 snp_every=10000):
    from krisp_amd import synth
    anc = None if independent else synth.ancestor(config, length)
    out = []
    for g in range(rank * per_rank, (rank + 1) * per_rank):
        # every rank holds half ingroup / half outgroup genomes of the 4N-genome family, so the
        # diagnostic filter can already prune locally (it is monotone: DESIGN.md "Multi-GPU")
        ing = (g % per_rank) < per_rank // 2
        codes = synth.genome_codes(config, g, length, ing, mu=mu, snp_every=snp_every,
                                   independent=independent, anc=anc)
        # --masked = SURVEY 8(d) secondary variant (ii): 0.1 % of the bases N in 1 kb runs, 5 % lower case
        out.append((g, ing, synth.codes_to_text(codes, records=records, n_frac=0.001 if masked else 0.0,
                                                lower_frac=0.05 if masked else 0.0, seed=100 * config + g)))
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(config, L, D, R, length, full_length, per_gpu, gpu_check=None):
    """The packed-key C oracle (oracle/kmer_oracle.c) on a bounded sample of the same workload:
    same generator and parameters, genomes shortened so that the run takes roughly 10-20 s on
    this host (calibrated on 1 Mbp genomes first).  One thread per genome for the sorts -- the
    reference's own parallelism, a process per genome (krisp_fasta.py:86-123) -- then the n-way
    intersection + filter + collect on one thread; `replicas` such families run side by side so
    that all host cores work (cores = threads actually used).  Also the reference's own
    CPU-runnable case, BASELINE configs[0] (test_data/krisp_fasta, 9.8e4 k-mers), end to end."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import kmer_oracle as K
    K.build()
    # the host cores this process may use: its affinity mask, capped at the one-GPU share of a box
    # (16 cores per GPU on the benchmark pool; KRISP_BENCH_CPU_THREADS overrides)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, int(os.environ.get("KRISP_BENCH_CPU_THREADS", "16"))))
    replicas = max(1, ncpu // per_gpu)
    cores = min(ncpu, replicas * per_gpu)

    kept = {}

    def one_family(fam, keep=False):
        with ThreadPoolExecutor(max_workers=per_gpu) as pool:       # ctypes releases the GIL
            keys = list(pool.map(lambda g: K.sorted_keys(g[2].tobytes(), L, D, R), fam))
        cands = K.intersect(keys, [f for _, f, _ in fam], L, D, R, apply_filter=True)
        recs = K.collect(keys, cands, L, D, R)
        if keep:
            kept.update(keys=keys, cands=cands, recs=recs)
        return sum(len(k) for k in keys)

    def run(fam, reps, keep=False):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=reps) as pool:
            n = sum(pool.map(lambda i: one_family(fam, keep and i == 0), range(reps)))
        return time.perf_counter() - t0, n

    if length <= 0:
        per_mbp = run(make_genomes(config, 0, 1, per_gpu, 1_000_000), replicas)[0] / per_gpu
        length = int(min(full_length, max(1_000_000, 15.0 / (per_gpu * per_mbp) * 1e6)))
        length -= length % 1_000_000
    full = gpu_check is not None and length == full_length
    dt, n = run(make_genomes(config, 0, 1, per_gpu, length), replicas, keep=full)
    # the sample IS the benchmarked workload (same generator, same length): its sorted keys, candidates and records are
    # compared with what the GPU left behind after the timed steps, bit for bit (`oracle_match`)
    match = gpu_check(kept) if full else {"checked": False, "why": f"the CPU sample ({length / 1e6:g} Mbp genomes) is "
                                          f"shorter than the benchmarked workload ({full_length / 1e6:g} Mbp)"}
    kept.clear()
    out = {"oracle_match": match, "value": n / dt, "unit": "k-mers/s", "cores": cores, "cores_on_host": os.cpu_count(), "kind": "port", "cpu": cpu_model(),
           "sample": f"{replicas} side-by-side replica(s) of {per_gpu} x {length / 1e6:g} Mbp genomes of the same "
                     f"generator, {L}/{D}/{R}, {n} k-mers in {dt:.1f} s (oracle/kmer_oracle.c: generate + LSD radix "
                     f"sort, one thread per genome; n-way intersect + filter + collect on one thread per replica); {cores} of the host's "
                     f"{os.cpu_count()} cores (the one-GPU share of the box)",
           "reference_python": REFERENCE_PYTHON}
    # BASELINE configs[0]: the reference's test data through the same oracle (file text -> records)
    try:
        from krisp_amd import fasta
        d = os.path.join(ROOT, "tests", "golden", "c1")
        files = [(f"{d}/ingroup{i}.fasta.gz", True) for i in (0, 1)] + [(f"{d}/outgroup{i}.fasta.gz", False) for i in (0, 1, 2)]
        t0 = time.perf_counter()
        keys = [K.sorted_keys(fasta.to_bases(fasta.read_records(f)).tobytes(), 25, 1, 2) for f, _ in files]
        c = K.intersect(keys, [f for _, f in files], 25, 1, 2, apply_filter=True)
        K.collect(keys, c, 25, 1, 2)
        dt1 = time.perf_counter() - t0
        out["configs0_test_data"] = {"kmers": int(sum(len(k) for k in keys)), "seconds": round(dt1, 4),
                                     "candidates": int(len(c)), "what": "test_data/krisp_fasta 25/1/2 through the oracle, one thread"}
    except Exception as e:  # noqa: BLE001
        out["configs0_test_data"] = {"error": str(e)}
    return out


# The wide path (amplicons longer than one key: BASELINE configs[2]) -- ALGORITHMIC bytes per k-mer record of its
# stages over ONE kr_wide_run with L = R (one flank spectrum serves both flanks, one canonical
# composite key per window start = half a key per record).  Streamed bytes as the kernels' arguments say; a
# dictionary look-up = one 32-byte DRAM sector (the unit the memory system moves, tools/randbench.hip).
WIDE_STAGE_BYTES = {
    "pack": 2 * (0.5 + 0.1875),     # both phases pack the genome again
    "hist8": 0.1875 + 8.0 + (0.1875 + 2 * 32 / 2 + 8.0) + 2 * 8.0,   # spectrum keys from the codes (pass 0 + slices read
                                    # them), composite keys: 2 look-ups and a 16-byte cache line per window start, the
                                    # slices' histograms over the pass-0 arrays of both phases
    "scatter1": (0.1875 + 8.0 + 16.0) + (8.0 + 4.0 + 8.0),            # pass 0 + pass 1 of the spectrum; the composite phase
                                    # from the cache, half a key per record
    "scatter2": 16.0 + 8.0,
    "localsort": 16.0 + 8.0,
    "hist2": 8.0 + 4.0,             # (key-space slices: fine offsets from the pass-1 output)
    "intersect": 8.0 + 4.0,
    "locate": 8.0 + 32 / 2 + 8.0,   # cached key + group look-up per window start + the window's group written back
}


def wide_stage_bytes(key_share, member_share):
    """Round 6: the same table for a run that kept its composite keys and its member windows as LISTS (DESIGN.md 3b;
    kr_wide_fetch KR_WIDE_KEYS_LISTED / KR_WIDE_LOCATED): ONE look-up of the canonical flank per window start, 16 bytes
    per KEY (key_share of the records have one) written by k_hist8w, read by pass 0 and by the locate pass, 16 bytes per
    member window (member_share) written by the locate pass and read twice; the composite phase's sort passes move
    key_share of what the dense model says."""
    b = dict(WIDE_STAGE_BYTES)
    b["hist8"] = 0.1875 + 8.0 + (0.1875 + 32 / 2 + 16.0 * key_share) + 8.0 + 8.0 * key_share
    b["scatter1"] = (0.1875 + 8.0 + 16.0) + (16.0 + 8.0 + 16.0) * key_share
    b["scatter2"] = 16.0 + 16.0 * key_share
    b["localsort"] = 16.0 + 16.0 * key_share
    b["hist2"] = 8.0 + 8.0 * key_share
    b["intersect"] = 8.0 + 8.0 * key_share
    b["locate"] = (16.0 + 32.0) * key_share + 3 * 16.0 * member_share
    return b


def cpu_baseline_wide(gen, L, D, R, per_gpu, mu, records, snp_every):
    """the reference's own text pipeline as restated in oracle/krisp_oracle.py (kstream -> sort -> merge tree ->
    filter -> render, pure Python, one core) on a down-scaled family of the same generator and geometry"""
    import tempfile
    from oracle import krisp_oracle as O
    from krisp_amd import synth
    length = 40_000
    with tempfile.TemporaryDirectory(prefix="krisp_bench_") as td:
        fam = make_genomes(gen, 0, 1, per_gpu, length, mu=mu, records=min(records, 4), snp_every=min(snp_every, 5000))
        ing, outg = [], []
        for g, is_in, text in fam:
            f = os.path.join(td, ("in" if is_in else "out") + f"{g}.fasta")
            synth.write_fasta(f, text)
            (ing if is_in else outg).append(f)
        t0 = time.perf_counter()
        res = O.run_krisp_fasta(ing, outg, L, D, R)
        dt = time.perf_counter() - t0
        n = sum(len(v) for v in res["sorted"].values())
    return {"value": n / dt, "unit": "k-mers/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"{per_gpu} x {length / 1e3:g} kbp genomes of the same generator, {L}/{D}/{R}: {n} k-mer records in "
                      f"{dt:.1f} s through oracle/krisp_oracle.py (the reference's text pipeline restated: per-character "
                      f"Python, sorted text lines, pairwise merge tree), one core",
            "reference_python": REFERENCE_PYTHON}


def self_launch(n, timeout_s):
    """`bench.py --gpus N` without a launcher: this process -- which has not touched the GPU and never will -- starts N
    fresh copies of itself, one rank each (RANK / LOCAL_RANK / WORLD_SIZE as torch.distributed.run sets them, a private
    rendezvous file), relays rank 0's JSON line, and ends every rank when one fails or the watchdog expires.  Returns
    the exit code: 0, the first failing rank's code, or 124 for the watchdog."""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    td = tempfile.mkdtemp(prefix="krisp_bench_launch_")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, rc = [], 0
    out0 = open(os.path.join(td, "rank0.out"), "w+")
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KRISP_LAUNCHER="torchrun",
                       KRISP_COMM_FILE=os.path.join(td, "comm"))
            # (own session = own process group: a rank and whatever it started can be ended by that group's id alone)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else sys.stderr, start_new_session=True))
        t_end = time.time() + timeout_s
        live = list(procs)
        while live and not rc:
            time.sleep(0.05)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    print(f"bench.py: rank {procs.index(p)} exited with {code}: ending the other ranks", file=sys.stderr)
                    break
            if live and not rc and time.time() > t_end:
                rc = 124
                print(f"bench.py: no result within {timeout_s} s: ending all ranks", file=sys.stderr)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            pending = [p for p in procs if p.poll() is None]
            if not pending:
                break
            for p in pending:
                try:
                    os.killpg(p.pid, sig)
                except OSError:
                    pass
            t1 = time.time() + 5
            while time.time() < t1 and any(p.poll() is None for p in pending):
                time.sleep(0.05)
        out0.seek(0)
        text = out0.read()
        if rc == 0:
            sys.stdout.write(text)
            sys.stdout.flush()
        elif text:
            sys.stderr.write(text)
        return rc
    finally:
        out0.close()
        shutil.rmtree(td, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 2 for the multi-GB configs 2 and 4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; 1 for configs 2 and 4)")
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[N]: 1 = 4 x 50 Mbp 25/1/2 per GPU (default at every --gpus), "
                         "3 = 4 x 100 Mbp per GPU (quoted on 8 GPUs), 2 = 8 x 500 Mbp 32/60/32 (wide path), 4 = 2 x 3 Gbp 28/1/2")
    ap.add_argument("--place-tries", type=int, default=None,
                    help="KR_OPT_PLACE_TRIES: candidate allocations of the pass-1 output buffer, the fastest is kept "
                         "(default: the library's own default, 8 -- what the command line runs with, too)")
    ap.add_argument("--lanes", type=int, default=None,
                    help="sort lanes (KR_OPT_LANES; default: the library's, 3): consecutive genome sorts overlap on the device")
    ap.add_argument("--length", type=int, default=None, help="bases per genome (overrides the config's: a custom workload)")
    ap.add_argument("--per-gpu", type=int, default=None, help="genomes per GPU (overrides the config's)")
    ap.add_argument("--ldr", type=int, nargs=3, default=None, help="conserved-left diagnostic conserved-right")
    ap.add_argument("--mu", type=float, default=0.01, help="substitution rate of the generator (SURVEY 8(d): 0.01)")
    ap.add_argument("--records", type=int, default=16, help="FASTA records per genome (8(d): 16)")
    ap.add_argument("--snp-every", type=int, default=10000, help="planted ingroup / outgroup SNP spacing (8(d): 10 kb)")
    ap.add_argument("--independent", action="store_true", help="independent random genomes")
    ap.add_argument("--masked", action="store_true", help="0.1 %% N in 1 kb runs + 5 %% lower case (soft-mask mapped)")
    ap.add_argument("--cpu-length", type=int, default=0,
                    help="bases per genome of the CPU baseline sample (0 = calibrate to ~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timers", action="store_true")
    ap.add_argument("--no-collect", action="store_true", help="leave kr_collect out of the step (round-1 definition)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "dir"],
                    help="dir: rehearse the N > 1 flow with several ranks sharing the visible GPU (messages through files)")
    ap.add_argument("--force-comm", action="store_true",
                    help="create the RCCL communicator even at world size 1 (plumbing self-test)")
    ap.add_argument("--no-configs3", action="store_true",
                    help="at --gpus 8 without --config: leave out the second block that times BASELINE configs[3]'s load")
    ap.add_argument("--force-configs3", action="store_true", help="time the configs[3] block at any world size (tests)")
    ap.add_argument("--configs3-length", type=int, default=0, help="genome length of that block (tests; default: configs[3]'s 100 Mbp)")
    ap.add_argument("--launch-timeout", type=int, default=1500,
                    help="self-launched ranks (--gpus N without a launcher) are ended after this many seconds")
    args = ap.parse_args()

    # RCCL writes its version banner (NCCL_DEBUG=VERSION, what these boxes export) and its warnings to STDOUT, in front of
    # the ONE JSON line: errors only, unless the caller asked for more than the banner
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "ERROR"

    from krisp_amd import distributed as D
    if args.gpus > 1 and not D.launched():
        sys.exit(self_launch(args.gpus, args.launch_timeout))       # (before anything here touches the GPU)
    from krisp_amd import _native
    rank, local_rank, world = D.env_rank_world()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    if args.transport == "dir":
        local_rank = 0                          # rehearsal: the ranks share the GPU

    cfg = args.config if args.config is not None else 1     # (ONE per-GPU workload for every N: a weak-scaling series)
    C = CONFIGS[cfg]
    custom = args.length is not None or args.per_gpu is not None or args.ldr is not None
    length = args.length if args.length is not None else C["length"]
    per_gpu = args.per_gpu if args.per_gpu is not None else C["per_gpu"]
    L, Dg, R = args.ldr if args.ldr is not None else C["ldr"]
    k = L + Dg + R
    wide = k > 32 or Dg > 16
    if wide and world > 1:
        print("bench.py: the wide path (k > 32) is benchmarked on one GPU (DESIGN.md 7)", file=sys.stderr)
        sys.exit(2)
    big = length * per_gpu >= 2_000_000_000
    steps = args.steps if args.steps is not None else (2 if big else 20)
    warmup = args.warmup if args.warmup is not None else (1 if big else 3)
    args.steps, args.warmup = steps, warmup
    config = C["gen"]
    genomes = make_genomes(config, rank, world, per_gpu, length, args.independent, args.masked,
                           mu=args.mu, records=args.records, snp_every=args.snp_every)
    eng = _native.Engine(device=local_rank)
    comm = world > 1 or args.force_comm
    if comm:
        D.connect(eng, rank, world, transport=args.transport)
    # (the library chooses among 8 allocations of its pass-1 output buffer by itself -- physical placement moves
    # pass 1 / pass 2 by up to 15 % --: bench and command line run the same code; --place-tries is an A/B switch)
    if args.place_tries is not None:
        eng.set_option(_native.OPT_PLACE_TRIES, args.place_tries)
    place_tries = args.place_tries if args.place_tries is not None else int(os.environ.get("KR_PLACE_TRIES", "8"))
    lanes = args.lanes if args.lanes is not None else int(os.environ.get("KR_LANES", "3"))
    eng.set_option(_native.OPT_LANES, lanes)
    max_bases = max(len(t) for _, _, t in genomes)
    if wide:
        eng.set_params_wide(L, Dg, R, omit_soft=False, max_bases=max_bases)
    else:
        eng.set_params(L, Dg, R, omit_soft=False, max_bases=max_bases)
    ids = []
    for g, ing, text in genomes:
        eng.upload(g, text)            # inputs resident in HBM before the timed region
        ids.append(g)
    flags = [ing for _, ing, _ in genomes]
    del genomes

    def barrier():
        if comm:
            eng.comm_barrier()         # stream sync, then every rank meets
        else:
            eng.sync()

    nrec = [0]

    def step():
        if wide:
            # one kr_wide_run: flank spectra -> dictionaries -> composite keys -> sort + intersect -> locate ->
            # filter -> hits (group, genome, position, strand) resident in HBM
            nrec[0] = eng.wide_run(ids, flags, apply_filter=True)
            return int(eng.wide_fetch(_native.WIDE_NGROUPS)[0])
        n, nrec[0] = D.sharded_step(eng, ids, flags, world, apply_filter=True, collect=not args.no_collect)
        return n

    stage_bytes_all = WIDE_STAGE_BYTES if wide else STAGE_BYTES

    for _ in range(args.warmup):
        step()
    barrier()
    # calibration (untimed, after the warm-up): every stage bracketed by HIP events -> the stage
    # table and the dominant kernel.  Event pairs around all ~60 launches of a step cost ~4 % of it,
    # so the timed region below brackets the launches of the dominant kernel only.
    # The calibration runs with ONE sort lane, every kernel alone on the device: with the library's lanes the sorts
    # of consecutive genomes overlap, a kernel's launch then lasts longer (it shares the device) while the step gets
    # shorter.  The timed region runs as the library does; its bracket around the dominant kernel is the contract's
    # `roofline` (live, overlapped), the calibration's the kernel by itself (`roofline.alone`).
    calib = {}
    dom_stage = None
    ncal = 3
    gen_8d = (args.mu, args.records, args.snp_every) == (0.01, 16, 10000) and not (args.independent or args.masked)
    if not args.no_stage_timers:
        eng.set_option(_native.OPT_LANES, 1)
        eng.stage_enable(True)
        eng.stage_reset()
        for _ in range(ncal):
            step()
        barrier()
        calib = {s: (v[0] / ncal, v[1] // ncal) for s, v in eng.stage_times().items() if v[1]}
        if wide and int(eng.wide_fetch(_native.WIDE_KEYS_LISTED)[0]):
            nrecords = max(int(sum(eng.wide_fetch(_native.WIDE_COUNTS))), 1)
            stage_bytes_all = wide_stage_bytes(int(eng.wide_fetch(_native.WIDE_KEYS_LISTED)[0]) / nrecords,
                                               int(eng.wide_fetch(_native.WIDE_LOCATED)[0]) / nrecords)
        dom_stage = max((s for s in calib if s in stage_bytes_all), key=lambda s: calib[s][0])
        eng.stage_select([dom_stage])
        eng.set_option(_native.OPT_LANES, lanes)
        step()
        eng.stage_reset()
        barrier()
    # the price of one blocking call of the exchange on this box (RCCL communicators: --force-comm at N = 1, any N > 1):
    # 28 KB = a list of ~1200 candidates (what configs[1]'s filter leaves per rank)
    comm_probe = None
    if comm and args.transport == "rccl":
        try:
            comm_probe = eng.comm_probe(28 << 10, 50)
        except _native.KrispHipError as e:
            comm_probe = {"error": str(e)}
    comm0 = eng.debug_comm() if comm else None
    selfx = None
    barrier()                           # (the timed region starts with every rank here and its stream drained)
    t0 = time.perf_counter()
    ncand = 0
    for _ in range(args.steps):
        ncand = step()
    comm1 = eng.debug_comm() if comm else None      # (before the closing barrier: that is bench.py's, not the step's)
    barrier()
    dt = time.perf_counter() - t0                   # (the K steps and their closing barrier, nothing else: ADVICE r5)
    if comm and args.transport == "rccl" and not wide and world == 1:
        # --force-comm at N = 1: one round of the tree with the rank as its own partner, on the real transport -- the step's
        # candidate list + header through ncclSend / ncclRecv, merged as a received list is (kr_debug_cands_selfexchange)
        try:
            before = int(ncand)
            selfx = {"candidates_before": before, "candidates_after": int(eng.cands_selfexchange(apply_filter=True))}
            selfx["ok"] = selfx["candidates_after"] == before
        except _native.KrispHipError as e:
            selfx = {"error": str(e)}
        barrier()
    stages = eng.stage_times() if not args.no_stage_timers else {}
    kmers_local = int(sum(eng.wide_fetch(_native.WIDE_COUNTS))) if wide else sum(eng.count(g) for g in ids)
    # measured streaming-copy rate of this box: the best of every copy form x grid the library times (kr_debug_copy_gbps),
    # over 1 GiB and over 800 MB arrays (the size of a genome's key array)
    copy_gbps, copy_which = None, None
    if rank == 0:
        for nb in (1 << 30, 800_000_000):
            v = eng.copy_gbps(nb, 10)
            if copy_gbps is None or v > copy_gbps:
                copy_gbps, copy_which = v, eng.copy_which()

    if comm and world > 1:
        dt = float(eng.comm_allreduce([dt], "max")[0])
        kmers_total = int(eng.comm_allreduce([float(kmers_local)], "sum")[0])
        records_total = int(eng.comm_allreduce([float(nrec[0])], "sum")[0])
    else:
        kmers_total, records_total = kmers_local, nrec[0]

    # the driver's sweep runs `--gpus 8` without --config, i.e. configs[1]'s per-GPU load (the weak-scaling
    # series).  BASELINE quotes configs[3] -- 32 x 100 Mbp, 4 per GPU -- ON 8 GPUs: at that world size the same process times
    # that load too (the first workload's genomes freed, the context's parameters set again), so the one 8-GPU run there may
    # be yields the BASELINE-quoted config in the same line (`configs3`).  Nothing of it is inside the region `value` times.
    configs3 = None
    c3 = CONFIGS[3]
    if (args.config is None and not custom and world == c3["gpus"] and gen_8d and not args.no_configs3) or args.force_configs3:
        try:
            # (what can fail on ONE rank -- memory for the larger genomes -- happens before the first exchange of this block,
            # and every rank learns whether all got through: nobody enters the steps' collectives alone)
            local_err, ids3, flags3 = None, [], []
            try:
                for g in ids:
                    eng.free(g)
                g3 = make_genomes(c3["gen"], rank, world, c3["per_gpu"], args.configs3_length or c3["length"])
                eng.set_params(*c3["ldr"], omit_soft=False, max_bases=max(len(t) for _, _, t in g3))
                for g, ing, text in g3:
                    eng.upload(g, text)
                    ids3.append(g)
                flags3 = [ing for _, ing, _ in g3]
                del g3
            except Exception as e:  # noqa: BLE001
                local_err = e
            failed = 1.0 if local_err is not None else 0.0
            if comm and world > 1:
                failed = float(eng.comm_allreduce([failed], "max")[0])
            if failed:
                raise _native.KrispHipError(f"configs[3] block not run: {local_err or 'another rank could not set it up'}")
            st3, wu3 = 5, 2
            for _ in range(wu3):
                D.sharded_step(eng, ids3, flags3, world, apply_filter=True, collect=True)
            barrier()
            t3 = time.perf_counter()
            n3 = 0
            for _ in range(st3):
                n3, _r3 = D.sharded_step(eng, ids3, flags3, world, apply_filter=True, collect=True)
            barrier()
            dt3 = time.perf_counter() - t3
            k3 = sum(eng.count(g) for g in ids3)
            if comm and world > 1:
                dt3 = float(eng.comm_allreduce([dt3], "max")[0])
                k3 = int(eng.comm_allreduce([float(k3)], "sum")[0])
            configs3 = {"workload": f"BASELINE configs[3]: {c3['what']}" + ("" if world == c3["gpus"] else
                                    f" -- its per-GPU load on {world} GPU(s)")
                                    + ("" if not args.configs3_length else f", genomes of {args.configs3_length / 1e6:g} Mbp (--configs3-length)"),
                        "value": k3 * st3 / dt3, "unit": "k-mers/s", "n_gpus": world, "steps": st3, "warmup": wu3,
                        "ms_per_step": dt3 / st3 * 1e3, "kmers_per_step": k3, "candidates": int(n3),
                        "note": "timed after the main workload of this line, in the same process and context; not part of `value`"}
        except _native.KrispHipError as e:
            configs3 = {"error": str(e)}

    def gpu_check(want):
        """the oracle's sorted keys / candidates / records of the full workload against what the last timed step left in
        HBM (bench.py's checker leg: the oracle is never on the timed path)"""
        import numpy as np
        res = {"checked": True, "what": "sorted keys of every genome, candidates (prefix, ingroup / outgroup masks) and records "
                                        "of the last timed step == oracle/kmer_oracle.c on the same genomes, bit for bit"}
        try:
            ok_keys = all(np.array_equal(eng.keys(g), want["keys"][i]) for i, g in enumerate(ids))
            got = eng.cands()
            ok_c = len(got) == len(want["cands"]) and all(np.array_equal(got[f], want["cands"][f]) for f in ("prefix", "in_mask", "out_mask"))
            recs = np.sort(eng.fetch_records(nrec[0]), order=["key", "genome"])
            wrec = np.sort(want["recs"], order=["key", "genome"])
            ok_r = np.array_equal(recs, wrec)
            res.update(sorted_keys=bool(ok_keys), candidates=bool(ok_c), records=bool(ok_r), match=bool(ok_keys and ok_c and ok_r),
                       n_candidates=int(len(got)), n_records=int(len(recs)))
        except Exception as e:  # noqa: BLE001
            res.update(match=False, error=str(e))
        return res

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = kmers_total * args.steps / dt
        model_b = 696.5 if wide else model_bytes_per_kmer(k)      # (128-bit key + position payload for 32/60/32)
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
            stage_bytes = dict(stage_bytes_all)
            if nslices > 1 and not wide:
                stage_bytes["scatter1"] = 0.1875 + 8.0 + 16.0
                stage_bytes["hist8"] = 0.1875 + 8.0
            per_launch = kmers_local * args.steps / launches
            achieved = stage_bytes[dom] * kmers_local * args.steps / (ms * 1e-3) / 1e9
            alone_ms = calib[dom][0] / max(calib[dom][1], 1)        # per launch, the kernel by itself (one lane)
            alone = stage_bytes[dom] * per_launch / (alone_ms * 1e-3) / 1e9
            # HBM bytes from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command,
            # profiles/make_traffic.py): per launch of the dominant kernel and per step.  The file names the build it was
            # measured with (hash of the HIP sources): with another build nothing is printed from it.
            traffic, step_roof, traffic_meta = None, None, None
            # (one PMC file per BASELINE config: traffic.json = configs[1], traffic_c2.json / traffic_c4.json from
            # tools/profile_config.sh -- their launches differ in size by phase and slice: bytes per STEP there)
            tfile = os.path.join(ROOT, "profiles", "traffic.json" if cfg == 1 else f"traffic_c{cfg}.json")
            if os.path.exists(tfile) and not custom and world == 1 and gen_8d:
                try:
                    from krisp_amd import build as kb
                    tj = json.load(open(tfile))
                    meta = tj.get("_meta", {})
                    if meta.get("source_sha16") != kb.source_sha16():
                        traffic_meta = {"stale": f"{os.path.basename(tfile)} was measured with another build of the library "
                                                 f"({meta.get('source_sha16')}, round {meta.get('round')}); this one is "
                                                 f"{kb.source_sha16()}: no traffic figures"}
                    else:
                        traffic_meta = meta
                        if cfg != 1:
                            traffic = None          # (a stage is several kernels there, a kernel serves several stages)
                        else:
                            tb = tj.get(dom, {}).get("bytes_per_launch")
                            traffic = round(tb / (alone_ms * 1e-3) / 1e9, 1) if tb else None
                        sb = tj.get("_step", {}).get("bytes_per_step")
                        if sb:
                            sg = sb / (ms_per_step * 1e-3) / 1e9
                            step_roof = {"what": "HBM bytes of ALL kernels of one step (PMC: FETCH_SIZE x 2 + WRITE_SIZE, "
                                                 "profiles/make_traffic.py) / ms_per_step of this run",
                                         "bytes_per_step": sb, "ms_per_step": round(ms_per_step, 4), "achieved": round(sg, 1),
                                         "frac": round(sg / HBM_PEAK_GBPS, 4), "frac_of_copy_peak": round(sg / copy_gbps, 4)}
                except Exception as e:  # noqa: BLE001
                    traffic_meta = {"error": str(e)}
            # Schema 6: the TOP LEVEL is the whole step -- the figure tied to the driver-timed `value`:
            # HBM bytes of one step / ms_per_step of this run.  Bytes: measured (PMC, all kernels of a step) when
            # profiles/traffic*.json was taken with this very build, else the design's algorithmic bytes per k-mer
            # (`basis` says which).  `kernel` = the dominant kernel BY ITSELF (one sort lane, calibration
            # steps of this run: what `rocprofv3 --stats` of `bench.py --lanes 1` shows), `live` = the same kernel inside
            # the timed region, where its launches share the device with other genomes' kernels.
            lazy = eng.debug_lazy()
            alg_bpk = None
            if not wide:
                sb = stage_bytes
                # (KR_OPT_LAZY_ORDER: the LDS sort reads and writes the anchor genome only -- one of per_gpu --, and no genome
                # at all where the intersection makes the anchor's slots itself: k_intersect3t<., UA>)
                ls_share = 0.0 if lazy["anchor_in_bucket_order"] else ((1.0 / per_gpu) if lazy["on"] and lazy["skipped"] else 1.0)
                alg_bpk = (sb["pack"] + sb["hist8"] + sb["scatter1"] + (2 * 0.1875 if nslices == 1 else sb["hist2"]) + sb["scatter2"]
                           + sb["localsort"] * ls_share + sb["intersect"])
            if step_roof is not None:
                top_bytes, basis = step_roof["bytes_per_step"], "measured: PMC bytes of all kernels of one step (profiles/, this build)"
            elif alg_bpk is not None:
                top_bytes, basis = alg_bpk * kmers_local, "algorithmic bytes per k-mer of the design (no PMC file of this build)"
            else:
                top_bytes, basis = stage_bytes[dom] * per_launch * calib[dom][1], "algorithmic bytes of the dominant stage only"
            top_gbps = top_bytes / (ms_per_step * 1e-3) / 1e9
            roof = {"bound": "hbm", "scope": "step", "achieved": round(top_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(top_gbps / HBM_PEAK_GBPS, 4),
                    "traffic": step_roof["bytes_per_step"] if step_roof is not None else None,
                    "basis": basis, "bytes_per_step": round(top_bytes), "bytes_per_kmer": round(top_bytes / max(kmers_local, 1), 2),
                    "algorithmic_bytes_per_kmer": None if alg_bpk is None else round(alg_bpk, 2),
                    "how": "HBM bytes of one step / ms_per_step of the timed region (the figure `value` follows from)",
                    "copy_peak_measured": round(copy_gbps, 1), "frac_of_copy_peak": round(top_gbps / copy_gbps, 4),
                    "copy_peak_which": copy_which, "copy_peak_guide": 6290.0, "key_space_slices": nslices,
                    "lazy_order": lazy,
                    "kernel": {"name": _native.STAGE_KERNELS[dom], "achieved": round(alone, 1), "frac": round(alone / HBM_PEAK_GBPS, 4),
                               "traffic": traffic, "avg_launch_ms": round(alone_ms, 4), "launches": calib[dom][1] * ncal,
                               "bytes_per_kmer": stage_bytes[dom], "kmers_per_launch": per_launch,
                               "frac_of_copy_peak": round(alone / copy_gbps, 4),
                               "how": "algorithmic bytes per launch / average launch time (HIP events on the library's stream "
                                      "around each launch) of the dominant kernel by itself: calibration steps of this run with "
                                      "one sort lane"},
                    "live": {"what": f"the same kernel inside the timed region ({lanes} sort lane(s)"
                                     + ("" if lanes <= 1 or wide else ": its launches run beside the kernels of other genomes' "
                                        "sorts, so a launch lasts longer while the step gets shorter") + ")",
                             "avg_launch_ms": round(avg_ms, 4), "launches": launches, "achieved": round(achieved, 1),
                             "frac": round(achieved / HBM_PEAK_GBPS, 4), "sort_lanes": lanes},
                    "step": step_roof, "traffic_source": traffic_meta,
                    "pipeline_model_bytes_per_kmer": model_b,
                    "pipeline_model_GBps": round(model_b * value / world / 1e9, 1),
                    "pipeline_model_frac": round(model_b * value / world / 1e9 / HBM_PEAK_GBPS, 4),
                    "stage_ms_per_step_calibration": {s: round(v[0], 4) for s, v in calib.items()}}
        name = baseline_config_name(cfg, custom, world, args.independent, args.masked, args.mu, args.records, args.snp_every)
        out = {
            # schema 6 (round 6): `roofline` top level = the whole STEP (bytes of one step / ms_per_step: the figure tied to the
            # driver-timed value), `roofline.kernel` = the dominant kernel by itself, `roofline.live` = the same kernel inside
            # the timed region, `roofline.step` = the PMC details; `configs3` = BASELINE configs[3]'s load in the same run at 8 GPUs
            "schema": 6,
            "metric": f"k-mers/s sorted+intersected at k={k}", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "rccl_ranks": eng.comm_rccl_ranks(),     # (ncclCommCount: 0 = no RCCL communicator in this run)
            # the ONE exchange of a step on rank 0 (kr_debug_comm): blocking calls and host time inside kr_cands_reduce +
            # kr_cands_bcast (the time includes waiting for the slowest rank's sorts: the tree is the step's first meeting)
            "exchange_call_cost": comm_probe, "exchange_self_round": selfx,
            "exchange": None if not comm or world == 1 else {
                "ms_per_step": round((comm1["exchange_us"] - comm0["exchange_us"]) / args.steps / 1e3, 4),
                "host_syncs_per_step": (comm1["syncs"] - comm0["syncs"]) / args.steps,
                "p2p_calls_per_step": (comm1["p2p"] - comm0["p2p"]) / args.steps,
                "collectives_per_step": (comm1["collectives"] - comm0["collectives"]) / args.steps,
                "message_entries": comm1["message_entries"], "transport": args.transport},
            "config": {"workload": f"{name}: {per_gpu} synthetic {length / 1e6:g} Mbp random "
                                   f"genomes per GPU (half in / half out over the {per_gpu * world}-genome "
                                   f"family, mu={args.mu:g}, {args.records} records, planted SNP / {args.snp_every}"
                                   + (", independent genomes" if args.independent else "")
                                   + (", 0.1 % N runs + 5 % lower case" if args.masked else "")
                                   + f"), {L}/{Dg}/{R} (k={k})",
                       "baseline_config": name, "baseline_config_index": None if custom else cfg,
                       "baseline_text": None if custom else C["what"],
                       "step": ("one kr_wide_run: flank spectrum + dictionaries + composite keys, sort + n-way intersect, "
                                "locate + filter + hits (resident in HBM)") if wide else
                               "sort every genome + n-way intersect + filter"
                               + ("" if args.no_collect else " + collect the candidate records (resident in HBM)"),
                       "kmers_per_step": kmers_total, "candidates": int(ncand), "records": int(records_total),
                       "place_tries": place_tries, "sort_lanes": lanes,
                       # (the wide path: member windows the locate pass listed -- 16 bytes each -- or 0: a group number per window start)
                       "wide_members_listed": int(eng.wide_fetch(_native.WIDE_LOCATED)[0]) if wide else None,
                       "wide_keys_listed": int(eng.wide_fetch(_native.WIDE_KEYS_LISTED)[0]) if wide else None,
                       # which placement class the run got: probe times of the candidate pass-1 buffers, 800 MB
                       # each at configs[1] -- ~0.19 ms fast, ~0.22 / ~0.25 ms the slower classes; the fastest go to the lanes
                       "placement": eng.debug_place(),
                       "parallelism": f"genome-sharded x{world}" + ("" if world == 1 else
                                      f" + tree-reduce of candidates ({args.transport})")},
            "roofline": roof,
            "configs3": configs3,
        }
        if world == 1 and not args.no_cpu_baseline and wide:
            out["cpu_baseline"] = cpu_baseline_wide(config, L, Dg, R, per_gpu, args.mu, args.records, args.snp_every)
        elif world == 1 and not args.no_cpu_baseline:
            check = gpu_check if gen_8d and not args.no_collect and configs3 is None else None      # (the CPU sample is 8(d)'s generator; a configs3 block has replaced the genomes)
            out["cpu_baseline"] = cpu_baseline(config, L, Dg, R, args.cpu_length, length, per_gpu, gpu_check=check)
            out["oracle_match"] = out["cpu_baseline"].pop("oracle_match", None)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if comm:
        eng.comm_barrier()
    eng.close()


if __name__ == "__main__":
    main()
