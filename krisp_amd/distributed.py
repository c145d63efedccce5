"""Multi-GPU sharding of the krisp_fasta path: one process per GPU, genomes sharded across
ranks, no collective on the sort / intersect data path, and ONE exchange step -- a binary-tree
reduction of candidate lists (the reference does the same reduction as a tree of pairwise file
merges on one host, krisp_fasta/intersectAmplicons.py:256-307).

The exchange itself lives in the library (csrc/h_comm.inc: RCCL send / recv between device
buffers, device-side list merges): `connect()` below only does the rendezvous -- rank 0 makes
the RCCL unique id and leaves it in a file the other ranks wait for -- so neither PyTorch nor
MPI is needed; `torch.distributed.run` (or any launcher that sets RANK / LOCAL_RANK /
WORLD_SIZE) merely starts the processes.

(The same tree written over torch.distributed, the executable statement of its semantics --
"tree-reduce of per-rank lists == n-way intersection" -- is test scaffolding and lives with the
tests: tests/dist_tree_reference.py, driven over gloo by tests/test_distributed.py.)
"""
import os
import secrets
import time


def shard(items, rank, world):
    """Round-robin genome -> rank assignment."""
    return [x for i, x in enumerate(items) if i % world == rank]


def env_rank_world():
    """(rank, local_rank, world) as torch.distributed.run / mpirun / srun export them.  A launcher's
    variables count only as a SET -- its size together with its rank: SLURM_NTASKS alone is set for
    every process inside an salloc / sbatch allocation, also for a plain `python bench.py` there,
    which is one rank of one.  KRISP_LAUNCHER = torchrun | mpi | slurm | none names the set to
    read (none: always (0, 0, 1))."""
    e = os.environ
    sets = {"torchrun": ("RANK", "WORLD_SIZE", "LOCAL_RANK"),
            "mpi": ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"),
            "slurm": ("SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID")}
    want = e.get("KRISP_LAUNCHER", "").strip().lower()
    if want == "none":
        return 0, 0, 1
    if want and want not in sets:
        raise ValueError(f"KRISP_LAUNCHER={want!r}: torchrun, mpi, slurm or none")
    for name in ([want] if want else ["torchrun", "mpi", "slurm"]):
        r, w, l = sets[name]
        if r in e and w in e:
            rank, world = int(e[r]), int(e[w])
            if not 0 <= rank < world:
                raise ValueError(f"{r}={rank} outside {w}={world}")
            return rank, int(e.get(l, str(rank))), world
    return 0, 0, 1


def launched():
    """True when a launcher's variable set (env_rank_world) is present: this process is one rank of a launch"""
    e = os.environ
    want = e.get("KRISP_LAUNCHER", "").strip().lower()
    if want == "none":
        return False
    sets = {"torchrun": ("RANK", "WORLD_SIZE"), "mpi": ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"),
            "slurm": ("SLURM_PROCID", "SLURM_NTASKS")}
    return any(r in e and w in e for nm, (r, w) in sets.items() if not want or nm == want)


def rendezvous_path(world):
    """Where rank 0 leaves the RCCL unique id.  KRISP_COMM_FILE names it explicitly; else a name
    every rank of ONE launch derives alike and no other launch shares: the launcher's pid (the
    parent of all ranks), the master port and the world size."""
    p = os.environ.get("KRISP_COMM_FILE")
    if p:
        return p
    tmp = os.environ.get("TMPDIR", "/tmp")
    return os.path.join(tmp, f"krisp_comm_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{world}")


def _read(path):
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError:
        return None


def _write_atomic(path, data):
    """a reader never sees a partial file; the temporary name is ours alone (O_EXCL, mode 0600)"""
    tmp = f"{path}.{os.getpid()}.{secrets.token_hex(4)}.tmp"
    try:
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    except FileNotFoundError:
        # (the directory went away under us: rank 0 of an earlier run under the same name has just tidied up)
        private_dir(os.path.dirname(path))
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def private_dir(d):
    """`d` as a directory of ours alone: created 0700, or -- when it exists -- owned by this user with no access for
    anybody else; refused otherwise (a directory somebody else made under a guessable name in /tmp could feed a run
    forged rendezvous answers or messages)."""
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"{d}: not a private directory of this user (owner {st.st_uid}, mode {st.st_mode & 0o777:o})")
    return d


def _rmdir_quiet(d):
    try:
        os.rmdir(d)
    except OSError:
        pass


def rendezvous(base, rank, world, payload=b"", timeout_s=600):
    """Rank 0's `payload` and a per-run nonce for every rank, through files under `base`.rv --
    proof against what an earlier, crashed run left there.  Rank r > 0 announces itself with a
    fresh random token (join_r); rank 0 answers each token it sees with token + nonce + payload
    (go_r) and re-answers when the token under join_r changes (it may first have read a stale
    one); rank r accepts only an answer that starts with ITS token and acknowledges the nonce
    (ack_r); rank 0 returns once every rank has acknowledged THIS nonce, then removes the files.
    Returns (nonce, payload)."""
    d = private_dir(base + ".rv")
    t0 = time.time()

    def late(what):
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: {what} (rendezvous {d}, {timeout_s} s)")
        time.sleep(0.003)

    if world == 1:
        return secrets.token_hex(8), payload
    if rank == 0:
        nonce = secrets.token_hex(8).encode()
        answered = {}
        while True:
            for r in range(1, world):
                tok = _read(os.path.join(d, f"join_{r}"))
                if tok and len(tok) == 16 and answered.get(r) != tok:
                    _write_atomic(os.path.join(d, f"go_{r}"), tok + nonce + payload)
                    answered[r] = tok
            if all(_read(os.path.join(d, f"ack_{r}")) == nonce for r in range(1, world)):
                break
            late("not every rank has joined")
        for r in range(1, world):
            for nm in (f"join_{r}", f"go_{r}", f"ack_{r}"):
                try:
                    os.unlink(os.path.join(d, nm))
                except OSError:
                    pass
        _rmdir_quiet(d)                     # (stays when another run under the same name has files in it)
        return nonce.decode(), payload
    tok = secrets.token_hex(8).encode()
    _write_atomic(os.path.join(d, f"join_{rank}"), tok)
    while True:
        g = _read(os.path.join(d, f"go_{rank}"))
        if g and g[:16] == tok:
            break
        late("no answer from rank 0")
    nonce, payload = g[16:32], g[32:]
    _write_atomic(os.path.join(d, f"ack_{rank}"), nonce)
    return nonce.decode(), payload


def connect(engine, rank, world, transport="rccl", path=None, timeout_s=600):
    """Give `engine` its communicator.  transport "rccl": one GPU per rank, rank 0's unique id
    travels with the rendezvous; "dir": the rehearsal transport (messages through files, ranks
    may share a GPU) in a directory named by the run's nonce, so no run ever reads another's
    messages."""
    from . import _native
    path = path or rendezvous_path(world)
    if transport not in ("rccl", "dir"):
        raise ValueError(f"unknown transport {transport!r}")
    if transport == "rccl" and os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        # (RCCL prints its version banner and its warnings on STDOUT, where the command line writes its CSV and bench.py its
        # one JSON line: errors only, unless the caller asked for more than the banner)
        os.environ["NCCL_DEBUG"] = "ERROR"
    if transport == "rccl" and world == 1:
        engine.comm_init(0, 1, _native.comm_unique_id())
        return
    cid = _native.comm_unique_id() if (transport == "rccl" and rank == 0) else b""
    nonce, cid = rendezvous(path, rank, world, bytes(cid), timeout_s)
    if transport == "dir":
        # (the library removes the run's message directory, and this parent once it is empty, with the communicator)
        engine.comm_init_dir(rank, world, os.path.join(private_dir(path + ".d"), nonce))
        return
    engine.comm_init(rank, world, cid)             # (collective: returns once every rank has joined)


def sharded_step(eng, ids, flags, world, apply_filter=True, collect=True):
    """One pass of the sharded hot path on THIS rank (`eng` holds the rank's genomes `ids`, uploaded):
    sort every genome, intersect them locally -- with the diagnostic filter already, which is safe: the
    predicate is monotone --, then the ONE exchange (tree reduction of the candidate lists to rank 0,
    csrc/h_comm.inc; the survivors broadcast) and the local collect of the survivors' records.
    bench.py --gpus N times exactly this function, tests/test_gpu_fullsize.py runs it at world 8.
    Returns (candidates after the reduction -- on rank 0 --, this rank's record count)."""
    for g in ids:
        eng.sort(g)
    n = eng.intersect(ids, flags, apply_filter=apply_filter)
    if world > 1:
        n = eng.cands_reduce(apply_filter=apply_filter)
        if collect:
            eng.cands_bcast()
    nrec = eng.collect(ids, fetch=False) if collect else 0          # (the records stay in HBM)
    return n, nrec
