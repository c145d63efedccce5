"""Multi-GPU sharding of the krisp_fasta path: one process per GPU, genomes sharded across
ranks, no collective on the sort / intersect data path, and ONE exchange step -- a binary-tree
reduction of candidate lists (the reference does the same reduction as a tree of pairwise file
merges on one host, krisp_fasta/intersectAmplicons.py:256-307).

The exchange itself lives in the library (csrc/h_comm.inc: RCCL send / recv between device
buffers, device-side list merges): `connect()` below only does the rendezvous -- rank 0 makes
the RCCL unique id and leaves it in a file the other ranks wait for -- so neither PyTorch nor
MPI is needed; `torch.distributed.run` (or any launcher that sets RANK / LOCAL_RANK /
WORLD_SIZE) merely starts the processes.

The functions that take a `dist` argument are the same tree written over torch.distributed
send / recv with host bounces.  They are not used by the product path any more; the CPU tests
(tests/test_distributed.py, gloo, a CPU stand-in engine) keep them as the executable statement
of the tree's semantics: "tree-reduce of per-rank lists == n-way intersection".
"""
import os
import time

import numpy as np

from ._native import CAND


def shard(items, rank, world):
    """Round-robin genome -> rank assignment."""
    return [x for i, x in enumerate(items) if i % world == rank]


def env_rank_world():
    """(rank, local_rank, world) as torch.distributed.run / mpirun / srun export them"""
    e = os.environ
    rank = int(e.get("RANK", e.get("OMPI_COMM_WORLD_RANK", e.get("SLURM_PROCID", "0"))))
    world = int(e.get("WORLD_SIZE", e.get("OMPI_COMM_WORLD_SIZE", e.get("SLURM_NTASKS", "1"))))
    local = int(e.get("LOCAL_RANK", e.get("OMPI_COMM_WORLD_LOCAL_RANK", e.get("SLURM_LOCALID", str(rank)))))
    return rank, local, world


def rendezvous_path(world):
    """Where rank 0 leaves the RCCL unique id.  KRISP_COMM_FILE names it explicitly; else a name
    every rank of ONE launch derives alike and no other launch shares: the launcher's pid (the
    parent of all ranks), the master port and the world size."""
    p = os.environ.get("KRISP_COMM_FILE")
    if p:
        return p
    tmp = os.environ.get("TMPDIR", "/tmp")
    return os.path.join(tmp, f"krisp_comm_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{world}")


def connect(engine, rank, world, transport="rccl", path=None, timeout_s=600):
    """Give `engine` its communicator.  transport "rccl": one GPU per rank, the unique id goes
    through the file `path`; "dir": the rehearsal transport (messages through files in the
    directory `path`; ranks may share a GPU)."""
    from . import _native
    path = path or rendezvous_path(world)
    if transport == "dir":
        engine.comm_init_dir(rank, world, path + ".d")
        return
    if transport != "rccl":
        raise ValueError(f"unknown transport {transport!r}")
    if world == 1:
        engine.comm_init(0, 1, _native.comm_unique_id())
        return
    if rank == 0:
        cid = _native.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(cid)
        os.replace(path + ".tmp", path)            # (atomic: a reader never sees a partial id)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: no RCCL id at {path} after {timeout_s} s")
            time.sleep(0.005)
        with open(path, "rb") as f:
            cid = f.read()
    engine.comm_init(rank, world, cid)             # (collective: returns once every rank has joined)
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass


# ----------------------------------------------------------------------------
# the same tree over torch.distributed (CPU tests; see the module docstring)
# ----------------------------------------------------------------------------
def _to_tensor(arr, device):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).reshape(-1).copy())
    return t.to(device) if device is not None else t


def _send(dist, arr, dst, device):
    import torch
    n = torch.tensor([len(arr)], dtype=torch.int64)
    n = n.to(device) if device is not None else n
    dist.send(n, dst)
    if len(arr):
        dist.send(_to_tensor(arr, device), dst)


def _recv(dist, src, device, dtype):
    import torch
    n = torch.zeros(1, dtype=torch.int64)
    n = n.to(device) if device is not None else n
    dist.recv(n, src)
    cnt = int(n.item())
    words = dtype.itemsize // 8
    if cnt == 0:
        return np.empty(0, dtype=dtype)
    buf = torch.empty(cnt * words, dtype=torch.int64)
    buf = buf.to(device) if device is not None else buf
    dist.recv(buf, src)
    return buf.cpu().numpy().view(dtype)


def tree_reduce_candidates(engine, dist, rank, world, apply_filter, device=None):
    """Every rank holds the candidates of its own genomes in `engine` (unfiltered, or
    already pruned with the diagnostic filter: the predicate "some column has disjoint
    ingroup / outgroup base sets" is monotone -- masks only grow under merging -- so a
    candidate that fails it on partial masks fails it globally and may be dropped at
    any stage).  After the call rank 0 holds the candidates present on every rank, masks
    OR-ed (and filtered when apply_filter); other ranks' candidate sets are spent.
    Returns the final count on rank 0, -1 elsewhere.  log2(world) rounds."""
    step = 1
    active = True
    while step < world:
        if active:
            if rank % (2 * step) == 0:
                partner = rank + step
                if partner < world:
                    other = _recv(dist, partner, device, CAND)
                    engine.merge_cands(other, apply_filter=apply_filter)
            else:
                _send(dist, engine.cands(), rank - step, device)
                active = False
        step *= 2
    if rank == 0:
        if apply_filter:
            return engine.merge_cands(None, apply_filter=True)
        return len(engine.cands())
    return -1


def broadcast_candidates(engine, dist, rank, world, device=None):
    """Rank 0's final candidates -> every rank's engine (for the local collect)."""
    import torch
    if world == 1:
        return
    cands = engine.cands() if rank == 0 else np.empty(0, dtype=CAND)
    n = torch.tensor([len(cands)], dtype=torch.int64)
    n = n.to(device) if device is not None else n
    dist.broadcast(n, 0)
    cnt = int(n.item())
    buf = _to_tensor(cands, device) if rank == 0 else torch.empty(cnt * 3, dtype=torch.int64)
    if rank != 0 and device is not None:
        buf = buf.to(device)
    if cnt:
        dist.broadcast(buf, 0)
    if rank != 0:
        engine.load_cands(buf.cpu().numpy().view(CAND) if cnt else np.empty(0, dtype=CAND))


def gather_records(records, dist, rank, world, device=None):
    """Per-rank (key, genome, count) records -> concatenation on rank 0."""
    from ._native import RECORD
    if world == 1:
        return records
    if rank == 0:
        parts = [records]
        for src in range(1, world):
            parts.append(_recv(dist, src, device, RECORD))
        return np.concatenate(parts)
    _send(dist, records, 0, device)
    return None
