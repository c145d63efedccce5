"""The candidate tree of csrc/h_comm.inc written over torch.distributed send / recv with host
bounces: TEST SCAFFOLDING (tests/test_distributed.py drives it over gloo with an oracle-backed
stand-in engine).  It states the semantics the library's RCCL exchange must have -- tree-reduce
of per-rank candidate lists == n-way intersection, broadcast, record gather -- and is not
imported by anything under krisp_amd/."""
import numpy as np

from krisp_amd._native import CAND, RECORD


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
    if world == 1:
        return records
    if rank == 0:
        parts = [records]
        for src in range(1, world):
            parts.append(_recv(dist, src, device, RECORD))
        return np.concatenate(parts)
    _send(dist, records, 0, device)
    return None
