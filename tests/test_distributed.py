"""The N > 1 path on CPU: two processes over gloo drive krisp_amd.distributed
(sharding, tree reduction of candidate lists, broadcast, record gather) with a
CPU stand-in engine built on the packed-key oracle; the result must equal the
single-process n-way intersection."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["KR_ROOT"])
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.environ["KR_ROOT"], "tests"))
    from krisp_amd import distributed as D, synth
    import dist_tree_reference as T
    from krisp_amd._native import CAND, RECORD
    from oracle import kmer_oracle as K

    L, Dg, R = 8, 1, 4

    class OracleEngine:
        """cands()/merge_cands()/load_cands() of _native.Engine, computed by the oracle"""
        def __init__(self, keys, flags):
            self.keys, self.flags = keys, flags
            self.c = K.intersect(keys, flags, L, Dg, R, apply_filter=False) if keys else np.empty(0, CAND)
        def cands(self):
            return self.c.astype(CAND)
        def load_cands(self, c):
            self.c = np.asarray(c).astype(K.CAND)
        def merge_cands(self, other=None, apply_filter=False):
            c = self.c
            if other is not None:
                other = np.asarray(other)
                idx = {int(p): i for i, p in enumerate(other["prefix"])}
                keep = []
                for row in c:
                    j = idx.get(int(row["prefix"]))
                    if j is not None:
                        keep.append((row["prefix"], row["in_mask"] | other["in_mask"][j],
                                     row["out_mask"] | other["out_mask"][j]))
                c = np.array(keep, dtype=K.CAND) if keep else np.empty(0, K.CAND)
            if apply_filter:
                ok = [any((((int(r["in_mask"]) & int(r["out_mask"])) >> (4 * col)) & 15) == 0 for col in range(Dg))
                      for r in c]
                c = c[np.array(ok, dtype=bool)] if len(c) else c
            self.c = c
            return len(c)

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ng = max(6, world)
    fam = synth.family(5, ng // 2, ng - ng // 2, 6000, records=2, mu=0.004, snp_every=500)
    mine = D.shard(list(range(len(fam))), rank, world)
    keys = [K.sorted_keys(fam[g][2].tobytes(), L, Dg, R) for g in mine]
    eng = OracleEngine(keys, [fam[g][1] for g in mine])
    n = T.tree_reduce_candidates(eng, dist, rank, world, apply_filter=True)
    T.broadcast_candidates(eng, dist, rank, world)
    final = eng.cands()
    recs = K.collect(keys, final.astype(K.CAND), L, Dg, R) if len(final) else np.empty(0, K.RECORD)
    recs = recs.astype(RECORD)
    recs["genome"] = np.array(mine, dtype=np.uint32)[recs["genome"]] if len(recs) else recs["genome"]
    allrec = T.gather_records(recs, dist, rank, world)
    if rank == 0:
        allkeys = [K.sorted_keys(t.tobytes(), L, Dg, R) for _, _, t in fam]
        want = K.intersect(allkeys, [f for _, f, _ in fam], L, Dg, R, apply_filter=True)
        assert n == len(want) and len(want) > 0, (n, len(want))
        assert np.array_equal(final["prefix"], want["prefix"])
        assert np.array_equal(final["in_mask"], want["in_mask"])
        assert np.array_equal(final["out_mask"], want["out_mask"])
        wrec = K.collect(allkeys, want, L, Dg, R)
        a = np.sort(allrec.astype(K.RECORD), order=["key", "genome"])
        b = np.sort(wrec, order=["key", "genome"])
        assert np.array_equal(a, b)
        print("DIST_OK", n, len(allrec))
    dist.barrier()
    dist.destroy_process_group()
''')


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tree_reduce_over_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KR_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0], outs[0]


def test_shard_is_round_robin():
    from krisp_amd.distributed import shard
    assert shard(list(range(10)), 1, 4) == [1, 5, 9]
    assert sum((shard(list(range(10)), r, 4) for r in range(4)), []).__len__() == 10


def test_launcher_variables_count_only_as_a_set(monkeypatch):
    """distributed.env_rank_world: SLURM_NTASKS alone (every process inside an allocation has it) is not a
    launch; a launcher's size counts together with its rank; KRISP_LAUNCHER pins the set"""
    from krisp_amd import distributed as D
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE",
              "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID", "KRISP_LAUNCHER"):
        monkeypatch.delenv(v, raising=False)
    assert D.env_rank_world() == (0, 0, 1)
    monkeypatch.setenv("SLURM_NTASKS", "8")
    assert D.env_rank_world() == (0, 0, 1)
    monkeypatch.setenv("SLURM_PROCID", "3")
    monkeypatch.setenv("SLURM_LOCALID", "1")
    assert D.env_rank_world() == (3, 1, 8)
    monkeypatch.setenv("OMPI_COMM_WORLD_RANK", "2")
    monkeypatch.setenv("OMPI_COMM_WORLD_SIZE", "4")
    assert D.env_rank_world() == (2, 2, 4)                 # (mpirun inside the allocation: its own set wins)
    monkeypatch.setenv("KRISP_LAUNCHER", "slurm")
    assert D.env_rank_world() == (3, 1, 8)
    monkeypatch.setenv("KRISP_LAUNCHER", "none")
    assert D.env_rank_world() == (0, 0, 1)
    monkeypatch.setenv("KRISP_LAUNCHER", "torchrun")
    assert D.env_rank_world() == (0, 0, 1)
    monkeypatch.setenv("RANK", "5")
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(ValueError):
        D.env_rank_world()


def test_rendezvous_ignores_what_a_crashed_run_left(tmp_path):
    """distributed.rendezvous: stale join / go / ack files of an earlier run under the same name change
    nothing -- every rank gets rank 0's payload and ONE fresh nonce"""
    import threading
    from krisp_amd import distributed as D
    base = str(tmp_path / "rv")
    os.makedirs(base + ".rv", mode=0o700)              # (as the crashed run made it: private to this user)
    for r in (1, 2):
        open(os.path.join(base + ".rv", f"join_{r}"), "wb").write(b"0123456789abcdef")
        open(os.path.join(base + ".rv", f"go_{r}"), "wb").write(b"0123456789abcdef" + b"f" * 16 + b"stale payload")
        open(os.path.join(base + ".rv", f"ack_{r}"), "wb").write(b"f" * 16)
    for trial in range(2):                                # (and a second run under the same name)
        out = [None] * 3

        def work(rank):
            out[rank] = D.rendezvous(base, rank, 3, b"payload of rank 0" if rank == 0 else b"", timeout_s=20)

        ts = [threading.Thread(target=work, args=(r,)) for r in (2, 1, 0)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(30)
        assert all(o is not None for o in out)
        assert {o[1] for o in out} == {b"payload of rank 0"}
        assert len({o[0] for o in out}) == 1 and out[0][0] != "f" * 16
    assert not os.path.exists(base + ".rv")             # (rank 0 takes the files and the directory away)
