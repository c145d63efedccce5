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
    from krisp_amd import distributed as D, synth
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
    n = D.tree_reduce_candidates(eng, dist, rank, world, apply_filter=True)
    D.broadcast_candidates(eng, dist, rank, world)
    final = eng.cands()
    recs = K.collect(keys, final.astype(K.CAND), L, Dg, R) if len(final) else np.empty(0, K.RECORD)
    recs = recs.astype(RECORD)
    recs["genome"] = np.array(mine, dtype=np.uint32)[recs["genome"]] if len(recs) else recs["genome"]
    allrec = D.gather_records(recs, dist, rank, world)
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
