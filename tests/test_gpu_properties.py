"""Property tests of the GPU path against the packed-key oracle: random geometries,
alphabets (soft mask, N runs), record layouts, genome counts and in/out assignments
(hypothesis, derandomised so the GPU box and this container draw the same examples)."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu


@st.composite
def workload(draw):
    L = draw(st.integers(0, 12))
    D = draw(st.integers(0, 6))
    R = draw(st.integers(0, 12))
    if L + D + R == 0:
        L = 1
    n_genomes = draw(st.integers(1, 6))
    alphabet = draw(st.sampled_from([b"ACGT", b"ACGTACGTACGTN", b"ACGTACGTacgt", b"ACGTACGTACGTacgtNn", b"AC", b"A"]))
    base_len = draw(st.integers(0, 3000))
    mu = draw(st.sampled_from([0.0, 0.002, 0.02, 0.2]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    omit = draw(st.booleans())
    n_in = draw(st.integers(0, n_genomes))
    return L, D, R, n_genomes, alphabet, base_len, mu, seed, omit, n_in


def _genomes(n, alphabet, base_len, mu, seed):
    rng = np.random.default_rng(seed)
    a = np.frombuffer(alphabet, dtype=np.uint8)
    anc = a[rng.integers(0, len(a), size=base_len)]
    out = []
    for _ in range(n):
        g = anc.copy()
        if base_len:
            m = rng.random(base_len) < mu
            g[m] = a[rng.integers(0, len(a), size=int(m.sum()))]
            for p in rng.integers(0, base_len, size=rng.integers(0, 4)):     # record separators
                g[p] = 10
            if rng.random() < 0.3:                                            # a duplicated stretch
                ln = int(rng.integers(1, max(2, base_len // 4)))
                s = int(rng.integers(0, base_len - ln + 1)); d = int(rng.integers(0, base_len - ln + 1))
                g[d:d + ln] = g[s:s + ln].copy()
        out.append(g)
    return out


@settings(max_examples=int(os.environ.get("KR_HYP_EXAMPLES", "60")), deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(workload())
def test_gpu_matches_oracle_on_random_workloads(w):
    from krisp_amd import _native as N
    from oracle import kmer_oracle as K
    L, D, R, n, alphabet, base_len, mu, seed, omit, n_in = w
    texts = _genomes(n, alphabet, base_len, mu, seed)
    flags = [i < n_in for i in range(n)]
    want_keys = [K.sorted_keys(t.tobytes(), L, D, R, omit=omit) for t in texts]
    with N.Engine() as e:
        e.set_params(L, D, R, omit_soft=omit, max_bases=max(1, max(len(t) for t in texts)))
        for i, t in enumerate(texts):
            assert e.add(i, t) == len(want_keys[i])
            assert np.array_equal(e.keys(i), want_keys[i])
        ids = list(range(n))
        for filt in (False, True):
            want = K.intersect(want_keys, flags, L, D, R, apply_filter=filt)
            assert e.intersect(ids, flags, apply_filter=filt) == len(want)
            got = e.cands()
            assert np.array_equal(got["prefix"], want["prefix"])
            assert np.array_equal(got["in_mask"], want["in_mask"])
            assert np.array_equal(got["out_mask"], want["out_mask"])
            recs = np.sort(e.collect(ids), order=["key", "genome"])
            wrec = np.sort(K.collect(want_keys, want, L, D, R), order=["key", "genome"])
            assert np.array_equal(recs, wrec)
