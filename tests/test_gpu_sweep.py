"""Randomised sweep: many small jobs of odd shapes (very short and ragged sequences, caterpillar and random guide
trees, protein / nucleotide scoring, fast and default mode) against the oracle -- the shapes golden files do not cover.
Everything that is an integer must match exactly; so must the scores."""
import os

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu


def _tree(n, rng, shape):
    nodes = list(range(n))
    tasks, nxt = [], n
    if shape == "caterpillar":                       # longest possible dependency chain, seq-profile merges
        cur = nodes[0]
        for x in nodes[1:]:
            tasks.append((cur, x, nxt) if rng.rand() < 0.5 else (x, cur, nxt))
            cur = nxt
            nxt += 1
    else:
        while len(nodes) > 1:
            i, j = rng.choice(len(nodes), 2, replace=False)
            tasks.append((nodes[i], nodes[j], nxt))
            nodes = [x for k, x in enumerate(nodes) if k not in (i, j)] + [nxt]
            nxt += 1
    return np.array(tasks, np.int32)


@pytest.mark.parametrize("seed", range(24))
def test_random_job_matches_oracle(oracle, seed):
    import kalign_amd
    rng = np.random.RandomState(1000 + seed)
    dna = bool(seed % 3 == 0)
    n = int(rng.choice([2, 3, 4, 7, 12, 20, 33]))
    base_len = int(rng.choice([1, 2, 5, 17, 64, 65, 130, 260]))
    alpha = 4 if dna else 20
    base = rng.randint(0, alpha, base_len).astype(np.uint8)
    codes = []
    for _ in range(n):
        s = base.copy()
        idx = rng.rand(len(s)) < rng.uniform(0.0, 0.5)
        s[idx] = rng.randint(0, alpha, int(idx.sum()))
        lo = rng.randint(0, max(1, len(s) // 3) + 1) if len(s) > 2 else 0
        hi = len(s) - (rng.randint(0, max(1, len(s) // 3) + 1) if len(s) > 2 else 0)
        s = s[lo:max(hi, lo + 1)]
        if rng.rand() < 0.3:                          # an insertion
            p = rng.randint(0, len(s) + 1)
            s = np.concatenate([s[:p], rng.randint(0, alpha, rng.randint(1, 12)).astype(np.uint8), s[p:]])
        codes.append(np.ascontiguousarray(s))
    tasks = _tree(n, rng, "caterpillar" if seed % 4 == 1 else "random")
    z = np.load(os.path.join(GOLDEN, "param_tables.npz"))
    subm = z["subm_1_0"] if dna else z["subm_0_3"]
    scal = (z["scal_1_0"] if dna else z["scal_0_3"]).copy()
    if seed % 5 == 2:
        scal[3] = 0.4                                 # dist_scale: scaled penalties per task
    dist = rng.uniform(0.1, 1.5, size=n).astype(np.float32)
    k = int(rng.choice([0, 0, 2, 5])) if n >= 3 else 0
    ctx = kalign_amd.Context(0)
    recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, dist, n_anchors=k, weight=2.0)
    ctx.close()
    if k:
        orecs, opaths, ogaps, _, _, _ = oracle.msa_tree_cons(codes, tasks, subm, scal, dist, k, 2.0)
    else:
        orecs, opaths, ogaps, _ = oracle.msa_tree(codes, tasks, subm, scal, dist)
    for t, (r, o) in enumerate(zip(recs, orecs)):
        assert (r.plen, r.kind, r.swapped, r.meet, r.transition) == (o.plen, o.kind, o.swapped, o.meet, o.transition), (seed, t)
        assert r.score == o.score, (seed, t, r.score, o.score)
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], opaths[o.path_off:o.path_off + o.plen + 2]), (seed, t)
    for a, b in zip(gaps, ogaps):
        assert np.array_equal(a, b), seed
