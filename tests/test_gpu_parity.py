"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the real
reference and against the oracle on seeded inputs.  Integer/byte outputs (paths, gap arrays,
meetup columns/transitions) bit-exact; float outputs: merged profiles, top-level f/b rows and
scores bit-exact as well (the HIP kernels keep the reference's evaluation order), confidence
within 1e-5 relative (it is a mean whose summation order differs)."""
import numpy as np
import pytest

from util import Golden, compare_recs, cons_cases, pair_cases, tree_cases

pytestmark = pytest.mark.gpu

EXACT = ["a", "b", "c", "len_a", "len_b", "nsip_a", "nsip_b", "plen", "kind", "swapped",
         "meet", "transition", "gap_scale", "subm_off", "score", "fhash", "bhash"]


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", tree_cases())
def test_tree_matches_reference_golden(ctx, oracle, name):
    from kalign_amd import api
    g = Golden(name)
    recs, paths, gaps = ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
    assert compare_recs(g, recs, paths, EXACT) == []
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    # merged profiles: hash of every non-root node + the dumped one bit for bit
    L = oracle.lib()
    for t, r in enumerate(recs[:-1]):
        prof = ctx.tree_profile(r.c, r.plen)
        n = 64 * (r.plen + 2)
        h = L.ko_fnv1a(prof.ctypes.data, 4 * n)
        assert h == int(g.rec("prof_hash")[t]), (name, t)
    t = int(g.dump_task)
    if t < len(recs) - 1:
        prof = ctx.tree_profile(recs[t].c, recs[t].plen)
        assert np.array_equal(prof[:len(g.dump)].view(np.uint32), g.dump.view(np.uint32))


@pytest.mark.parametrize("name", tree_cases() + cons_cases())
def test_tree_exact_confidence_flag(ctx, name):
    """KA_FLAG_EXACT_CONFIDENCE: ka_tree_run's task confidences are the reference's floats bit for bit (the meetup
    margins sorted into the recursion order before they are added), next to everything in EXACT"""
    from kalign_amd import api
    g = Golden(name)
    cons = hasattr(g, "n_anchors") and int(g.n_anchors) > 0
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_EXACT_CONFIDENCE | api.FLAG_DEVICE_GAPS)
    if cons:
        ctx.tree_build_consistency(int(g.n_anchors), float(g.weight))
    ctx.tree_run()
    ctx.tree_sync()
    recs, paths, gaps = ctx.tree_download()
    fields = [f for f in EXACT if f not in ("fhash", "bhash")] + ["confidence"]
    assert compare_recs(g, recs, paths, fields, tol_fields=()) == []
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.rec("confidence").astype(np.float32))
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert ctx.fallback_runs() == 0


@pytest.mark.parametrize("name", pair_cases())
def test_pairwise_matches_reference_golden(ctx, name):
    g = Golden(name)
    paths, scores = ctx.pairwise_batch(g.codes, g.ia, g.ib, g.subm, g.scal[0], g.scal[1], g.scal[2])
    o = 0
    for k, p in enumerate(paths):
        n = int(g.plen[k]) + 2
        assert np.array_equal(p, g.paths[o:o + n]), k
        o += n


def random_tree(n, rng):
    """random binary guide tree in TASK_ORDER_TREE order"""
    nodes = list(range(n))
    tasks, nxt = [], n
    while len(nodes) > 1:
        i, j = rng.choice(len(nodes), 2, replace=False)
        a, b = nodes[i], nodes[j]
        tasks.append((a, b, nxt))
        nodes = [x for k, x in enumerate(nodes) if k not in (i, j)] + [nxt]
        nxt += 1
    return np.array(tasks, np.int32)


@pytest.mark.parametrize("n,length,dna,seed", [(48, 260, False, 3), (24, 700, True, 4), (96, 90, False, 5), (12, 1500, True, 6)])
def test_tree_matches_oracle_seeded(ctx, oracle, n, length, dna, seed):
    from kalign_amd import api, synth
    rng = np.random.RandomState(seed)
    seqs = synth.family(n, length, dna=dna, seed=seed)
    alpha = "ACGT" if dna else "ARNDCQEGHILKMFPSTWYV"
    codes = [np.array([alpha.index(ch) for ch in s], np.uint8) for s in seqs]
    tasks = random_tree(n, rng)
    z = np.load(__import__("os").path.join(__import__("util").GOLDEN, "param_tables.npz"))
    subm = z["subm_1_0"] if dna else z["subm_0_3"]
    scal = (z["scal_1_0"] if dna else z["scal_0_3"]).copy()
    dist = rng.uniform(0.2, 1.2, size=n).astype(np.float32)
    recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, dist, flags=api.FLAG_DEBUG_ROWS)
    orecs, opaths, ogaps, _ = oracle.msa_tree(codes, tasks, subm, scal, dist)
    for t, (r, o) in enumerate(zip(recs, orecs)):
        for f in EXACT:
            assert getattr(r, f) == getattr(o, f), (t, f)
        assert abs(r.confidence - o.confidence) <= 1e-5 * max(1.0, abs(o.confidence))
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], opaths[o.path_off:o.path_off + o.plen + 2]), t
    for a, b in zip(gaps, ogaps):
        assert np.array_equal(a, b)


def test_pairwise_matches_oracle_seeded(ctx, oracle):
    from kalign_amd import synth
    seqs = synth.family(40, 350, seed=9)
    alpha = "ARNDCQEGHILKMFPSTWYV"
    codes = [np.array([alpha.index(ch) for ch in s], np.uint8) for s in seqs]
    z = np.load(__import__("os").path.join(__import__("util").GOLDEN, "param_tables.npz"))
    subm, scal = z["subm_0_3"], z["scal_0_3"]
    ia = np.repeat(np.arange(40), 5).astype(np.int32)
    ib = np.tile(np.arange(5), 40).astype(np.int32)
    keep = ia != ib
    ia, ib = ia[keep], ib[keep]
    paths, scores = ctx.pairwise_batch(codes, ia, ib, subm, scal[0], scal[1], scal[2])
    opaths, oscores = oracle.pairwise_batch(codes, ia, ib, subm, float(scal[0]), float(scal[1]), float(scal[2]))
    for k in range(len(ia)):
        assert np.array_equal(paths[k], opaths[k]), k
    assert np.array_equal(scores, oscores)
