"""GPU parity in the reference's DEFAULT mode (anchor consistency): anchor selection, the N x K
position maps, and the whole guide tree with the per-task consistency bonus -- against golden
vectors of the real reference and against the oracle on seeded inputs.  Everything the bonus
touches must still be bit-exact: paths, gap arrays, meetups, scores, f/b rows, merged profiles."""
import os

import numpy as np
import pytest

from util import GOLDEN, Golden, compare_recs, cons_cases

pytestmark = pytest.mark.gpu

EXACT = ["a", "b", "c", "len_a", "len_b", "nsip_a", "nsip_b", "plen", "kind", "swapped",
         "meet", "transition", "gap_scale", "subm_off", "score", "fhash", "bhash"]


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", cons_cases())
def test_consistency_tree_matches_reference_golden(ctx, oracle, name):
    from kalign_amd import api
    g = Golden(name)
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
    ctx.tree_build_consistency(int(g.n_anchors), float(g.weight))
    ids, maps = ctx.tree_consistency()
    assert np.array_equal(ids, g.anchor_ids)
    for got_row, want_row in zip(maps, g.maps_list()):
        for got, want in zip(got_row, want_row):
            assert np.array_equal(got, want)
    ctx.tree_run()
    recs, paths, gaps = ctx.tree_download()
    assert compare_recs(g, recs, paths, EXACT) == []
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    L = oracle.lib()
    for t, r in enumerate(recs[:-1]):
        prof = ctx.tree_profile(r.c, r.plen)
        assert L.ko_fnv1a(prof.ctypes.data, 4 * 64 * (r.plen + 2)) == int(g.rec("prof_hash")[t]), (name, t)
    # running the same job twice gives the same answer (the residue->column table is reset per run)
    ctx.tree_run()
    recs2, paths2, gaps2 = ctx.tree_download()
    assert compare_recs(g, recs2, paths2, EXACT) == []


def _random_tree(n, rng):
    nodes = list(range(n))
    tasks, nxt = [], n
    while len(nodes) > 1:
        i, j = rng.choice(len(nodes), 2, replace=False)
        tasks.append((nodes[i], nodes[j], nxt))
        nodes = [x for k, x in enumerate(nodes) if k not in (i, j)] + [nxt]
        nxt += 1
    return np.array(tasks, np.int32)


@pytest.mark.parametrize("n,length,dna,seed,k", [(40, 260, False, 13, 5), (16, 900, True, 14, 5), (80, 100, False, 15, 2), (10, 1300, False, 16, 4)])
def test_consistency_tree_matches_oracle_seeded(ctx, oracle, n, length, dna, seed, k):
    """random guide trees (unbalanced: long chains of seq-profile merges), strips + packed passes + clusters"""
    from kalign_amd import api, synth
    rng = np.random.RandomState(seed)
    seqs = synth.family(n, length, dna=dna, seed=seed)
    alpha = "ACGT" if dna else "ARNDCQEGHILKMFPSTWYV"
    codes = [np.array([alpha.index(ch) for ch in s], np.uint8) for s in seqs]
    tasks = _random_tree(n, rng)
    z = np.load(os.path.join(GOLDEN, "param_tables.npz"))
    subm = z["subm_1_0"] if dna else z["subm_0_3"]
    scal = (z["scal_1_0"] if dna else z["scal_0_3"]).copy()
    dist = rng.uniform(0.2, 1.2, size=n).astype(np.float32)
    recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, dist, flags=api.FLAG_DEBUG_ROWS, n_anchors=k, weight=2.0)
    ids, maps = ctx.tree_consistency()
    orecs, opaths, ogaps, oids, omaps, _ = oracle.msa_tree_cons(codes, tasks, subm, scal, dist, k, 2.0)
    assert np.array_equal(ids, oids)
    for ra, rb in zip(maps, omaps):
        for a, b in zip(ra, rb):
            assert np.array_equal(a, b)
    for t, (r, o) in enumerate(zip(recs, orecs)):
        for f in EXACT:
            assert getattr(r, f) == getattr(o, f), (t, f, getattr(r, f), getattr(o, f))
        assert abs(r.confidence - o.confidence) <= 1e-5 * max(1.0, abs(o.confidence))
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], opaths[o.path_off:o.path_off + o.plen + 2]), t
    for a, b in zip(gaps, ogaps):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", cons_cases())
def test_carried_votes_give_the_reference_answer(name, monkeypatch):
    """KA_CARRY=1 (off by default): a node's anchor votes follow from its operands' tables (first / last voter per column, marked cells
    settled by a sweep) instead of a count over all members at every task -- same positions, same confidences, same alignment; then a
    refinement pass on top of it (the refined nodes rebuild their tables)."""
    import kalign_amd
    from kalign_amd import api
    monkeypatch.setenv("KA_CARRY", "1")
    c = kalign_amd.Context(0)
    try:
        g = Golden(name)
        c.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
        c.tree_build_consistency(int(g.n_anchors), float(g.weight))
        for _ in range(2):
            c.tree_run()
            recs, paths, gaps = c.tree_download()
            assert compare_recs(g, recs, paths, EXACT) == []
            for got, want in zip(gaps, g.gaps_list()):
                assert np.array_equal(got, want)
        c.tree_refine(1)
        c.tree_sync()
        _, _, gaps_carried = c.tree_download()
    finally:
        c.close()
    monkeypatch.delenv("KA_CARRY")
    c = kalign_amd.Context(0)
    try:
        c.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
        c.tree_build_consistency(int(g.n_anchors), float(g.weight))
        c.tree_run()
        c.tree_refine(1)
        c.tree_sync()
        _, _, gaps_counted = c.tree_download()
    finally:
        c.close()
    for a, b in zip(gaps_carried, gaps_counted):
        assert np.array_equal(a, b)


def test_consistency_declines_like_the_reference(ctx):
    """no seq_distances / fewer than 3 sequences / K <= 0: no table, plain tree (anchor_consistency.c:206-217)"""
    g = Golden("tree_prot32x200")
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, None)
    ctx.tree_build_consistency(5, 2.0)
    assert ctx.tree_consistency() is None
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    ctx.tree_build_consistency(0, 2.0)
    assert ctx.tree_consistency() is None
    with pytest.raises(RuntimeError):
        ctx.tree_build_consistency(129, 2.0)                     # (KA_CONS_MAX_ANCHORS = 128 is what the second kernel set walks per DP row)


def test_forest_in_default_mode_has_one_table_per_alignment(ctx):
    """Two independent alignments as one forest job, both in default mode: each selects its own anchors among its
    own sequences and builds its own position maps; results per tree equal the single-tree goldens."""
    from kalign_amd import guide
    g = Golden("cons_prot32x200")
    codes, tasks, dist, spans = guide.forest([(g.codes, g.tasks, g.seq_distances)] * 2)
    recs, paths, gaps = ctx.msa_tree(codes, tasks, g.subm, g.scal, dist, n_anchors=int(g.n_anchors), weight=float(g.weight))
    ids, maps = ctx.tree_consistency()
    n = len(g.codes)
    assert np.array_equal(ids, np.concatenate([g.anchor_ids, g.anchor_ids + n]))
    for (s0, t0, ns, nt) in spans:
        for t in range(nt):
            r = recs[t0 + t]
            assert r.plen == g.rec("plen")[t] and r.score == g.rec("score")[t]
            assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], g.path(t)), t
        for got, want in zip(gaps[s0:s0 + ns], g.gaps_list()):
            assert np.array_equal(got, want)
        for got_row, want_row in zip(maps[s0:s0 + ns], g.maps_list()):
            for got, want in zip(got_row, want_row):
                assert np.array_equal(got, want - 0)          # positions are within the anchor sequence: no offset


def test_a_node_with_65536_members_votes_in_32_bits(ctx, oracle):
    """65537 short sequences, a balanced tree over 65536 of them and the last one joining at the root: the root task's
    profile side has 65536 members, where a cell's `total` and `agree` counts no longer fit 16 bits each
    (anchor_consistency.c:352-470 counts in ints)"""
    N, LEN = 65537, 8
    rng = np.random.RandomState(5)
    root = rng.randint(0, 20, size=LEN)
    codes = []
    for i in range(N):
        s = root.copy()
        m = rng.random_sample(LEN) < 0.15
        s[m] = rng.randint(0, 20, size=int(m.sum()))
        if rng.random_sample() < 0.3:
            s = np.delete(s, rng.randint(LEN))
        codes.append(s.astype(np.uint8))
    nodes, tasks, nxt = list(range(N - 1)), [], N
    while len(nodes) > 1:
        new = []
        for k in range(0, len(nodes) - 1, 2):
            tasks.append((nodes[k], nodes[k + 1], nxt)); new.append(nxt); nxt += 1
        if len(nodes) & 1:
            new.append(nodes[-1])
        nodes = new
    tasks.append((nodes[0], N - 1, nxt))
    tasks = np.array(tasks, np.int32)
    z = np.load(os.path.join(GOLDEN, "param_tables.npz"))
    subm, scal = z["subm_0_3"], z["scal_0_3"].copy()
    dist = rng.uniform(0.2, 1.2, size=N).astype(np.float32)
    recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, dist, n_anchors=3, weight=2.0)
    orecs, opaths, ogaps, oids, omaps, _ = oracle.msa_tree_cons(codes, tasks, subm, scal, dist, 3, 2.0)
    assert max(orecs[-1].nsip_a, orecs[-1].nsip_b) == 65536
    for t in list(range(0, len(recs), 997)) + list(range(len(recs) - 40, len(recs))):
        r, o = recs[t], orecs[t]
        for f in ("plen", "meet", "transition", "score", "nsip_a", "nsip_b"):
            assert getattr(r, f) == getattr(o, f), (t, f)
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], opaths[o.path_off:o.path_off + o.plen + 2]), t
    for a, b in zip(gaps[::257] + gaps[-3:], ogaps[::257] + ogaps[-3:]):
        assert np.array_equal(a, b)
