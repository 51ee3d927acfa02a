"""Pins the oracle (oracle/kalign_oracle.c) against golden vectors produced by
the REAL reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from util import Golden, compare_recs, cons_cases, pair_cases, refine_cases, tree_cases

EXACT = ["a", "b", "c", "len_a", "len_b", "nsip_a", "nsip_b", "plen", "kind", "swapped",
         "meet", "transition", "gap_scale", "subm_off", "score", "prof_hash", "fhash", "bhash"]


@pytest.mark.parametrize("name", tree_cases())
def test_tree_matches_reference(oracle, name):
    g = Golden(name)
    recs, paths, gaps, dump = oracle.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances,
                                              dump_task=int(g.dump_task))
    assert compare_recs(g, recs, paths, EXACT) == []
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    n = len(g.dump)
    assert np.array_equal(dump[:n].view(np.uint32), g.dump.view(np.uint32))
    # finalise_alignment + rank order == the reference's aligned rows
    rows_sorted = oracle.rows_from_gaps(g.sorted_seqs(), gaps)
    rows = [None] * len(rows_sorted)
    for i, r in enumerate(g.ranks):
        rows[int(r)] = rows_sorted[i]
    assert rows == [str(x) for x in g.rows]


@pytest.mark.parametrize("name", cons_cases())
def test_consistency_tree_matches_reference(oracle, name):
    """default mode: anchor selection, N x K position maps, per-task bonus matrices, bonus-aware DP"""
    g = Golden(name)
    recs, paths, gaps, ids, maps, bh = oracle.msa_tree_cons(g.codes, g.tasks, g.subm, g.scal, g.seq_distances,
                                                            int(g.n_anchors), float(g.weight))
    assert np.array_equal(ids, g.anchor_ids)
    for got_row, want_row in zip(maps, g.maps_list()):
        for got, want in zip(got_row, want_row):
            assert np.array_equal(got, want)
    assert np.array_equal(bh, g.bonus_hash)
    assert compare_recs(g, recs, paths, EXACT) == []
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    rows_sorted = oracle.rows_from_gaps(g.sorted_seqs(), gaps)
    rows = [None] * len(rows_sorted)
    for i, r in enumerate(g.ranks):
        rows[int(r)] = rows_sorted[i]
    assert rows == [str(x) for x in g.rows]


@pytest.mark.parametrize("name", pair_cases())
def test_pairwise_matches_reference(oracle, name):
    g = Golden(name)
    paths, scores = oracle.pairwise_batch(g.codes, g.ia, g.ib, g.subm, float(g.scal[0]), float(g.scal[1]), float(g.scal[2]))
    o = 0
    for k, p in enumerate(paths):
        n = int(g.plen[k]) + 2
        assert np.array_equal(p, g.paths[o:o + n]), k
        o += n


def test_code_path_edge_cases(oracle):
    import ctypes as C
    L = oracle.lib()
    # all rows unmatched then all columns skipped is not producible by Gotoh; check simple shapes
    for raw, la, lb, want in [
        ([-1, 1, 2, 3], 3, 3, [3, 0, 0, 0, 3]),
        ([-1, 2, 3, -1], 3, 4, [4, 33, 0, 0, 34, 3]),      # quirk: no trailing gap-in-a after an unmatched last row
        ([-1, -1, 1, 2], 3, 2, [3, 34, 0, 0, 3]),
        ([-1, 1, 4], 2, 5, [5, 0, 1, 1, 0, 33, 3]),
    ]:
        r = np.array(raw + [0] * 8, np.int32)
        out = np.zeros(la + lb + 3, np.int32)
        L.ko_code_path(r.ctypes.data_as(C.c_void_p), la, lb, out.ctypes.data_as(C.c_void_p))
        assert out[:len(want)].tolist() == want


def test_bpm_distances_match_reference(oracle):
    """distance estimation: bpm_block through calc_distance, 1024 pairs, lengths 1..1500"""
    g = Golden("bpm_mixed")
    assert np.array_equal(oracle.bpm_batch(g.codes, g.ia, g.ib), g.dist)


def _realign_cases():
    import os
    from util import GOLDEN
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("realign_") and f.endswith(".npz"))


@pytest.mark.parametrize("name", _realign_cases())
def test_realign_tree_matches_reference(oracle, name):
    """the oracle's restatement of compute_aln_pairwise_dist + build_tree_from_pairwise against what the reference's
    own kalign_run_realign loop produced"""
    import os
    from util import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    tasks, sd, dm = oracle.aln_guide_tree([str(r) for r in z["rows_sorted"]])
    assert np.array_equal(dm.view(np.uint32), z["dm"].view(np.uint32))
    assert np.array_equal(tasks, z["tasks2"])
    assert np.array_equal(sd.view(np.uint32), z["seq_distances2"].view(np.uint32))


def test_realign_tree_live_against_the_reference(oracle):
    """more shapes than the goldens hold, when oracle/_ref is there to ask: duplicated sequences (tied distances),
    two sequences, a family whose first alignment has long terminal gaps"""
    from oracle import refdrv
    from kalign_amd import synth
    if not refdrv.available():
        pytest.skip("oracle/_ref not built")
    fam = synth.family(25, 70, seed=41)
    cases = [fam + fam[:9], synth.family(2, 40, seed=42), [s[i % 30:] for i, s in enumerate(synth.family(33, 120, seed=43))],
             synth.family(64, 50, dna=True, seed=44)]
    for seqs in cases:
        job = refdrv.RefJob(seqs)
        job.run_tree()
        rows, dm, _, _ = job.realign_tree()
        tasks, sd, odm = oracle.aln_guide_tree(rows)
        assert np.array_equal(odm.view(np.uint32), dm.view(np.uint32))
        assert np.array_equal(tasks, job.tasks)
        assert np.array_equal(sd.view(np.uint32), job.seq_distances.view(np.uint32))
        job.close()


def test_tree_live_against_the_reference(oracle):
    """the oracle's dispatcher against the real one on seeded inputs the goldens do not hold (both modes, protein and
    DNA, ragged lengths, distance-scaled penalties), when oracle/_ref is there to ask"""
    from oracle import refdrv
    from kalign_amd import synth
    if not refdrv.available():
        pytest.skip("oracle/_ref not built")
    fam = synth.family(28, 110, seed=52)
    cases = [(synth.family(19, 60, seed=51), dict(), 0), (synth.family(21, 90, dna=True, seed=53), dict(), 4),
             ([s[:20 + (11 * i) % 90] for i, s in enumerate(fam)], dict(), 5),
             (synth.family(17, 75, seed=54), dict(dist_scale=0.5, vsm_amax=1.5), 0),
             (synth.family(3, 30, seed=55), dict(), 0)]
    for seqs, kw, k in cases:
        job = refdrv.RefJob(seqs, **kw)
        scal = np.array([job.gpo, job.gpe, job.tgpe, job.dist_scale, job.vsm_amax, job.use_seq_weights], np.float32)
        if k:
            job.build_consistency(k, 2.0)
        recs, paths, gaps, _ = job.run_tree_traced()
        if k:
            orecs, opaths, ogaps, _, _, _ = oracle.msa_tree_cons(job.codes, job.tasks, job.subm, scal, job.seq_distances, k, 2.0)
        else:
            orecs, opaths, ogaps, _ = oracle.msa_tree(job.codes, job.tasks, job.subm, scal, job.seq_distances)
        for r, o in zip(recs, orecs):
            assert (r.plen, r.meet, r.transition, r.kind, r.swapped, r.score) == (o.plen, o.meet, o.transition, o.kind, o.swapped, o.score)
            assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], opaths[o.path_off:o.path_off + o.plen + 2])
        for a, b in zip(gaps, ogaps):
            assert np.array_equal(a, b)
        job.close()


@pytest.mark.parametrize("name", refine_cases())
def test_refinement_matches_reference(oracle, name):
    """refine_alignment: second pass with convert_raw_path coding, five trials per refined edge
    (round-robin flips of uncertain meetups in DFS order), SP scoring, best trial kept -- gap arrays, task confidences,
    lengths and rows of the real reference (KALIGN_REFINE_ALL and _CONFIDENT, with and without consistency)"""
    g = Golden(name)
    recs, paths, gaps = oracle.msa_tree_refine(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, mode=int(g.mode),
                                               conf_in=g.conf_before, n_anchors=int(g.n_anchors), weight=float(g.weight))
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)
    assert all(recs[t].plen == g.plen_after[g.tasks[t][2]] for t in range(len(recs)))
    assert int(g.n_differ) > 0 or int(g.mode) == 3               # the refinement really changed the alignment
    rows_sorted = oracle.rows_from_gaps(g.sorted_seqs(), gaps)
    rows = [None] * len(rows_sorted)
    for i, r in enumerate(g.ranks):
        rows[int(r)] = rows_sorted[i]
    assert rows == [str(x) for x in g.rows]


# ---- Hirschberg prefix reuse (round 5): the rule the kernels use, proven exact on the CPU first ----
@pytest.fixture
def reuse(oracle):
    oracle.set_prefix_reuse(True)
    yield oracle
    oracle.set_prefix_reuse(False)


@pytest.mark.parametrize("name", tree_cases())
def test_prefix_reuse_leaves_every_tree_golden_bit_identical(reuse, name):
    """A child takes its parent's saved row (forward: after (n-1)/2 rows, backward: after n/2 rows; last column's ga := -FLT_MAX)
    instead of running the pass -- ~21 % of the DP cells are never computed, and nothing the reference produces changes:
    coded paths, meetups, scores, merged profiles, gap arrays (aln_controller.c:194-436, aln_seqseq.c:40-58,108-117)."""
    g = Golden(name)
    recs, paths, gaps, dump = reuse.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, dump_task=int(g.dump_task))
    run, reused = reuse.prefix_reuse_cells()
    assert compare_recs(g, recs, paths, EXACT) == []
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert reused > 0 and reused < run
    # (2 x rows x columns without reuse; ~1.58 x with it on square tasks: at least a tenth of the cells must have gone)
    assert reused / float(run + reused) > 0.10, (run, reused)


@pytest.mark.parametrize("name", cons_cases())
def test_prefix_reuse_with_the_consistency_bonus(reuse, name):
    """... with the bonus matrix in every pass, incl. its 1-based / 0-based column quirk and the wrap-around cell"""
    g = Golden(name)
    recs, paths, gaps, ids, maps, bh = reuse.msa_tree_cons(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, int(g.n_anchors), float(g.weight))
    assert np.array_equal(bh, g.bonus_hash)
    assert compare_recs(g, recs, paths, EXACT) == []
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name", pair_cases())
def test_prefix_reuse_in_the_pair_batch(reuse, name):
    g = Golden(name)
    paths, scores = reuse.pairwise_batch(g.codes, g.ia, g.ib, g.subm, float(g.scal[0]), float(g.scal[1]), float(g.scal[2]))
    o = 0
    for k, p in enumerate(paths):
        n = int(g.plen[k]) + 2
        assert np.array_equal(p, g.paths[o:o + n]), k
        o += n


@pytest.mark.parametrize("name", refine_cases()[:8])
def test_prefix_reuse_in_refinement_trials(reuse, name):
    """flip trials (aln_seqseq.c:376-414): a flipped meetup changes the children's windows, never what a saved row holds --
    the rule only asks that the child's pass have the row count the parent saved"""
    g = Golden(name)
    recs, paths, gaps = reuse.msa_tree_refine(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, mode=int(g.mode),
                                              conf_in=g.conf_before, n_anchors=int(g.n_anchors), weight=float(g.weight))
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)


# ---- carried anchor votes (round 5; the device's KA_CARRY=1): the recurrence, proven exact on the CPU ----
@pytest.fixture
def carried(oracle):
    oracle.set_carried_votes(True)
    yield oracle
    oracle.set_carried_votes(False)


@pytest.mark.parametrize("name", cons_cases())
def test_carried_votes_leave_every_consistency_golden_bit_identical(carried, name):
    """A node's anchor positions from its operands' tables (first / last voter per anchor and column, sip[c] = rev(sip[a]) ++ rev(sip[b]),
    aln_run.c:428-436) instead of get_node_anchor_positions' count over all members (anchor_consistency.c:352-470): the same bonus
    matrices (hashes of the reference's dense ones), the same paths, scores, gap arrays."""
    g = Golden(name)
    recs, paths, gaps, ids, maps, bh = carried.msa_tree_cons(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, int(g.n_anchors), float(g.weight))
    cells, counted = carried.carried_votes_cells()
    assert np.array_equal(bh, g.bonus_hash)
    assert compare_recs(g, recs, paths, EXACT) == []
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert cells > 0 and counted <= cells


@pytest.mark.parametrize("name", [n for n in refine_cases() if int(Golden(n).n_anchors) > 0][:6])
def test_carried_votes_in_a_refinement_pass(carried, name):
    g = Golden(name)
    recs, paths, gaps = carried.msa_tree_refine(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, mode=int(g.mode),
                                                conf_in=g.conf_before, n_anchors=int(g.n_anchors), weight=float(g.weight))
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)


def test_carried_votes_need_counts_where_voters_disagree(carried):
    """On a divergent family some cells cannot be settled from the operands' cells alone (the device marks them and sweeps once):
    the tally says how many -- and the answer is still the count's."""
    from kalign_amd import synth
    import os
    seqs = synth.dssim(48, 120, dna=False, seed=5)
    alpha = "ARNDCQEGHILKMFPSTWYV"
    codes = [np.array([alpha.index(ch) for ch in s], np.uint8) for s in seqs]
    n = len(codes)
    tasks, nodes, nxt = [], list(range(n)), n
    rng = np.random.RandomState(3)
    while len(nodes) > 1:
        i, j = rng.choice(len(nodes), 2, replace=False)
        tasks.append((nodes[i], nodes[j], nxt))
        nodes = [x for q, x in enumerate(nodes) if q not in (i, j)] + [nxt]
        nxt += 1
    tasks = np.array(tasks, np.int32)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "param_tables.npz"))
    subm, scal = z["subm_0_3"], z["scal_0_3"].copy()
    dist = rng.uniform(0.2, 1.2, size=n).astype(np.float32)
    out1 = carried.msa_tree_cons(codes, tasks, subm, scal, dist, 5, 2.0)
    cells, counted = carried.carried_votes_cells()
    carried.set_carried_votes(False)
    out0 = carried.msa_tree_cons(codes, tasks, subm, scal, dist, 5, 2.0)
    assert np.array_equal(out1[5], out0[5])                       # bonus hashes of every task
    assert np.array_equal(out1[1], out0[1])
    for a, b in zip(out1[2], out0[2]):
        assert np.array_equal(a, b)
    assert 0 < counted < cells, (cells, counted)
