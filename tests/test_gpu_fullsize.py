"""Full-size checks on the GPU box (BASELINE.json configs): the bench workload against the REAL
reference's dispatcher (the prebuilt oracle/_ref travels with the repo; its create_msa_tree runs
in well under a second on the host cores) and through size-independent properties."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def check_path_invariants(recs, paths):
    """Every coded path consumes exactly len_a positions of a and len_b of b, has the declared
    length and terminator, and only the op codes the reference can emit (0, 1, 2, 33, 34)."""
    for r in recs:
        p = paths[r.path_off:r.path_off + r.plen + 2]
        assert p[0] == r.plen and p[-1] == 3
        ops = p[1:-1]
        assert np.isin(ops, [0, 1, 2, 33, 34]).all()
        assert int((ops == 0).sum() + ((ops & 2) != 0).sum()) == r.len_a
        assert int((ops == 0).sum() + ((ops & 1) != 0).sum()) == r.len_b
        lead = np.flatnonzero(ops == 0)
        if len(lead):
            assert ((ops[:lead[0]] & 32) != 0).all() and ((ops[lead[-1] + 1:] & 32) != 0).all()
            assert ((ops[lead[0]:lead[-1] + 1] & 32) == 0).all()


def run_case(nseq, length, dna, seed=1, reference=True, n_anchors=0):
    import bench
    import kalign_amd
    codes, tasks, dist = bench.make_workload(nseq, length, dna, seed)
    subm, scal = bench.scoring(dna)
    ctx = kalign_amd.Context(0)
    recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, dist, n_anchors=n_anchors, weight=2.0)
    ctx.close()
    check_path_invariants(recs, paths)
    # all rows of the final alignment have the same length, residues are preserved by construction
    lens = np.array([len(c) for c in codes])
    total = np.array([int(g.sum()) for g in gaps]) + lens
    assert (total == total[0]).all() and total[0] == recs[-1].plen
    # parents consume their children's alignment lengths
    plen = {r.c: r.plen for r in recs}
    for r in recs:
        assert r.len_a == (plen[r.a] if r.a in plen else lens[r.a])
        assert r.len_b == (plen[r.b] if r.b in plen else lens[r.b])
    if reference:
        from oracle import refdrv
        if not refdrv.available():
            pytest.skip("oracle/_ref not built")
        job = refdrv.EncodedJob(codes, tasks, dist, biotype=1 if dna else 0, type_=0 if dna else -1,
                                n_threads=min(16, os.cpu_count() or 1))
        if n_anchors:
            job.build_consistency(n_anchors, 2.0)
        ref_gaps, _ = job.run_tree()
        job.close()
        for got, want in zip(gaps, ref_gaps):
            assert np.array_equal(got, want)


def test_config1_protein_1024x400_matches_reference():
    """BASELINE.json configs[1] shape (the bench.py default): bit-identical gap arrays."""
    run_case(1024, 400, False)


def test_headline_shape_protein_4096x400_matches_reference():
    """The shape `value` is quoted on (north_star: 4096 x 400 aa; bench.py's default workload, where the same comparison runs
    inside the CPU leg): every gap array of the whole tree against the real reference's create_msa_tree on the host cores."""
    run_case(4096, 400, False)


def test_headline_shape_through_the_throughput_kernel(monkeypatch):
    """The same tree with the levels of the queued launch on the opt-in throughput kernel (KA_TP=1, unit 10: ka_lstrip's lean
    profile-profile strips, three workgroups per CU): the reference's gap arrays bit for bit."""
    monkeypatch.setenv("KA_TP", "1")
    run_case(4096, 400, False)


def test_config2_shape_dna_256x2000_matches_reference():
    """configs[2] shape (--type dna, ~2000 nt, profile-profile dominated), scaled to 256 sequences
    so the CPU reference finishes in seconds: long anti-diagonals, multi-strip passes, clusters."""
    run_case(256, 2000, True)


def test_config1_default_mode_512x400_matches_reference():
    """The CLI's default mode (5 consistency anchors, weight 2.0) at the configs[1] shape, scaled to 512
    sequences because the reference builds its N x K position maps serially (~1 ms per pair)."""
    run_case(512, 400, False, n_anchors=5)


def test_default_mode_2304_sequences_votes_shared_by_member_ranges_match_reference(monkeypatch):
    """Default mode on a tree of >= 2048 sequences: the top tasks run on clusters of 20 / 30 workgroups and the votes of an
    (operand, anchor) pair are shared by member ranges (ka_cons_votes_split: partial tables in LDS, merged through the task's
    HBM table).  Against the reference, and the same job with the cluster limit at 16 (one workgroup per pair)."""
    run_case(2304, 120, False, n_anchors=5)
    # the same job again and again on one context: the partial tables meet through atomics in HBM behind cluster barriers --
    # a race would show as a run that differs
    import bench
    import kalign_amd
    codes, tasks, dist = bench.make_workload(2304, 120, False, 1)
    subm, scal = bench.scoring(False)
    ctx = kalign_amd.Context(0)
    try:
        ctx.tree_upload(codes, tasks, subm, scal, dist)
        ctx.tree_build_consistency(5, 2.0)
        first = None
        for rep in range(6):
            ctx.tree_run()
            recs, paths, _ = ctx.tree_download(want_gaps=False)
            sig = (np.ascontiguousarray(paths).tobytes(), tuple((r.plen, r.meet, r.transition, r.score) for r in recs))
            if first is None:
                first = sig
            assert sig == first, rep
        assert ctx.fallback_runs() == 0
    finally:
        ctx.close()
    monkeypatch.setenv("KA_MAX_CLUSTER", "16")
    run_case(2304, 120, False, n_anchors=5)


@pytest.mark.parametrize("nseq,length,dna,k", [(384, 200, False, 8), (96, 900, True, 10), (160, 300, False, 6),
                                               (256, 250, False, 12), (128, 500, True, 16), (200, 180, False, 32), (700, 300, False, 24),
                                               (160, 150, False, 48), (192, 120, False, 100), (140, 200, True, 128)])
def test_more_than_five_anchors_match_reference(nseq, length, dna, k):
    """`--consistency K` with 5 < K <= 128 (32 until round 6): the second set of consistency kernels -- round 5: a DP row's bonus entries are no longer held
    in registers but walked, sorted by column, along with the row's columns (KaBonus::STREAM) -- anchor selection, position maps,
    votes (ten anchors per sweep), bonus entries and the whole tree against the reference."""
    run_case(nseq, length, dna, n_anchors=k)


def test_config2_full_size_dna_4096x2000_properties():
    """configs[2] at its full size (4096 DNA x ~2000, 2e10 useful cells, root profile > 10k columns): no CPU
    run of this size fits a test, so the size-independent properties decide: every coded path consumes
    exactly its operands, parents consume their children's lengths, all rows of the woven alignment
    have the root's length."""
    run_case(4096, 2000, True, reference=False)


def test_config3_shape_protein_16384x500_properties():
    """configs[3] shape on ONE GPU (the 8-GPU form shards it by subtree, kalign_amd/dist.py)."""
    run_case(16384, 500, False, reference=False)


def test_two_sequences_and_tiny_inputs():
    import kalign_amd
    import bench
    subm, scal = bench.scoring(False)
    ctx = kalign_amd.Context(0)
    from oracle import oracledrv
    for lens in [(1, 1), (1, 7), (9, 1), (3, 2), (64, 65), (129, 128), (257, 300)]:
        rng = np.random.RandomState(sum(lens))
        codes = [rng.randint(0, 20, n).astype(np.uint8) for n in lens]
        tasks = np.array([[0, 1, 2]], np.int32)
        recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, None)
        orecs, opaths, ogaps, _ = oracledrv.msa_tree(codes, tasks, subm, scal, None)
        assert np.array_equal(paths[:recs[0].plen + 2], opaths[:orecs[0].plen + 2]), lens
        assert recs[0].score == orecs[0].score
    ctx.close()


def test_error_behaviour():
    """FAIL (non-zero + message) like the reference's convention, never a crash or a hang."""
    import kalign_amd
    import bench
    subm, scal = bench.scoring(False)
    ctx = kalign_amd.Context(0)
    a = np.arange(5, dtype=np.uint8)
    with pytest.raises(kalign_amd.KalignAmdError):          # parents before children
        ctx.msa_tree([a, a, a], np.array([[3, 2, 4], [0, 1, 3]], np.int32), subm, scal)
    with pytest.raises(kalign_amd.KalignAmdError):          # code outside the alphabet
        ctx.msa_tree([a, (a + 30).astype(np.uint8)], np.array([[0, 1, 2]], np.int32), subm, scal)
    with pytest.raises(kalign_amd.KalignAmdError):          # more tasks than a tree over 3 sequences can have
        ctx.msa_tree([a, a, a], np.array([[0, 1, 3], [3, 2, 4], [4, 0, 5]], np.int32), subm, scal)
    with pytest.raises(kalign_amd.KalignAmdError):          # a node consumed by two tasks
        ctx.msa_tree([a, a, a, a], np.array([[0, 1, 4], [0, 2, 5]], np.int32), subm, scal)
    with pytest.raises(kalign_amd.KalignAmdError):          # a node aligned to itself
        ctx.msa_tree([a, a, a], np.array([[0, 0, 3]], np.int32), subm, scal)
    # fewer tasks than numseq-1 is a forest (here: one pair plus a sequence that stays alone)
    recs, paths, gaps = ctx.msa_tree([a, a, a], np.array([[0, 1, 3]], np.int32), subm, scal)
    assert recs[0].plen == 5 and all(int(g.sum()) == 0 for g in gaps)
    ctx.close()


def test_arena_overflow_grows_and_reruns():
    """Device arenas (profiles, paths, scratch) are bump-allocated by the kernels; when one overflows the run is
    repeated with a bigger arena (ka_tree_sync).  Start with arenas that are far too small: the result must be
    the reference's, and a failed task must not leave the clusters waiting at its parent's join point hanging."""
    import kalign_amd
    from util import Golden
    g = Golden("tree_prot64_gon")
    ctx = kalign_amd.Context(0)
    ctx.debug_set_hooks(1)                                   # KA_DEBUG_SMALL_ARENAS
    recs, paths, gaps = ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    ctx.close()
    for t, r in enumerate(recs):
        assert r.plen == g.rec("plen")[t]
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], g.path(t)), t
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)


def test_starved_join_is_reported_and_the_run_replanned():
    """A join of the chained launch whose sibling never arrives (what a non-resident workgroup looks like from the
    device): the wait is bounded (~2 s), the run is re-planned without joins / clusters and repeated."""
    import kalign_amd
    from util import Golden
    g = Golden("tree_prot64_gon")
    ctx = kalign_amd.Context(0)
    ctx.debug_set_hooks(2)                                   # KA_DEBUG_STARVE_ROOT_JOIN
    recs, paths, gaps = ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    assert ctx.fallback_runs() == 1                          # visible to the caller, not a silent cliff
    ctx.debug_set_hooks(0)
    ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    assert ctx.fallback_runs() == 1                          # the next job is back on the fast plan
    ctx.close()
    for t, r in enumerate(recs):
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], g.path(t)), t
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)


def test_a_gpu_that_stays_shared_costs_two_stalls_not_one_per_job():
    """The job after a fallback tries the fast plan again (above).  When that one stalls too the context stays on the plan
    without waits for 4 jobs (then 16, 64) before the next try -- a GPU somebody else keeps using, or a runtime that keeps
    the two launches apart, must not cost every job its two seconds (ka_tree_sync, ka_tree_upload)."""
    import kalign_amd
    from util import Golden
    g = Golden("tree_prot64_gon")
    ctx = kalign_amd.Context(0)
    ctx.debug_set_hooks(2)                                   # KA_DEBUG_STARVE_ROOT_JOIN, for every job from here
    for want in (1, 2, 2, 2, 2, 2, 3):                       # stall, stall, four jobs held on the shared plan, the next try stalls again
        recs, paths, gaps = ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
        assert ctx.fallback_runs() == want
        for got, ref in zip(gaps, g.gaps_list()):
            assert np.array_equal(got, ref)
    ctx.debug_set_hooks(0)
    ctx.close()
