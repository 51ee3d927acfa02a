"""Nothing may depend on what a device arena held before the run (KA_DEBUG_POISON_ARENAS: the profile, scratch and path arenas start
every run as 0xff bytes -- NaN as floats, -1 as ints).  Round 6 stopped writing the profile records of SEQUENCES (the merge makes a
sequence's record from its residue, ka_update_profile / make_profile_n, aln_setup.c:40-99): every reader of a leaf's record must do
the same, on every path -- gap columns at the ends of their runs (update_n's two-step adjustments, aln_setup.c:230-436), refinement,
jobs with a consistency table, forests."""
import numpy as np
import pytest

from util import Golden, compare_recs, tree_cases, cons_cases, refine_cases

pytestmark = pytest.mark.gpu

EXACT = ["len_a", "len_b", "nsip_a", "nsip_b", "plen", "kind", "swapped", "meet", "transition", "gap_scale", "subm_off", "score"]
POISON = 16


@pytest.mark.parametrize("name", tree_cases())
def test_tree_goldens_on_poisoned_arenas(name, oracle):
    import kalign_amd
    g = Golden(name)
    ctx = kalign_amd.Context(0)
    try:
        ctx.debug_set_hooks(POISON)
        for _ in range(2):
            recs, paths, gaps = ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
            assert ctx.fallback_runs() == 0
            assert compare_recs(g, recs, paths, EXACT) == []
            for got, want in zip(gaps, g.gaps_list()):
                assert np.array_equal(got, want)
        L = oracle.lib()
        for t, r in enumerate(recs[:-1]):                  # every merged profile, bit for bit (the root's is not made)
            prof = ctx.tree_profile(r.c, r.plen)
            assert L.ko_fnv1a(prof.ctypes.data, 4 * 64 * (r.plen + 2)) == int(g.rec("prof_hash")[t]), (name, t)
    finally:
        ctx.close()


@pytest.mark.parametrize("name", cons_cases())
def test_consistency_goldens_on_poisoned_arenas(name):
    import kalign_amd
    g = Golden(name)
    ctx = kalign_amd.Context(0)
    try:
        ctx.debug_set_hooks(POISON)
        ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
        ctx.tree_build_consistency(int(g.n_anchors), float(g.weight))
        for _ in range(2):
            ctx.tree_run()
            recs, paths, gaps = ctx.tree_download()
            assert compare_recs(g, recs, paths, EXACT) == []
            for got, want in zip(gaps, g.gaps_list()):
                assert np.array_equal(got, want)
    finally:
        ctx.close()


def test_refinement_on_poisoned_arenas():
    """a refinement pass after a first pass: same gaps as on a context whose arenas were left alone"""
    import bench
    import kalign_amd
    codes, tasks, dist = bench.make_workload(192, 150, False, 9)
    subm, scal = bench.scoring(False)
    out = []
    for hooks in (0, POISON):
        ctx = kalign_amd.Context(0)
        try:
            ctx.debug_set_hooks(hooks)
            ctx.tree_upload(codes, tasks, subm, scal, dist)
            ctx.tree_run()
            ctx.tree_refine(1)
            ctx.tree_sync()
            out.append(ctx.tree_download()[2])
        finally:
            ctx.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(1200, 300, False, 0), (600, 250, False, 5), (700, 500, True, 0)])
def test_synthetic_trees_on_poisoned_arenas(shape):
    """queued + chained launches, clusters, strips with helpers, subtrees: against the run on untouched arenas"""
    import bench
    import kalign_amd
    nseq, length, dna, anchors = shape
    codes, tasks, dist = bench.make_workload(nseq, length, dna, 4)
    subm, scal = bench.scoring(dna)
    out = []
    for hooks in (0, POISON):
        ctx = kalign_amd.Context(0)
        try:
            ctx.debug_set_hooks(hooks)
            recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, dist, n_anchors=anchors, weight=2.0)
            assert ctx.fallback_runs() == 0
            out.append(([(r.plen, r.meet, r.transition, r.score) for r in recs], gaps))
        finally:
            ctx.close()
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
