"""ka_multi_* (kalign_amd/csrc/ka_multi.cpp): the GPUs of one node under ONE caller -- what the drop-in glue uses for
create_msa_tree / anchor_consistency_build when it sees more than one device.  On the one GPU of the test box the ranks are
threads on device 0 over the library's in-process transport (loopback); records, coded paths and gap arrays must be the
single-GPU ones bit for bit whatever the number of ranks (lib/src/aln_run.c:95-109; tests/dssim_test.c:41-86)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _job(n=128, length=160, seed=9):
    import bench
    import kalign_amd
    from kalign_amd import guide, synth
    seqs = synth.dssim(n, length, seed=seed)
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
    seqs = [seqs[i] for i in order]
    codes = guide.encode(seqs, dna=False)
    ctx = kalign_amd.Context(0)
    tasks, sd = ctx.guide_tree(guide.encode_tree(seqs, dna=False), n_threads=4)
    subm, scal = bench.scoring(False)
    return ctx, codes, tasks, sd, subm, scal


def _single(ctx, codes, tasks, sd, subm, scal, anchors):
    ctx.tree_upload(codes, tasks, subm, scal, sd)
    maps = None
    if anchors:
        ctx.tree_build_consistency(anchors, 2.0)
        maps = ctx.tree_consistency()
    ctx.tree_run()
    recs, paths, gaps = ctx.tree_download(want_gaps=True)
    return recs, paths, gaps, maps


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("anchors", [0, 5])
def test_one_caller_several_ranks(world, anchors):
    from kalign_amd import api
    ctx, codes, tasks, sd, subm, scal = _job()
    recs0, paths0, gaps0, maps0 = _single(ctx, codes, tasks, sd, subm, scal, anchors)
    ctx.close()
    m = api.Multi(world, loopback=True)
    try:
        keep = False
        if anchors:
            # the drop-in's order: anchor_consistency_build first (a pairing task list), then create_msa_tree keeps the table
            ids, maps = m.consistency(codes, tasks, subm, scal, sd, anchors, 2.0)
            assert np.array_equal(ids, maps0[0])
            assert np.array_equal(maps, np.concatenate([mk for row in maps0[1] for mk in row]))
            keep = True
        for rep in range(2):                                # (repeatable; the second run re-uses the table the ranks hold)
            m.tree_run(codes, tasks, subm, scal, sd, n_anchors=anchors, weight=2.0, keep_consistency=keep)
            recs, paths, gaps = m.download()
            assert [(r.plen, r.meet, r.transition, r.score) for r in recs] == [(r.plen, r.meet, r.transition, r.score) for r in recs0]
            for r, r0 in zip(recs, recs0):
                assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], paths0[r0.path_off:r0.path_off + r0.plen + 2])
            for g, g0 in zip(gaps, gaps0):
                assert np.array_equal(g, g0)
        assert m.runs() == 2
    finally:
        m.close()


def test_world_of_one_is_the_plain_run():
    from kalign_amd import api
    ctx, codes, tasks, sd, subm, scal = _job(96, 200, seed=5)
    recs0, paths0, gaps0, _ = _single(ctx, codes, tasks, sd, subm, scal, 0)
    ctx.close()
    m = api.Multi(1)
    try:
        m.tree_run(codes, tasks, subm, scal, sd)
        recs, paths, gaps = m.download()
        for g, g0 in zip(gaps, gaps0):
            assert np.array_equal(g, g0)
    finally:
        m.close()
