"""The C multi-GPU layer (ka_dist_*, include/kalign_amd.h) on the one GPU of the box: a rank's subtrees as ONE planned
run, the whole sharded step with a world of one (with and without a real RCCL communicator), and 2 / 4 ranks as threads
of this process over the in-process loopback transport (RCCL refuses two ranks on one device).  Results must not depend
on the number of ranks: the reference's thread-count invariance (lib/src/aln_run.c:95-109)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _job(n=96, length=200, seed=5):
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


def _whole(ctx, codes, tasks, sd, subm, scal, anchors=0):
    ctx.tree_upload(codes, tasks, subm, scal, sd)
    if anchors:
        ctx.tree_build_consistency(anchors, 2.0)
    ctx.tree_run()
    recs, paths, _ = ctx.tree_download(want_gaps=False)
    return [(r.plen, r.meet, r.transition, r.score) for r in recs], [paths[r.path_off:r.path_off + r.plen + 2].copy() for r in recs]


def _same(got_recs, got_paths, want):
    sig, paths = want
    assert [(r.plen, r.meet, r.transition, r.score) for r in got_recs] == sig
    for t, r in enumerate(got_recs):
        assert np.array_equal(got_paths[r.path_off:r.path_off + r.plen + 2], paths[t]), t


def test_planned_subsets_give_the_whole_tree():
    """two subtrees as planned runs (queued / chained launches where they apply), the merges above the cut one by one"""
    from kalign_amd import api
    ctx, codes, tasks, sd, subm, scal = _job(192, 150)
    want = _whole(ctx, codes, tasks, sd, subm, scal)
    run_rank, top = api.dist_plan_subtrees([len(c) for c in codes], tasks, 2)
    ctx.tree_reset()
    for r in (0, 1):
        ids = [t for t in range(len(tasks)) if run_rank[t] == r and t not in set(top)]
        ctx.tree_plan_tasks(ids)
        ctx.tree_run_planned()
        ctx.tree_sync()
    for t in top:
        ctx.tree_run_tasks([t])
    recs, paths = ctx.tree_download_tasks(list(range(len(tasks))))
    _same(recs, paths, want)
    ctx.tree_run()                                          # a whole-tree run plans the whole tree again
    recs2, paths2, _ = ctx.tree_download(want_gaps=False)
    _same(recs2, paths2, want)
    ctx.close()


@pytest.mark.parametrize("anchors", [0, 5])
@pytest.mark.parametrize("rccl", [False, True])
def test_world_of_one(anchors, rccl):
    """the whole C path with one rank: no communicator at all, and a real RCCL communicator of one rank (ncclAllReduce /
    ncclBroadcast execute on HBM buffers)"""
    from kalign_amd import api
    ctx, codes, tasks, sd, subm, scal = _job()
    want = _whole(ctx, codes, tasks, sd, subm, scal, anchors)
    ctx.tree_upload(codes, tasks, subm, scal, sd)
    d = api.Dist(ctx, 0, 1, api.dist_unique_id() if rccl else None)
    d.plan()
    for _ in range(2):                                      # (steps are repeatable)
        if anchors:
            d.consistency(anchors, 2.0)
        d.tree_run()
    recs, paths = d.download()
    _same(recs, paths, want)
    d.close()
    ctx.close()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("anchors", [0, 5])
def test_ranks_as_threads_over_the_loopback(world, anchors):
    """every rank its own context (same GPU), the transport an in-process stand-in: subtrees, hand-overs of profiles
    (and, in default mode, of residue -> column tables) above the cut, the all-reduced records and paths"""
    import kalign_amd
    from kalign_amd import api
    ctx0, codes, tasks, sd, subm, scal = _job(128, 160, seed=9)
    want = _whole(ctx0, codes, tasks, sd, subm, scal, anchors)
    ctx0.close()
    L = api.load_library()
    loop = L.ka_dist_loopback_new(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            ctx = kalign_amd.Context(0, shared=True)        # (several contexts share the GPU: no co-residency assumptions)
            ctx.tree_upload(codes, tasks, subm, scal, sd)
            d = api.Dist(ctx, r, world, loopback=loop)
            d.plan()
            for _ in range(2):
                if anchors:
                    d.consistency(anchors, 2.0)
                d.tree_run()
            out[r] = d.download()
            d.close()
            ctx.close()
        except Exception as e:                              # noqa
            errs.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    L.ka_dist_loopback_free(loop)
    assert not errs, errs
    assert all(o is not None for o in out)
    for recs, paths in out:
        _same(recs, paths, want)


@pytest.mark.parametrize("anchors", [0, 5])
def test_an_overflow_on_one_rank_repeats_the_step_on_all(anchors):
    """rank 1 alone starts with device arenas that are certainly too small (KA_DEBUG_SMALL_ARENAS): its part of the step
    overflows.  The ranks agree on the outcome before the gather's collectives, rank 1 grows its arenas, EVERY rank runs the
    step again, and the result is the whole tree's -- no rank fails, or hangs in a collective, alone."""
    import kalign_amd
    from kalign_amd import api
    world = 2
    ctx0, codes, tasks, sd, subm, scal = _job(128, 160, seed=9)
    want = _whole(ctx0, codes, tasks, sd, subm, scal, anchors)
    ctx0.close()
    L = api.load_library()
    loop = L.ka_dist_loopback_new(world)
    out, errs, retries = [None] * world, [], [0] * world

    def rank_main(r):
        try:
            ctx = kalign_amd.Context(0, shared=True)
            if r == 1:
                ctx.debug_set_hooks(1)                      # KA_DEBUG_SMALL_ARENAS
            ctx.tree_upload(codes, tasks, subm, scal, sd)
            d = api.Dist(ctx, r, world, loopback=loop)
            d.plan()
            if anchors:
                d.consistency(anchors, 2.0)
            d.tree_run()
            retries[r] = d.retries()
            out[r] = d.download()
            d.close()
            ctx.close()
        except Exception as e:                              # noqa
            errs.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    L.ka_dist_loopback_free(loop)
    assert not errs, errs
    assert retries[0] >= 1 and retries[0] == retries[1], retries
    for recs, paths in out:
        _same(recs, paths, want)
