"""The sharded path over REAL RCCL between several devices (subtrees per GPU, profiles sent / received above the cut
with ncclSend / ncclRecv, position maps broadcast, records and paths all-reduced) -- the one thing the one-GPU boxes of the pool
cannot run: there RCCL only ever sees a world of one (tests/test_gpu_dist_c.py::test_world_of_one) and the ranks of tests/
test_gpu_dist_c.py / test_gpu_multi.py meet over the library's in-process transport.  These tests SKIP unless the box has at
least two devices; on a node they run the same comparisons against the single-GPU answer, which must not depend on the number of
ranks (lib/src/aln_run.c:95-109; the reference's thread-count invariance, tests/dssim_test.c:41-86)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


needs_two = pytest.mark.skipif(_devices() < 2, reason="needs at least two GPUs (RCCL refuses two ranks on one device)")


def _job(n=160, length=180, seed=11):
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


def _whole(ctx, codes, tasks, sd, subm, scal, anchors):
    ctx.tree_upload(codes, tasks, subm, scal, sd)
    if anchors:
        ctx.tree_build_consistency(anchors, 2.0)
    ctx.tree_run()
    recs, paths, gaps = ctx.tree_download(want_gaps=True)
    return recs, paths, gaps


def _same(recs, paths, recs0, paths0):
    assert [(r.plen, r.meet, r.transition, r.score) for r in recs] == [(r.plen, r.meet, r.transition, r.score) for r in recs0]
    for r, r0 in zip(recs, recs0):
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], paths0[r0.path_off:r0.path_off + r0.plen + 2])


@needs_two
@pytest.mark.parametrize("anchors", [0, 5])
def test_ranks_on_their_own_devices_over_rccl(anchors):
    """ka_dist_* with one rank per device, the ranks threads of this process (ncclCommInitRank from every thread at once)"""
    import kalign_amd
    from kalign_amd import api
    world = min(_devices(), 4)
    ctx0, codes, tasks, sd, subm, scal = _job()
    recs0, paths0, _ = _whole(ctx0, codes, tasks, sd, subm, scal, anchors)
    ctx0.close()
    uid = api.dist_unique_id()
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            ctx = kalign_amd.Context(r)
            ctx.tree_upload(codes, tasks, subm, scal, sd)
            d = api.Dist(ctx, r, world, uid)
            d.plan()
            for _ in range(2):                              # (steps are repeatable)
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
    [t.join(600) for t in th]
    assert not errs, errs
    for r in range(world):                                  # every rank holds the whole answer
        _same(out[r][0], out[r][1], recs0, paths0)


@needs_two
@pytest.mark.parametrize("anchors", [0, 5])
def test_one_caller_several_devices(anchors):
    """ka_multi_*: what the drop-in glue calls when it sees more than one device"""
    from kalign_amd import api
    world = min(_devices(), 4)
    ctx0, codes, tasks, sd, subm, scal = _job(seed=12)
    recs0, paths0, gaps0 = _whole(ctx0, codes, tasks, sd, subm, scal, anchors)
    ctx0.close()
    m = api.Multi(world)
    try:
        keep = False
        if anchors:
            m.consistency(codes, tasks, subm, scal, sd, anchors, 2.0)
            keep = True
        m.tree_run(codes, tasks, subm, scal, sd, n_anchors=anchors, weight=2.0, keep_consistency=keep)
        recs, paths, gaps = m.download()
        _same(recs, paths, recs0, paths0)
        for g, g0 in zip(gaps, gaps0):
            assert np.array_equal(g, g0)
    finally:
        m.close()


def test_rccl_is_resolved_once():
    """a process that already maps an RCCL (PyTorch brings its own) must not get a second one: the C layer takes the symbols that
    are there (ka_dist.cpp:rccl_load); a communicator of one rank runs on it"""
    import kalign_amd
    from kalign_amd import api
    ctx, codes, tasks, sd, subm, scal = _job(48, 120, seed=3)
    ctx.tree_upload(codes, tasks, subm, scal, sd)
    d = api.Dist(ctx, 0, 1, api.dist_unique_id())
    d.plan()
    d.tree_run()
    recs, paths = d.download()
    assert len(recs) == len(tasks)
    d.close()
    ctx.close()
    with open("/proc/self/maps") as f:
        libs = {ln.split()[-1] for ln in f if "librccl" in ln}
    assert len(libs) <= 1, libs
