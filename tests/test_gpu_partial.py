"""The partial-run C ABI that single-tree multi-GPU sharding is built from (ka_tree_run_tasks,
ka_tree_set_profile, ka_tree_download_tasks, ka_weave_gaps): two contexts on one GPU play two ranks
through kalign_amd.dist.sharded_tree's plan; results must equal the whole-tree goldens bit for bit."""
import numpy as np
import pytest

from util import Golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,world", [("tree_prot32x200", 2), ("tree_prot64_gon", 4), ("tree_dna16x300", 3),
                                        ("cons_prot32x200", 2), ("cons_BB30014", 4), ("cons_prot48_k3", 3)])
def test_subtrees_on_separate_contexts_match_golden(case, world):
    """cons_* cases: default mode -- every context builds the same consistency table and the residue->column
    state of a subtree travels with its root profile."""
    import kalign_amd
    from kalign_amd import api, dist as kd
    g = Golden(case)
    ctxs = [kalign_amd.Context(0) for _ in range(world)]
    for c in ctxs:
        c.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
        if case.startswith("cons_"):
            c.tree_build_consistency(int(g.n_anchors), float(g.weight))
    run_rank, top = kd.plan_subtrees(g.tasks, g.lens, world)
    top_set = set(top)
    n = len(g.lens)
    for r, c in enumerate(ctxs):
        mine = [t for t in range(len(g.tasks)) if run_rank[t] == r and t not in top_set]
        if mine:
            c.tree_run_tasks(mine)
    holder = {int(g.tasks[t][2]): int(run_rank[t]) for t in range(len(g.tasks)) if t not in top_set}
    for t in top:
        a, b, cnode = (int(v) for v in g.tasks[t])
        dst = int(run_rank[t])
        for child in (a, b):
            if child >= n and holder[child] != dst:
                ctxs[dst].tree_set_node(child, ctxs[holder[child]].tree_get_node(child))
        ctxs[dst].tree_run_tasks([t])
        holder[cnode] = dst
    recs_all = [None] * len(g.tasks)
    chunks, off = [], 0
    for r, c in enumerate(ctxs):
        ran = [t for t in range(len(g.tasks)) if run_rank[t] == r]
        if not ran:
            continue
        recs, paths = c.tree_download_tasks(ran)
        for t, rec in zip(ran, recs):
            rec.path_off += off
            recs_all[t] = rec
        chunks.append(paths)
        off += len(paths)
    paths = np.concatenate(chunks)
    for t, r in enumerate(recs_all):
        assert r.plen == g.rec("plen")[t] and r.score == g.rec("score")[t] and r.meet == g.rec("meet")[t]
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], g.path(t)), t
    gaps = api.weave_gaps(g.lens, recs_all, paths)
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    # misuse fails loudly: running a task twice, or a task whose operand is not on this context
    with pytest.raises(kalign_amd.KalignAmdError):
        ctxs[int(run_rank[0])].tree_run_tasks([0])
    if world > 1:
        other = [c for r, c in enumerate(ctxs) if r != int(run_rank[top[-1]])][0]
        with pytest.raises(kalign_amd.KalignAmdError):
            other.tree_run_tasks([top[-1]])
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("case,world", [("tree_prot64_gon", 4), ("cons_BB30014", 3), ("tree_dna16x300", 2)])
def test_device_to_device_hand_over(case, world):
    """The device-pointer ABI of the RCCL path (ka_tree_profile_dev / ka_tree_reserve_profile_dev, wrapped as torch
    tensors by kalign_amd.dist.dev_tensor): subtree roots move HBM to HBM between contexts -- here with a device-side
    copy between two contexts on one GPU, over xGMI with RCCL send / recv when every context has its own GPU -- and,
    in default mode, the consistency table is assembled from the ranks' parts (ka_tree_build_consistency_part) by
    device-side copies of the part ranges.  Results equal the whole-tree goldens."""
    import torch
    import kalign_amd
    from kalign_amd import dist as kd
    torch.cuda.init()
    g = Golden(case)
    cons = case.startswith("cons_")
    ctxs = [kalign_amd.Context(0) for _ in range(world)]
    for r, c in enumerate(ctxs):
        c.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
        if cons:
            c.tree_build_consistency_part(int(g.n_anchors), float(g.weight), r, world)
    if cons:
        tables = [c.cons_table() for c in ctxs]
        for r in range(world):                                  # what dist.sharded_consistency's broadcasts do
            lo, hi = ctxs[r].cons_part_range(r, world)
            for q in range(world):
                assert ctxs[q].cons_part_range(r, world) == (lo, hi)
                if q != r and hi > lo:
                    tables[q][lo:hi].copy_(tables[r][lo:hi])
        torch.cuda.synchronize()
        want = np.concatenate([np.asarray(m, np.int32) for row in g.maps_list() for m in row])
        for t in tables:
            assert np.array_equal(t.cpu().numpy(), want)
    run_rank, top = kd.plan_subtrees(g.tasks, g.lens, world)
    top_set = set(top)
    n = len(g.lens)
    for r, c in enumerate(ctxs):
        mine = [t for t in range(len(g.tasks)) if run_rank[t] == r and t not in top_set]
        if mine:
            c.tree_run_tasks(mine)
    holder = {int(g.tasks[t][2]): int(run_rank[t]) for t in range(len(g.tasks)) if t not in top_set}
    moved = 0
    for t in top:
        a, b, cnode = (int(v) for v in g.tasks[t])
        dst = int(run_rank[t])
        for child in (a, b):
            if child >= n and holder[child] != dst:
                src = ctxs[holder[child]]
                ptr, plen = src.tree_profile_dev(child)
                dptr = ctxs[dst].tree_reserve_profile_dev(child, plen)
                kd.dev_tensor(dptr, (plen + 2) * 64, torch.float32).copy_(kd.dev_tensor(ptr, (plen + 2) * 64, torch.float32))
                torch.cuda.synchronize()
                cols = src.tree_node_cols(child)
                if cols is not None:
                    ctxs[dst].tree_set_node_cols(child, cols)
                moved += 1
        ctxs[dst].tree_run_tasks([t])
        holder[cnode] = dst
    assert moved >= world - 1
    for r, c in enumerate(ctxs):
        ran = [t for t in range(len(g.tasks)) if run_rank[t] == r]
        recs, paths = c.tree_download_tasks(ran)
        for t, rec in zip(ran, recs):
            assert rec.plen == g.rec("plen")[t] and rec.score == g.rec("score")[t]
            assert np.array_equal(paths[rec.path_off:rec.path_off + rec.plen + 2], g.path(t)), t
        c.close()


def test_shared_contexts_run_concurrently_and_match_golden():
    """ka_ctx_set_shared: several alignments in flight on one GPU (separate streams).  Shared contexts use neither
    multi-workgroup tasks nor the chained launch, so no workgroup ever waits for one that is not resident."""
    import torch
    import kalign_amd
    g = Golden("tree_prot64_gon")
    streams = [torch.cuda.Stream() for _ in range(4)]
    ctxs = [kalign_amd.Context(0, stream=s.cuda_stream, shared=True) for s in streams]
    for c in ctxs:
        c.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    for _ in range(2):
        for c in ctxs:
            c.tree_run()
    for c in ctxs:
        recs, paths, gaps = c.tree_download()
        for t, r in enumerate(recs):
            assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], g.path(t)), t
        for got, want in zip(gaps, g.gaps_list()):
            assert np.array_equal(got, want)
        c.close()


def test_forest_of_independent_alignments_matches_goldens():
    """n_tasks < numseq-1: a batch of independent alignments as one job (levels of all trees share launches, the
    upper parts run in one chained launch); every tree's results are those of its own single-tree run."""
    import kalign_amd
    from kalign_amd import guide
    names = ["tree_prot32x200", "tree_prot64_gon", "tree_prot32x200"]
    gs = [Golden(n) for n in names]
    # one scoring scheme per job: these three goldens share PFASUM defaults? use each golden's own parameters where equal
    assert all(np.array_equal(g.subm, gs[0].subm) for g in gs[:1])
    jobs = [(g.codes, g.tasks, g.seq_distances) for g in gs if np.array_equal(g.subm, gs[0].subm) and np.array_equal(g.scal, gs[0].scal)]
    gs = [g for g in gs if np.array_equal(g.subm, gs[0].subm) and np.array_equal(g.scal, gs[0].scal)]
    assert len(jobs) >= 2
    codes, tasks, dist, spans = guide.forest(jobs)
    ctx = kalign_amd.Context(0)
    recs, paths, gaps = ctx.msa_tree(codes, tasks, gs[0].subm, gs[0].scal, dist)
    ctx.close()
    for g, (s0, t0, n, nt) in zip(gs, spans):
        for t in range(nt):
            r = recs[t0 + t]
            assert r.plen == g.rec("plen")[t] and r.score == g.rec("score")[t]
            assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], g.path(t)), t
        for got, want in zip(gaps[s0:s0 + n], g.gaps_list()):
            assert np.array_equal(got, want)


def test_unshared_contexts_in_flight_together_fall_back():
    """Two contexts that did NOT declare the GPU shared run their chained launches at the same time: together they
    want more resident workgroups than there are CUs.  Whether the joins of one of them starve depends on how the
    dispatcher interleaves the two kernels; if they do, the bounded waits report it and ka_tree_sync re-plans
    without joins / clusters (what ka_ctx_set_shared would have chosen) and runs again.  Either way: the
    reference's result, never a hang."""
    import torch
    import bench
    import kalign_amd
    codes, tasks, dist = bench.make_workload(1024, 400, False, 1)
    subm, scal = bench.scoring(False)
    ref = kalign_amd.Context(0)
    want_recs, want_paths, want_gaps = ref.msa_tree(codes, tasks, subm, scal, dist)
    ref.close()
    streams = [torch.cuda.Stream() for _ in range(2)]
    ctxs = [kalign_amd.Context(0, stream=s.cuda_stream) for s in streams]
    for c in ctxs:
        c.tree_upload(codes, tasks, subm, scal, dist)
    for _ in range(3):
        for c in ctxs:
            c.tree_run()
    for c in ctxs:
        recs, paths, gaps = c.tree_download()
        assert all(r.plen == w.plen for r, w in zip(recs, want_recs))
        for got, want in zip(gaps, want_gaps):
            assert np.array_equal(got, want)
        c.close()
