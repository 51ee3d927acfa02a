"""Refinement on the device: ka_tree_refine against the real reference's refine_alignment
(aln_refine.c:34-325) -- goldens made by tests/golden/make_golden.py from the reference itself."""
import numpy as np
import pytest

from util import Golden, cons_cases, refine_cases, tree_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


def run_refine(ctx, g, first_pass=True):
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    if int(g.n_anchors) > 0:
        ctx.tree_build_consistency(int(g.n_anchors), float(g.weight))
    if first_pass:
        ctx.tree_run()
        recs, paths, gaps = ctx.tree_download()
        for got, want in zip(gaps, _split(g.gaps_first, g.lens)):
            assert np.array_equal(got, want)
        # the first pass's confidences are sums in a different order: close to, not identical with, the reference's
        assert np.allclose([r.confidence for r in recs], g.conf_before, rtol=1e-5, atol=1e-6)
    ctx.tree_refine(int(g.mode), g.conf_before)
    return ctx.tree_download()


def _split(flat, lens):
    out, o = [], 0
    for n in lens:
        out.append(flat[o:o + int(n) + 1])
        o += int(n) + 1
    return out


@pytest.mark.parametrize("name", refine_cases())
def test_refinement_matches_reference(ctx, name):
    """every edge re-aligned with the five flip trials of refine_edge, the best sum-of-pairs trial kept: gap arrays,
    profile lengths and the kept trial's confidence equal the real reference's, coded paths (convert_raw_path flags)
    equal the pinned oracle's"""
    g = Golden(name)
    recs, paths, gaps = run_refine(ctx, g)
    for t, r in enumerate(recs):
        assert r.plen == g.plen_after[g.tasks[t][2]], (name, t)
        want = g.paths[int(g.path_off[t]):int(g.path_off[t]) + r.plen + 2]
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], want), (name, t)
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)
    assert int(g.n_differ) > 0 or int(g.mode) == 3      # (the inline trials never beat the baseline on these inputs)
    # mode + 256 = KA_REFINE_ADAPTIVE (--adaptive-budget): 1 .. 8 trials per edge


def test_refine_is_repeatable_and_run_returns_to_the_first_pass(ctx):
    """ka_tree_refine resets the device state like ka_tree_run: refining twice gives the same answer, it does not need
    a preceding ka_tree_run, and a later ka_tree_run is the plain first pass again"""
    g = Golden("refine_prot32x200_all")
    _, _, gaps1 = run_refine(ctx, g, first_pass=False)
    ctx.tree_refine(1)
    _, paths2, gaps2 = ctx.tree_download()
    for a, b, want in zip(gaps1, gaps2, g.gaps_list()):
        assert np.array_equal(a, want) and np.array_equal(b, want)
    ctx.tree_run()
    _, _, gaps3 = ctx.tree_download()
    for got, want in zip(gaps3, _split(g.gaps_first, g.lens)):
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name", [n for n in refine_cases() if n.endswith("_conf")])
def test_confident_mode_computes_its_own_threshold(ctx, name):
    """KALIGN_REFINE_CONFIDENT without confidences from the caller: the first pass is repeated depth first, its task
    confidences are the reference's exact float sums, and the median rule picks the same edges"""
    g = Golden(name)
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    if int(g.n_anchors) > 0:
        ctx.tree_build_consistency(int(g.n_anchors), float(g.weight))
    ctx.tree_refine(2)
    recs, paths, gaps = ctx.tree_download()
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)


@pytest.mark.parametrize("name", tree_cases() + cons_cases())
def test_depth_first_pass_has_exact_confidences(ctx, name):
    """mode 4: the first pass through the depth-first engine -- same paths and gaps as ka_tree_run, and the mean
    meetup margin of every task bit-identical with the reference's (summed in the reference's order)"""
    g = Golden(name)
    cons = hasattr(g, "n_anchors") and int(g.n_anchors) > 0
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    if cons:
        ctx.tree_build_consistency(int(g.n_anchors), float(g.weight))
    ctx.tree_refine(4)
    recs, paths, gaps = ctx.tree_download()
    for t, r in enumerate(recs):
        assert r.plen == g.rec("plen")[t] and r.score == g.rec("score")[t]
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], g.path(t)), (name, t)
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.rec("confidence").astype(np.float32))


@pytest.mark.parametrize("name", ["refine_prot32x200_all", "refine_cons_prot24_conf", "refine_dna16x300_inline"])
def test_serial_trials_give_the_same_answer(ctx, name, monkeypatch):
    """small trees leave CUs idle, so the tests above run the flip trials of an edge side by side on several
    workgroups; KA_REFINE_SERIAL=1 is the one-workgroup-per-edge path big levels take"""
    import os
    monkeypatch.setenv("KA_REFINE_SERIAL", "1")
    assert os.environ["KA_REFINE_SERIAL"] == "1"
    ctx.reload_env()                                   # (the environment is read once per context)
    g = Golden(name)
    try:
        recs, paths, gaps = run_refine(ctx, g, first_pass=False)
    finally:
        monkeypatch.delenv("KA_REFINE_SERIAL")
        ctx.reload_env()
    for t, r in enumerate(recs):
        want = g.paths[int(g.path_off[t]):int(g.path_off[t]) + r.plen + 2]
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], want), (name, t)
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)


@pytest.mark.parametrize("switches", [{"KA_NO_INC": "1"}, {"KA_NO_LDFS": "1"}, {"KA_NO_INC": "1", "KA_NO_LDFS": "1"},
                                      {"KA_NO_WDFS": "1"}, {"KA_NO_LS0": "1"}, {"KA_NO_INC": "1", "KA_REFINE_SERIAL": "1"},
                                      {"KA_REFINE_SERIAL": "1"}])
@pytest.mark.parametrize("name", refine_cases())
def test_every_schedule_of_the_flip_trials_gives_the_reference_answer(name, switches, monkeypatch):
    """the flip trials have four engines stacked on each other -- the incremental walk over the baseline's uncertain
    meetups (re-running only the subtrees it flips), the depth-first recursion across the workgroup, its wave-local
    subtrees with operands in HBM, and the same with everything in LDS -- and the baseline two (level-synchronous with
    sorted margins, depth first): every golden again with each layer switched off in turn"""
    import kalign_amd
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    ctx = kalign_amd.Context(0)                        # (reads the environment)
    try:
        g = Golden(name)
        recs, paths, gaps = run_refine(ctx, g, first_pass=False)
        for t, r in enumerate(recs):
            want = g.paths[int(g.path_off[t]):int(g.path_off[t]) + r.plen + 2]
            assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], want), (name, t, switches)
        for got, want in zip(gaps, g.gaps_list()):
            assert np.array_equal(got, want), (name, switches)
        assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after), (name, switches)
    finally:
        ctx.close()


def test_big_tree_parallel_and_serial_trials_agree(ctx, monkeypatch):
    """512 x 300: the lower levels have more edges than CUs (one workgroup per edge), the upper ones run their trials in
    parallel; forcing everything serial must not change a single gap"""
    from kalign_amd import guide, synth
    import bench
    seqs = synth.dssim(512, 300, seed=4)
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
    seqs = [seqs[i] for i in order]
    codes = guide.encode(seqs, dna=False)
    tasks, sd = ctx.guide_tree(guide.encode_tree(seqs, dna=False), n_threads=4)
    subm, scal = bench.scoring(False)
    ctx.tree_upload(codes, tasks, subm, scal, sd)
    ctx.tree_refine(1)
    recs1, paths1, gaps1 = ctx.tree_download()
    monkeypatch.setenv("KA_REFINE_SERIAL", "1")
    ctx.reload_env()
    ctx.tree_refine(1)
    recs2, paths2, gaps2 = ctx.tree_download()
    monkeypatch.delenv("KA_REFINE_SERIAL")
    ctx.reload_env()
    assert all(np.array_equal(a, b) for a, b in zip(gaps1, gaps2))
    assert [(r.plen, r.confidence, r.meet, r.score) for r in recs1] == [(r.plen, r.confidence, r.meet, r.score) for r in recs2]
    ctx.tree_run()
    _, _, gaps0 = ctx.tree_download()
    assert any(not np.array_equal(a, b) for a, b in zip(gaps0, gaps1))


def test_refine_rows_match_reference(ctx):
    """the aligned rows written from the refined gap arrays are the reference's output rows"""
    g = Golden("refine_cons_prot48_all")
    run_refine(ctx, g)
    rows_sorted = ctx.tree_aligned_rows(g.sorted_seqs())
    rows = [None] * len(rows_sorted)
    for i, r in enumerate(g.ranks):
        rows[int(r)] = rows_sorted[i].decode()
    assert rows == [str(x) for x in g.rows]


def test_refine_argument_errors(ctx):
    from kalign_amd.api import KalignAmdError
    g = Golden("refine_BB11001_all")
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    with pytest.raises(KalignAmdError):
        ctx.tree_refine(5)
    with pytest.raises(KalignAmdError):
        ctx.tree_refine(0)
    with pytest.raises(KalignAmdError):
        ctx.tree_refine(2, g.conf_before[:-1])


@pytest.mark.parametrize("mode,anchors,nseq,length,dna", [(1, 0, 768, 300, False), (2, 5, 384, 300, False), (3, 0, 384, 250, False),
                                                          (1, 5, 256, 600, True), (1, 8, 192, 250, False)],
                         ids=["all_768x300", "confident_cons_384x300", "inline_384x250", "all_cons_dna256x600", "all_cons8_192x250"])
def test_refinement_against_the_live_reference(ctx, mode, anchors, nseq, length, dna):
    """bigger trees than the goldens (levels with more edges than CUs: serial trials; upper levels: trials in parallel),
    against refine_alignment / create_msa_tree_inline_refine of the real reference run on the box's host cores: gap arrays,
    profile lengths and the fp32 task confidences must be identical"""
    from kalign_amd import synth
    from oracle import refdrv
    assert refdrv.available(), "oracle/_ref missing"
    seqs = synth.dssim(nseq, length, dna=dna, seed=9)
    job = refdrv.RefJob(seqs, n_threads=16, **({"type_": 0} if dna else {}))
    if anchors:
        job.build_consistency(anchors, 2.0)
    job.run_tree()
    want_gaps, conf_before, conf_after, plen = job.refine(mode)
    scal = np.array([job.gpo, job.gpe, job.tgpe, job.dist_scale, job.vsm_amax, job.use_seq_weights], np.float32)
    ctx.tree_upload(job.codes, job.tasks, job.subm, scal, job.seq_distances)
    if anchors:
        ctx.tree_build_consistency(anchors, 2.0)
    ctx.tree_refine(mode)                                         # (mode 2: the device computes the threshold itself)
    recs, paths, gaps = ctx.tree_download()
    job.close()
    for got, want in zip(gaps, want_gaps):
        assert np.array_equal(got, want)
    assert all(recs[t].plen == plen[job.tasks[t][2]] for t in range(len(recs)))
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), conf_after)


def test_refine_survives_arena_growth(ctx):
    """KA_DEBUG_SMALL_ARENAS: the arenas overflow, ka_tree_sync grows them and repeats the refinement pass (not the
    first pass) -- same answer"""
    g = Golden("refine_cons_prot48_all")
    ctx.debug_set_hooks(1)
    try:
        recs, paths, gaps = run_refine(ctx, g, first_pass=False)
    finally:
        ctx.debug_set_hooks(0)
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)


def test_refine_of_a_forest_equals_separate_jobs(ctx):
    """two independent alignments as one forest job (n_tasks < numseq - 1, two roots): refining the forest gives each
    tree the result of its own golden"""
    from kalign_amd import guide
    ga, gb = Golden("refine_prot32x200_all"), Golden("refine_ragged_all")
    assert np.array_equal(ga.subm, gb.subm) and np.array_equal(ga.scal, gb.scal)
    codes, tasks, dist, spans = guide.forest([(ga.codes, ga.tasks, ga.seq_distances), (gb.codes, gb.tasks, gb.seq_distances)])
    ctx.tree_upload(codes, tasks, ga.subm, ga.scal, dist)
    ctx.tree_refine(1)
    recs, paths, gaps = ctx.tree_download()
    for g, (s0, t0, ns, nt) in zip((ga, gb), spans):
        for got, want in zip(gaps[s0:s0 + ns], g.gaps_list()):
            assert np.array_equal(got, want)
        assert np.array_equal(np.array([r.confidence for r in recs[t0:t0 + nt]], np.float32), g.conf_after)


def test_confident_marks_survive_a_watchdog_fallback():
    """KALIGN_REFINE_CONFIDENT with a member of a multi-workgroup edge that never starts (what a non-resident workgroup
    looks like from the device): the member barrier's watchdog reports it, ka_tree_sync re-plans with one workgroup per
    edge and repeats the pass -- and the repeated pass must still refine exactly the edges at or below the median
    confidence (the marks are not part of the launch plan)."""
    import kalign_amd
    g = Golden("refine_cons_prot24_conf")
    c = kalign_amd.Context(0)
    try:
        c.debug_set_hooks(4)                                 # KA_DEBUG_STARVE_REFINE_MEMBER
        recs, paths, gaps = run_refine(c, g, first_pass=False)
        assert c.fallback_runs() == 1
    finally:
        c.debug_set_hooks(0)
        c.close()
    for t, r in enumerate(recs):
        want = g.paths[int(g.path_off[t]):int(g.path_off[t]) + r.plen + 2]
        assert np.array_equal(paths[r.path_off:r.path_off + r.plen + 2], want), t
    for got, want in zip(gaps, g.gaps_list()):
        assert np.array_equal(got, want)
    assert np.array_equal(np.array([r.confidence for r in recs], np.float32), g.conf_after)
    assert int(g.n_differ) > 0
