"""GPU: ka_guide_tree (both distance batches of build_tree_kmeans on the device) reproduces the reference's task
lists, and the whole chain sequences -> guide tree -> task tree -> aligned rows reproduces the reference's output."""
import numpy as np
import pytest

from util import Golden, cons_cases, guide_cases, tree_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", guide_cases())
def test_device_guide_tree_matches_reference(ctx, name):
    g = Golden(name)
    tasks, sd = ctx.guide_tree(g.tree_seqs, n_threads=4, dm_scale=g.dm_scale if hasattr(g, "dm_scale") else None)
    assert np.array_equal(tasks, g.tasks)
    assert np.array_equal(sd.view(np.uint32), g.seq_distances.view(np.uint32))


@pytest.mark.parametrize("name", guide_cases())
def test_device_bisection_matches_reference(ctx, name, monkeypatch):
    """the 2-means bisection on the device (ka_kmeans.hip; KA_KMEANS=1 forces it below its size threshold): the goldens of the
    real reference, task list and seq_distances bit for bit"""
    from kalign_amd import api
    g = Golden(name)
    if len(g.tree_seqs) < 64:
        pytest.skip("fewer than 32 anchors / nothing to bisect")
    monkeypatch.setenv("KA_KMEANS", "1")
    tasks, sd = ctx.guide_tree(g.tree_seqs, n_threads=4, dm_scale=g.dm_scale if hasattr(g, "dm_scale") else None)
    assert api.guide_last_bisect()[1]
    assert np.array_equal(tasks, g.tasks)
    assert np.array_equal(sd.view(np.uint32), g.seq_distances.view(np.uint32))


@pytest.mark.parametrize("nseq,length,dna", [(300, 120, False), (2500, 150, False), (1500, 200, True), (6000, 100, False)])
def test_device_bisection_equals_the_host_one(ctx, nseq, length, dna, monkeypatch):
    """the same tree from the device's candidates-side-by-side bisection as from the host's restatement of the reference
    (which the guide goldens pin to the real build_tree_kmeans): levels with hundreds of sets, lopsided splits, sets that
    stop at the 50-sequence threshold at different depths"""
    from kalign_amd import api, guide, synth
    seqs = synth.dssim(nseq, length, dna=dna, seed=3) if nseq <= 4096 else synth.dssim_fast(nseq, length, dna=dna, seed=3)
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
    tcodes = guide.encode_tree([seqs[i] for i in order], dna=dna)
    monkeypatch.setenv("KA_KMEANS", "0")
    t0, sd0 = ctx.guide_tree(tcodes, n_threads=8)
    assert not api.guide_last_bisect()[1]
    monkeypatch.setenv("KA_KMEANS", "1")
    t1, sd1 = ctx.guide_tree(tcodes, n_threads=8)
    assert api.guide_last_bisect()[1]
    assert np.array_equal(t0, t1)
    assert np.array_equal(sd0.view(np.uint32), sd1.view(np.uint32))


@pytest.mark.parametrize("name", tree_cases() + cons_cases())
def test_sequences_to_rows(ctx, name):
    """kalign_run's alignment phase end to end on the device: guide tree, (consistency,) task tree, final rows"""
    g = Golden(name)
    if len(g.lens) < 2 or g.seq_distances is None:
        pytest.skip("no tree")
    tasks, sd = ctx.guide_tree(g.tree_seqs)
    assert np.array_equal(tasks, g.tasks)
    k = int(g.n_anchors) if hasattr(g, "n_anchors") else 0
    ctx.msa_tree(g.codes, tasks, g.subm, g.scal, sd, n_anchors=k, weight=float(g.weight) if k else 2.0)
    rows = ctx.tree_aligned_rows(g.sorted_seqs())
    got = [None] * len(rows)
    for i, r in enumerate(g.ranks):
        got[int(r)] = rows[i].decode()
    assert got == [str(x) for x in g.rows]


def test_ensemble_members_on_a_context(ctx):
    """kalign_ensemble's member loop (ensemble.c:286-339) through dist.member_on_context: the default member equals
    the golden; members with scaled penalties and a noisy tree equal the real kalign_run_seeded when oracle/_ref
    is there to ask."""
    from kalign_amd import dist as kd
    from oracle import refdrv
    g = Golden("tree_prot32x200")
    n = len(g.lens)
    members = [dict(scal=g.scal.copy())]
    specs = [(0.8, 43, 0.2), (1.25, 44, 0.6)]
    have_ref = refdrv.available()
    for f, seed, sigma in specs:
        s = g.scal.copy()
        s[:3] *= np.float32(f)
        scale = refdrv.noise_multipliers(seed, sigma, n * min(32, n)) if have_ref else np.ones(n * min(32, n), np.float32)
        members.append(dict(scal=s, dm_scale=scale))
    run = kd.member_on_context(ctx, g.tree_seqs, g.codes, g.sorted_seqs(), g.subm)
    rows = kd.ensemble_members(run, members, 0, 1)

    def input_order(r):
        out = [None] * n
        for i, k in enumerate(g.ranks):
            out[int(k)] = r[i].decode()
        return out
    assert input_order(rows[0]) == [str(x) for x in g.rows]
    if not have_ref:
        pytest.skip("oracle/_ref not built: members 1.. unchecked")
    seqs = [str(s) for s in g.seqs]
    for k, (m, (f, seed, sigma)) in enumerate(zip(members[1:], specs), start=1):
        job = refdrv.RefJob(seqs, gpo=float(m["scal"][0]), gpe=float(m["scal"][1]), tgpe=float(m["scal"][2]),
                            tree_seed=seed, tree_noise=sigma)
        job.run_tree()
        assert input_order(rows[k]) == job.finalise(), k
        job.close()
