"""The guide-tree builder (ka_guide_tree_from = build_tree_kmeans, bisectingKmeans.c:177-271) against the task lists
the REAL reference produced (tests/golden/guide_*.npz, and the trees inside every tree_/cons_ golden).  CPU only: the
host-side clustering is product code; the distances come from the oracle's bpm_block restatement here and from the
device in tests/test_gpu_guide.py."""
import numpy as np
import pytest

from util import Golden, cons_cases, guide_cases, tree_cases


def oracle_dist(oracle, seqs):
    return lambda ia, ib: oracle.bpm_batch(seqs, ia, ib)


@pytest.mark.parametrize("name", guide_cases())
@pytest.mark.parametrize("n_threads", [1, 4])
def test_guide_tree_matches_reference(oracle, name, n_threads):
    from kalign_amd import api
    g = Golden(name)
    # guide_noisy_*: build_tree_kmeans_noisy, the multipliers drawn by the reference's generator are part of the golden
    scale = g.dm_scale if hasattr(g, "dm_scale") else None
    tasks, sd = api.guide_tree_from(g.lens, oracle_dist(oracle, g.tree_seqs), n_threads=n_threads, dm_scale=scale)
    assert np.array_equal(tasks, g.tasks)
    assert np.array_equal(sd.view(np.uint32), g.seq_distances.view(np.uint32))       # bit for bit


@pytest.mark.parametrize("name", tree_cases() + cons_cases())
def test_trees_of_the_alignment_goldens(oracle, name):
    """the same for the guide trees the alignment goldens were run on (BAliBASE families, DNA, RNA, ragged lengths)"""
    from kalign_amd import api
    g = Golden(name)
    if len(g.lens) < 2 or g.seq_distances is None:
        pytest.skip("no tree")
    tasks, sd = api.guide_tree_from(g.lens, oracle_dist(oracle, g.tree_seqs))
    assert np.array_equal(tasks, g.tasks)
    assert np.array_equal(sd.view(np.uint32), g.seq_distances.view(np.uint32))


def test_live_against_the_reference_when_it_is_built(oracle):
    """random families through the real build_tree_kmeans (oracle/_ref) and through ours"""
    from oracle import refdrv
    from kalign_amd import api, synth
    if not refdrv.available():
        pytest.skip("oracle/_ref not built")
    for n, length, dna, seed, noise in ((57, 70, False, 101, 0.0), (140, 50, True, 102, 0.0), (260, 40, False, 103, 0.0),
                                        (180, 60, False, 104, 0.3), (75, 90, True, 105, 0.8)):
        job = refdrv.RefJob(synth.family(n, length, dna=dna, seed=seed), tree_seed=seed if noise else 0, tree_noise=noise)
        scale = refdrv.noise_multipliers(seed, noise, n * min(32, n)) if noise else None
        tasks, sd = api.guide_tree_from(job.lens, oracle_dist(oracle, job.tree_codes), n_threads=2, dm_scale=scale)
        assert np.array_equal(tasks, job.tasks), (n, length, dna, noise)
        assert np.array_equal(sd, job.seq_distances)
        job.close()


def test_guide_tree_error_behaviour():
    import kalign_amd
    from kalign_amd import api
    with pytest.raises(kalign_amd.KalignAmdError, match="bad arguments"):
        api.guide_tree_from(np.array([5], np.int32), lambda ia, ib: np.zeros(len(ia), np.int32))
    with pytest.raises(kalign_amd.KalignAmdError, match="zero-length"):
        api.guide_tree_from(np.array([5, 0, 3], np.int32), lambda ia, ib: np.zeros(len(ia), np.int32))

    def broken(ia, ib):
        raise RuntimeError("no distances today")
    with pytest.raises(kalign_amd.KalignAmdError, match="distance source failed"):
        api.guide_tree_from(np.array([5, 4, 3], np.int32), broken)
    with pytest.raises(kalign_amd.KalignAmdError, match="multipliers"):
        api.guide_tree_from(np.array([5, 4, 3], np.int32), lambda ia, ib: np.zeros(len(ia), np.int32), dm_scale=np.ones(4))


@pytest.mark.parametrize("name", tree_cases() + cons_cases())
def test_tree_alphabet_classes(name):
    """guide.encode_tree puts two letters into one class exactly when the reference's tree alphabet does"""
    from kalign_amd import guide
    g = Golden(name)
    dna = int(g.biotype) != 0 if hasattr(g, "biotype") else False
    ours = guide.encode_tree(g.sorted_seqs(), dna=dna)
    a = np.concatenate(ours).astype(np.int64)
    b = np.concatenate(g.tree_seqs).astype(np.int64)
    assert len(a) == len(b)
    pairs = set(zip(a.tolist(), b.tolist()))
    assert len(pairs) == len(set(p[0] for p in pairs)) == len(set(p[1] for p in pairs))       # a bijection of classes


def test_live_edge_shapes(oracle):
    """shapes around the builder's thresholds and degenerate inputs, against the real build_tree_kmeans: exactly 50 and
    51 sequences (UPGMA below 50), 128 (parallel candidate splits), many identical sequences (tied distances, splits
    decided by the index parity rule), two tight clusters (a 2-means that converges at once), one long outlier"""
    from oracle import refdrv
    from kalign_amd import api, synth
    if not refdrv.available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.RandomState(9)
    fam = synth.family(60, 80, seed=31)
    a, b = synth.family(45, 60, seed=32), synth.family(45, 140, seed=33)
    cases = {
        "n50": synth.family(50, 60, seed=34), "n51": synth.family(51, 60, seed=35), "n128": synth.family(128, 40, seed=36),
        "identical": [fam[0]] * 70 + fam[:10],
        "few_distinct": [fam[i % 3] for i in range(90)],
        "two_clusters": a + b,
        "outlier": synth.family(80, 50, seed=37) + ["".join("ACDEFGHIKLMNPQRSTVWY"[k] for k in rng.randint(0, 20, size=900))],
    }
    for name, seqs in cases.items():
        job = refdrv.RefJob(seqs)
        for nt in (1, 3):
            tasks, sd = api.guide_tree_from(job.lens, oracle_dist(oracle, job.tree_codes), n_threads=nt)
            assert np.array_equal(tasks, job.tasks), (name, nt)
            assert np.array_equal(sd, job.seq_distances), (name, nt)
        job.close()
