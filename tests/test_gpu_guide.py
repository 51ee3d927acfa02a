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
    tasks, sd = ctx.guide_tree(g.tree_seqs, n_threads=4)
    assert np.array_equal(tasks, g.tasks)
    assert np.array_equal(sd.view(np.uint32), g.seq_distances.view(np.uint32))


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
