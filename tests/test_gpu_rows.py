"""GPU parity of ka_tree_aligned_rows (finalise_alignment + make_linear_sequence, msa_op.c:546-598, built on the
device from the residue->column tables) against the aligned rows the real reference wrote into the goldens."""
import numpy as np
import pytest

from util import Golden, cons_cases, tree_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


def in_input_order(g, rows_sorted):
    rows = [None] * len(rows_sorted)
    for i, r in enumerate(g.ranks):
        rows[int(r)] = rows_sorted[i].decode()
    return rows


@pytest.mark.parametrize("name", tree_cases())
def test_rows_match_reference_golden(ctx, name):
    g = Golden(name)
    ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)
    rows = ctx.tree_aligned_rows(g.sorted_seqs())
    assert in_input_order(g, rows) == [str(x) for x in g.rows]


@pytest.mark.parametrize("name", cons_cases())
def test_rows_in_default_mode(ctx, name):
    g = Golden(name)
    ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, n_anchors=int(g.n_anchors), weight=float(g.weight))
    rows = ctx.tree_aligned_rows(g.sorted_seqs())
    assert in_input_order(g, rows) == [str(x) for x in g.rows]


def test_rows_of_a_forest_and_other_gap_characters(ctx):
    """Every alignment of a forest job gets rows of its own length; a sequence in no task is its own row."""
    from kalign_amd import guide
    a, b = Golden("tree_prot32x200"), Golden("tree_BB11001")
    lone = a.codes[0][:17]
    codes, tasks, dist, spans = guide.forest([(a.codes, a.tasks, a.seq_distances), (b.codes, b.tasks, b.seq_distances),
                                              ([lone], np.zeros((0, 3), np.int32), np.zeros(1, np.float32))])
    letters = a.sorted_seqs() + b.sorted_seqs() + ["W" * len(lone)]
    # the goldens were made with their own parameter sets: run the forest with each and compare that part
    for g, (s0, t0, ns, nt) in ((a, spans[0]), (b, spans[1])):
        ctx.msa_tree(codes, tasks, g.subm, g.scal, dist)
        rows = ctx.tree_aligned_rows(letters, gap=b".")
        assert in_input_order(g, rows[s0:s0 + ns]) == [str(x).replace("-", ".") for x in g.rows]
        assert rows[-1] == b"W" * len(lone)


def test_rows_error_behaviour(ctx):
    import kalign_amd
    from kalign_amd import api
    g = Golden("tree_dna4")
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances)          # no KA_FLAG_DEVICE_GAPS
    ctx.tree_run()
    with pytest.raises(kalign_amd.KalignAmdError, match="DEVICE_GAPS"):
        ctx.tree_aligned_rows(g.sorted_seqs())
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEVICE_GAPS)
    with pytest.raises(kalign_amd.KalignAmdError, match="no finished run"):
        ctx.tree_aligned_rows(g.sorted_seqs())
    ctx.tree_run()
    with pytest.raises(kalign_amd.KalignAmdError, match="do not match"):
        ctx.tree_aligned_rows(g.sorted_seqs()[:-1])
    assert in_input_order(g, ctx.tree_aligned_rows(g.sorted_seqs())) == [str(x) for x in g.rows]
