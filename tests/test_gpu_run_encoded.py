"""GPU: ka_run_encoded -- kalign_run_seeded / kalign_run_realign from "sequences encoded" to "rows finalised" as one
call -- against the rows the real reference produced for the tree, consistency and realignment goldens."""
import os

import numpy as np
import pytest

from util import GOLDEN, Golden, cons_cases, refine_cases, tree_cases

pytestmark = pytest.mark.gpu

REALIGN = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("realign_") and f.endswith(".npz"))


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


def input_order(ranks, rows):
    out = [None] * len(rows)
    for i, r in enumerate(ranks):
        out[int(r)] = rows[i].decode()
    return out


@pytest.mark.parametrize("name", tree_cases() + cons_cases())
def test_run_seeded(ctx, name):
    g = Golden(name)
    if len(g.lens) < 2 or g.seq_distances is None:
        pytest.skip("no tree")
    k = int(g.n_anchors) if hasattr(g, "n_anchors") else 0
    rows = ctx.run_encoded(g.tree_seqs, g.codes, g.sorted_seqs(), g.subm, g.scal, n_anchors=k,
                           weight=float(g.weight) if k else 2.0, n_threads=2)
    assert input_order(g.ranks, rows) == [str(x) for x in g.rows]


@pytest.mark.parametrize("name", refine_cases())
def test_run_seeded_with_refinement(ctx, name):
    """kalign_run_seeded(..., refine, adaptive_budget): KALIGN_REFINE_ALL / _CONFIDENT (with and without the adaptive
    budget) after the alignment, KALIGN_REFINE_INLINE instead of it -- one call, rows of the real reference"""
    g = Golden(name)
    k = int(g.n_anchors)
    rows = ctx.run_encoded(g.tree_seqs, g.codes, g.sorted_seqs(), g.subm, g.scal, n_anchors=k,
                           weight=float(g.weight) if k else 2.0, n_threads=2, refine=int(g.mode))
    assert input_order(g.ranks, rows) == [str(x) for x in g.rows]


@pytest.mark.parametrize("name", REALIGN)
def test_run_realign(ctx, name):
    from kalign_amd import guide
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    off = np.concatenate([[0], np.cumsum(z["lens"])])
    codes = [z["codes"][off[i]:off[i + 1]] for i in range(len(z["lens"]))]
    letters = [str(z["seqs"][r]) for r in z["ranks"]]
    rows = ctx.run_encoded(guide.encode_tree(letters, dna=int(z["biotype"]) != 0), codes, letters, z["subm"], z["scal"],
                           n_anchors=int(z["n_anchors"]), weight=float(z["weight"]), realign=1)
    assert input_order(z["ranks"], rows) == [str(r) for r in z["final_rows"]]


def test_run_encoded_row_buffer_too_narrow(ctx):
    """a caller that guessed the alignment length too short gets KA_ERR_ROWS_STRIDE and the length, and fetches the rows
    without running again"""
    import ctypes as C
    from kalign_amd import api
    g = Golden("tree_dna4")
    tflat, off, lens = api._flatten(g.tree_seqs)
    cflat, _, _ = api._flatten(g.codes)
    letters = g.sorted_seqs()
    lflat = np.frombuffer("".join(letters).encode(), np.uint8).copy()
    sub = np.ascontiguousarray(g.subm, np.float32).reshape(-1)
    sc = np.ascontiguousarray(g.scal, np.float32)
    n = len(g.codes)
    alen = np.zeros(n, np.int32)
    rows = np.zeros((n, 4), np.uint8)
    p = api._ptr
    rc = ctx.L.ka_run_encoded(ctx.h, n, p(tflat), p(cflat), p(lflat), p(off), p(lens), p(sub), p(sc), 0, 2.0, 0, None, 1,
                              ord("-"), p(rows), 4, p(alen))
    assert rc == 3 and b"row_stride" in ctx.L.ka_last_error()
    assert int(alen[0]) == len(str(g.rows[0]))
    ctx._job = dict(lens=lens, ntasks=n - 1, n=n)
    assert input_order(g.ranks, ctx.tree_aligned_rows(letters)) == [str(x) for x in g.rows]
