"""GPU parity of the realignment pass (kalign_run_realign, aln_wrap.c:361-527): identity distances from a finished
alignment (compute_aln_pairwise_dist), the UPGMA tree built on them (build_tree_from_pairwise) -- both on the device --
and the second alignment on that tree, against goldens the real reference produced (tests/golden/realign_*.npz)."""
import os

import numpy as np
import pytest

from util import GOLDEN

pytestmark = pytest.mark.gpu

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("realign_") and f.endswith(".npz"))


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    off = np.concatenate([[0], np.cumsum(z["lens"])])
    codes = [z["codes"][off[i]:off[i + 1]] for i in range(len(z["lens"]))]
    letters = [str(z["seqs"][r]) for r in z["ranks"]]            # sorted order
    return z, codes, letters


@pytest.mark.parametrize("name", CASES)
def test_tree_from_given_rows(ctx, name):
    """host rows in: distances bit for bit, the reference's task list, its seq_distances"""
    z, _, _ = load(name)
    rows = [str(r).encode() for r in z["rows_sorted"]]
    tasks, sd, dm = ctx.aln_guide_tree(rows, want_dm=True)
    assert np.array_equal(dm.view(np.uint32), z["dm"].view(np.uint32))
    assert np.array_equal(tasks, z["tasks2"])
    assert np.array_equal(sd.view(np.uint32), z["seq_distances2"].view(np.uint32))


@pytest.mark.parametrize("name", CASES)
def test_align_realign_align(ctx, name):
    """the whole iteration with everything resident: align, rows, tree from the rows in HBM, align again"""
    from kalign_amd import api
    z, codes, letters = load(name)
    k = int(z["n_anchors"])
    ctx.msa_tree(codes, z["tasks1"], z["subm"], z["scal"], z["seq_distances1"], n_anchors=k, weight=float(z["weight"]))
    rows = ctx.tree_aligned_rows(letters)
    assert [r.decode() for r in rows] == [str(r) for r in z["rows_sorted"]]
    tasks2, sd2 = ctx.aln_guide_tree()
    assert np.array_equal(tasks2, z["tasks2"])
    assert np.array_equal(sd2.view(np.uint32), z["seq_distances2"].view(np.uint32))
    # second pass: new tree, new distances, the consistency table of the first pass
    ctx.tree_upload(codes, tasks2, z["subm"], z["scal"], sd2, flags=api.FLAG_DEVICE_GAPS | api.FLAG_KEEP_CONSISTENCY)
    ctx.tree_run()
    rows2 = ctx.tree_aligned_rows(letters)
    got = [None] * len(rows2)
    for i, r in enumerate(z["ranks"]):
        got[int(r)] = rows2[i].decode()
    assert got == [str(r) for r in z["final_rows"]]


def test_realign_error_behaviour(ctx):
    import kalign_amd
    from kalign_amd import api
    z, codes, letters = load("realign_prot40")
    ctx.tree_upload(codes, z["tasks1"], z["subm"], z["scal"], z["seq_distances1"], flags=api.FLAG_DEVICE_GAPS)
    ctx.tree_run()
    with pytest.raises(kalign_amd.KalignAmdError, match="no rows on the device"):
        ctx.aln_guide_tree()
    with pytest.raises(kalign_amd.KalignAmdError, match="one length"):
        ctx.aln_guide_tree([b"AC-", b"AC"])
    # a table must not be kept for other sequences
    ctx.tree_build_consistency(3, 2.0)
    other = [c.copy() for c in codes]
    other[0] = other[0][::-1].copy()
    with pytest.raises(kalign_amd.KalignAmdError, match="sequences differ"):
        ctx.tree_upload(other, z["tasks1"], z["subm"], z["scal"], z["seq_distances1"], flags=api.FLAG_KEEP_CONSISTENCY)


@pytest.mark.parametrize("name", ["realign_prot40", "realign_dna24_cons"])
def test_precise_member_through_dist(ctx, name):
    """a `--precise` ensemble member (kalign_run_realign, one iteration) through dist.member_on_context, starting
    from letters: tree alphabet, k-means tree, align, rows, UPGMA tree, align, rows"""
    from kalign_amd import dist as kd, guide
    z, codes, letters = load(name)
    dna = int(z["biotype"]) != 0
    run = kd.member_on_context(ctx, guide.encode_tree(letters, dna=dna), codes, letters, z["subm"],
                               n_anchors=int(z["n_anchors"]), weight=float(z["weight"]), realign=1)
    rows = kd.ensemble_members(run, [dict(scal=z["scal"])], 0, 1)[0]
    got = [None] * len(rows)
    for i, r in enumerate(z["ranks"]):
        got[int(r)] = rows[i].decode()
    assert got == [str(r) for r in z["final_rows"]]


def test_tiny_and_degenerate_row_sets(ctx):
    # two rows: one merge; distance = 1 - 2/3 over the three columns where both have a residue
    tasks, sd, dm = ctx.aln_guide_tree([b"AC-GT", b"ACTC-"], want_dm=True)
    assert tasks.tolist() == [[0, 1, 2]]
    want = np.float32(1.0) - np.float32(2) / np.float32(3)
    assert dm[0, 1] == want and dm[1, 0] == want and dm[0, 0] == 0 and sd.tolist() == [want, want]
    # nothing aligned between two rows: distance 1 (aln_apair_dist.c:82-84); identical rows: 0 and the first pair wins
    rows = [b"AC---", b"---GT", b"AC---", b"AC---"]
    tasks, sd, dm = ctx.aln_guide_tree(rows, want_dm=True)
    assert dm[0, 1] == 1.0 and dm[0, 2] == 0.0 and dm[2, 3] == 0.0
    assert tasks[0].tolist() == [0, 2, 4]                         # the first minimal pair in row-major order
    # a width that is no multiple of the kernel's 128-column step, a row count that is no multiple of its 16-row tile
    rng = np.random.RandomState(1)
    rows = [bytes(rng.choice(list(b"ACDE-"), size=301).astype(np.uint8)) for _ in range(37)]
    tasks, sd, dm = ctx.aln_guide_tree(rows, want_dm=True)
    a = np.frombuffer(b"".join(rows), np.uint8).reshape(37, 301)
    for i, j in ((0, 1), (5, 36), (17, 18), (35, 36)):
        both = (a[i] != 45) & (a[j] != 45)
        want = np.float32(1.0) - np.float32(int((both & (a[i] == a[j])).sum())) / np.float32(int(both.sum()))
        assert dm[i, j] == want and dm[j, i] == want
    assert sorted(tasks[:, 2].tolist()) == list(range(37, 73))


@pytest.mark.parametrize("n,alnlen,seed", [(300, 257, 11), (97, 1000, 12), (513, 130, 13)])
def test_tree_from_random_rows_matches_the_oracle(ctx, oracle, n, alnlen, seed):
    """seeded rows with many tied distances (few letters, short rows, duplicated rows): the device's distances and
    UPGMA task list against the oracle's restatement, bit for bit"""
    rng = np.random.RandomState(seed)
    base = rng.choice(list(b"ACD-"), size=(n // 3 + 1, alnlen)).astype(np.uint8)
    rows = base[rng.randint(0, len(base), size=n)].copy()
    flip = rng.random_sample(rows.shape) < 0.02
    rows[flip] = rng.choice(list(b"ACD-"), size=int(flip.sum())).astype(np.uint8)
    rows = [bytes(r) for r in rows]
    tasks, sd, dm = ctx.aln_guide_tree(rows, want_dm=True)
    otasks, osd, odm = oracle.aln_guide_tree(rows)
    assert np.array_equal(dm.view(np.uint32), odm.view(np.uint32))
    assert np.array_equal(sd.view(np.uint32), osd.view(np.uint32))
    assert np.array_equal(tasks, otasks)


@pytest.mark.parametrize("n", [300, 1100, 2100, 4200, 6200])
def test_upgma_in_one_workgroup_and_in_per_merge_launches_agree(ctx, n):
    """the two schedules of the device UPGMA (one workgroup for all merges, in its five sizes; one launch per merge, the
    path above 6144 rows) give the same task list on rows with many tied distances"""
    import os
    rng = np.random.RandomState(n)
    base = rng.choice(list(b"ACDE-"), size=(n // 3, 60)).astype(np.uint8)
    rows = base[rng.randint(0, len(base), size=n)].copy()
    flip = rng.random_sample(rows.shape) < 0.03
    rows[flip] = rng.choice(list(b"ACDE-"), size=int(flip.sum())).astype(np.uint8)
    rows = [bytes(r) for r in rows]
    t1, sd1 = ctx.aln_guide_tree(rows)
    os.environ["KA_UPGMA_LAUNCHES"] = "1"
    try:
        ctx.reload_env()
        t2, sd2 = ctx.aln_guide_tree(rows)
    finally:
        del os.environ["KA_UPGMA_LAUNCHES"]
        ctx.reload_env()
    assert np.array_equal(t1, t2) and np.array_equal(sd1, sd2)
