"""Hand-over between neighbouring strips through LDS (ka_strip<.., HO>; KA_HO in the environment, on by default): strips of one
pass that run on neighbouring waves of a workgroup pass the boundary row -- the same 64-column batches -- through a ring in LDS
with progress words in LDS, instead of the HBM row buffer behind release / acquire fences.  Only HOW a strip learns its boundary
changes: every meetup, path and gap array must stay the reference's bit for bit, with the hand-over on (1), with four strips per
workgroup (2) and off (0).  The cases have tasks with several 128-row (and, with KA_Q1, 64-row) strips per pass, one workgroup
and clusters, protein (20 / 23 residue classes) and nucleotide profiles, and passes longer than the ring (256 slots: the slot
re-use check is exercised by the ~2000-column DNA tasks)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

# (KA_HW: strips with helper waves, ka_wstrip.h -- on by default; the LDS hand-over of ka_strip is what runs with KA_HW=0 and on
# levels with more than four items per workgroup)
MODES = [{"KA_HO": "1"}, {"KA_HO": "2"}, {"KA_HO": "0"}, {"KA_HO": "1", "KA_Q1": "3"}, {"KA_HO": "2", "KA_Q1": "1"},
         {"KA_HO": "1", "KA_MAX_CLUSTER": "1"}, {"KA_HO": "1", "KA_NO_CHAIN": "1"},
         {"KA_HW": "0"}, {"KA_HW": "0", "KA_HO": "2"}, {"KA_HW": "0", "KA_HO": "0"}, {"KA_HW": "0", "KA_MAX_CLUSTER": "1"},
         {"KA_HW": "1", "KA_MAX_CLUSTER": "2"}, {"KA_HW": "1", "KA_MAX_CLUSTER": "4"},
         # (round 4: clusters of up to 32 workgroups -- the default of big jobs -- and the spare workgroups by the ranking alone)
         {"KA_MAX_CLUSTER": "32"}, {"KA_MAX_CLUSTER": "24", "KA_CRIT_GREEDY": "0"}]


def reference_gaps(codes, tasks, dist, dna):
    from oracle import refdrv
    if not refdrv.available():
        pytest.skip("oracle/_ref not built")
    job = refdrv.EncodedJob(codes, tasks, dist, biotype=1 if dna else 0, type_=0 if dna else -1,
                            n_threads=min(16, os.cpu_count() or 1))
    gaps, _ = job.run_tree()
    job.close()
    return gaps


@pytest.mark.parametrize("nseq,length,dna", [(768, 400, False), (192, 2000, True)])
def test_lds_handover_matches_reference(nseq, length, dna, monkeypatch):
    import bench
    import kalign_amd
    codes, tasks, dist = bench.make_workload(nseq, length, dna, 3)
    subm, scal = bench.scoring(dna)
    want = reference_gaps(codes, tasks, dist, dna)
    ctx = kalign_amd.Context(0)
    try:
        ctx.tree_upload(codes, tasks, subm, scal, dist)
        for mode in MODES:
            for k, v in mode.items():
                monkeypatch.setenv(k, v)
            ctx.reload_env()
            for rep in range(3):
                ctx.tree_run()
                recs, paths, gaps = ctx.tree_download()
                for i, (got, w) in enumerate(zip(gaps, want)):
                    assert np.array_equal(got, w), (mode, rep, i)
                assert ctx.fallback_runs() == 0, (mode, rep)
            for k in mode:
                monkeypatch.delenv(k, raising=False)
    finally:
        ctx.close()


def test_lds_handover_with_b_z_x_residues(monkeypatch):
    """23 residue classes (B, Z, X present): the NRES = 23 instance of the strip."""
    import bench
    import kalign_amd
    codes, tasks, dist = bench.make_workload(384, 400, False, 5)
    rng = np.random.RandomState(11)
    codes = [c.copy() for c in codes]
    for c in codes:
        idx = rng.randint(0, len(c), max(len(c) // 50, 1))
        c[idx] = rng.randint(20, 23, len(idx)).astype(c.dtype)
    subm, scal = bench.scoring(False)
    want = reference_gaps(codes, tasks, dist, False)
    ctx = kalign_amd.Context(0)
    try:
        ctx.tree_upload(codes, tasks, subm, scal, dist)
        for mode in MODES[:4]:
            for k, v in mode.items():
                monkeypatch.setenv(k, v)
            ctx.reload_env()
            ctx.tree_run()
            recs, paths, gaps = ctx.tree_download()
            for i, (got, w) in enumerate(zip(gaps, want)):
                assert np.array_equal(got, w), (mode, i)
            assert ctx.fallback_runs() == 0, mode
            for k in mode:
                monkeypatch.delenv(k, raising=False)
    finally:
        ctx.close()
