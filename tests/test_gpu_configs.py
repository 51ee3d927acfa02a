"""GPU: BASELINE.json's configurations at (or near) their full sizes against the REAL reference (oracle/_ref) run on
the box's host cores -- sized so that the reference finishes in about a minute per case.

  C5  `--precise --ensemble 8`, 2048 x ~300 aa: members of kalign_ensemble's loop (ensemble.c:286-339) are
      kalign_run_realign runs with the member's gap penalties (resolve_run_params, ensemble.c:55-76), default-mode
      consistency (5 anchors) and one realignment iteration; member k through ka_run_encoded == the reference.
  C3  DNA x ~2000 nt, --type dna --fast: 1024 sequences (the 4096-sequence tree runs through property checks in
      test_gpu_fullsize.py); final rows == the reference.
  C4  protein x ~500, default mode: 4096 sequences --fast and 2048 in default mode; final rows == the reference.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# ensemble.c:32-45 (gpo, gpe, tgpe multipliers of member k; member 0 runs the defaults)
RUN_PARAMS = {0: (1.0, 1.0, 1.0), 1: (0.5, 1.5, 0.8), 2: (1.5, 0.5, 1.2), 5: (0.8, 1.2, 1.0)}


@pytest.fixture(scope="module")
def ctx():
    import kalign_amd
    c = kalign_amd.Context(0)
    yield c
    c.close()


def _sorted(seqs):
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))       # msa_sort_len_name; names = input order
    return order, [seqs[i] for i in order]


def _input_order(order, rows):
    out = [None] * len(rows)
    for k, i in enumerate(order):
        out[i] = rows[k].decode()
    return out


@pytest.fixture(scope="module")
def c5_set():
    from kalign_amd import synth
    return synth.dssim(2048, 300, seed=1)


@pytest.mark.parametrize("member", [0, 1, 2])
def test_c5_precise_ensemble_member_at_workload(ctx, c5_set, member):
    import bench
    from kalign_amd import guide
    from oracle import refdrv
    assert refdrv.available(), "oracle/_ref missing"
    seqs = c5_set
    order, srt = _sorted(seqs)
    subm, scal = bench.scoring(False)
    f = RUN_PARAMS[member]
    scal = scal.copy()
    scal[0] *= np.float32(f[0]); scal[1] *= np.float32(f[1]); scal[2] *= np.float32(f[2])
    rows = ctx.run_encoded(guide.encode_tree(srt), guide.encode(srt), srt, subm, scal, n_anchors=5, weight=2.0, realign=1,
                           n_threads=16)
    # the reference: kalign_run_realign's sequence (aln_wrap.c:361-527) with the member's penalties
    job = refdrv.RefJob(seqs, gpo=float(scal[0]), gpe=float(scal[1]), tgpe=float(scal[2]), n_threads=16)
    job.build_consistency(5, 2.0)
    job.run_tree()
    job.realign_tree(want_dm=False)
    job.run_tree()
    want = job.finalise()
    job.close()
    got = _input_order(order, rows)
    assert len(got[0]) == len(want[0])
    assert got == want


@pytest.mark.parametrize("n,length,dna,anchors", [(1024, 2000, True, 0), (4096, 500, False, 0), (2048, 500, False, 5)],
                         ids=["c3_dna1024x2000_fast", "c4_prot4096x500_fast", "c4_prot2048x500_default"])
def test_c3_c4_rows_against_the_reference(ctx, n, length, dna, anchors):
    import bench
    from kalign_amd import guide, synth
    from oracle import refdrv
    assert refdrv.available(), "oracle/_ref missing"
    seqs = synth.dssim(n, length, dna=dna, seed=2)
    order, srt = _sorted(seqs)
    subm, scal = bench.scoring(dna)
    rows = ctx.run_encoded(guide.encode_tree(srt, dna=dna), guide.encode(srt, dna=dna), srt, subm, scal, n_anchors=anchors,
                           weight=2.0, realign=0, n_threads=16)
    job = refdrv.RefJob(seqs, type_=0 if dna else -1, n_threads=16)
    if anchors:
        job.build_consistency(anchors, 2.0)
    job.run_tree()
    want = job.finalise()
    job.close()
    assert _input_order(order, rows) == want
