"""The throughput kernel (round 6; unit 10 of kalign_amd/csrc/ka_kernels.hip, opt-in: KA_TP=1): launches of the 4-wave kind run
three workgroups per CU with the lean profile-profile strip of ka_lstrip.h (record-major 88-column ring fed by the strip itself,
steps in octets, the last row through an LDS out ring) and their passes and meetups as real functions.  A launch takes that kernel
when a guide-tree level has more tasks than the GPU has CUs: forests of the reference's goldens (every copy must come out as the
golden does: aln_seqseq.c / aln_seqprofile.c / aln_profileprofile.c through the Hirschberg recursion of aln_controller.c, bit for
bit) and a synthetic tree against the default kernels."""
import numpy as np
import pytest

from util import Golden, compare_recs, tree_cases

pytestmark = pytest.mark.gpu

EXACT = ["len_a", "len_b", "nsip_a", "nsip_b", "plen", "kind", "swapped", "meet", "transition", "gap_scale", "subm_off", "score"]
# goldens whose scoring tables and penalties are the plain ones of the forest below (every copy shares one subm / scal)
COPIES = 40


@pytest.mark.parametrize("name", tree_cases())
@pytest.mark.parametrize("tp", ["1", "0"])
def test_forest_of_goldens(name, tp, monkeypatch):
    """40 copies of a golden tree as one forest: the lower levels hold more tasks than CUs and go to the 4-wave kind of launch --
    the throughput kernel with KA_TP=1 (alphabets without B / Z / X; the others stay on the 4-wave kernel, same answer)"""
    import kalign_amd
    from kalign_amd import guide
    monkeypatch.setenv("KA_TP", tp)
    g = Golden(name)
    if len(g.lens) < 8:
        pytest.skip("too few tasks per level even as a forest")
    sd = g.seq_distances
    # dependency levels of the tree; a level goes to the 4-wave kind of launch when it holds more tasks than the GPU has CUs
    # (256) and not only seq-seq ones -- enough copies that the widest mixed level of the forest does
    n, depth, mixed = len(g.lens), {}, {}
    for a, b, c in np.asarray(g.tasks):
        d = 1 + max(depth.get(int(a), 0), depth.get(int(b), 0))
        depth[int(c)] = d
        if a >= n or b >= n:
            mixed[d] = mixed.get(d, 0) + 1
    copies = max(COPIES, 300 // max(mixed.values()) + 1)
    fc, ft, fd, spans = guide.forest([(g.codes, g.tasks, sd)] * copies if sd is not None else [(g.codes, g.tasks)] * copies)
    ctx = kalign_amd.Context(0)
    try:
        before = ctx.tp_launches()
        recs, paths, gaps = ctx.msa_tree(fc, ft, g.subm, g.scal, fd)
        assert ctx.fallback_runs() == 0
        nres = int(max(int(c.max()) for c in g.codes)) + 1
        took_tp = ctx.tp_launches() > before
        assert took_tp == (tp == "1" and nres <= 20), (nres, took_tp)
        want_gaps = g.gaps_list()
        for (s0, t0, ns, nt) in spans:
            sub = recs[t0:t0 + nt]
            assert compare_recs(g, sub, paths, EXACT) == [], (name, t0)
            for got, want in zip(gaps[s0:s0 + ns], want_gaps):
                assert np.array_equal(got, want)
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", [(1536, 300, False), (2048, 700, False), (2048, 600, True)])
def test_throughput_kernel_against_the_default_kernels(shape, monkeypatch):
    """synthetic families (wider windows, several strips per pass, nucleotides): KA_TP=1 against KA_TP=0 on one context"""
    import os
    import bench
    import kalign_amd
    nseq, length, dna = shape
    codes, tasks, dist = bench.make_workload(nseq, length, dna, 3)
    subm, scal = bench.scoring(dna)
    ctx = kalign_amd.Context(0)
    try:
        out = {}
        for tp in ("0", "1"):
            monkeypatch.setenv("KA_TP", tp)
            ctx.reload_env()
            before = ctx.tp_launches()
            recs, paths, gaps = ctx.msa_tree(codes, tasks, subm, scal, dist)
            assert ctx.fallback_runs() == 0
            assert (ctx.tp_launches() > before) == (tp == "1")
            out[tp] = ([(r.plen, r.meet, r.transition, r.score) for r in recs], [paths[r.path_off:r.path_off + r.plen + 2].copy() for r in recs], gaps)
        assert out["0"][0] == out["1"][0]
        for a, b in zip(out["0"][1], out["1"][1]):
            assert np.array_equal(a, b)
        for a, b in zip(out["0"][2], out["1"][2]):
            assert np.array_equal(a, b)
    finally:
        ctx.close()


@pytest.mark.parametrize("switches", [{"KA_SPINE": "8", "KA_RESERVE": "16"}, {"KA_SPINE": "3"}, {"KA_RESERVE": "24"}, {"KA_QORDER": "2"},
                                      {"KA_SPINE": "6", "KA_RESERVE": "8", "KA_TP": "1"}])
def test_round6_schedule_experiments_give_the_same_alignment(switches, monkeypatch):
    """the planner experiments of round 6 (all off by default): the most critical entries' spines as tasks of the chained
    launch (chain_need 1 + a done flag from the queue), CUs of XCC 0 that the queued launch leaves to the head of the chain, one
    order over all the queue's levels -- WHO runs WHEN changes, the alignment must not, and no run may fall back"""
    import bench
    import kalign_amd
    codes, tasks, dist = bench.make_workload(2560, 300, False, 5)
    subm, scal = bench.scoring(False)
    ctx = kalign_amd.Context(0)
    try:
        want = ctx.msa_tree(codes, tasks, subm, scal, dist)
        for k, v in switches.items():
            monkeypatch.setenv(k, v)
        ctx.reload_env()
        for _ in range(2):
            got = ctx.msa_tree(codes, tasks, subm, scal, dist)
            assert ctx.fallback_runs() == 0
            assert [(r.plen, r.meet, r.transition, r.score) for r in got[0]] == [(r.plen, r.meet, r.transition, r.score) for r in want[0]]
            assert np.array_equal(got[1], want[1])
            for a, b in zip(got[2], want[2]):
                assert np.array_equal(a, b)
    finally:
        ctx.close()
