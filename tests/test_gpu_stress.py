"""Schedule stress: every tree / consistency golden of the real reference again and again under randomised launch plans
-- cluster sizes, with and without the queued launch, the chained launch, the half kernel, the 4-wave leaf kernel, the
wave-local subtrees, the multi-wave meetup scan, 64-row strips, the LDS hand-over between strips -- and the answer must be the reference's bit for bit
every time (paths, meetups, scores, top-level f / b rows, gap arrays).  What varies is only WHO computes WHEN: a missing
barrier or an LDS region reused too early shows up here as a run that differs (round 2's multi-wave meetup experiment
failed one golden deterministically; this is the net that was missing).  KA_STRESS_REPS raises the repetitions."""
import os

import numpy as np
import pytest

from util import Golden, compare_recs, cons_cases, tree_cases

pytestmark = pytest.mark.gpu

EXACT = ["plen", "kind", "swapped", "meet", "transition", "score", "fhash", "bhash"]
SWITCHES = {"KA_MAX_CLUSTER": ["1", "2", "4", "8", "16", "24", "32", None], "KA_CRIT_GREEDY": ["0", None, None], "KA_NO_HALF": ["1", None], "KA_NO_QUEUE": ["1", None],
            "KA_NO_CHAIN": ["1", None, None], "KA_NO_LEAN": ["1", None, None], "KA_LEAN4": ["0", None], "KA_SUBTREE": ["0", "2", None, None],
            "KA_MW": ["0", None, None], "KA_Q1": ["0", "1", "2", "3", "4", None], "KA_CHAIN_G1": ["1", None], "KA_NO_CRIT": ["1", None],
            "KA_HO": ["0", "1", "2", None], "KA_HW": ["0", "1", None],
            # round 5: launch shapes of the 4-wave kernels, the chained launch beside the queued one, the queue's order
            "KA_QW": ["2", "1", None, None], "KA_LW": ["2", "1", None, None], "KA_OVERLAP": ["0", "1", None], "KA_QORDER": ["0", None, None],
            # anchor votes carried up the tree instead of counted at every task (off by default)
            "KA_CARRY": ["1", "3", None]}           # (3: marked cells always settled by the sweep, never one by one)


@pytest.mark.parametrize("name", tree_cases() + cons_cases())
def test_randomised_schedules_give_the_reference_answer(name, monkeypatch):
    import kalign_amd
    from kalign_amd import api
    reps = int(os.environ.get("KA_STRESS_REPS", "20"))
    g = Golden(name)
    cons = name.startswith("cons_")
    import zlib
    rng = np.random.RandomState(zlib.crc32(name.encode()))
    ctx = kalign_amd.Context(0)
    try:
        ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
        if cons:
            ctx.tree_build_consistency(int(g.n_anchors), float(g.weight))
        for rep in range(reps):
            chosen = {}
            for k, vals in SWITCHES.items():
                v = vals[rng.randint(len(vals))]
                if v is None:
                    monkeypatch.delenv(k, raising=False)
                else:
                    monkeypatch.setenv(k, v)
                    chosen[k] = v
            ctx.reload_env()
            ctx.tree_run()
            recs, paths, gaps = ctx.tree_download()
            assert compare_recs(g, recs, paths, EXACT) == [], (name, rep, chosen)
            for got, want in zip(gaps, g.gaps_list()):
                assert np.array_equal(got, want), (name, rep, chosen)
            assert ctx.fallback_runs() == 0, (name, rep, chosen)
    finally:
        for k in SWITCHES:
            monkeypatch.delenv(k, raising=False)
        ctx.close()


# ---- the big-shape schedule paths: clusters of 8-16 workgroups, hand-overs between workgroups (helper
# waves: write-through stores + flags; ka_strip: release fences), the multi-wave meetup scan over thousands of columns, four
# strips per workgroup, tasks beyond 4000 rows -- none of which the goldens (<= 64 sequences, <= 300 columns) reach.  Live
# reference (oracle/_ref: the real create_msa_tree), the same randomised switches, gap arrays bit for bit every time.
BIG_SWITCHES = dict(SWITCHES, KA_MAX_CLUSTER=["2", "8", "16", "32", None, None], KA_HO=["0", "1", "2", None], KA_HW=["0", "1", None, None],
                    KA_Q1=["0", "4", None, None], KA_MW=["0", None, None])


def _caterpillar(n):
    """a guide tree that adds one sequence at a time: its last tasks align one long profile to a sequence / small profile --
    with 2000-column sequences and gaps the root passes 4000 rows"""
    tasks, cur, nxt = [], 0, n
    for i in range(1, n):
        tasks.append((cur, i, nxt))
        cur, nxt = nxt, nxt + 1
    return np.array(tasks, np.int32)


def _big_jobs():
    import bench
    jobs = []
    codes, tasks, dist = bench.make_workload(768, 400, False, 3)
    jobs.append(("768x400 aa", codes, tasks, dist, False))
    codes, tasks, dist = bench.make_workload(256, 2000, True, 5)
    jobs.append(("256x2000 nt", codes, tasks, dist, True))
    # pairs first (profiles), then a caterpillar over the pair profiles' roots: profile-profile tasks that grow past 4000 rows
    from kalign_amd import synth, guide
    seqs = synth.dssim(96, 2400, dna=True, seed=7)
    codes = guide.encode(seqs, dna=True)
    n = len(codes)
    t, nxt, roots = [], n, []
    for i in range(0, n, 2):
        t.append((i, i + 1, nxt)); roots.append(nxt); nxt += 1
    cur = roots[0]
    for r in roots[1:]:
        t.append((cur, r, nxt)); cur = nxt; nxt += 1
    jobs.append(("96x2400 nt caterpillar of pair profiles", codes, np.array(t, np.int32), np.random.RandomState(7).uniform(0.3, 0.9, n).astype(np.float32), True))
    return jobs


@pytest.mark.parametrize("which", [0, 1, 2], ids=["768x400aa", "256x2000nt", "caterpillar_gt4000_rows"])
def test_randomised_schedules_on_big_shapes(which, monkeypatch):
    import bench
    import kalign_amd
    from oracle import refdrv
    if not refdrv.available():
        pytest.skip("oracle/_ref not built")
    label, codes, tasks, dist, dna = _big_jobs()[which]
    job = refdrv.EncodedJob(codes, tasks, dist, biotype=1 if dna else 0, type_=0 if dna else -1, n_threads=min(16, os.cpu_count() or 1))
    want, _ = job.run_tree()
    job.close()
    subm, scal = bench.scoring(dna)
    reps = int(os.environ.get("KA_STRESS_REPS_BIG", "10"))
    rng = np.random.RandomState(100 + which)
    ctx = kalign_amd.Context(0)
    try:
        ctx.tree_upload(codes, tasks, subm, scal, dist)
        longest = 0
        for rep in range(reps):
            chosen = {}
            for k, vals in BIG_SWITCHES.items():
                v = vals[rng.randint(len(vals))]
                if v is None:
                    monkeypatch.delenv(k, raising=False)
                else:
                    monkeypatch.setenv(k, v)
                    chosen[k] = v
            ctx.reload_env()
            ctx.tree_run()
            recs, paths, gaps = ctx.tree_download()
            longest = max(longest, max(max(r.len_a, r.len_b) for r in recs))
            for i, (got, w) in enumerate(zip(gaps, want)):
                assert np.array_equal(got, w), (label, rep, chosen, i)
            assert ctx.fallback_runs() == 0, (label, rep, chosen)
        if which == 2:
            assert longest > 4000, longest                       # (the "workgroup per four strips" rule of long tasks was reached)
    finally:
        for k in BIG_SWITCHES:
            monkeypatch.delenv(k, raising=False)
        ctx.close()


def test_chained_launch_that_arrives_before_the_queue_helps_it():
    """Overlapping launches (round 5): the chained launch goes out beside the queued one on a low-priority stream.  A dispatcher that
    brings its workgroups in FIRST leaves the queue eight CUs -- a slow round for one tree, a watchdog fallback for a forest (seen
    once in bench.py's 16-tree point, reproduced at will with KA_OVERLAP_HELP=0 on four 4096 x 2000 trees).  The chained workgroups
    therefore take queue tasks while more than a round of the queue is left.  KA_DEBUG_CHAIN_FIRST enqueues the chain BEFORE the queue:
    same alignment, three launches, no fallback."""
    import kalign_amd
    from kalign_amd import api, guide, synth
    seqs = synth.dssim(4096, 300, dna=False, seed=11)
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
    seqs = [seqs[i] for i in order]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "param_tables.npz"))
    subm, scal = z["subm_0_3"], z["scal_0_3"].copy()
    ctx = kalign_amd.Context(0)
    try:
        tasks, sd = ctx.guide_tree(guide.encode_tree(seqs, dna=False), n_threads=8)
        codes = guide.encode(seqs, dna=False)
        # two copies as a forest: the queue holds a few rounds of tasks
        fc, ft, fd, _ = guide.forest([(codes, tasks, sd)] * 2)
        ctx.tree_upload(fc, ft, subm, scal, fd)
        ctx.tree_run(); ctx.tree_sync()
        assert ctx.tree_kernel_ms()[1] == 3
        recs0, paths0, gaps0 = ctx.tree_download()
        ctx.debug_set_hooks(8)                                        # KA_DEBUG_CHAIN_FIRST
        helped = 0
        for _ in range(3):
            ctx.tree_run(); ctx.tree_sync()
            assert ctx.tree_kernel_ms()[1] == 3 and ctx.fallback_runs() == 0
            helped += ctx.helped_tasks()
            recs, paths, gaps = ctx.tree_download()
            assert [r.plen for r in recs] == [r.plen for r in recs0]
            assert np.array_equal(paths, paths0)
            for a, b in zip(gaps, gaps0):
                assert np.array_equal(a, b)
        ctx.debug_set_hooks(0)
        assert helped > 0, "the chained launch's workgroups were resident first and took no queue task"
    finally:
        ctx.close()
