"""CPU checks of the Python-side host logic: synthetic inputs, guide trees, forests, and -- through the oracle -- that
a forest job is nothing but its trees run side by side."""
import numpy as np


def test_bisecting_tree_is_a_valid_task_list():
    from kalign_amd import guide
    for n in (2, 3, 17, 256):
        tasks = guide.bisecting_tree(n, seed=n)
        assert tasks.shape == (n - 1, 3)
        made = set(range(n))
        for t, (a, b, c) in enumerate(tasks):
            assert a in made and b in made and c == n + t          # children before parents, c = numseq + t
            made.discard(int(a)); made.discard(int(b)); made.add(int(c))
        assert made == {2 * n - 2}                                  # one root


def test_dssim_like_generator_is_seeded_and_has_the_right_shape():
    from kalign_amd import synth
    a = synth.dssim(64, 200, seed=3)
    b = synth.dssim(64, 200, seed=3)
    c = synth.dssim(64, 200, seed=4)
    assert a == b and a != c
    lens = np.array([len(s) for s in a])
    assert len(a) == 64 and 150 < lens.mean() < 250 and lens.min() > 0
    assert set("".join(a)) <= set("ARNDCQEGHILKMFPSTWYV")


def test_forest_renumbers_nodes_and_runs_like_separate_jobs(oracle):
    """guide.forest: node ids stay unique, tasks stay in children-before-parents order; each tree of the forest
    aligns exactly as it does alone (checked with the oracle's per-tree runs)."""
    from kalign_amd import guide
    from util import Golden
    g1, g2 = Golden("tree_prot32x200"), Golden("tree_ragged")
    codes, tasks, dist, spans = guide.forest([(g1.codes, g1.tasks, g1.seq_distances), (g2.codes, g2.tasks, g2.seq_distances)])
    n1, n2 = len(g1.codes), len(g2.codes)
    assert len(codes) == n1 + n2 and len(tasks) == len(g1.tasks) + len(g2.tasks)
    assert spans == [(0, 0, n1, len(g1.tasks)), (n1, len(g1.tasks), n2, len(g2.tasks))]
    made = set(range(n1 + n2))
    for a, b, c in tasks:
        assert a in made and b in made and c not in made and c >= n1 + n2
        made.add(int(c))
    # leaves of the second job were shifted, internal nodes of both jobs do not collide
    assert tasks[len(g1.tasks)][0] >= n1 or tasks[len(g1.tasks)][0] >= n1 + n2
    assert len({int(c) for _, _, c in tasks}) == len(tasks)
    assert dist is not None and len(dist) == n1 + n2


def test_member_on_context_call_sequence():
    """dist.member_on_context drives a context exactly like kalign_run_seeded / kalign_run_realign drive the library:
    checked with a recording stand-in (no GPU): order of calls, the noisy tree only without realignment, the second
    pass keeping the consistency table."""
    import numpy as np
    from kalign_amd import api, dist as kd

    class Recorder:
        def __init__(self):
            self.calls = []

        def guide_tree(self, tree_codes, n_threads=1, dm_scale=None):
            self.calls.append(("guide_tree", dm_scale is not None))
            return "tasks0", "sd0"

        def msa_tree(self, codes, tasks, subm, scal, sd, n_anchors=0, weight=2.0):
            self.calls.append(("msa_tree", tasks, sd, n_anchors))

        def tree_aligned_rows(self, letters):
            self.calls.append(("rows",))
            return ["row%d" % len(self.calls)]

        def aln_guide_tree(self):
            self.calls.append(("aln_guide_tree",))
            return "tasks%d" % len(self.calls), "sd%d" % len(self.calls)

        def tree_upload(self, codes, tasks, subm, scal, sd, flags=0):
            self.calls.append(("upload", tasks, sd, flags))

        def tree_run(self):
            self.calls.append(("run",))

    member = dict(scal=np.zeros(6, np.float32), dm_scale=np.ones(4, np.float32))
    r = Recorder()
    rows = kd.member_on_context(r, "tc", "c", "l", "subm", n_anchors=5)(member)
    assert r.calls == [("guide_tree", True), ("msa_tree", "tasks0", "sd0", 5), ("rows",)] and rows == ["row3"]
    r = Recorder()
    rows = kd.member_on_context(r, "tc", "c", "l", "subm", n_anchors=5, realign=2)(member)
    keep = api.FLAG_DEVICE_GAPS | api.FLAG_KEEP_CONSISTENCY
    assert r.calls == [("guide_tree", False), ("msa_tree", "tasks0", "sd0", 5), ("rows",),
                       ("aln_guide_tree",), ("upload", "tasks4", "sd4", keep), ("run",), ("rows",),
                       ("aln_guide_tree",), ("upload", "tasks8", "sd8", keep), ("run",), ("rows",)]
    assert rows == ["row11"]


def test_integration_md_quotes_the_compiled_glue():
    """INTEGRATION.md section 1 must be, verbatim, the functions of oracle/dropin/kalign_amd_glue.c that `make -C oracle
    dropin` compiles (tools/make_integration.py regenerates it)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_integration", os.path.join(root, "tools", "make_integration.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.render() == open(os.path.join(root, "INTEGRATION.md")).read()


def test_c_subtree_plan_matches_the_python_plan():
    """ka_dist_plan_subtrees (the C layer's planning function, pure host logic) against kalign_amd.dist.plan_subtrees on
    seeded random guide trees: same rank for every task, same tasks above the cut -- for 1 .. 8 ranks"""
    import numpy as np
    from kalign_amd import api, dist, guide
    rng = np.random.RandomState(7)
    for n in (3, 5, 17, 64, 257):
        lens = rng.randint(40, 700, n)
        for world in (1, 2, 3, 4, 8):
            tasks = guide.bisecting_tree(n, seed=int(rng.randint(1 << 30)), jitter=0.3)
            rr_py, top_py = dist.plan_subtrees(tasks, lens, world)
            rr_c, top_c = api.dist_plan_subtrees(lens, tasks, world)
            assert list(rr_c) == [int(x) for x in rr_py], (n, world)
            assert top_c == [int(t) for t in top_py], (n, world)
