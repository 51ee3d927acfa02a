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
SWITCHES = {"KA_MAX_CLUSTER": ["1", "2", "4", "8", "16", None], "KA_NO_HALF": ["1", None], "KA_NO_QUEUE": ["1", None],
            "KA_NO_CHAIN": ["1", None, None], "KA_NO_LEAN": ["1", None, None], "KA_LEAN4": ["0", None], "KA_SUBTREE": ["0", None, None],
            "KA_MW": ["0", None, None], "KA_Q1": ["0", "1", "2", "3", "4", None], "KA_CHAIN_G1": ["1", None], "KA_NO_CRIT": ["1", None],
            "KA_HO": ["0", "1", "2", None], "KA_HW": ["0", "1", None]}


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
