"""Debugging aid: one golden under one KA_* setting again and again; which fields of which tasks ever differ from the golden."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, kalign_amd
from kalign_amd import api
from util import Golden, compare_recs
name, reps = sys.argv[1], int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("="); os.environ[k] = v
g = Golden(name)
ctx = kalign_amd.Context(0)
ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
bad = collections.Counter()
for rep in range(reps):
    ctx.tree_run()
    recs, paths, gaps = ctx.tree_download()
    for t, f, *_ in compare_recs(g, recs, paths, ["plen", "kind", "swapped", "meet", "transition", "score", "fhash", "bhash"]):
        r = recs[t]
        bad[(t, f, r.kind, r.len_a, r.len_b)] += 1
print(name, sys.argv[3:], "runs", reps, "mismatches:", dict(bad) if bad else "none", "fallback", ctx.fallback_runs())
