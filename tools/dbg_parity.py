"""Which tasks of the golden trees differ (kind, shape, first differing field)?  usage: dbg_parity.py [KA_X=..,KA_Y=..]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []):
    k, v = kv.split("="); os.environ[k] = v
import numpy as np, kalign_amd
from kalign_amd import api
from util import Golden, compare_recs, tree_cases
EXACT = ["plen", "kind", "swapped", "meet", "transition", "score", "fhash", "bhash"]
ctx = kalign_amd.Context(0)
for name in tree_cases():
    g = Golden(name)
    recs, paths, gaps = ctx.msa_tree(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
    probs = compare_recs(g, recs, paths, EXACT)
    bad = sorted({p[0] for p in probs})
    print("%-28s tasks %3d  bad %3d" % (name, len(recs), len(bad)), flush=True)
    for t in bad[:6]:
        r = recs[t]
        print("      task %3d kind %d  %4d x %-4d nsip %d+%d  fields %s" % (t, r.kind, r.len_a, r.len_b, r.nsip_a, r.nsip_b, [p[1] for p in probs if p[0] == t][:6]))
ctx.close()
