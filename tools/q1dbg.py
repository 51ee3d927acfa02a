"""Debugging aid: the top-level f / b rows (KA_FLAG_DEBUG_ROWS) of every task of a golden under two KA_Q1 settings, diffed."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, kalign_amd
from kalign_amd import api
from util import Golden
def rows(tag, name):
    fn = "/tmp/rows_%s.bin" % tag
    if os.path.exists(fn): os.remove(fn)
    os.environ["KA_DUMP_ROWS"] = fn
    ctx = kalign_amd.Context(0)
    g = Golden(name)
    ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
    ctx.tree_run(); recs, _, _ = ctx.tree_download()
    ctx.close()
    out, raw, o = {}, open(fn, "rb").read(), 0
    while o < len(raw):
        t, n = np.frombuffer(raw, np.int64, 2, o); o += 16
        out[int(t)] = np.frombuffer(raw, np.float32, 2 * int(n), o).reshape(2, -1, 3); o += 8 * int(n)
    return out, recs
name = sys.argv[1] if len(sys.argv) > 1 else "tree_BB12006"
os.environ["KA_Q1"] = "0"; a, recs = rows("q0", name)
os.environ["KA_Q1"] = "4"; b, _ = rows("q4", name)
nbad = 0
for t in sorted(a):
    d = np.argwhere(a[t].view(np.uint32) != b[t].view(np.uint32))
    if len(d):
        nbad += 1
        r = recs[t]
        print("task", t, "kind", r.kind, "lens", r.len_a, r.len_b, "differs at", len(d), "entries")
        for side in (0, 1):
            cols = sorted(set(int(x[1]) for x in d if x[0] == side))
            print("    %s row: %d columns differ: %s" % ("fb"[side], len(cols), cols[:12]))
print("tasks with differing rows:", nbad, "of", len(a))
