"""Debugging aid: top-level rows of one golden run after run (KA_DUMP_ROWS); where do runs differ from the first one?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, kalign_amd
from kalign_amd import api
from util import Golden
name, reps = sys.argv[1], int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("="); os.environ[k] = v
fn = "/tmp/rows_probe.bin"
os.environ["KA_DUMP_ROWS"] = fn
g = Golden(name)
ctx = kalign_amd.Context(0)
ctx.tree_upload(g.codes, g.tasks, g.subm, g.scal, g.seq_distances, flags=api.FLAG_DEBUG_ROWS)
def one():
    if os.path.exists(fn): os.remove(fn)
    ctx.tree_run(); recs, _, _ = ctx.tree_download()
    out, raw, o = {}, open(fn, "rb").read(), 0
    while o < len(raw):
        t, n = np.frombuffer(raw, np.int64, 2, o); o += 16
        out[int(t)] = np.frombuffer(raw, np.float32, 2 * int(n), o).reshape(2, -1, 3).copy(); o += 8 * int(n)
    return out, recs
ref, recs = one()
want = {t: (int(g.rec("fhash")[t]), int(g.rec("bhash")[t])) for t in ref}
ok = all((recs[t].fhash, recs[t].bhash) == want[t] for t in ref)
print("first run matches the golden's hashes:", ok)
for rep in range(reps):
    cur, _ = one()
    for t in sorted(cur):
        d = np.argwhere(cur[t].view(np.uint32) != ref[t].view(np.uint32))
        if len(d):
            print("run", rep, "task", t, "lens", recs[t].len_a, recs[t].len_b, ":", len(d), "entries differ")
            for side in (0, 1):
                cols = sorted(set(int(x[1]) for x in d if x[0] == side))
                if cols: print("    %s row columns %s" % ("fb"[side], cols[:20]), "...", cols[-3:], "comps", sorted(set(int(x[2]) for x in d if x[0] == side)))
            x = d[0]
            print("    e.g. [%s col %d comp %d] first run %r this run %r" % ("fb"[x[0]], x[1], x[2], ref[t][tuple(x)], cur[t][tuple(x)]))
