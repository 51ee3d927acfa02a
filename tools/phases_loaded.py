"""Where a task's time goes when the GPU is full: mean phase times (KA_FLAG_TIMING) of the tasks of every guide-tree level of the
4096 x 400 protein tree, alone and with COPIES copies of the tree in flight as one forest.  usage: phases_loaded.py [copies 16]
Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd, torch
torch.cuda.init()
from kalign_amd import api, guide
COPIES = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = kalign_amd.Context(0)
job = bench.make_job(ctx, 4096, 400, False, seed=1)
subm, scal = bench.scoring(False)
GHZ = 2.4
for n in (1, COPIES):
    fc, ft, fd, _ = guide.forest([(job["codes"], job["tasks"], job["seq_distances"])] * n)
    ctx.tree_upload(fc, ft, subm, scal, fd, flags=api.FLAG_TIMING)
    for _ in range(2): ctx.tree_run(); ctx.tree_sync()
    recs, paths, _ = ctx.tree_download(want_gaps=False)
    tm = ctx.tree_timing()
    lvl = {}
    tl = []
    for r in recs:
        l = 1 + max(lvl.get(r.a, 0), lvl.get(r.b, 0)); lvl[r.c] = l; tl.append(l)
    tl = np.array(tl); kind = np.array([r.kind for r in recs])
    print("== %d tree(s) in flight: launches %s ms" % (n, ["%.2f" % x for x in ctx.tree_launch_ms()] if os.environ.get("KA_LAUNCH_EV") else ctx.tree_kernel_ms()))
    print("   level  tasks  kind  mean us: total  prep  pass  meet  other-hirsch  code  merge   cells/task")
    for l in range(1, 8):
        for k in range(3):
            m = (tl == l) & (kind == k)
            if not m.sum(): continue
            x = tm[m].astype(np.float64) / GHZ / 1e3
            tot = x[:, :4].sum(1)
            print("   L%-2d   %6d   k%d          %7.0f %5.0f %5.0f %5.0f %8.0f %9.0f %6.0f   %9.0f" % (
                l, m.sum(), k, tot.mean(), x[:, 0].mean(), x[:, 4].mean(), x[:, 5].mean(), (x[:, 1] - x[:, 4] - x[:, 5]).mean(), x[:, 2].mean(), x[:, 3].mean(),
                tm[m][:, 7].mean()))
ctx.close()
