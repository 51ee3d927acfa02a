"""Cycles per wavefront step of the strip pipeline, measured on synthetic profile-profile tasks of chosen shape
(two random protein pairs, rows x cols): root-task level times (KA_FLAG_TIMING) against the step model
C + 64 + (K-1) * 127.  Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd, torch
torch.cuda.init()
from kalign_amd import api
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
rng = np.random.RandomState(3)
GHZ = 2.4
for rows, cols in [(120, 3000), (250, 3000), (500, 3000), (1000, 3000), (2000, 3000), (250, 500), (500, 500)]:
    base_r = rng.randint(0, 20, rows).astype(np.uint8)
    base_c = rng.randint(0, 20, cols).astype(np.uint8)
    def mutate(b):
        x = b.copy(); m = rng.rand(len(x)) < 0.2; x[m] = rng.randint(0, 20, m.sum()); return x
    codes = [mutate(base_r), mutate(base_r), mutate(base_c), mutate(base_c)]
    tasks = np.array([[0, 1, 4], [2, 3, 5], [4, 5, 6]], np.int32)
    sd = np.full(4, 0.5, np.float32)
    ctx.tree_upload(codes, tasks, subm, scal, sd, flags=api.FLAG_TIMING)
    for _ in range(3): ctx.tree_run(); ctx.tree_sync()
    recs, _, _ = ctx.tree_download(want_gaps=False)
    tm = ctx.tree_timing()
    r = recs[-1]
    La, Lb = min(r.len_a, r.len_b), max(r.len_a, r.len_b)
    g = (int(tm[-1, 6]) >> 8) & 255
    out = []
    for l, (nsub, cp, cm) in enumerate(ctx.root_levels[:3]):
        prow = (La >> (l + 1)) + 1
        K = (prow + 127) // 128
        C = Lb >> l
        steps = C + min(64, (prow + 1) // 2) + (K - 1) * 127
        out.append("L%d n=%d pass %.0f us (%.0f cyc/step, K=%d) meet %.0f us" % (l, nsub, cp / GHZ / 1e3, cp / max(steps, 1), K, cm / GHZ / 1e3))
    print("%4d x %4d  (DP %d x %d, G %d): %s" % (rows, cols, La, Lb, g, "; ".join(out)), flush=True)
ctx.close()
