"""One profile-profile task whose passes are lone strips (DP 120 x 3000 by default: one 60-row strip per direction), run
REPS times: the job behind the per-step instruction / stall counters (rocprofv3 --pmc) of profiles/r03b_one_strip_*.
usage: one_strip.py [rows 120] [cols 3000] [reps 20].  Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd, torch
torch.cuda.init()
from kalign_amd import api
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
rng = np.random.RandomState(3)
base_r = rng.randint(0, 20, rows).astype(np.uint8)
base_c = rng.randint(0, 20, cols).astype(np.uint8)
def mutate(b):
    x = b.copy(); m = rng.rand(len(x)) < 0.2; x[m] = rng.randint(0, 20, m.sum()); return x
codes = [mutate(base_r), mutate(base_r), mutate(base_c), mutate(base_c)]
tasks = np.array([[0, 1, 4], [2, 3, 5], [4, 5, 6]], np.int32)
ctx.tree_upload(codes, tasks, subm, scal, np.full(4, 0.5, np.float32), flags=api.FLAG_TIMING)
for _ in range(reps):
    ctx.tree_run(); ctx.tree_sync()
tm = ctx.tree_timing()
print("reps %d  root task: levels (n, pass us, meet us) %s" % (reps, [(int(n), round(cp / 2.4e3, 1), round(cm / 2.4e3, 1)) for n, cp, cm in ctx.root_levels[:6]]))
ctx.close()
