"""bench.py's saturation leg alone, several times over, with the context's fallback counter: a 16-tree forest that falls back to the
per-level plan (19 launches) shows here.  python tools/saturation_check.py [REPEATS 3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
torch.cuda.init()
import bench, kalign_amd
from kalign_amd import guide
ctx0 = kalign_amd.Context(0)
job = bench.make_job(ctx0, 4096, 400, False, seed=1)
ctx0.close()
subm, scal = bench.scoring(False)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ctx = kalign_amd.Context(0)
    for n in (1, 2, 4, 8, 16):
        fc, ft, fd, _ = guide.forest([(job["codes"], job["tasks"], job["seq_distances"])] * n)
        ctx.tree_upload(fc, ft, subm, scal, fd)
        t0 = time.perf_counter(); ctx.tree_run(); ctx.tree_sync(); first = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(2):
            ctx.tree_run()
        ctx.tree_sync()
        dt = (time.perf_counter() - t0) / 2
        print("rep %d  %2d trees: first run %.1f ms, then %.1f ms per round, launches %d, fallbacks so far %d" % (rep, n, first * 1e3, dt * 1e3, ctx.tree_kernel_ms()[1], ctx.fallback_runs()), flush=True)
    ctx.close()
