"""KA_TP switched on a live context (ka_debug_reload_env): time of every run, launches, tasks the chained launch took over, fallbacks.
usage: tp_switch.py [COPIES 16] [SEQUENCE e.g. 0,1,0,1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
torch.cuda.init()
import bench, kalign_amd
from kalign_amd import guide
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seq = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,0,1").split(",")]
os.environ["KA_TP"] = str(seq[0])
ctx = kalign_amd.Context(0)
job = bench.make_job(ctx, 4096, 400, False, seed=1)
subm, scal = bench.scoring(False)
fc, ft, fd, _ = guide.forest([(job["codes"], job["tasks"], job["seq_distances"])] * n)
ctx.tree_upload(fc, ft, subm, scal, fd)
for tp in seq:
    os.environ["KA_TP"] = str(tp)
    ctx.reload_env()
    for rep in range(3):
        t0 = time.perf_counter(); ctx.tree_run(); ctx.tree_sync(); dt = time.perf_counter() - t0
        print("KA_TP=%d run %d: %.1f ms, launches %d, helped %d, fallbacks so far %d" % (tp, rep, dt * 1e3, ctx.tree_kernel_ms()[1], ctx.helped_tasks(), ctx.fallback_runs()), flush=True)
ctx.close()
