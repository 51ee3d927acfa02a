"""Wall time of the host-facing stages of one job (upload incl. host-side task preparation, run, download) --
arguments: [nseq 4096] [len 400].  Run on the GPU box from the repo root."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
import bench, kalign_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = kalign_amd.Context(0)
job = bench.make_job(ctx, n, L, False, 1)
subm, scal = bench.scoring(False)
for rep in range(3):
    t0 = time.perf_counter(); ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"]); t1 = time.perf_counter()
    ctx.tree_run(); ctx.tree_sync(); t2 = time.perf_counter()
    recs, paths, _ = ctx.tree_download(want_gaps=False); t3 = time.perf_counter()
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"], flags=4); ctx.tree_run(); ctx.tree_sync(); t4 = time.perf_counter()
    recs, paths, gaps = ctx.tree_download(want_gaps=True); t5 = time.perf_counter()
    print("upload %.2f ms  run+sync %.2f  download(no gaps) %.2f   | device-gaps job: download with gaps %.2f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t5-t4)*1e3))
