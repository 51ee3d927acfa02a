import os, sys
sys.path.insert(0, os.getcwd())
os.environ["KA_PLAN_VERBOSE"] = "1"
import bench, kalign_amd
ctx = kalign_amd.Context(0)
job = bench.make_job(ctx, 4096, 2000, True, seed=1)
subm, scal = bench.scoring(True)
ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])
ctx.close()
