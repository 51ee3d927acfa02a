import os, sys
sys.path.insert(0, os.getcwd())
os.environ["KA_PLAN_VERBOSE"] = "1"
import bench, kalign_amd
ctx = kalign_amd.Context(0)
import sys
n, l, dna = (int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))) if len(sys.argv) > 3 else (4096, 2000, True)
job = bench.make_job(ctx, n, l, dna, seed=1)
subm, scal = bench.scoring(dna)
ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])
ctx.close()
