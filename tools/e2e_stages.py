"""Wall time of every stage of the end-to-end leg (host buffers), 1024 x 400 protein, default mode."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import kalign_amd
from kalign_amd import api, guide, synth
import bench

seqs = synth.dssim(1024, 400, seed=1)
order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
seqs = [seqs[i] for i in order]
tcodes = guide.encode_tree(seqs); codes = guide.encode(seqs)
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
for rep in range(4):
    t = []
    def lap(name, t0=[time.perf_counter()]):
        now = time.perf_counter(); t.append((name, (now - t0[0]) * 1e3)); t0[0] = now
    lap("-")
    tasks, sd = ctx.guide_tree(tcodes, n_threads=16); lap("guide")
    ctx.tree_upload(codes, tasks, subm, scal, sd, flags=api.FLAG_DEVICE_GAPS); lap("upload")
    ctx.tree_build_consistency(5, 2.0); lap("cons")
    ctx.tree_run(); ctx.tree_sync(); lap("run")
    recs, paths, gaps = ctx.tree_download(); lap("download")
    rows = ctx.tree_aligned_rows(seqs); lap("rows")
    print(" ".join("%s=%.2f" % x for x in t[1:]))
