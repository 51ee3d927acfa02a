"""The realignment pass at C5's member size (2048 x ~300 aa): device timings next to the reference's (serial) ones."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import kalign_amd
from kalign_amd import api, guide, synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
with_ref = len(sys.argv) > 3 and sys.argv[3] == "ref"
seqs = synth.dssim(n, L, seed=1)
order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
srt = [seqs[i] for i in order]
tcodes = guide.encode_tree(srt); codes = guide.encode(srt)
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
for rep in range(2):
    t = []; t0 = [time.perf_counter()]
    def lap(name):
        now = time.perf_counter(); t.append((name, (now - t0[0]) * 1e3)); t0[0] = now
    tasks, sd = ctx.guide_tree(tcodes, n_threads=16); lap("guide")
    ctx.msa_tree(codes, tasks, subm, scal, sd); lap("align1")
    rows = ctx.tree_aligned_rows(srt); lap("rows1")
    tasks2, sd2 = ctx.aln_guide_tree(); lap("aln_tree")
    ctx.tree_upload(codes, tasks2, subm, scal, sd2, flags=api.FLAG_DEVICE_GAPS | api.FLAG_KEEP_CONSISTENCY); ctx.tree_run(); ctx.tree_sync(); lap("align2")
    rows2 = ctx.tree_aligned_rows(srt); lap("rows2")
    print(" ".join("%s=%.1f" % x for x in t), "alnlen %d -> %d" % (len(rows[0]), len(rows2[0])), flush=True)
if with_ref:
    from oracle import refdrv
    job = refdrv.RefJob(seqs, n_threads=16)
    t0 = time.perf_counter(); job.run_tree(); a1 = time.perf_counter() - t0
    r, _, sdist, stree = job.realign_tree(want_dm=False)
    t0 = time.perf_counter(); job.run_tree(); a2 = time.perf_counter() - t0
    fin = job.finalise()
    got = [None] * n
    for k, i in enumerate(order):
        got[i] = rows2[k].decode()
    print("reference: tree %.0f ms align1 %.0f ms aln_dist %.0f ms upgma %.0f ms align2 %.0f ms; identical rows: %s" % (
        job.tree_seconds * 1e3, a1 * 1e3, sdist * 1e3, stree * 1e3, a2 * 1e3, got == fin), flush=True)
