"""Where one `--precise` ensemble member spends its time (2048 x ~300 by default), phase by phase, and what 2/4/8 members
side by side on contexts sharing the GPU buy.  python tools/ensemble_phases.py [NSEQ LEN]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import kalign_amd
from kalign_amd import api, guide

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
length = int(sys.argv[2]) if len(sys.argv) > 2 else 300
inp = bench.workload_letters(nseq, length, False, 5)
order = sorted(range(len(inp)), key=lambda i: (-len(inp[i]), i))
seqs = [inp[i] for i in order]
tcodes, codes = guide.encode_tree(seqs, dna=False), guide.encode(seqs, dna=False)
subm, scal = bench.scoring(False)
scal = np.array(scal, np.float32)


def member(ctx, out=None, nthr=8):
    t = [time.perf_counter()]
    def lap(name):
        t.append(time.perf_counter())
        if out is not None:
            out.setdefault(name, []).append((t[-1] - t[-2]) * 1e3)
    tasks, sd = ctx.guide_tree(tcodes, n_threads=nthr)
    lap("guide_tree")
    ctx.msa_tree(codes, tasks, subm, scal, sd, n_anchors=5, weight=2.0)
    lap("msa_tree(5 anchors)")
    if out is not None:
        out.setdefault("  tree kernel ms (1st)", []).append(ctx.tree_kernel_ms()[0])
    ctx.tree_aligned_rows(seqs)
    lap("rows")
    tasks2, sd2 = ctx.aln_guide_tree()
    lap("aln_dist + upgma")
    ctx.tree_upload(codes, tasks2, subm, scal, sd2, flags=api.FLAG_DEVICE_GAPS | api.FLAG_KEEP_CONSISTENCY)
    lap("upload 2")
    ctx.tree_run(); ctx.tree_sync()
    lap("run 2")
    if out is not None:
        out.setdefault("  tree kernel ms (2nd)", []).append(ctx.tree_kernel_ms()[0])
        depth = {}
        for a, b, c in tasks2:
            depth[int(c)] = 1 + max(depth.get(int(a), 0), depth.get(int(b), 0))
        out["  upgma tree depth"] = [max(depth.values())]
        depth = {}
        for a, b, c in tasks:
            depth[int(c)] = 1 + max(depth.get(int(a), 0), depth.get(int(b), 0))
        out["  k-means tree depth"] = [max(depth.values())]
    ctx.tree_aligned_rows(seqs)
    lap("rows 2")


ctxs = [kalign_amd.Context(0, shared=True) for _ in range(8)]
alone = kalign_amd.Context(0)
for c in ctxs + [alone]:
    member(c)
for what, c in (("a context that has the GPU to itself", alone), ("a shared context (a launch per guide-tree level)", ctxs[0])):
    ph = {}
    for _ in range(3):
        member(c, ph)
    print("%d x %d, one member on %s, phases (ms, 3 runs):" % (nseq, length, what))
    for k, v in ph.items():
        print("  %-28s %s" % (k, " ".join("%8.2f" % x for x in v)))
for n in (1, 2, 4, 8):
    for nthr in (8, 1):
        th = [threading.Thread(target=member, args=(ctxs[k], None, nthr)) for k in range(n)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        print("%d members side by side (host threads per member %d): %.1f ms" % (n, nthr, (time.perf_counter() - t0) * 1e3))
