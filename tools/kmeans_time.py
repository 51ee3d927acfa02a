"""ka_guide_tree with the 2-means bisection on the host and on the device (KA_KMEANS=0 / 1): wall time of the whole call and of
the bisection inside it, and whether the two trees are the same.  usage: kmeans_time.py [NSEQ LEN ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd
from kalign_amd import api, guide
ctx = kalign_amd.Context(0)
args = [int(x) for x in sys.argv[1:]] or [4096, 400, 16384, 500]
for nseq, length in zip(args[::2], args[1::2]):
    inp = bench.workload_letters(nseq, length, False, 1)
    order = sorted(range(len(inp)), key=lambda i: (-len(inp[i]), i))
    tcodes = guide.encode_tree([inp[i] for i in order], dna=False)
    res = {}
    for mode in ("0", "1"):
        os.environ["KA_KMEANS"] = mode
        ctx.guide_tree(tcodes, n_threads=bench.host_threads())
        t0 = time.perf_counter()
        tasks, sd = ctx.guide_tree(tcodes, n_threads=bench.host_threads())
        ms = (time.perf_counter() - t0) * 1e3
        bis, dev = api.guide_last_bisect()
        res[mode] = (tasks, sd)
        print("%d x %d  KA_KMEANS=%s: guide tree %.1f ms, bisection %.1f ms (%s)" % (nseq, length, mode, ms, bis, "device" if dev else "host"), flush=True)
    print("   same tree:", bool(np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1].view(np.uint32), res["1"][1].view(np.uint32))))
ctx.close()
