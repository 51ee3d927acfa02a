"""ka_aln_guide_tree (pairwise distances of aligned rows + UPGMA on the device) timed with the merges in one workgroup
and with one launch per merge.  usage: upgma_time.py [N ...]   (run on the GPU box from the repo root)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, kalign_amd

ctx = kalign_amd.Context(0)
for n in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096]:
    rng = np.random.RandomState(n)
    base = rng.choice(list(b"ACDEFGHIK-"), size=(n // 2, 400)).astype(np.uint8)
    rows = base[rng.randint(0, len(base), size=n)].copy()
    flip = rng.random_sample(rows.shape) < 0.1
    rows[flip] = rng.choice(list(b"ACDEFGHIK-"), size=int(flip.sum())).astype(np.uint8)
    rows = [bytes(r) for r in rows]
    out = {}
    for mode in ("one workgroup", "launch per merge"):
        if mode == "launch per merge":
            os.environ["KA_UPGMA_LAUNCHES"] = "1"
        else:
            os.environ.pop("KA_UPGMA_LAUNCHES", None)
        ctx.reload_env()
        ctx.aln_guide_tree(rows)
        t0 = time.perf_counter()
        for _ in range(3):
            tasks, sd = ctx.aln_guide_tree(rows)
        out[mode] = ((time.perf_counter() - t0) / 3 * 1e3, tasks)
    same = np.array_equal(out["one workgroup"][1], out["launch per merge"][1])
    print("N=%5d  whole call (upload, distances, UPGMA, download): one workgroup %8.2f ms   launch per merge %8.2f ms   %s" % (
        n, out["one workgroup"][0], out["launch per merge"][0], "same tree" if same else "TREES DIFFER"), flush=True)
ctx.close()
