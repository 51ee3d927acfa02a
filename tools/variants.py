"""One job, many KA_* settings: ms per step, per-launch times and a parity check between the variants.
usage: variants.py NSEQ LEN DNA 'K=V,K=V;K=V;...'   (';' separates variants, an empty variant = the defaults).
VAR_COPIES=n: n independent copies of the job in flight as one forest (the saturation leg of bench.py under the variants).
VAR_ANCHORS=k: default mode -- the consistency table (k anchors, weight 2) is built once, the timed steps are the task tree with the bonus.
VAR_PAIRS=k: not a tree -- the N x k seq-seq batch of anchor consistency (ka_pairwise_batch) under every variant: kernel ms, GCUPS.
Run on the GPU box from the repo root."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KA_LAUNCH_EV"] = "1"
import numpy as np, bench, kalign_amd, torch
torch.cuda.init()

NSEQ, LEN, DNA = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
variants = sys.argv[4].split(";") if len(sys.argv) > 4 else [""]
STEPS = int(os.environ.get("VAR_STEPS", "8"))
ctx = kalign_amd.Context(0)
job = bench.make_job(ctx, NSEQ, LEN, DNA, seed=1)
subm, scal = bench.scoring(DNA)
COPIES = int(os.environ.get("VAR_COPIES", "1"))
PAIRS = int(os.environ.get("VAR_PAIRS", "0"))
if PAIRS > 0:
    codes = job["codes"]
    n = len(codes)
    ia = np.repeat(np.arange(n), PAIRS).astype(np.int32)
    ib = np.tile(np.arange(PAIRS), n).astype(np.int32)
    keep = ia != ib
    ia, ib = ia[keep], ib[keep]
    lens = np.array([len(c) for c in codes], np.int64)
    cells = float((lens[ia] * lens[ib]).sum())
    ref = None
    touched = set()
    for v in variants:
        for k in touched:
            os.environ.pop(k, None)
        touched = set()
        for kv in [x for x in v.split(",") if x]:
            k, val = kv.split("=")
            os.environ[k] = val
            touched.add(k)
        ctx.reload_env()
        best = 1e30
        for _ in range(3):
            out = ctx.pairwise_batch(codes, ia, ib, subm, scal[0], scal[1], scal[2])
            best = min(best, ctx.pairwise_kernel_ms())
        sig = (zlib.crc32(np.concatenate(out[0]).tobytes()), zlib.crc32(np.asarray(out[1]).tobytes()))
        if ref is None:
            ref = sig
        print("%-60s pairs %d  kernel %8.3f ms  %7.2f GCUPS  %s" % (v or "(defaults)", len(ia), best, cells / best / 1e6,
              "same result" if sig == ref else "RESULT DIFFERS"), flush=True)
    ctx.close()
    sys.exit(0)
if COPIES > 1:
    from kalign_amd import guide
    fc, ft, fd, _ = guide.forest([(job["codes"], job["tasks"], job["seq_distances"])] * COPIES)
    ctx.tree_upload(fc, ft, subm, scal, fd)
else:
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])
if int(os.environ.get("VAR_ANCHORS", "0")) > 0:
    ctx.tree_build_consistency(int(os.environ["VAR_ANCHORS"]), 2.0)
ref = None
touched = set()
for v in variants:
    for k in touched:
        os.environ.pop(k, None)
    touched = set()
    for kv in [x for x in v.split(",") if x]:
        k, val = kv.split("=")
        os.environ[k] = val
        touched.add(k)
    ctx.reload_env()
    for _ in range(2):
        ctx.tree_run(); ctx.tree_sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        ctx.tree_run()
    ctx.tree_sync()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / STEPS * 1e3
    recs, paths, gaps = ctx.tree_download()
    sig = (zlib.crc32(np.ascontiguousarray(paths).tobytes()), zlib.crc32(np.concatenate(gaps).tobytes()),
           tuple((r.plen, r.meet, r.transition, r.score) for r in recs[-4:]))
    if ref is None:
        ref = sig
    cells = float(sum(r.len_a * r.len_b for r in recs))
    print("%-60s %8.3f ms/step  %7.2f GCUPS  launches %s  fallback %d  %s" % (
        v or "(defaults)", ms, cells / ms / 1e6, ["%.2f" % x for x in ctx.tree_launch_ms()], ctx.fallback_runs(),
        "same result" if sig == ref else "RESULT DIFFERS"), flush=True)
ctx.close()
