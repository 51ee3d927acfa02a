"""Mean per-task phase times (KA_FLAG_TIMING) by task kind and launch of the headline tree: prep / passes / meetups / path coding / merge, us.
arguments: [nseq 4096] [len 400] [dna 0].  Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd
from kalign_amd import api
NSEQ = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
LEN = int(sys.argv[2]) if len(sys.argv) > 2 else 400
DNA = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
codes, tasks, dist = bench.make_workload(NSEQ, LEN, DNA, 1)
subm, scal = bench.scoring(DNA)
ctx = kalign_amd.Context(0)
ctx.tree_upload(codes, tasks, subm, scal, dist, flags=api.FLAG_TIMING)
for setting in (sys.argv[4].split(";") if len(sys.argv) > 4 else [""]):       # 'K=V,K=V;K=V': one table per setting of KA_* switches
    for kv in filter(None, setting.split(",")):
        os.environ[kv.split("=")[0]] = kv.split("=")[1]
    ctx.reload_env()
    print("--", setting or "defaults")
    for _ in range(3):
        ctx.tree_run(); ctx.tree_sync()
    recs, _, _ = ctx.tree_download(want_gaps=False)
    tm = ctx.tree_timing()[:len(recs)] / 2.4e3
    kind = np.array([r.kind for r in recs]); G = (tm[:, 6] * 2.4e3).astype(np.int64) >> 8 & 255
    print("kernel ms", ctx.tree_kernel_ms())
    for k, name in ((0, "seq-seq"), (1, "seq-profile"), (2, "profile-profile")):
        for lo, hi, what in ((1, 1, "one workgroup"), (2, 255, "clusters")):
            m = (kind == k) & (G >= lo) & (G <= hi)
            if not m.any():
                continue
            print("%-16s %-14s n=%5d  prep %6.1f  passes %6.1f  meetups %5.1f  coding %5.1f  merge %5.1f  total %6.1f us" % (
                name, what, m.sum(), tm[m, 0].mean(), tm[m, 4].mean(), tm[m, 5].mean(), tm[m, 2].mean(), tm[m, 3].mean(), tm[m, :4].sum(1).mean()))
ctx.close()
