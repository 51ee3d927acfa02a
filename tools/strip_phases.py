"""Where a strip spends its time (KA_PROF build of unit 0: hipcc ... -DKA_UNIT=0 -DKA_PROF, linked as libkalign_amd_prof.so and copied
over libkalign_amd.so on the GPU box): per wave of the leading workgroup of a synthetic profile-profile task -- loop start / end,
steps and cycles inside the branch-free step pairs, the event steps of the steady phase and what they spend where.
KA_MAX_CLUSTER=1 puts all strips of both passes on the eight waves of one workgroup.  Run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench, kalign_amd, torch
torch.cuda.init()
from kalign_amd import api
subm, scal = bench.scoring(False)
ctx = kalign_amd.Context(0)
rng = np.random.RandomState(3)
JOBS = []
# PHASES_SHAPES=618x451,917x825: other task shapes than the two built in
SHAPES = [tuple(int(v) for v in x.split("x")) for x in os.environ["PHASES_SHAPES"].split(",")] if os.environ.get("PHASES_SHAPES") else [(1000, 3000), (500, 500)]
for rows, cols in SHAPES:
    base_r = rng.randint(0, 20, rows).astype(np.uint8); base_c = rng.randint(0, 20, cols).astype(np.uint8)
    def mutate(b):
        x = b.copy(); m = rng.rand(len(x)) < 0.2; x[m] = rng.randint(0, 20, m.sum()); return x
    JOBS.append((rows, cols, [mutate(base_r), mutate(base_r), mutate(base_c), mutate(base_c)]))
tasks = np.array([[0, 1, 4], [2, 3, 5], [4, 5, 6]], np.int32)
if os.environ.get("PHASES_REAL"):
    # the root task of the headline tree (4096 x 400, the reference's guide tree): the strips of its leading workgroup
    rj = bench.make_job(ctx, 4096, 400, False, seed=1)
    JOBS = [(0, 0, None)]
# KA_HW=1 (default): strips with helper waves (ka_wstrip.h) -- their counters: cycles in the octets (waits included), waits and
# cycles waited for operands, head and tail; KA_HW=0: ka_strip and its event steps
for hw in os.environ.get("PHASES_HW", "1,0").split(","):
    os.environ["KA_HW"] = hw
    ctx.reload_env()
    print("== KA_HW=%s%s" % (hw, " KA_MAX_CLUSTER=" + os.environ["KA_MAX_CLUSTER"] if "KA_MAX_CLUSTER" in os.environ else ""))
    for rows, cols, codes in JOBS:
        if codes is None:
            ctx.tree_upload(rj["codes"], rj["tasks"], subm, scal, rj["seq_distances"], flags=api.FLAG_TIMING)
        else:
            ctx.tree_upload(codes, tasks, subm, scal, np.full(4, 0.5, np.float32), flags=api.FLAG_TIMING)
        for _ in range(3): ctx.tree_run(); ctx.tree_sync()
        ctx.tree_timing()
        print("%d x %d: " % (rows, cols) + ", ".join("level %d: n %d pass %.0f us meet %.0f us" % (l, ctx.root_levels[l][0], ctx.root_levels[l][1] / 2.4e3, ctx.root_levels[l][2] / 2.4e3) for l in range(8) if ctx.root_levels[l][0]))
        for lvl in range(int(os.environ.get("PHASES_LEVELS", "1"))):
            for w in range(8):
                p = ctx.prof[lvl][w]
                if p[5] == 0: continue
                total = p[3] - p[2]
                x = ctx.prof[lvl + 4][w]
                if hw != "0":
                    print("  L%d wave %d: loop start at %.1f us, end at %.1f us (barrier done %.1f) | loop %.1f us: octets %d steps at %.0f cyc/step (%d waits, %d of them for the row above, %.0f cycles each); head %d cycles, tail %d, rest %d" % (
                        lvl, w, (p[2] - p[0]) / 2.4e3, (p[3] - p[0]) / 2.4e3, (p[4] - p[0]) / 2.4e3, total / 2.4e3, p[7], p[6] / max(p[7], 1),
                        x[3], x[1], x[0] / max(x[3], 1), x[2], x[4], total - p[6] - x[2] - x[4]))
                    continue
                print("  L%d wave %d: loop start at %.1f us, end at %.1f us (barrier done %.1f) | loop %.1f us: pair loops %d steps at %.0f cyc/step; everything else %d cycles" % (
                    lvl, w, (p[2] - p[0]) / 2.4e3, (p[3] - p[0]) / 2.4e3, (p[4] - p[0]) / 2.4e3, total / 2.4e3, p[7], p[6] / max(p[7], 1), total - p[6]))
                print("      steady phase: %d event steps at %.0f cycles each" % (x[3], x[2] / max(x[3], 1)))
                print("      in the event steps: ring issue %.0f cycles per event, collection + flush %.0f per event; top wait %.0f, start..chain end %.0f, vmcnt wait %.0f" % (x[6] / max(x[3], 1), x[7] / max(x[3], 1), x[0] / max(x[3], 1), x[4] / max(x[3], 1), x[1] / max(x[3], 1)))
ctx.close()
