"""Refinement on the device against the reference's refine_alignment: time of ka_tree_refine (modes 1, 2, 3, 4) next to
the first pass, and -- for sizes the reference finishes -- refine_alignment itself on the host, results compared.
Usage: python tools/refine_time.py [nseq len [cpu]]   (GPU box; oracle/_ref for the cpu leg)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kalign_amd  # noqa: E402
from kalign_amd import guide  # noqa: E402


def level_of(tasks, nseq):
    lvl = np.zeros(2 * nseq - 1, np.int32)
    out = np.zeros(len(tasks), np.int32)
    for t, (a, b, c) in enumerate(tasks):
        lvl[c] = max(lvl[a], lvl[b]) + 1
        out[t] = lvl[c]
    return out


def main():
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    cpu = len(sys.argv) > 3 and sys.argv[3] == "cpu"
    ctx = kalign_amd.Context(0)
    job = bench.make_job(ctx, nseq, length, False, 1)
    subm, scal = bench.scoring(False)
    from kalign_amd import api
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"], flags=api.FLAG_TIMING)
    for _ in range(2):
        ctx.tree_run()
        ctx.tree_sync()
    ms0, _ = ctx.tree_kernel_ms()
    recs, _, gaps0 = ctx.tree_download()
    cells = float(sum(r.len_a * r.len_b for r in recs))
    print("first pass %dx%d: %.2f ms kernels, %.1f GCUPS" % (nseq, length, ms0, cells / ms0 / 1e6))
    out = {}
    for mode, name in ((4, "depth-first first pass"), (1, "refine all"), (2, "refine confident"), (3, "inline")):
        for _ in range(2):
            t0 = time.perf_counter()
            ctx.tree_refine(mode)
            ctx.tree_sync()
            wall = (time.perf_counter() - t0) * 1e3
        ms, nl = ctx.tree_kernel_ms()
        r, _, g = ctx.tree_download()
        out[mode] = (r, g)
        changed = sum(not np.array_equal(a, b) for a, b in zip(g, gaps0))
        print("mode %d (%s): %.1f ms kernels (last launch sequence, %d launches), %.1f ms wall incl. set-up; alnlen %d; "
              "sequences whose gaps differ from the first pass: %d" % (mode, name, ms, nl, wall, r[-1].plen, changed))
        tm = ctx.tree_timing()
        if tm is not None:
            # the finishing member of every task; longest task of every tree level, summed over the levels
            tm = np.asarray(tm)[:, :7].astype(np.float64) / 2400.0   # s_memtime at 2.4 GHz (as tools/levels_real.py)
            lv = level_of(job["tasks"], nseq)
            tot = np.zeros(7)
            for L in range(1, lv.max() + 1):
                idx = np.where(lv == L)[0]
                top = idx[np.argmax(tm[idx].sum(axis=1))]
                tot += tm[top]
                if mode == 1 and os.environ.get("REFINE_LEVELS"):
                    print("   L%2d n=%4d longest: %5dx%-5d nsip %4d+%-4d k%d  us: prep %4.0f sp tables %5.0f recursions %6.0f coding %4.0f sp scoring %5.0f waiting %4.0f record+merge %4.0f" % (
                        (L, len(idx), r[top].len_a, r[top].len_b, r[top].nsip_a, r[top].nsip_b, r[top].kind) + tuple(tm[top])))
            print("   longest task per level, summed (us): prep %.0f  sp tables %.0f  recursions %.0f  path coding %.0f  sp scoring %.0f  "
                  "waiting %.0f  record+merge %.0f" % tuple(tot))
    if cpu:
        from oracle import refdrv
        for nt in (16, 1):
            j = refdrv.EncodedJob(job["codes"], job["tasks"], job["seq_distances"], biotype=0, type_=-1, n_threads=nt)
            _, secs = j.run_tree()
            t0 = time.perf_counter()
            g, cb, ca, plen = j.refine(1)
            rs = time.perf_counter() - t0
            j.close()
            same = all(np.array_equal(a, b) for a, b in zip(g, out[1][1]))
            print("reference, %d threads: create_msa_tree %.2f s, refine_alignment(ALL) %.2f s; gaps identical to the device: %s"
                  % (nt, secs, rs, same))
    ctx.close()


if __name__ == "__main__":
    main()
