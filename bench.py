#!/usr/bin/env python
"""bench.py -- GCUPS of the progressive-alignment hot path on MI355X.

One "step" = one pass of the hot path over one synthetic N x L sequence set: every task of the
guide tree (seq-seq, seq-profile and profile-profile Gotoh/Hirschberg DP, profile merge, path
coding) through ka_tree_run(), with sequences, task list and scoring tables already resident in
HBM.  GCUPS = sum over tasks of len_a*len_b ("useful" cell updates) / time.

    python bench.py [--gpus N --steps K --warmup W] [--nseq 1024 --len 400 --dna]

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank aligns its own
independent sequence set of the same shape (the unit the path partitions into without a
data-path collective) -> weak scaling; the timing is barrier-bracketed and max-reduced.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(recs):
    """per task D*(wa*La + wb*Lb + 48*Lb) + 4*(La+Lb+2) + 256*(alnlen+2) [merged
    profile written on the device; not for the root], D = ceil(log2 La)+1, w = 1 B/position for a
    sequence operand, 256 B/position for a profile operand; La/Lb = DP rows/cols."""
    total = 0.0
    for i, r in enumerate(recs):
        la, lb = (r.len_b, r.len_a) if r.swapped else (r.len_a, r.len_b)
        wa = 1 if r.kind == 0 else 256
        wb = 256 if r.kind == 2 else 1
        depth = math.ceil(math.log2(max(la, 2))) + 1
        total += depth * (wa * la + wb * lb + 48 * lb) + 4 * (la + lb + 2)
        if i != len(recs) - 1:
            total += 256 * (r.plen + 2)
    return total


def pmc_traffic(args):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE collected in separate passes for exactly this workload, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the profiler on itself, so it
    reports the committed measurement for the default workload and null for any other."""
    if (args.nseq, args.len, args.dna) != (1024, 400, False):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            return float(json.load(fh)["hbm_bytes_per_launch_corrected"])
    except Exception:
        return None


def workload_letters(nseq, length, dna, seed):
    from kalign_amd import synth
    # the reference's own benchmark generator, restated: independent samples of one profile HMM
    # (tests/dssim.c)
    return synth.dssim(nseq, length, dna=dna, seed=seed)


def make_workload(nseq, length, dna, seed, seqs=None):
    from kalign_amd import guide
    if seqs is None:
        seqs = workload_letters(nseq, length, dna, seed)
    codes = guide.encode(seqs, dna=dna)
    tasks = guide.bisecting_tree(nseq, seed=seed)
    dist = np.random.RandomState(seed).uniform(0.3, 0.9, nseq).astype(np.float32)
    return codes, tasks, dist


def scoring(dna):
    """Default matrices of the reference (aln_param.c): PFASUM43 7/1.25/1 for protein,
    the +5/-4 8/6/0 DNA matrix for --type dna; tables are data fixtures dumped from the reference."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "param_tables.npz"))
    key = "1_0" if dna else "0_3"
    return z["subm_" + key], z["scal_" + key].copy()


def cpu_baseline(codes, tasks, dist, dna, cells):
    """The reference's own dispatcher (oracle/_ref = the real Kalign sources compiled as they
    lie) on the host cores, same task list, bounded to ~10-30 s of CPU work."""
    ncores = os.cpu_count() or 1
    try:
        from oracle import refdrv
        if not refdrv.available():
            raise RuntimeError("no oracle/_ref")
        best = None
        tried = []
        budget = time.time() + 25.0
        for nt in sorted({ncores, min(ncores, 16), 1}, reverse=True):
            reps = 0
            while reps < 2 and time.time() < budget:
                job = refdrv.EncodedJob(codes, tasks, dist, biotype=1 if dna else 0, type_=0 if dna else -1, n_threads=nt)
                _, secs = job.run_tree()
                job.close()
                reps += 1
                tried.append((nt, secs))
                if best is None or secs < best[1]:
                    best = (nt, secs)
        return {"value": cells / best[1] / 1e9, "unit": "GCUPS", "cores": best[0], "kind": "reference",
                "sample": "full workload, create_msa_tree of the reference (OpenMP), best of %s (threads, s)" % (
                    ["%d:%.3f" % t for t in tried])}
    except Exception as e:  # pragma: no cover - only when the prebuilt reference is missing
        from oracle import oracledrv
        n = min(len(codes), 128)
        from kalign_amd import guide
        sub_tasks = guide.bisecting_tree(n, seed=1)
        subm, scal = scoring(dna)
        t0 = time.time()
        recs, _, _, _ = oracledrv.msa_tree(codes[:n], sub_tasks, subm, scal, dist[:n])
        secs = time.time() - t0
        c = sum(r.len_a * r.len_b for r in recs)
        return {"value": c / secs / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
                "sample": "first %d sequences, oracle C restatement, 1 thread (%s)" % (n, e)}


def pairwise_leg(ctx, codes, subm, scal, args, k_anchors=5):
    """N x K independent seq-seq alignments (every sequence against K anchors), full Hirschberg +
    path coding per pair.  Reports the kernel rate (HIP events) and the host-buffer wall rate."""
    n = len(codes)
    ia = np.repeat(np.arange(n), k_anchors).astype(np.int32)
    ib = np.tile(np.arange(k_anchors), n).astype(np.int32)
    keep = ia != ib
    ia, ib = ia[keep], ib[keep]
    lens = np.array([len(c) for c in codes], np.int64)
    cells = float((lens[ia] * lens[ib]).sum())
    ctx.pairwise_batch(codes, ia, ib, subm, scal[0], scal[1], scal[2])           # warm-up (allocations)
    t0 = time.perf_counter()
    ctx.pairwise_batch(codes, ia, ib, subm, scal[0], scal[1], scal[2])
    wall = time.perf_counter() - t0
    kms = ctx.pairwise_kernel_ms()
    depth = math.ceil(math.log2(max(int(lens.min()), 2))) + 1
    abytes = float((depth * (np.minimum(lens[ia], lens[ib]) + 49 * np.maximum(lens[ia], lens[ib])) + 4 * (lens[ia] + lens[ib] + 2)).sum())
    info = {"pairs": int(len(ia)), "useful_cells": cells, "kernel_ms": kms, "gcups_kernel": cells / kms / 1e6,
            "wall_ms_host_buffers": wall * 1e3, "gcups_wall_host_buffers": cells / wall / 1e9,
            "roofline_frac_hbm": abytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if not args.no_cpu:
        try:
            from oracle import refdrv
            m = min(len(ia), 256)
            nt = min(os.cpu_count() or 1, 16)
            _, secs = refdrv.pairwise_batch(codes, ia[:m], ib[:m], subm, float(scal[0]), float(scal[1]), float(scal[2]),
                                            n_threads=nt, want_paths=False)
            c = float((lens[ia[:m]] * lens[ib[:m]]).sum())
            info["cpu_reference_gcups"] = c / secs / 1e9
            info["cpu_reference_sample"] = "%d pairs, aln_runner of the reference, %d OpenMP threads (the reference itself runs this loop serially)" % (m, nt)
        except Exception as e:      # pragma: no cover
            info["cpu_reference_gcups"] = None
            info["cpu_reference_sample"] = str(e)
    return info


def default_mode_leg(ctx, codes, tasks, subm, scal, seq_dist, args, k_anchors=5, weight=2.0):
    """The reference CLI's default mode: anchor_consistency_build (N x K seq-seq alignments -> position
    maps) followed by the guide tree with the consistency bonus in every DP.  Outside the timed region
    of the headline metric; reports wall times with host buffers for the build and HBM-resident runs
    for the tree, next to the reference doing the same on the host cores."""
    lens = np.array([len(c) for c in codes], np.int64)
    ctx.tree_upload(codes, tasks, subm, scal, seq_dist)
    ctx.tree_build_consistency(k_anchors, weight)                  # warm-up (allocations)
    t0 = time.perf_counter()
    ctx.tree_build_consistency(k_anchors, weight)
    build_s = time.perf_counter() - t0
    ids, _ = ctx.tree_consistency()
    pair_cells = float(sum(int(lens[i]) * int(lens[a]) for i in range(len(codes)) for a in ids if a != i))
    ctx.tree_run(); ctx.tree_sync()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.tree_run()
    ctx.tree_sync()
    tree_s = (time.perf_counter() - t0) / reps
    recs, _, _ = ctx.tree_download(want_gaps=False)
    tree_cells = float(sum(r.len_a * r.len_b for r in recs))
    info = {"anchors": k_anchors, "weight": weight,
            "build_consistency_ms_host_buffers": build_s * 1e3, "tree_ms": tree_s * 1e3,
            "useful_cells": pair_cells + tree_cells,
            "gcups_tree": tree_cells / tree_s / 1e9,
            "gcups_total": (pair_cells + tree_cells) / (build_s + tree_s) / 1e9}
    if not args.no_cpu:
        try:
            from oracle import refdrv
            nt = min(os.cpu_count() or 1, 16)
            job = refdrv.EncodedJob(codes, tasks, seq_dist, biotype=1 if args.dna else 0, type_=0 if args.dna else -1, n_threads=nt)
            t0 = time.perf_counter()
            job.build_consistency(k_anchors, weight)
            cb = time.perf_counter() - t0
            _, ct = job.run_tree()
            job.close()
            info["cpu_reference"] = {"build_consistency_ms": cb * 1e3, "tree_ms": ct * 1e3,
                                     "gcups_total": (pair_cells + tree_cells) / (cb + ct) / 1e9, "threads": nt,
                                     "note": "the reference builds the position maps serially (anchor_consistency.c:246-267); the tree uses OpenMP tasks"}
        except Exception as e:      # pragma: no cover
            info["cpu_reference"] = str(e)
    return info


def end_to_end_leg(ctx, seqs, subm, scal, args, k_anchors=5, weight=2.0):
    """kalign_run's whole alignment phase from letters to aligned rows, host buffers in and out: the guide tree
    (build_tree_kmeans: two distance batches on the device, 2-means / UPGMA on the host), anchor consistency, the
    task tree on that guide tree, finalise_alignment on the device -- next to the reference doing the same on the
    host cores.  The guide tree here is the reference's own k-means tree, not the synthetic one of the headline."""
    from kalign_amd import api, guide
    # the order kalign_run gives the sequences before it builds the tree (msa_sort_len_name, msa_sort.c:62-80:
    # longest first, ties by name = input index here)
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))
    inp = seqs
    seqs = [inp[i] for i in order]
    tcodes = guide.encode_tree(seqs, dna=args.dna)
    codes = guide.encode(seqs, dna=args.dna)
    nt = min(os.cpu_count() or 1, 16)
    got = {}

    def once():
        t = {}
        t0 = time.perf_counter()
        tasks, sd = ctx.guide_tree(tcodes, n_threads=nt)
        t["guide_tree_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        ctx.tree_upload(codes, tasks, subm, scal, sd, flags=api.FLAG_DEVICE_GAPS)
        ctx.tree_build_consistency(k_anchors, weight)
        ctx.tree_run()
        recs, _, _ = ctx.tree_download()
        t["align_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        rows = ctx.tree_aligned_rows(seqs)
        t["rows_ms"] = (time.perf_counter() - t0) * 1e3
        t["total_ms"] = t["guide_tree_ms"] + t["align_ms"] + t["rows_ms"]
        t["alnlen"] = len(rows[0])
        got["rows"] = rows
        return t

    once()                                                         # warm-up (allocations)
    info = once()
    info["anchors"] = k_anchors
    info["host_threads_for_the_bisection"] = nt
    if not args.no_cpu:
        try:
            from oracle import refdrv
            t0 = time.perf_counter()
            job = refdrv.RefJob(inp, type_=0 if args.dna else -1, n_threads=nt)
            prep = time.perf_counter() - t0
            t0 = time.perf_counter()
            job.build_consistency(k_anchors, weight)
            job.run_tree()
            aln = time.perf_counter() - t0
            t0 = time.perf_counter()
            rows = job.finalise()
            fin = time.perf_counter() - t0
            ours = [None] * len(inp)
            for k, i in enumerate(order):
                ours[i] = got["rows"][k].decode()
            info["rows_identical_to_reference"] = bool(ours == rows)
            info["cpu_reference"] = {"guide_tree_ms": job.tree_seconds * 1e3, "align_ms": aln * 1e3, "rows_ms": fin * 1e3,
                                     "total_ms": (job.tree_seconds + aln + fin) * 1e3, "alnlen": len(rows[0]), "threads": nt,
                                     "input_checks_and_encoding_ms": (prep - job.tree_seconds) * 1e3}
            job.close()
        except Exception as e:      # pragma: no cover
            info["cpu_reference"] = str(e)
    return info


def realign_leg(ctx, seqs, subm, scal, args):
    """One member run of `--precise`: kalign_run_realign with one iteration, fast mode (aln_wrap.c:361-527) -- guide
    tree, alignment, rows, identity distances + UPGMA tree from those rows (both on the device), alignment on that
    tree, rows -- next to the reference doing the same on the host cores."""
    from kalign_amd import api, guide
    order = sorted(range(len(seqs)), key=lambda i: (-len(seqs[i]), i))       # msa_sort_len_name
    inp = seqs
    seqs = [inp[i] for i in order]
    tcodes = guide.encode_tree(seqs, dna=args.dna)
    codes = guide.encode(seqs, dna=args.dna)
    nt = min(os.cpu_count() or 1, 16)
    got = {}

    def once():
        t = {}

        def lap(name, t0=[time.perf_counter()]):
            now = time.perf_counter()
            t[name] = (now - t0[0]) * 1e3
            t0[0] = now
        lap("_")
        tasks, sd = ctx.guide_tree(tcodes, n_threads=nt)
        lap("guide_tree_ms")
        ctx.msa_tree(codes, tasks, subm, scal, sd)
        ctx.tree_aligned_rows(seqs)
        lap("align1_and_rows_ms")
        tasks2, sd2 = ctx.aln_guide_tree()
        lap("aln_dist_and_upgma_ms")
        ctx.tree_upload(codes, tasks2, subm, scal, sd2, flags=api.FLAG_DEVICE_GAPS | api.FLAG_KEEP_CONSISTENCY)
        ctx.tree_run()
        got["rows"] = ctx.tree_aligned_rows(seqs)
        lap("align2_and_rows_ms")
        del t["_"]
        t["total_ms"] = sum(t.values())
        return t

    once()
    info = once()
    if not args.no_cpu:
        try:
            from oracle import refdrv
            job = refdrv.RefJob(inp, type_=0 if args.dna else -1, n_threads=nt)
            t0 = time.perf_counter(); job.run_tree(); a1 = time.perf_counter() - t0
            _, _, sdist, stree = job.realign_tree(want_dm=False)
            t0 = time.perf_counter(); job.run_tree(); a2 = time.perf_counter() - t0
            rows = job.finalise()
            ours = [None] * len(inp)
            for k, i in enumerate(order):
                ours[i] = got["rows"][k].decode()
            info["rows_identical_to_reference"] = bool(ours == rows)
            info["cpu_reference"] = {"guide_tree_ms": job.tree_seconds * 1e3, "align1_ms": a1 * 1e3, "aln_dist_ms": sdist * 1e3,
                                     "upgma_ms": stree * 1e3, "align2_ms": a2 * 1e3, "threads": nt,
                                     "total_ms": (job.tree_seconds + a1 + sdist + stree + a2) * 1e3,
                                     "note": "compute_aln_pairwise_dist and upgma are serial in the reference"}
            job.close()
        except Exception as e:      # pragma: no cover
            info["cpu_reference"] = str(e)
    return info


def concurrent_sets_leg(codes, tasks, subm, scal, seq_dist, local_rank, nsets=8, reps=3):
    """Throughput when independent alignments are available (ensemble members, a batch of families): `nsets`
    copies of the workload as ONE forest job (include/kalign_amd.h: n_tasks < numseq-1) -- the levels of all trees
    share launches and the upper parts of all trees run in one chained launch.  Reported next to the headline
    value, never instead of it."""
    import kalign_amd
    from kalign_amd import guide
    fc, ft, fd, _ = guide.forest([(codes, tasks, seq_dist)] * nsets)
    ctx = kalign_amd.Context(local_rank)
    ctx.tree_upload(fc, ft, subm, scal, fd)
    ctx.tree_run(); ctx.tree_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.tree_run()
    ctx.tree_sync()
    dt = (time.perf_counter() - t0) / reps
    recs, _, _ = ctx.tree_download(want_gaps=False)
    cells = float(sum(r.len_a * r.len_b for r in recs))
    kern_ms, n_launch = ctx.tree_kernel_ms()
    ctx.close()
    return {"sets_in_flight": nsets, "ms_per_round": dt * 1e3, "gcups": cells / dt / 1e9, "launches": n_launch,
            "note": "independent copies of the workload scheduled as one forest job in one context"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nseq", type=int, default=1024)
    ap.add_argument("--len", type=int, default=400)
    ap.add_argument("--dna", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pairs", action="store_true")
    ap.add_argument("--no-default-mode", action="store_true")
    args = ap.parse_args()

    import torch
    import kalign_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist_on = world > 1
    from kalign_amd import dist as kd
    if dist_on:
        import torch.distributed as dist
        kd.init("nccl", device=torch.device("cuda", local_rank))      # "nccl" is RCCL on ROCm

    seqs = workload_letters(args.nseq, args.len, args.dna, seed=1 + rank)
    codes, tasks, seq_dist = make_workload(args.nseq, args.len, args.dna, seed=1 + rank, seqs=seqs)
    subm, scal = scoring(args.dna)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = kalign_amd.Context(local_rank, stream=stream)
    ctx.tree_upload(codes, tasks, subm, scal, seq_dist)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.tree_run()
        ctx.tree_sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.tree_run()
    ctx.tree_sync()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = kd.reduce_scalar(elapsed, "max", device="cuda")      # MAX over ranks

    recs, paths, _ = ctx.tree_download(want_gaps=False)
    cells = float(sum(r.len_a * r.len_b for r in recs))
    kern_ms, n_launch = ctx.tree_kernel_ms()          # HIP events on the launch stream, last step
    total_cells = kd.reduce_scalar(cells, "sum", device="cuda")    # every rank aligned its own set

    # secondary measurement (rank 0 only, outside the timed region): the N x K seq-seq batch of
    # anchor consistency (anchor_consistency.c:246-267) through ka_pairwise_batch
    # (the secondary legs and the CPU baselines run at N = 1 only: at N > 1 the other ranks would sit in the final barrier)
    solo = rank == 0 and world == 1
    pair_info = None
    if solo and not args.no_pairs:
        pair_info = pairwise_leg(ctx, codes, subm, scal, args)
    dm_info = None
    if solo and not args.no_default_mode and not args.no_pairs:
        dm_info = default_mode_leg(ctx, codes, tasks, subm, scal, seq_dist, args)
    cs_info = None
    if solo and not args.no_pairs:
        cs_info = concurrent_sets_leg(codes, tasks, subm, scal, seq_dist, local_rank)
    e2e_info = None
    ra_info = None
    if solo and not args.no_default_mode and not args.no_pairs:
        e2e_info = end_to_end_leg(ctx, seqs, subm, scal, args)
        ra_info = realign_leg(ctx, seqs, subm, scal, args)

    if rank == 0:
        abytes = algorithmic_bytes(recs)
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        kinds = np.bincount([r.kind for r in recs], minlength=3)
        out = {
            "metric": "GCUPS (DP cell updates/s) on NxL synthetic MSA",
            "value": total_cells * args.steps / elapsed / 1e9,
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%d %s seqs x ~%d, whole guide tree per step: %d seq-seq + %d seq-profile + %d profile-profile DP tasks (Hirschberg), profile merge, path coding; one independent set per GPU"
                            % (args.nseq, "DNA" if args.dna else "protein", args.len, kinds[0], kinds[1], kinds[2]),
                "nseq": args.nseq, "len": args.len, "type": "dna" if args.dna else "protein",
                "useful_cells_per_step": cells, "tree_levels": n_launch,
            },
            "roofline": {
                "bound": "hbm", "kernel": "ka_task_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args),
                "algorithmic_bytes_per_step": abytes, "launches_per_step": n_launch,
                "avg_launch_ms": kern_ms / max(n_launch, 1),
                "kernel_ms_per_step": kern_ms,
            },
        }
        if pair_info:
            out["seqseq_batch"] = pair_info
        if dm_info:
            out["default_mode"] = dm_info
        if cs_info:
            out["concurrent_sets"] = cs_info
        if e2e_info:
            out["end_to_end"] = e2e_info
        if ra_info:
            out["realign_member"] = ra_info
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(codes, tasks, seq_dist, args.dna, cells)
        print(json.dumps(out))
    ctx.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
