#!/usr/bin/env python
"""bench.py -- GCUPS of the progressive-alignment hot path on MI355X.

One "step" = one pass of the hot path over one synthetic N x L sequence set: every task of the
guide tree (seq-seq, seq-profile and profile-profile Gotoh/Hirschberg DP, profile merge, path
coding) through ka_tree_run(), with sequences, task list and scoring tables already resident in
HBM.  GCUPS = sum over tasks of len_a*len_b ("useful" cell updates) / time.

Headline workload (N = 1): the shape north_star's target is quoted on -- 4096 protein sequences
x ~400 aa (DSSim family), the reference's `--fast` mode, on the REFERENCE'S OWN guide tree
(build_tree_kmeans through ka_guide_tree; its time is reported separately).
Secondary legs with their own roofline objects: C2 (1024 x 400) and C3 (4096 DNA x 2000).

    python bench.py [--gpus N --steps K --warmup W] [--nseq 4096 --len 400 --dna]

N > 1 (launched by torch.distributed.run, one rank per GPU): see multi_gpu_main() -- ONE guide
tree (C4's shape) sharded over the ranks by subtree (kalign_amd.dist.sharded_tree) -> strong scaling.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

# (the C5 leg runs eight contexts side by side, each on its own stream; the HIP runtime folds streams onto 4 hardware queues unless
# told otherwise before it starts -- the drop-in's glue asks for the same 8)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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


def pmc_traffic(key, workload=None, kernel_ms=None):
    """HBM bytes per STEP (all launches of one ka_tree_run) from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE and WRITE_SIZE collected in separate passes for exactly this workload, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950; profiles/collect.sh).  bench.py cannot run the
    profiler on itself, so it reports the committed measurement -- but only of THIS workload on THIS build: the
    committed record names its workload and the kernel time per step it was collected at; another workload string,
    or a live kernel time more than 5 % away, gives null (a stale profile is not this run's traffic)."""
    for tag in ("r06", "r05", "r04", "r03", "r02"):              # the newest collection that has this workload
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")) as fh:
                rec = json.load(fh)[key]
        except Exception:
            continue
        if workload is not None and rec.get("workload") != workload:
            return None
        ref_ms = rec.get("kernel_ms_per_step")
        if kernel_ms is not None and ref_ms and abs(kernel_ms - ref_ms) > 0.05 * ref_ms:
            return None
        if kernel_ms is not None and not ref_ms:
            return None                                          # (records without a kernel time cannot be matched to a build)
        return float(rec["hbm_bytes_per_step_corrected"])
    return None


def workload_letters(nseq, length, dna, seed):
    from kalign_amd import synth
    # the reference's own benchmark generator, restated: independent samples of one profile HMM
    # (tests/dssim.c); the big sets through the vectorised sampler of the same model
    if nseq > 4096:
        return synth.dssim_fast(nseq, length, dna=dna, seed=seed)
    return synth.dssim(nseq, length, dna=dna, seed=seed)


def make_workload(nseq, length, dna, seed, seqs=None):
    """Synthetic balanced guide tree (round 1's headline; kept for the concurrent-sets leg and tests)."""
    from kalign_amd import guide
    if seqs is None:
        seqs = workload_letters(nseq, length, dna, seed)
    codes = guide.encode(seqs, dna=dna)
    tasks = guide.bisecting_tree(nseq, seed=seed)
    dist = np.random.RandomState(seed).uniform(0.3, 0.9, nseq).astype(np.float32)
    return codes, tasks, dist


def host_threads():
    return min(os.cpu_count() or 1, 16)


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo); None when it cannot be told."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys, core = None, None
        return len(pairs) or None
    except OSError:
        return None


def make_job(ctx, nseq, length, dna, seed):
    """Letters -> what kalign_run hands create_msa_tree: the sequences in msa_sort_len_name order (longest first,
    msa_sort.c:62-80), encoded, and the reference's own guide tree (build_tree_kmeans, bisectingKmeans.c:177-271,
    through ka_guide_tree: distances on the device, 2-means / UPGMA on the host).  The tree's time is reported,
    never part of a timed step ("tree building excluded")."""
    from kalign_amd import guide
    inp = workload_letters(nseq, length, dna, seed)
    order = sorted(range(len(inp)), key=lambda i: (-len(inp[i]), i))
    seqs = [inp[i] for i in order]
    tcodes = guide.encode_tree(seqs, dna=dna)
    codes = guide.encode(seqs, dna=dna)
    ctx.guide_tree(tcodes, n_threads=host_threads())               # warm-up (allocations)
    t0 = time.perf_counter()
    tasks, sd = ctx.guide_tree(tcodes, n_threads=host_threads())
    gt_ms = (time.perf_counter() - t0) * 1e3
    return {"input": inp, "order": order, "seqs": seqs, "codes": codes, "tasks": tasks, "seq_distances": sd,
            "guide_tree_ms": gt_ms, "nseq": nseq, "len": length, "dna": dna}


def timed_tree(ctx, job, subm, scal, steps, warmup, barrier=None):
    """W untimed + K timed passes of ka_tree_run over the job, inputs resident in HBM."""
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])
    for _ in range(warmup):
        ctx.tree_run()
        ctx.tree_sync()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.tree_run()
    ctx.tree_sync()
    if barrier:
        barrier()
    elapsed = time.perf_counter() - t0
    recs, _, gaps = ctx.tree_download(want_gaps=True)  # (after the clock: the gap arrays are what the CPU leg compares)
    kern_ms, n_launch = ctx.tree_kernel_ms()          # HIP events on the launch stream, last step
    return {"elapsed": elapsed, "recs": recs, "gaps": gaps, "kern_ms": kern_ms, "n_launch": n_launch,
            "cells": float(sum(r.len_a * r.len_b for r in recs))}


def roofline_of(res, key, workload=None):
    abytes = algorithmic_bytes(res["recs"])
    achieved = abytes / (res["kern_ms"] * 1e-3) / 1e9
    traffic = pmc_traffic(key, workload, res["kern_ms"])
    return {"bound": "hbm", "kernel": "ka_task_kernel (all launches of one step)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_unit": "HBM bytes per step (FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)",
            "algorithmic_bytes_per_step": abytes, "launches_per_step": res["n_launch"],
            "avg_launch_ms": res["kern_ms"] / max(res["n_launch"], 1), "kernel_ms_per_step": res["kern_ms"]}


# VALU instructions per 128-cell wavefront step of the strips (ISA listings, DESIGN.md sections 4, 4f, 7): profile-profile 78 (20
# residues, helper-wave strips; 100 in ka_strip), seq-profile 36, seq-seq 31.5 -- and Hirschberg executes ~2.06 x the useful cells
# of a task in the tree kernels (1.6 x in the pair batch, which takes rows over from the parent's pass).
VALU_PER_STEP = {0: 31.5, 1: 36.0, 2: 78.0}
VALU_PER_STEP_SOURCE = "constants typed into bench.py (ISA listings of round 4)"


def _valu_from_build():
    """The profile-profile figure from the BUILD (the constant drifts silently when the kernels change): __graft_entry__.build()
    runs tools/check_hot_loops.py on unit 0 and leaves its octet table in kalign_amd/hot_loops.json -- the steady-state octet (no scratch
    access) of the two-row strip at the job's alphabet is the one with 20 (23 with B / Z / X) v_pk_mul_f32 per step."""
    global VALU_PER_STEP_SOURCE
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kalign_amd", "hot_loops.json")
    try:
        with open(path) as f:
            octets = json.load(f)["octets"]
        steady = [o["instructions_per_step"] for o in octets if o["v_pk_mul_per_step"] == 20 and o["scratch_per_step"] == 0]
        if steady:
            VALU_PER_STEP[2] = float(min(steady))
            VALU_PER_STEP_SOURCE = "kalign_amd/hot_loops.json (tools/check_hot_loops.py on the built unit 0: steady two-row octet, 20 residues); seq-seq / seq-profile: ISA listings of round 4"
    except (OSError, ValueError, KeyError):
        pass


_valu_from_build()
EXECUTED_PER_USEFUL = 2.06
ISSUE_PER_S = 256 * 4 * 2.4e9 / 4.0                                   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles


def roofline_valu(recs, gcups, copies=1):
    """The roofline that binds this path (at 200 GCUPS the task mix still moves only 20 % of the HBM peak): VALU
    issue.  `peak` = useful cells per second if every SIMD issued strip steps back to back at full lane occupancy, for THIS task mix:
    sum over kinds of useful cells x 2.06 executed per useful x instructions per cell.  What separates `achieved` from it: lanes
    idle in partial strips and at the ends of a pass, the recursion's barriers and meetups, path coding and merges, and -- for one
    tree -- the dependency chain of the guide tree."""
    cells = [0.0, 0.0, 0.0]
    for r in recs:
        cells[r.kind] += float(r.len_a) * r.len_b
    instr = sum(c * EXECUTED_PER_USEFUL * VALU_PER_STEP[k] / 128.0 for k, c in enumerate(cells))
    total = sum(cells)
    peak = total / (instr / ISSUE_PER_S) / 1e9
    return {"bound": "valu_issue", "achieved": gcups, "peak": peak, "unit": "GCUPS", "frac": gcups / peak,
            "executed_cells_per_useful_cell": EXECUTED_PER_USEFUL,
            "valu_instructions_per_128_cell_step": {"seq_seq": VALU_PER_STEP[0], "seq_profile": VALU_PER_STEP[1], "profile_profile": VALU_PER_STEP[2]},
            "valu_instructions_source": VALU_PER_STEP_SOURCE,
            "useful_cells_by_kind": {"seq_seq": cells[0] * copies, "seq_profile": cells[1] * copies, "profile_profile": cells[2] * copies}}


def workload_name(job, res):
    kinds = np.bincount([r.kind for r in res["recs"]], minlength=3)
    return ("%d %s seqs x ~%d (DSSim), --fast mode, the reference's k-means guide tree; whole tree per step: %d seq-seq + %d "
            "seq-profile + %d profile-profile DP tasks (Hirschberg), profile merge, path coding"
            % (job["nseq"], "DNA" if job["dna"] else "protein", job["len"], kinds[0], kinds[1], kinds[2]))


def pp_rate_leg(ctx, job, subm, scal):
    """What north_star's target is quoted on: the rate of the profile-profile DP itself.  One more run of the same job
    with per-task phase timers (KA_FLAG_TIMING; outside the timed steps -- the timers cost a few per cent): every
    task's wall time on its cluster.  `per_task`: profile-profile cells / the summed run times of the profile-profile
    tasks (the rate at which ONE such task proceeds on its workgroups, tasks counted one after the other);
    `critical_path`: the same over the profile-profile tasks on the dependency-driven critical path; `root`: the root
    task alone.  Cells are the useful ones (len_a * len_b); Hirschberg executes ~2.06 x as many."""
    from kalign_amd import api
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"], flags=api.FLAG_TIMING)
    for _ in range(2):
        ctx.tree_run()
        ctx.tree_sync()
    recs, _, _ = ctx.tree_download(want_gaps=False)
    tm = ctx.tree_timing()
    ghz = 2.4                                                    # shader clock of the s_memtime counter
    tot = tm[:, :4].sum(1) / ghz / 1e3                           # us per task
    cells = np.array([float(r.len_a) * r.len_b for r in recs])
    pp = np.array([r.kind == 2 for r in recs])
    done, crit_parent = {}, {}
    for r, t in zip(recs, tot):
        a, b = done.get(r.a, 0.0), done.get(r.b, 0.0)
        done[r.c] = max(a, b) + t
        crit_parent[r.c] = r.a if a >= b else r.b
    by_c = {r.c: i for i, r in enumerate(recs)}
    node, crit = recs[-1].c, []
    while node in by_c:
        crit.append(by_c[node])
        node = crit_parent[node]
    crit = np.array(crit)
    cpp = crit[pp[crit]]
    out = {"gcups_per_task": float(cells[pp].sum() / max(tot[pp].sum(), 1e-9) / 1e3),
           "gcups_critical_path_tasks": float(cells[cpp].sum() / max(tot[cpp].sum(), 1e-9) / 1e3),
           "gcups_root_task": float(cells[-1] / max(tot[-1], 1e-9) / 1e3),
           "critical_path_ms": float(done[recs[-1].c] / 1e3), "critical_path_tasks": int(len(crit)),
           "profile_profile_tasks": int(pp.sum()), "root_task_ms": float(tot[-1] / 1e3),
           "note": "per-task wall times from KA_FLAG_TIMING (one extra run with timers); useful cells"}
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])   # (back to the untimed job)
    return out


def pp_share(res):
    pp = float(sum(r.len_a * r.len_b for r in res["recs"] if r.kind == 2))
    return pp / max(res["cells"], 1.0)


def transfers_leg(ctx, job, subm, scal, reps=5):
    """SURVEY.md 8(d)'s t_DP bracket: H2D of the sequences and the task list, host-side task preparation, the task
    tree, D2H of records and coded paths -- ONE ka_msa_tree call of the C ABI with host buffers (the binding's own
    Python-side flattening / unpacking is outside the bracket, it is not what a C caller pays).  Reported beside
    `value`, never as `value`."""
    import ctypes as C
    from kalign_amd import api
    flat, off, lens = api._flatten(job["codes"])
    tasks = np.ascontiguousarray(job["tasks"], np.int32)
    sd = np.ascontiguousarray(job["seq_distances"], np.float32)
    sub = np.ascontiguousarray(subm, np.float32).reshape(-1)
    sc = np.ascontiguousarray(scal, np.float32)
    nt = len(tasks)
    recs = (api.TaskRec * nt)()
    cap = int(lens.sum()) * 40 + 4 * nt + 1024
    paths = np.zeros(cap, np.int32)
    p = api._ptr

    def once():
        rc = ctx.L.ka_msa_tree(ctx.h, len(lens), p(flat), p(off), p(lens), p(sd), nt, p(tasks), p(sub), p(sc), 0, recs, p(paths), cap, None)
        if rc:
            raise RuntimeError(ctx.L.ka_last_error().decode())
    once()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    dt = (time.perf_counter() - t0) / reps
    cells = float(sum(r.len_a * r.len_b for r in recs))
    return {"ms": dt * 1e3, "gcups": cells / dt / 1e9,
            "note": "one ka_msa_tree call, host buffers: upload + host-side task preparation + run + download of records and coded paths"}


def scoring(dna):
    """Default matrices of the reference (aln_param.c): PFASUM43 7/1.25/1 for protein,
    the +5/-4 8/6/0 DNA matrix for --type dna; tables are data fixtures dumped from the reference."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "param_tables.npz"))
    key = "1_0" if dna else "0_3"
    return z["subm_" + key], z["scal_" + key].copy()


def cpu_baseline(codes, tasks, dist, dna, cells, budget_s=25.0, gpu_gaps=None):
    """The reference's own dispatcher (oracle/_ref = the real Kalign sources compiled as they lie:
    create_msa_tree, OpenMP) on the host cores, same task list, bounded to ~10-30 s of CPU work.
    Fails loudly when oracle/_ref is missing (no silent fall-back to another baseline).
    gpu_gaps: the gap arrays of the job the GPU just timed -- the reference's own gaps[] of every run are compared
    with them (`gaps_identical_to_reference`: the bench checks what it times)."""
    from oracle import refdrv
    if not refdrv.available():
        raise RuntimeError("oracle/_ref/libkalign_ref.so is missing: run `make -C oracle ref` in the build container "
                           "(or pass --no-cpu)")
    ncores = os.cpu_count() or 1
    best = None
    tried = []
    identical = None
    budget = time.time() + budget_s
    for nt in sorted({ncores, min(ncores, 64), min(ncores, 16), 1}, reverse=True):
        reps = 0
        while reps < 2 and time.time() < budget:
            job = refdrv.EncodedJob(codes, tasks, dist, biotype=1 if dna else 0, type_=0 if dna else -1, n_threads=nt)
            ref_gaps, secs = job.run_tree()
            job.close()
            if gpu_gaps is not None:
                same = len(ref_gaps) == len(gpu_gaps) and all(np.array_equal(a, b) for a, b in zip(ref_gaps, gpu_gaps))
                identical = same if identical is None else (identical and same)
            reps += 1
            tried.append((nt, secs))
            if best is None or secs < best[1]:
                best = (nt, secs)
    return {"value": cells / best[1] / 1e9, "unit": "GCUPS", "cores": best[0], "host_logical_cpus": ncores,
            "host_physical_cores": physical_cores(), "gaps_identical_to_reference": identical,
            "kind": "reference",
            "sample": "full workload (same task list), create_msa_tree of the reference (OpenMP tasks); best of (threads:seconds) %s; "
                      "`cores` = the thread count of the best run" % (["%d:%.3f" % t for t in tried])}


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
    if not args.no_cpu and len(codes) <= 1024:        # (the reference builds its position maps serially: 100 s at 4096 x 400)
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


def saturation_leg(job, subm, scal, local_rank, cells_one, counts=(1, 2, 4, 8, 16), reps=2):
    """Saturated throughput: 1 .. 16 independent copies of the headline tree in flight as ONE forest job (a batch of
    families, ensemble members): the single-tree number is bounded by the tree's dependency chain, this one by the
    kernels.  GCUPS at every point until the curve flattens; `value` is never taken from here."""
    import kalign_amd
    from kalign_amd import guide
    pts = []
    ctx = kalign_amd.Context(local_rank)
    try:
        for n in counts:
            fc, ft, fd, _ = guide.forest([(job["codes"], job["tasks"], job["seq_distances"])] * n)
            ctx.tree_upload(fc, ft, subm, scal, fd)
            ctx.tree_run(); ctx.tree_sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.tree_run()
            ctx.tree_sync()
            dt = (time.perf_counter() - t0) / reps
            cells = cells_one * n                              # (every copy is the same tree: same useful cells)
            kern_ms, n_launch = ctx.tree_kernel_ms()
            pts.append({"trees_in_flight": n, "ms_per_round": dt * 1e3, "gcups": cells / dt / 1e9, "launches": n_launch})
        # north_star quotes its 200 GCUPS on PROFILE-PROFILE DP: the saturated point once more with the per-task phase timers
        # (KA_FLAG_TIMING; outside the timed rounds) -- the round's wall time is shared out over the three kinds of task by the time
        # their tasks held a workgroup (the slots are what the round is made of), and the profile-profile cells are divided by the
        # profile-profile share.  Its own roofline object: SURVEY 8(d)'s bytes of those tasks over that time.
        pp = None
        try:
            from kalign_amd import api
            nb = pts[-1]["trees_in_flight"]
            fc, ft, fd, _ = guide.forest([(job["codes"], job["tasks"], job["seq_distances"])] * nb)
            ctx.tree_upload(fc, ft, subm, scal, fd, flags=api.FLAG_TIMING)
            for _ in range(2):
                ctx.tree_run()
                ctx.tree_sync()
            t0 = time.perf_counter()
            ctx.tree_run(); ctx.tree_sync()
            dt_t = time.perf_counter() - t0
            recs, _, _ = ctx.tree_download(want_gaps=False)
            tm = ctx.tree_timing()[:, :4].sum(1).astype(np.float64)
            kind = np.array([r.kind for r in recs])
            cells_k = np.array([float(r.len_a) * r.len_b for r in recs])
            share = float(tm[kind == 2].sum() / max(tm.sum(), 1.0))
            pp_cells = float(cells_k[kind == 2].sum())
            pp_s = dt_t * share
            pp_bytes = algorithmic_bytes([r for r in recs if r.kind == 2])
            pp = {"trees_in_flight": nb, "gcups": pp_cells / pp_s / 1e9, "profile_profile_cells_per_round": pp_cells,
                  "share_of_task_time": share, "ms_per_round_with_timers": dt_t * 1e3,
                  "time_by_kind_share": {"seq_seq": float(tm[kind == 0].sum() / tm.sum()), "seq_profile": float(tm[kind == 1].sum() / tm.sum()), "profile_profile": share},
                  "roofline": {"bound": "hbm", "achieved": pp_bytes / pp_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pp_bytes / pp_s / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_round": pp_bytes, "traffic": None},
                  "roofline_valu": roofline_valu([r for r in recs if r.kind == 2], pp_cells / pp_s / 1e9),
                  "note": "profile-profile tasks only: the round's wall time x the share of task time (KA_FLAG_TIMING) the profile-profile tasks held workgroups for; useful cells"}
        except Exception as e:                                     # pragma: no cover
            pp = {"error": repr(e)}
    finally:
        ctx.close()
    best = max(pts, key=lambda p: p["gcups"])
    return {"workload": "independent copies of the headline job (4096 x ~400 protein, --fast) as one forest", "points": pts,
            "gcups_saturated": best["gcups"], "at_trees_in_flight": best["trees_in_flight"],
            "profile_profile_only": pp,
            "valu_issue_bound_gcups": 256 * 4 * 2.4e9 / 4 / 100 * 128 / 1e9,
            "note": "valu_issue_bound: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction / ~100 instructions per "
                    "128-cell step (DESIGN.md section 4); the HBM roofline fraction of the same point is gcups x ~7.9 B per cell / 8 TB/s"}


def secondary_tree_leg(ctx, name, nseq, length, dna, args, steps=3, warmup=1):
    """A named secondary workload with its own roofline object (C2 / C3 of BASELINE.json)."""
    job = make_job(ctx, nseq, length, dna, seed=1)
    subm, scal = scoring(dna)
    res = timed_tree(ctx, job, subm, scal, steps, warmup)
    out = {"workload": workload_name(job, res), "value": res["cells"] * steps / res["elapsed"] / 1e9, "unit": "GCUPS",
           "ms_per_step": res["elapsed"] / steps * 1e3, "steps": steps, "useful_cells_per_step": res["cells"],
           "profile_profile_share_of_cells": pp_share(res), "guide_tree_ms": job["guide_tree_ms"],
           "roofline": roofline_of(res, name, workload_name(job, res))}
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(job["codes"], job["tasks"], job["seq_distances"], dna, res["cells"], budget_s=12.0, gpu_gaps=res["gaps"])
        out["gaps_identical_to_reference"] = out["cpu_baseline"]["gaps_identical_to_reference"]
    return out, job


def refine_leg(ctx, job, subm, scal, args):
    """refine_alignment (KALIGN_REFINE_ALL, aln_refine.c:36-346) on the C2 tree: ka_tree_refine on the device next to the
    reference's own function on the host (both after a first pass; only the refinement is timed), gaps compared."""
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])
    ctx.tree_refine(1)
    ctx.tree_sync()                                              # warm-up (arena sizes settle)
    t0 = time.perf_counter()
    ctx.tree_refine(1)
    ctx.tree_sync()
    wall = time.perf_counter() - t0
    kms, nl = ctx.tree_kernel_ms()
    recs, _, gaps = ctx.tree_download()
    out = {"mode": "KALIGN_REFINE_ALL: every edge re-aligned with five flip trials, best sum-of-pairs trial kept",
           "device_ms": wall * 1e3, "kernel_ms": kms, "launches": nl, "alnlen": int(recs[-1].plen)}
    if not args.no_cpu:
        from oracle import refdrv
        nt = min(16, host_threads())
        j = refdrv.EncodedJob(job["codes"], job["tasks"], job["seq_distances"], biotype=1 if job["dna"] else 0,
                              type_=0 if job["dna"] else -1, n_threads=nt)
        j.run_tree()
        t0 = time.perf_counter()
        g, _, _, _ = j.refine(1)
        out["reference_refine_alignment_ms"] = (time.perf_counter() - t0) * 1e3
        out["reference_threads"] = nt
        j.close()
        out["gaps_identical_to_reference"] = bool(all(np.array_equal(a, b) for a, b in zip(g, gaps)))
        out["speedup_vs_reference"] = out["reference_refine_alignment_ms"] / out["device_ms"]
    return out


def c4_single_gpu_leg(ctx, steps=3):
    """The workload `bench.py --gpus N` (N > 1) shards, on ONE GPU: 16384 protein x ~500, default mode -- one step =
    anchor_consistency_build + the task tree, as in multi_gpu_main.  The N = 1 point of the strong-scaling curve."""
    job = make_job(ctx, 16384, 500, False, seed=1)
    subm, scal = scoring(False)
    lens = np.array([len(c) for c in job["codes"]], np.int64)
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])

    split = [0.0, 0.0]

    def step():
        t0 = time.perf_counter()
        ctx.tree_build_consistency(5, 2.0)
        t1 = time.perf_counter()
        ctx.tree_run()
        ctx.tree_sync()
        split[0] += t1 - t0
        split[1] += time.perf_counter() - t1
    step()
    split[:] = [0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    recs, _, _ = ctx.tree_download(want_gaps=False)
    cells = float(sum(r.len_a * r.len_b for r in recs))
    ids, _ = ctx.tree_consistency()
    pair_cells = float(sum(int(lens[i]) * int(lens[a]) for i in range(len(lens)) for a in ids if a != i))
    cons_ms, tree_ms = split[0] / steps * 1e3, split[1] / steps * 1e3
    out = {"workload": "16384 protein seqs x ~500 (DSSim), default mode (5 anchors), the reference's guide tree, one GPU",
           "value": (cells + pair_cells) / dt / 1e9, "unit": "GCUPS", "ms_per_step": dt * 1e3, "steps": steps,
           "consistency_ms": cons_ms, "tree_ms": tree_ms,
           "useful_cells_tree": cells, "useful_cells_consistency_batch": pair_cells, "guide_tree_ms": job["guide_tree_ms"]}
    try:
        out["guide_tree_bisection_ms"], out["guide_tree_bisection_on_device"] = api_bisect()
    except Exception:                                              # pragma: no cover
        pass
    out["sharding_projection"] = sharding_projection(ctx, job, subm, scal, lens, cons_ms)
    return out


# kalign_ensemble's member table (ensemble.c:32-45): (gpo, gpe, tgpe) multipliers and the tree-noise sigma of member k mod 12
ENSEMBLE_MEMBERS = [(1.0, 1.0, 1.0, 0.0), (0.5, 1.5, 0.8, 0.20), (1.5, 0.5, 1.2, 0.20), (0.7, 0.7, 0.5, 0.25), (1.4, 1.4, 1.5, 0.25),
                    (0.8, 1.2, 1.0, 0.30), (1.3, 0.8, 0.7, 0.30), (0.6, 1.0, 1.3, 0.15), (1.0, 0.6, 0.6, 0.15), (1.8, 1.0, 1.0, 0.35),
                    (1.0, 1.8, 1.8, 0.35), (0.4, 0.4, 0.3, 0.20)]


def c5_ensemble_leg(local_rank, n_members=8, nseq=2048, length=300):
    """BASELINE config 5 on ONE GPU: `--precise --ensemble 8` on 2048 x ~300 -- eight complete members (noisy guide tree, 5-anchor
    consistency, task tree, rows, one realignment pass on the UPGMA tree of the rows: kalign_run_realign, aln_wrap.c:361-527), each
    with its own gap penalties (ensemble.c:32-76).  Members are independent: no collective.  Measured one after the other on one
    context, and side by side on eight contexts (eight host threads; what the drop-in's kalign_ensemble does with one context per
    device on a node -- here the contexts share the one GPU, i.e. the GPU is given eight members' worth of independent work)."""
    import threading
    import kalign_amd
    from kalign_amd import guide, dist
    inp = workload_letters(nseq, length, False, 5)
    order = sorted(range(len(inp)), key=lambda i: (-len(inp[i]), i))
    seqs = [inp[i] for i in order]
    tcodes, codes = guide.encode_tree(seqs, dna=False), guide.encode(seqs, dna=False)
    subm, scal = scoring(False)
    rng = np.random.default_rng(7)
    members = []
    for k in range(n_members):
        g, e, t, sigma = ENSEMBLE_MEMBERS[k % 12]
        sc = np.array(scal, np.float32).copy()
        sc[0], sc[1], sc[2] = scal[0] * g, scal[1] * e, scal[2] * t
        members.append({"scal": sc, "dm_scale": None if not k else np.maximum(rng.normal(1.0, sigma, (len(seqs), min(32, len(seqs)))), 0.1).astype(np.float32)})
    lens = np.array([len(c) for c in codes], np.int64)
    # (shared contexts: own stream each, no workgroup waits for another -- every guide-tree level is a launch; `alone` has the GPU to itself)
    ctxs = [kalign_amd.Context(local_rank, shared=True) for _ in range(n_members)]
    alone = kalign_amd.Context(local_rank)
    runs = [dist.member_on_context(c, tcodes, codes, seqs, subm, n_anchors=5, weight=2.0, n_threads=max(host_threads() // n_members, 1), realign=1) for c in ctxs]
    run_alone = dist.member_on_context(alone, tcodes, codes, seqs, subm, n_anchors=5, weight=2.0, n_threads=host_threads(), realign=1)
    cells = [0.0] * n_members

    pair = 5.0 * float(lens.sum()) * float(lens.mean())           # the N x 5 seq-seq batch (anchors are about as long as the rest)
    try:
        for k in range(n_members):                                 # warm-up (allocations) + the cells of every member
            runs[k](members[k])
            cells[k] = 2.0 * float(ctxs[k].tree_cells() or 0.0) + pair
        run_alone(members[0])
        t0 = time.perf_counter()
        for k in range(n_members):
            run_alone(members[k])
        serial = time.perf_counter() - t0
        th = [threading.Thread(target=runs[k], args=(members[k],)) for k in range(n_members)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        side = time.perf_counter() - t0
        fallbacks = alone.fallback_runs() + sum(c.fallback_runs() for c in ctxs)
    finally:
        for c in ctxs + [alone]:
            c.close()
    total = float(sum(cells))
    return {"workload": "%d ensemble members (`--precise --ensemble %d`: 5 anchors, one realignment pass) on %d protein seqs x ~%d, one GPU" % (n_members, n_members, nseq, length),
            "members": n_members, "useful_cells": total,
            "ms_one_after_the_other": serial * 1e3, "gcups_one_after_the_other": total / serial / 1e9,
            "ms_side_by_side": side * 1e3, "gcups_side_by_side": total / side / 1e9,
            "ms_per_member_alone": serial * 1e3 / n_members, "ms_per_member_side_by_side": side * 1e3 / n_members,
            "fallback_runs": fallbacks,
            "note": "guide trees (distances on the device, bisection / UPGMA on the host) are inside these times; the consensus stage (POAR, host) is not; "
                    "useful cells = 2 x the first tree's (the realignment pass aligns the same sequences again) + the N x 5 seq-seq batch"}


def api_bisect():
    from kalign_amd import api
    return api.guide_last_bisect()


def sharding_projection(ctx, job, subm, scal, lens, cons_ms, worlds=(2, 4, 8)):
    """What `bench.py --gpus N` should show for this job, from the N = 1 measurements alone (no multi-GPU box in the pool):
    the consistency batch divides by N (plus the broadcast of the 164 MB table at ~100 GB/s of xGMI ring bandwidth); the
    tree is cut exactly as ka_dist_plan cuts it (api.dist_plan_subtrees); a rank's subtrees take the larger of (its share
    of the summed task times of the one-GPU run / the concurrency that run had) and its own dependency chain; the tasks
    above the cut run one after the other on their owners, each after its children, with the smaller child's profile
    moved at 50 GB/s.  Per-task times: one default-mode run with KA_FLAG_TIMING.  A projection, labelled as one."""
    from kalign_amd import api
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"], flags=api.FLAG_TIMING)
    ctx.tree_build_consistency(5, 2.0)
    for _ in range(2):
        ctx.tree_run()
        ctx.tree_sync()
    recs, _, _ = ctx.tree_download(want_gaps=False)
    tm = ctx.tree_timing()
    kern_ms, _ = ctx.tree_kernel_ms()
    tot = tm[:, :4].sum(1) / 2.4 / 1e6                           # ms per task
    concurrency = max(float(tot.sum()) / max(kern_ms, 1e-9), 1.0)
    n = len(lens)
    idx = {r.c: i for i, r in enumerate(recs)}
    table_mb = float(lens.sum()) * 5 * 4 / 1e6
    out = {"one_gpu": {"consistency_ms": cons_ms, "tree_kernel_ms": kern_ms, "mean_tasks_in_flight": concurrency},
           "note": "projection from the one-GPU run, not a measurement"}
    for w in worlds:
        run_rank, top = api.dist_plan_subtrees([int(x) for x in lens], job["tasks"], w)
        top = set(int(t) for t in top)
        work = np.zeros(w)
        chain = {}                                               # node -> dependency chain (ms) inside its rank's subtrees
        for i, r in enumerate(recs):
            if i in top:
                continue
            work[run_rank[i]] += tot[i]
            chain[r.c] = max(chain.get(r.a, 0.0), chain.get(r.b, 0.0)) + tot[i]
        sub = np.zeros(w)
        for i, r in enumerate(recs):
            if i not in top:
                sub[run_rank[i]] = max(sub[run_rank[i]], chain[r.c])
        chain_ms = float(sub.max())                             # the longest dependency chain inside any rank's subtrees
        sub = np.maximum(sub, work / concurrency)
        done = {}
        for i, r in enumerate(recs):                             # (task order = dependency order)
            if i not in top:
                done[r.c] = sub[run_rank[i]]
                continue
            ta, tb = done.get(r.a, 0.0), done.get(r.b, 0.0)
            smaller = min(r.len_a, r.len_b) if r.a in idx and r.b in idx else 0
            move_ms = (smaller + 2) * 256 / 50e9 * 1e3 + 0.05
            done[r.c] = max(ta, tb) + move_ms + tot[i]
        tree_ms = done[recs[-1].c]
        cons = cons_ms / w + table_mb / 1e3 / 100.0 * 1e3 * (w - 1) / w
        gather_ms = 0.3
        out["n%d" % w] = {"consistency_ms": cons, "subtree_ms": float(sub.max()), "subtree_chain_ms": chain_ms, "top_ms": float(tree_ms - sub.max()), "gather_ms": gather_ms,
                          "ms_per_step": cons + tree_ms + gather_ms,
                          "speedup_vs_one_gpu": (cons_ms + kern_ms) / (cons + tree_ms + gather_ms)}
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nseq", type=int, default=4096)
    ap.add_argument("--len", type=int, default=400)
    ap.add_argument("--dna", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline line only (profiling runs)")
    ap.add_argument("--quick", action="store_true", help="= --no-legs --no-cpu: the headline line in seconds (profiling iterations)")
    ap.add_argument("--no-c3", action="store_true")
    ap.add_argument("--no-c4", action="store_true")
    ap.add_argument("--scale-workload", action="store_true", help="N > 1: shard --nseq x --len instead of C4 (tests)")
    ap.add_argument("--scale-fast", action="store_true", help="N > 1: --fast mode instead of the default mode")
    args = ap.parse_args()
    if args.quick:
        args.no_legs = args.no_cpu = True

    import torch
    import kalign_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if world > 1 and os.environ.get("KA_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = 0                                             # tests: every rank on the one GPU of the box (gloo)
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("KA_BENCH_FORCE_MULTI"):       # (the latter: the RCCL code path on a one-GPU box, world 1)
        return multi_gpu_main(args, rank, world, local_rank)

    stream = torch.cuda.current_stream().cuda_stream
    ctx = kalign_amd.Context(local_rank, stream=stream)
    job = make_job(ctx, args.nseq, args.len, args.dna, seed=1)
    subm, scal = scoring(args.dna)

    def barrier():
        torch.cuda.synchronize()

    res = timed_tree(ctx, job, subm, scal, args.steps, args.warmup, barrier)
    elapsed, cells = res["elapsed"], res["cells"]
    key = "headline" if (args.nseq, args.len, args.dna) == (4096, 400, False) else "other"
    out = {
        "metric": "GCUPS (DP cell updates/s) on NxL synthetic MSA",
        "value": cells * args.steps / elapsed / 1e9,
        "unit": "GCUPS",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload_name(job, res),
            "nseq": args.nseq, "len": args.len, "type": "dna" if args.dna else "protein", "mode": "--fast (no anchor consistency)",
            "guide_tree": "reference's bisecting k-means tree (ka_guide_tree), built outside the timed region",
            "guide_tree_ms": job["guide_tree_ms"],
            "useful_cells_per_step": cells, "launches_per_step": res["n_launch"],
            "profile_profile_share_of_cells": pp_share(res),
            "gcups_profile_profile_lower_bound": pp_share(res) * cells * args.steps / elapsed / 1e9,
        },
        "roofline": roofline_of(res, key, workload_name(job, res)),
        "roofline_valu": roofline_valu(res["recs"], cells * args.steps / elapsed / 1e9),
    }
    out["config"]["gcups_profile_profile"] = pp_rate_leg(ctx, job, subm, scal)
    if not args.no_legs:
        # SURVEY.md 8(d)'s t_DP bracket (H2D of sequences and tasks + run + D2H of records and paths).  The bench contract
        # keeps `value` on inputs resident in HBM ("the PCIe-inclusive rate ... is never `value`"); the bracket the survey
        # and the round-2 verdict quote against the target is this one, at the top level next to it.
        out["t_dp_with_transfers"] = transfers_leg(ctx, job, subm, scal)
        out["value_survey_8d_bracket"] = out["t_dp_with_transfers"]["gcups"]
        out["ms_per_step_survey_8d_bracket"] = out["t_dp_with_transfers"]["ms"]
        out["saturation"] = saturation_leg(job, subm, scal, local_rank, cells)
        out["saturation"]["roofline_valu"] = roofline_valu(res["recs"], out["saturation"]["gcups_saturated"], out["saturation"]["at_trees_in_flight"])
        out["default_mode"] = default_mode_leg(ctx, job["codes"], job["tasks"], subm, scal, job["seq_distances"], args)
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(job["codes"], job["tasks"], job["seq_distances"], args.dna, cells, gpu_gaps=res["gaps"])
        out["gaps_identical_to_reference"] = out["cpu_baseline"]["gaps_identical_to_reference"]
    if not args.no_legs:
        c2, c2job = secondary_tree_leg(ctx, "c2_1024x400", 1024, 400, False, args, steps=5, warmup=2)
        out["c2_1024x400_protein"] = c2
        s2, sc2 = scoring(False)
        out["seqseq_batch"] = pairwise_leg(ctx, c2job["codes"], s2, sc2, args)
        out["end_to_end"] = end_to_end_leg(ctx, c2job["input"], s2, sc2, args)
        out["realign_member"] = realign_leg(ctx, c2job["input"], s2, sc2, args)
        out["refine_all_c2"] = refine_leg(ctx, c2job, s2, sc2, args)
        bc, bt, bd = make_workload(1024, 400, False, seed=1)
        out["concurrent_sets"] = concurrent_sets_leg(bc, bt, s2, sc2, bd, local_rank)
        if not args.no_c3:
            out["c3_4096x2000_dna"], _ = secondary_tree_leg(ctx, "c3_dna_4096x2000", 4096, 2000, True, args, steps=3, warmup=1)
        if not args.no_c4:
            out["c4_single_gpu"] = c4_single_gpu_leg(ctx)
            out["c5_ensemble8_2048x300"] = c5_ensemble_leg(local_rank)
    # (runs that stalled on a wait between workgroups and were repeated on the plan without them: 0 unless the GPU is shared)
    out["fallback_runs"] = ctx.fallback_runs()
    print(json.dumps(out))
    ctx.close()


def multi_gpu_main(args, rank, world, local_rank):
    """N > 1: one process per GPU (torch.distributed, backend "nccl" = RCCL).  STRONG scaling of ONE alignment -- C4 of
    BASELINE.json: 16384 protein sequences x ~500, the reference's default mode (5 consistency anchors), the
    reference's guide tree:
      * anchor_consistency_build: the N x K seq-seq batch sharded over the ranks, every rank's share of the position
        maps broadcast in place HBM to HBM (dist.sharded_consistency);
      * the task tree cut into one subtree per rank (dist.sharded_tree); above the cut the profile of a subtree root
        moves device to device (RCCL send / recv) to the rank that runs the parent;
      * records and coded paths gathered on every rank.
    One step = all of that (SURVEY.md 8d's t_DP for default mode); `value` = the alignment's useful cells / time --
    the same cell count at every N.  The N = 1 number of this workload is the `c4_single_gpu` leg of the N = 1 line."""
    import torch
    import torch.distributed as dist
    import kalign_amd
    from kalign_amd import api
    from kalign_amd import dist as kd
    # RCCL prints a version banner to stdout when a communicator comes up; the contract is ONE JSON line on stdout:
    # everything else this process (and the libraries under it) writes to fd 1 goes to stderr, the line to the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    backend = os.environ.get("KA_BENCH_BACKEND", "nccl")          # tests on one GPU: gloo, every rank on cuda:0
    same_gpu = backend != "nccl"
    if same_gpu:
        torch.cuda.set_device(0)
        kd.init(backend, force=True)
    else:
        kd.init("nccl", device=torch.device("cuda", local_rank), force=True)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = kalign_amd.Context(0 if same_gpu else local_rank, stream=stream, shared=same_gpu)
    # RCCL: collectives and hand-overs on HBM buffers.  gloo (tests): host tensors -- gloo fills a device tensor from the CPU
    # side, past the GPU's caches
    coll_dev = "cpu" if same_gpu else "cuda"
    nseq, length = (args.nseq, args.len) if args.scale_workload else (16384, 500)
    job = make_job(ctx, nseq, length, False, seed=1)               # every rank: the same sequences, the same guide tree
    subm, scal = scoring(False)
    lens = np.array([len(c) for c in job["codes"]], np.int64)
    anchors = 0 if args.scale_fast else 5
    ctx.tree_upload(job["codes"], job["tasks"], subm, scal, job["seq_distances"])

    # The C multi-GPU layer (ka_dist_*: the cut planned once per job, a rank's subtrees as ONE planned run, RCCL called from
    # C on HBM buffers) whenever the backend is RCCL; the torch.distributed path of kalign_amd/dist.py for the gloo tests
    # (and as the fall-back should the C layer fail to come up: `dist_layer` says which one ran).
    cd, dist_layer = None, "python (kalign_amd/dist.py over torch.distributed)"
    if not same_gpu and not os.environ.get("KA_BENCH_PYDIST"):
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.as_tensor(api.dist_unique_id()))
            dist.broadcast(uid, src=0)
            torch.cuda.synchronize()
            cd = api.Dist(ctx, rank, world, uid.cpu().numpy())
            cd.plan()
            dist_layer = "C (ka_dist_*: RCCL from C, one planned run per rank)"
        except Exception as e:                                     # pragma: no cover
            sys.stderr.write("rank %d: C multi-GPU layer unavailable (%s): torch.distributed path\n" % (rank, e))
            cd = None
        ok = kd.reduce_scalar(1.0 if cd is not None else 0.0, "sum", device=coll_dev) >= world - 0.5
        if not ok and cd is not None:                              # (all ranks take the same path)
            cd.close(); cd = None

    def step():
        if cd is not None:
            if anchors:
                cd.consistency(anchors, 2.0)
            cd.tree_run()
            return None, None
        ctx.tree_reset()
        if anchors:
            kd.sharded_consistency(ctx, anchors, 2.0, rank, world)
        return kd.sharded_tree(ctx, job["tasks"], lens, rank, world, api.TaskRec, device=coll_dev)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    # the first step of the C layer is guarded: a rank on which it fails says so, and ALL ranks continue on the
    # torch.distributed path (an error, not a hang: every RCCL call of the layer is matched rank by rank in one global order)
    if cd is not None:
        failed = 0.0
        try:
            step()
        except Exception as e:                                     # pragma: no cover
            sys.stderr.write("rank %d: C multi-GPU layer failed in its first step (%s): torch.distributed path\n" % (rank, e))
            failed = 1.0
        if kd.reduce_scalar(failed, "sum", device=coll_dev) > 0.5:
            try:
                cd.close()
            except Exception:                                      # pragma: no cover
                pass
            cd, dist_layer = None, "python (kalign_amd/dist.py over torch.distributed; the C layer failed in its first step)"
    for _ in range(max(args.warmup, 1)):
        recs, paths = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs, paths = step()
    barrier()
    elapsed = kd.reduce_scalar(time.perf_counter() - t0, "max", device=coll_dev)
    if cd is not None:
        recs, paths = cd.download()                                # (every rank holds every record and path after each step)
    cells = float(sum(r.len_a * r.len_b for r in recs))
    pair_cells = 0.0
    if anchors:
        ids = ctx.tree_consistency()[0] if ctx.tree_consistency() else []
        pair_cells = float(sum(int(lens[i]) * int(lens[a]) for i in range(len(lens)) for a in ids if a != i))
    # every rank holds every record and path: a checksum over all of them must agree across the ranks
    chk = int(np.asarray(paths, np.int64).sum() % (1 << 31)) ^ int(sum(r.plen for r in recs))
    chk_max = kd.reduce_scalar(float(chk), "max", device=coll_dev)
    same_as_one_gpu = None
    if rank == 0:
        # the same job as ONE whole-tree run on this rank's GPU (outside the timed region): results must not depend on N
        def single():
            if anchors:
                ctx.tree_build_consistency(anchors, 2.0)
            ctx.tree_run()
            ctx.tree_sync()
        single()
        ts = time.perf_counter()
        for _ in range(2):
            single()
        single_ms = (time.perf_counter() - ts) / 2 * 1e3
        r1, p1, _ = ctx.tree_download(want_gaps=False)
        diff = [t for t, (a, b) in enumerate(zip(r1, recs))
                if a.plen != b.plen or not np.array_equal(p1[a.path_off:a.path_off + a.plen + 2], paths[b.path_off:b.path_off + b.plen + 2])]
        same_as_one_gpu = bool(len(r1) == len(recs) and not diff)
        if diff:
            run_rank, top = kd.plan_subtrees(job["tasks"], lens, world)
            sys.stderr.write("sharded run differs from the single-GPU run at tasks %s (of %d); ranks %s; above the cut: %s\n" % (
                diff[:12], len(recs), [int(run_rank[t]) for t in diff[:12]], [t in set(top) for t in diff[:12]]))
    # WEAK scaling beside it: the regime in which north_star's ">= 6x at 8 GPUs" can be answered whatever the
    # guide tree's serial top costs -- one independent alignment per GPU (the headline tree, --fast: what a caller with many
    # alignments, or the members of an ensemble, see), no data-path collective, the same barrier and max-over-ranks timing.
    weak = None
    try:
        wjob = make_job(ctx, 4096, 400, False, seed=1)
        ctx.tree_upload(wjob["codes"], wjob["tasks"], subm, scal, wjob["seq_distances"])
        for _ in range(2):
            ctx.tree_run()
        ctx.tree_sync()
        barrier()
        tw = time.perf_counter()
        wsteps = max(args.steps, 3)
        for _ in range(wsteps):
            ctx.tree_run()
        ctx.tree_sync()
        barrier()
        w_elapsed = kd.reduce_scalar(time.perf_counter() - tw, "max", device=coll_dev)
        wrecs, _, _ = ctx.tree_download(want_gaps=False)
        wcells = float(sum(r.len_a * r.len_b for r in wrecs))
        weak = {"workload": "one 4096 x ~400 protein tree (--fast) per GPU, independent replicas, no collective in the timed steps",
                "scaling": "weak", "gcups": wcells * world * wsteps / w_elapsed / 1e9, "ms_per_step": w_elapsed / wsteps * 1e3,
                "gcups_per_gpu": wcells * wsteps / w_elapsed / 1e9, "steps": wsteps}
    except Exception as e:                                         # pragma: no cover
        sys.stderr.write("rank %d: weak-scaling leg failed (%s)\n" % (rank, e))
    if rank == 0:
        kinds = np.bincount([r.kind for r in recs], minlength=3)
        real_stdout.write(json.dumps({
            "metric": "GCUPS (DP cell updates/s) on NxL synthetic MSA", "value": (cells + pair_cells) * args.steps / elapsed / 1e9,
            "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d protein seqs x ~%d (DSSim), ONE alignment sharded over %d GPUs: %s, the reference's k-means guide "
                                   "tree cut into one subtree per GPU (%d seq-seq + %d seq-profile + %d profile-profile DP tasks), profiles "
                                   "handed over HBM to HBM above the cut, records and paths gathered" % (
                                       nseq, length, world, ("default mode, the N x %d consistency batch sharded + position maps broadcast" % anchors)
                                       if anchors else "--fast mode", kinds[0], kinds[1], kinds[2]),
                       "nseq": nseq, "len": length, "mode": "default (5 anchors)" if anchors else "--fast",
                       "useful_cells_tree": cells, "useful_cells_consistency_batch": pair_cells,
                       "guide_tree_ms_every_rank": job["guide_tree_ms"],
                       "identical_results_on_all_ranks": bool(chk_max == float(chk)),
                       "identical_to_a_single_gpu_run": same_as_one_gpu,
                       "dist_layer": dist_layer},
            # the same job as one whole-tree run on rank 0's GPU (no sharding layer at all), timed after the steps; at
            # N = 1 (KA_BENCH_FORCE_MULTI=1) the difference is what the sharding layer itself costs
            "single_gpu_step_ms": single_ms,
            "sharding_overhead_ms": (elapsed / args.steps * 1e3 - single_ms) if world == 1 else None,
            "weak_scaling_replicas": weak,
        }) + "\n")
        real_stdout.flush()
    if cd is not None:
        cd.close()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
