"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" for the CPU tests).

What shards:
  * the N x K seq-seq batch of anchor consistency and any list of independent pairwise tasks:
    contiguous, cost-balanced slices per rank, no data-path collective, results gathered at the
    end (paths are small: 4*(La+Lb+2) bytes per pair);
  * independent sequence sets / ensemble members: one per rank (what bench.py scales).
  * a single guide tree (sharded_tree): cut into one subtree per rank balanced by estimated DP
    cells; every rank runs its subtree with profiles resident in its own HBM; the remaining
    top merges form a reduction in which one child profile moves point-to-point (send/recv,
    <= ~1 MB) to the rank that runs the parent; rank-independent results are gathered at the end.
See DESIGN.md "Multi-GPU".
"""
import os

import numpy as np


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device=None, force=False):
    """init_process_group from the torchrun environment; returns (rank, world).  force: also for a world of one."""
    import torch.distributed as dist
    rank, world, _ = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend or "nccl", **kw)
    return rank, world


def partition(costs, world):
    """Contiguous partition of units 0..n-1 into `world` slices with balanced total cost.
    Returns [(lo, hi)] per rank; every unit is in exactly one slice; slices may be empty."""
    costs = np.asarray(costs, np.float64)
    n = len(costs)
    if world <= 1:
        return [(0, n)]
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def reduce_scalar(value, op="max", device="cpu"):
    """all_reduce of one float (MAX for the timing, SUM for work counters)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def sharded_pairwise(compute, lens, ia, ib, rank, world, device="cpu"):
    """Shard `npairs` independent seq-seq alignments across ranks and gather all coded paths on
    every rank (the N x K loop of anchor_consistency_build, lib/src/anchor_consistency.c:246-267).

    compute(lo, hi) -> (list of coded paths for pairs lo..hi-1, scores array) runs the local
    slice (on this rank's GPU through kalign_amd.Context.pairwise_batch).
    Returns (paths for ALL pairs in order, scores for all pairs)."""
    import torch
    import torch.distributed as dist
    lens = np.asarray(lens)
    ia = np.asarray(ia)
    ib = np.asarray(ib)
    costs = lens[ia].astype(np.float64) * lens[ib]
    lo, hi = partition(costs, world)[rank]
    paths, scores = compute(lo, hi) if hi > lo else ([], np.zeros(0, np.float32))
    if world == 1:
        return paths, np.asarray(scores, np.float32)
    # flatten: paths are ragged -> (sizes, flat) with a padded all_gather
    sizes = np.array([len(p) for p in paths], np.int64)
    flat = np.concatenate(paths).astype(np.int32) if len(paths) else np.zeros(0, np.int32)
    meta = torch.tensor([len(sizes), len(flat)], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_n = int(max(m[0].item() for m in metas))
    max_f = int(max(m[1].item() for m in metas))

    def gather(arr, n, dtype):
        buf = torch.zeros(max(n, 1), dtype=dtype, device=device)
        if len(arr):
            buf[:len(arr)] = torch.as_tensor(arr, dtype=dtype, device=device)
        out = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        return [o.cpu().numpy() for o in out]

    g_sizes = gather(sizes, max_n, torch.int64)
    g_flat = gather(flat, max_f, torch.int32)
    g_scores = gather(np.asarray(scores, np.float32), max_n, torch.float32)
    all_paths, all_scores = [], []
    for r in range(world):
        n = int(metas[r][0].item())
        o = 0
        for k in range(n):
            s = int(g_sizes[r][k])
            all_paths.append(g_flat[r][o:o + s].copy())
            o += s
        all_scores.append(g_scores[r][:n])
    return all_paths, np.concatenate(all_scores) if all_scores else np.zeros(0, np.float32)


# ------------------------------------------------------------------------------------------------
# single guide tree over several GPUs
# ------------------------------------------------------------------------------------------------
def plan_subtrees(tasks, lens, world):
    """Cut the guide tree into at most `world` subtrees balanced by estimated DP cells.

    tasks: (n_tasks, 3) array of (a, b, c) in TASK_ORDER_TREE order.  Returns (run_rank, top) where
    run_rank[t] is the rank that runs task t and top is the list of task ids above the cut, in
    tree order.  Deterministic: every rank computes the same plan from the same inputs."""
    tasks = np.asarray(tasks)
    numseq = len(lens)
    nt = len(tasks)
    task_of = {int(c): t for t, (_, _, c) in enumerate(tasks)}
    est_len = {i: float(lens[i]) for i in range(numseq)}
    work = {i: 0.0 for i in range(numseq)}
    members = {i: 1 for i in range(numseq)}
    for a, b, c in tasks:
        a, b, c = int(a), int(b), int(c)
        est_len[c] = 1.05 * max(est_len[a], est_len[b])
        work[c] = work[a] + work[b] + est_len[a] * est_len[b]
        members[c] = members[a] + members[b]
    root = int(tasks[-1][2])
    frontier, top = [root], []
    while len([x for x in frontier if x >= numseq]) < world:
        cand = [x for x in frontier if x >= numseq]
        if not cand:
            break
        x = max(cand, key=lambda v: (work[v], -v))
        t = task_of[x]
        if len(cand) - 1 + sum(1 for y in (int(tasks[t][0]), int(tasks[t][1])) if y >= numseq) < len(cand):
            break                                              # splitting would not add a subtree (both children are leaves)
        frontier.remove(x)
        frontier += [int(tasks[t][0]), int(tasks[t][1])]
        top.append(t)
    top.sort()
    roots = sorted([x for x in frontier if x >= numseq], key=lambda v: (-work[v], v))
    run_rank = np.full(nt, -1, np.int64)
    holder = {}

    def assign(node, r):
        stack = [node]
        while stack:
            v = stack.pop()
            if v < numseq:
                continue
            t = task_of[v]
            run_rank[t] = r
            stack += [int(tasks[t][0]), int(tasks[t][1])]

    for r, x in enumerate(roots):
        assign(x, r % world)
        holder[x] = r % world
    for t in top:                                              # tree order: children first
        a, b, c = (int(v) for v in tasks[t])
        ha, hb = holder.get(a), holder.get(b)
        if ha is None and hb is None:
            r = 0
        elif ha is None or (hb is not None and members[b] > members[a]):
            r = hb
        else:
            r = ha
        run_rank[t] = r
        holder[c] = r
    return run_rank, top


def _gather_bytes(arr, world, device):
    """all_gather of ragged uint8 arrays -> list of numpy uint8 arrays, one per rank"""
    import torch
    import torch.distributed as dist
    n = torch.tensor([len(arr)], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    mx = max(int(x.item()) for x in ns)
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=device)
    if len(arr):
        buf[:len(arr)] = torch.as_tensor(np.ascontiguousarray(arr), dtype=torch.uint8, device=device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [o.cpu().numpy()[:int(k.item())] for o, k in zip(out, ns)]


class _DevArray:
    """a raw device pointer as a CUDA-array-interface object (torch.as_tensor aliases it, no copy)"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def dev_tensor(ptr, n, dtype):
    """torch tensor over `n` elements of library-owned HBM at `ptr` (float32 / int32)"""
    import torch
    typestr = {torch.float32: "<f4", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device="cuda")


def sharded_consistency(ex, n_anchors, weight, rank, world):
    """anchor_consistency_build over `world` GPUs: every rank aligns its share of the N x K seq-seq
    batch (contiguous, length-balanced ranges of sequences) and fills their position maps in its copy of the table;
    then every rank's range is broadcast IN PLACE into the other ranks' tables (HBM to HBM over RCCL / xGMI; the
    table never visits the host).  ex: kalign_amd.Context after tree_upload, or anything with cons_build_part(part,
    nparts), cons_table() -> 1-D int32 torch tensor aliasing the table, cons_part_range(part, nparts)."""
    import torch.distributed as dist
    if world == 1:
        ex.cons_build_part(n_anchors, weight, rank, world)
        return
    # a part can fail on ONE rank (e.g. a share that holds only anchors): every rank must learn of it before anybody
    # enters a broadcast the failed rank never joins
    import torch
    err = None
    try:
        ex.cons_build_part(n_anchors, weight, rank, world)
    except Exception as e:                                        # noqa: BLE001 -- re-raised below, on every rank
        err = e
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    bad = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(bad, op=dist.ReduceOp.SUM)
    if int(bad.item()):
        if err is not None:
            raise err
        raise RuntimeError("sharded consistency: the part of another rank could not be built")
    table = ex.cons_table()
    if table is None:                                             # the job declined (no distances / fewer than 3 sequences)
        return
    # RCCL moves the ranges HBM to HBM in place.  Any other backend (gloo: the one-GPU tests) goes through host tensors:
    # gloo would write into the device table from the CPU side, past the GPU's caches.
    direct = (not table.is_cuda) or dist.get_backend() == "nccl"
    for r in range(world):
        lo, hi = ex.cons_part_range(r, world)
        if hi <= lo:
            continue
        if direct:
            dist.broadcast(table[lo:hi], src=r)
        else:
            tmp = table[lo:hi].cpu() if r == dist.get_rank() else torch.empty(hi - lo, dtype=table.dtype)
            dist.broadcast(tmp, src=r)
            if r != dist.get_rank():
                table[lo:hi].copy_(tmp)
    if table.is_cuda:
        torch.cuda.synchronize()                                  # the table is complete before any kernel reads it


def sharded_tree(ex, tasks, lens, rank, world, rec_type, device="cpu"):
    """One guide tree over `world` ranks (one process per GPU).

    ex: this rank's executor over an uploaded job -- kalign_amd.Context after tree_upload, or anything
    with the same four methods: tree_run_tasks(ids), tree_get_node(node) -> float32[(plen+2)*64],
    tree_set_node(node, prof), tree_download_tasks(ids) -> (recs, paths).
    rec_type: the ctypes record class (kalign_amd.api.TaskRec) used to move records between ranks.
    Returns (recs in task order with path_off into paths, paths) on every rank; identical for any world."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    tasks = np.asarray(tasks)
    numseq = len(lens)
    run_rank, top = plan_subtrees(tasks, lens, world)
    top_set = set(top)
    mine = [t for t in range(len(tasks)) if run_rank[t] == rank and t not in top_set]
    if mine:
        ex.tree_run_tasks(mine)
    holder = {}
    for t in range(len(tasks)):
        if t not in top_set:
            holder[int(tasks[t][2])] = int(run_rank[t])
    for t in top:
        a, b, c = (int(v) for v in tasks[t])
        dst = int(run_rank[t])
        for child in (a, b):
            src = holder.get(child)
            if child < numseq or src is None or src == dst or world == 1:
                continue
            # (device pointers only over RCCL: gloo would fill the arena from the CPU side, past the GPU's caches)
            d2d = hasattr(ex, "tree_profile_dev") and str(device).startswith("cuda") and dist.get_backend() == "nccl"
            if d2d:
                # HBM to HBM: the profile is sent from where it lies in the source's arena into room reserved in the
                # destination's arena (RCCL send / recv over xGMI); only the residue -> column table of a default-mode
                # job (ints, scattered over the member sequences on the device) is packed through the host
                if rank == src:
                    ptr, plen = ex.tree_profile_dev(child)
                    cols = ex.tree_node_cols(child)
                    ncols = 0 if cols is None else len(cols)
                    dist.send(torch.tensor([plen, ncols], dtype=torch.int64, device=device), dst)
                    dist.send(dev_tensor(ptr, (plen + 2) * 64, torch.float32), dst)
                    if ncols:
                        dist.send(torch.as_tensor(cols, dtype=torch.int32, device=device), dst)
                elif rank == dst:
                    meta = torch.zeros(2, dtype=torch.int64, device=device)
                    dist.recv(meta, src)
                    plen, ncols = int(meta[0].item()), int(meta[1].item())
                    ptr = ex.tree_reserve_profile_dev(child, plen)
                    dist.recv(dev_tensor(ptr, (plen + 2) * 64, torch.float32), src)
                    if ncols:
                        cbuf = torch.zeros(ncols, dtype=torch.int32, device=device)
                        dist.recv(cbuf, src)
                        ex.tree_set_node_cols(child, cbuf.cpu().numpy())
                    torch.cuda.synchronize()
            elif rank == src:
                prof = ex.tree_get_node(child)
                dist.send(torch.tensor([len(prof)], dtype=torch.int64, device=device), dst)
                dist.send(torch.as_tensor(prof, dtype=torch.float32, device=device), dst)
            elif rank == dst:
                n = torch.zeros(1, dtype=torch.int64, device=device)
                dist.recv(n, src)
                buf = torch.zeros(int(n.item()), dtype=torch.float32, device=device)
                dist.recv(buf, src)
                ex.tree_set_node(child, buf.cpu().numpy())
        if rank == dst:
            ex.tree_run_tasks([t])
        holder[c] = dst
    ran = [t for t in range(len(tasks)) if run_rank[t] == rank]
    recs, paths = ex.tree_download_tasks(ran) if ran else ([], np.zeros(0, np.int32))
    if world == 1:
        return list(recs), np.asarray(paths, np.int32)
    rec_bytes = np.frombuffer(b"".join(bytes(r) for r in recs), np.uint8) if recs else np.zeros(0, np.uint8)
    g_ids = _gather_bytes(np.asarray(ran, np.int32).view(np.uint8), world, device)
    g_recs = _gather_bytes(rec_bytes, world, device)
    g_paths = _gather_bytes(np.asarray(paths, np.int32).view(np.uint8), world, device)
    rs = C.sizeof(rec_type)
    all_recs = [None] * len(tasks)
    chunks, off = [], 0
    for r in range(world):
        ids = g_ids[r].view(np.int32)
        rp = g_paths[r].view(np.int32)
        for i, t in enumerate(ids):
            rec = rec_type.from_buffer_copy(g_recs[r][i * rs:(i + 1) * rs].tobytes())
            rec.path_off += off
            all_recs[int(t)] = rec
        chunks.append(rp)
        off += len(rp)
    return all_recs, (np.concatenate(chunks) if chunks else np.zeros(0, np.int32))


# ------------------------------------------------------------------------------------------------
# ensemble members, one per GPU (kalign_ensemble's loop, lib/src/ensemble.c:286-339)
# ------------------------------------------------------------------------------------------------
def member_on_context(ctx, tree_codes, codes, letters, subm, n_anchors=0, weight=2.0, n_threads=1, realign=0):
    """run_member for ensemble_members on a kalign_amd.Context: one member of the reference's ensemble loop.
    realign == 0: kalign_run_seeded -- guide tree (noisy when the member carries dm_scale), consistency, task tree,
    final rows.  realign > 0 (`--precise`): kalign_run_realign -- after the first alignment, `realign` times: identity
    distances of the rows, UPGMA tree, alignment on that tree with the first pass's consistency table
    (aln_wrap.c:449-504).  A member is a dict: scal (gpo, gpe, tgpe, dist_scale, vsm_amax, use_seq_weights) and
    optionally dm_scale."""
    from . import api

    def run(member):
        tasks, sd = ctx.guide_tree(tree_codes, n_threads=n_threads, dm_scale=None if realign else member.get("dm_scale"))
        ctx.msa_tree(codes, tasks, subm, member["scal"], sd, n_anchors=n_anchors, weight=weight)
        rows = ctx.tree_aligned_rows(letters)
        for _ in range(realign):
            tasks, sd = ctx.aln_guide_tree()
            ctx.tree_upload(codes, tasks, subm, member["scal"], sd, flags=api.FLAG_DEVICE_GAPS | api.FLAG_KEEP_CONSISTENCY)
            ctx.tree_run()
            rows = ctx.tree_aligned_rows(letters)
        return rows
    return run


def ensemble_members(run_member, members, rank, world, device="cpu"):
    """Member k runs on rank k % world (replicas with different parameters: no data-path collective); the aligned
    rows of every member are gathered on every rank for the consensus stage, which stays on the host
    (POAR tables, lib/src/ensemble.c:341-).  Returns rows[k] = list of bytes, one per sequence."""
    mine = [k for k in range(len(members)) if k % world == rank]
    local = {k: run_member(members[k]) for k in mine}
    if world == 1:
        return [local[k] for k in range(len(members))]
    # flatten this rank's members: per member (n rows, row length) + the row bytes
    head = np.array([[k, len(local[k]), len(local[k][0]) if local[k] else 0] for k in mine], np.int64).reshape(-1, 3)
    body = np.frombuffer(b"".join(b"".join(local[k]) for k in mine), np.uint8) if mine else np.zeros(0, np.uint8)
    heads = _gather_bytes(head.view(np.uint8).reshape(-1), world, device)
    bodies = _gather_bytes(body, world, device)
    out = [None] * len(members)
    for r in range(world):
        h = heads[r].view(np.int64).reshape(-1, 3)
        o = 0
        for k, n, width in h:
            rows = bodies[r][o:o + int(n) * int(width)].reshape(int(n), int(width))
            out[int(k)] = [row.tobytes() for row in rows]
            o += int(n) * int(width)
    return out
