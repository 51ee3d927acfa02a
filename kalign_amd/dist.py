"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" for the CPU tests).

What shards:
  * the N x K seq-seq batch of anchor consistency and any list of independent pairwise tasks:
    contiguous, cost-balanced slices per rank, no data-path collective, results gathered at the
    end (paths are small: 4*(La+Lb+2) bytes per pair);
  * independent sequence sets / ensemble members: one per rank (what bench.py scales).
A single guide tree is NOT sharded in this round (its top is a serial chain of big tasks); see
DESIGN.md "Multi-GPU".
"""
import os

import numpy as np


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device=None):
    """init_process_group from the torchrun environment; returns (rank, world)."""
    import torch.distributed as dist
    rank, world, _ = env_world()
    if world > 1 and not dist.is_initialized():
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
