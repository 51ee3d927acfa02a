"""The N > 1 path on CPU: world_size 2 over gloo (what RCCL does on the GPU box).  The compute
leg is the oracle (tests may use it); what is under test is the sharding, the gather and the
barrier/MAX-reduce timing helpers of kalign_amd/dist.py that bench.py uses."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from kalign_amd import dist as kd
    from oracle import oracledrv
    from util import Golden
    r, w = kd.init(backend="gloo")
    assert (r, w) == (rank, world)
    g = Golden("pairs_prot12x90")

    def compute(lo, hi):
        return oracledrv.pairwise_batch(g.codes, g.ia[lo:hi], g.ib[lo:hi], g.subm,
                                        float(g.scal[0]), float(g.scal[1]), float(g.scal[2]))

    paths, scores = kd.sharded_pairwise(compute, g.lens, g.ia, g.ib, rank, world)
    tmax = kd.reduce_scalar(1.0 + rank, "max")
    tsum = kd.reduce_scalar(10.0, "sum")
    dist.barrier()
    if rank == 0:
        np.savez(out, n=len(paths), flat=np.concatenate(paths), scores=scores, tmax=tmax, tsum=tsum)
    dist.destroy_process_group()


def test_partition_covers_all_units():
    from kalign_amd.dist import partition
    rng = np.random.RandomState(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 100):
            costs = rng.uniform(1, 10, n)
            parts = partition(costs, world)
            assert len(parts) == world
            assert parts[0][0] == 0 and parts[-1][1] == n
            for (a, b), (c, d) in zip(parts[:-1], parts[1:]):
                assert b == c and a <= b
            if n >= 4 * world and world > 1:
                loads = [costs[a:b].sum() for a, b in parts]
                assert max(loads) <= 2.0 * costs.sum() / world


@pytest.mark.timeout(120)
def test_sharded_pairwise_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    from util import Golden
    out = str(tmp_path / "r0.npz")
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(out)
    g = Golden("pairs_prot12x90")
    assert int(z["n"]) == len(g.ia)
    assert np.array_equal(z["flat"], g.paths)          # identical to the reference's paths, in pair order
    assert float(z["tmax"]) == 2.0 and float(z["tsum"]) == 20.0


# ------------------------------------------------------------------------------------------------
# one guide tree over two ranks (kalign_amd.dist.sharded_tree)
# ------------------------------------------------------------------------------------------------
def _tree_worker(rank, world, port, out, case):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from kalign_amd import dist as kd
    from oracle import oracledrv
    from oracle_executor import OracleExecutor
    from util import Golden
    kd.init(backend="gloo")
    g = Golden(case)
    ex = OracleExecutor(g.codes, g.tasks, g.subm, g.scal, g.rec("gap_scale"), g.rec("subm_off"))
    recs, paths = kd.sharded_tree(ex, g.tasks, g.lens, rank, world, oracledrv.TaskRec)
    ran_here = len(ex.done)
    dist.barrier()
    np.savez(out % rank, plen=[r.plen for r in recs], path_off=[r.path_off for r in recs],
             score=np.array([r.score for r in recs], np.float32), abc=[[r.a, r.b, r.c] for r in recs],
             paths=paths, ran_here=ran_here)
    dist.destroy_process_group()


def test_plan_subtrees_is_a_partition():
    from kalign_amd.dist import plan_subtrees
    from util import Golden
    for case in ("tree_prot64_gon", "tree_ragged", "tree_BB11001"):
        g = Golden(case)
        for world in (1, 2, 3, 8):
            run_rank, top = plan_subtrees(g.tasks, g.lens, world)
            assert ((run_rank >= 0) & (run_rank < world)).all()
            assert sorted(top) == top and (not top or top[-1] == len(g.tasks) - 1 or world == 1)
            # below the cut a task runs where its children ran (no transfers inside a subtree)
            where = {int(c): int(run_rank[t]) for t, (_, _, c) in enumerate(g.tasks)}
            for t, (a, b, c) in enumerate(g.tasks):
                if t in top:
                    continue
                for child in (int(a), int(b)):
                    if child >= len(g.lens):
                        assert where[child] == run_rank[t]


@pytest.mark.timeout(180)
@pytest.mark.parametrize("case", ["tree_prot32x200", "tree_prot24_scaled"])
def test_sharded_tree_world2_gloo(tmp_path, case):
    """Two ranks, one subtree each, the profile of one subtree root crosses ranks for the root task:
    records, coded paths and gap arrays identical to the single-process reference run."""
    import torch.multiprocessing as mp
    from kalign_amd import api
    from util import Golden
    out = str(tmp_path / "r%d.npz")
    port = 29500 + ((os.getpid() + 77) % 500)
    mp.spawn(_tree_worker, args=(2, port, out, case), nprocs=2, join=True)
    g = Golden(case)
    z0, z1 = np.load(out % 0), np.load(out % 1)
    assert int(z0["ran_here"]) > 0 and int(z1["ran_here"]) > 0           # both ranks did work
    assert int(z0["ran_here"]) + int(z1["ran_here"]) == len(g.tasks)
    for z in (z0, z1):                                                    # every rank holds the full result
        assert np.array_equal(z["plen"], g.rec("plen"))
        assert np.array_equal(z["score"], g.rec("score"))
        assert np.array_equal(z["abc"], g.tasks)
        recs = []
        for t in range(len(g.tasks)):
            o, n = int(z["path_off"][t]), int(z["plen"][t])
            assert np.array_equal(z["paths"][o:o + n + 2], g.path(t)), t
            r = api.TaskRec()
            r.a, r.b, r.c = (int(v) for v in g.tasks[t])
            r.path_off, r.plen = o, n
            recs.append(r)
        gaps = api.weave_gaps(g.lens, recs, z["paths"])                   # host-only C function, no GPU needed
        for got, want in zip(gaps, g.gaps_list()):
            assert np.array_equal(got, want)


@pytest.mark.timeout(240)
def test_sharded_tree_world4_gloo(tmp_path):
    """Four ranks: three levels of the tree above the cut, profiles cross ranks twice on the way to the root; every
    rank ends up with the whole result, identical to the single-process golden."""
    import torch.multiprocessing as mp
    from util import Golden
    case = "tree_prot64_gon"
    out = str(tmp_path / "q%d.npz")
    port = 29500 + ((os.getpid() + 131) % 500)
    mp.spawn(_tree_worker, args=(4, port, out, case), nprocs=4, join=True)
    g = Golden(case)
    zs = [np.load(out % r) for r in range(4)]
    assert all(int(z["ran_here"]) > 0 for z in zs) and sum(int(z["ran_here"]) for z in zs) == len(g.tasks)
    for z in zs:
        assert np.array_equal(z["plen"], g.rec("plen")) and np.array_equal(z["score"], g.rec("score"))
        for t in range(len(g.tasks)):
            o, n = int(z["path_off"][t]), int(z["plen"][t])
            assert np.array_equal(z["paths"][o:o + n + 2], g.path(t)), t


# ------------------------------------------------------------------------------------------------
# the N x K consistency batch over several ranks (kalign_amd.dist.sharded_consistency) -- default mode
# ------------------------------------------------------------------------------------------------
class _ConsExecutor:
    """CPU stand-in of kalign_amd.Context's consistency interface: its share of the position maps comes from the
    oracle's seq-seq alignments (tests only); the table is a torch tensor like the device table of the real thing."""

    def __init__(self, g):
        import torch
        self.g = g
        self.lens = np.asarray(g.lens, np.int64)
        self.K = int(g.n_anchors)
        self.off = np.concatenate([[0], np.cumsum(self.lens * self.K)])
        self.table = torch.full((int(self.off[-1]),), -7, dtype=torch.int32)

    def _seq_range(self, part, nparts):
        # (the library's rule: contiguous ranges with balanced total length, ka_api.cpp:cons_part_seqs)
        total, n = int(self.lens.sum()), len(self.lens)

        def cut(r):
            if r <= 0:
                return 0
            if r >= nparts:
                return n
            target, acc, i = total * r // nparts, 0, 0
            while i < n and acc < target:
                acc += int(self.lens[i]); i += 1
            return i
        return cut(part), cut(part + 1)

    def cons_part_range(self, part, nparts):
        lo, hi = self._seq_range(part, nparts)
        return int(self.off[lo]), int(self.off[hi])

    def cons_table(self):
        return self.table

    def cons_build_part(self, n_anchors, weight, part, nparts):
        import torch
        from oracle import oracledrv
        g = self.g
        ids = [int(x) for x in g.anchor_ids]
        lo, hi = self._seq_range(part, nparts)
        for i in range(lo, hi):
            for k, a in enumerate(ids):
                if a == i:
                    m = np.arange(self.lens[i], dtype=np.int32)
                else:
                    paths, _ = oracledrv.pairwise_batch(g.codes, np.array([i], np.int32), np.array([a], np.int32), g.subm,
                                                        float(g.scal[0]), float(g.scal[1]), float(g.scal[2]))
                    p, m, pa, pb = paths[0], np.full(self.lens[i], -1, np.int32), 0, 0
                    for c in p[1:]:                                    # path -> position map, anchor_consistency.c:93-114
                        if c == 3:
                            break
                        if c == 0:
                            m[pa] = pb; pa += 1; pb += 1
                        elif c & 1:
                            pb += 1
                        else:
                            pa += 1
                o = int(self.off[i]) + k * int(self.lens[i])
                self.table[o:o + int(self.lens[i])] = torch.as_tensor(m)


def _cons_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from kalign_amd import dist as kd
    from util import Golden
    kd.init(backend="gloo")
    ex = _ConsExecutor(Golden("cons_prot32x200"))
    kd.sharded_consistency(ex, ex.K, 2.0, rank, world)
    dist.barrier()
    np.savez(out % rank, table=ex.table.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_sharded_consistency_world4_gloo(tmp_path):
    """default mode: four ranks align a quarter of the N x K batch each; after the in-place broadcasts every rank holds
    the complete position-map table, equal to the reference's (golden cons_prot32x200)"""
    import torch.multiprocessing as mp
    from util import Golden
    out = str(tmp_path / "c%d.npz")
    port = 29500 + ((os.getpid() + 303) % 500)
    mp.spawn(_cons_worker, args=(4, port, out), nprocs=4, join=True)
    g = Golden("cons_prot32x200")
    want = np.concatenate([np.asarray(m, np.int32) for row in g.maps_list() for m in row])
    for r in range(4):
        assert np.array_equal(np.load(out % r)["table"], want), r


# ------------------------------------------------------------------------------------------------
# ensemble members, one per rank (kalign_amd.dist.ensemble_members)
# ------------------------------------------------------------------------------------------------
def _members():
    """three members in the spirit of ensemble.c:55-76: the default one, and two with scaled penalties and a noisy tree"""
    from util import Golden
    g = Golden("tree_prot32x200")
    rng = np.random.RandomState(3)
    out = [dict(scal=g.scal.copy())]
    for f in (0.8, 1.3):
        s = g.scal.copy()
        s[:3] *= f
        out.append(dict(scal=s, dm_scale=np.maximum(0.1, rng.normal(1.0, 0.3, len(g.lens) * min(32, len(g.lens)))).astype(np.float32)))
    return g, out


def _oracle_member(g):
    """run_member built from the oracle and the host-side tree builder (the CPU stand-in of dist.member_on_context)"""
    from kalign_amd import api
    from oracle import oracledrv

    def run(member):
        tasks, sd = api.guide_tree_from(g.lens, lambda ia, ib: oracledrv.bpm_batch(g.tree_seqs, ia, ib), dm_scale=member.get("dm_scale"))
        _, _, gaps, _ = oracledrv.msa_tree(g.codes, tasks, g.subm, member["scal"], sd)
        return [r.encode() for r in oracledrv.rows_from_gaps(g.sorted_seqs(), gaps)]
    return run


def _ens_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from kalign_amd import dist as kd
    kd.init(backend="gloo")
    g, members = _members()
    ran = []
    run = _oracle_member(g)

    def counted(m):
        ran.append(1)
        return run(m)
    rows = kd.ensemble_members(counted, members, rank, world)
    dist.barrier()
    np.savez(out % rank, ran=len(ran), **{"m%d" % k: np.array(r) for k, r in enumerate(rows)})
    dist.destroy_process_group()


def test_ensemble_members_world2_gloo(tmp_path):
    """members are dealt round-robin, every rank ends up with every member's rows, and they equal a single-process run"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ens%d.npz")
    port = 29500 + ((os.getpid() + 211) % 500)
    mp.spawn(_ens_worker, args=(2, port, out), nprocs=2, join=True)
    g, members = _members()
    from kalign_amd import dist as kd
    want = kd.ensemble_members(_oracle_member(g), members, 0, 1)
    assert len(set(len(r) for r in want[0])) == 1 and want[0] != want[1]          # members really differ
    z0, z1 = np.load(out % 0), np.load(out % 1)
    assert int(z0["ran"]) == 2 and int(z1["ran"]) == 1
    for k in range(len(members)):
        for z in (z0, z1):
            assert [bytes(x) for x in z["m%d" % k]] == want[k]
