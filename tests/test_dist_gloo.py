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
