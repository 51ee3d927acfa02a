"""GPU: the N > 1 path of bench.py end to end on ONE GPU -- world_size ranks over gloo, all of them on cuda:0 (contexts
in shared mode), driving the real library through kalign_amd.dist: the consistency batch sharded with in-place broadcasts
of the device table, the guide tree cut into one subtree per rank, subtree roots handed over through the device-pointer
ABI (ka_tree_profile_dev / ka_tree_reserve_profile_dev), records and paths gathered.  The same code runs over RCCL
with one GPU per rank (the driver's multi-GPU bench); here the result must equal a whole-tree run on one context."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,extra", [(2, []), (4, []), (2, ["--scale-fast"])], ids=["w2_default", "w4_default", "w2_fast"])
def test_sharded_alignment_equals_single_gpu_run(world, extra):
    env = dict(os.environ, KA_BENCH_BACKEND="gloo")
    port = 29600 + (os.getpid() + 17 * world + len(extra)) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--scale-workload", "--nseq", "512", "--len", "300"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["config"]["identical_results_on_all_ranks"] is True
    assert d["config"]["identical_to_a_single_gpu_run"] is True


def test_rccl_path_with_a_world_of_one():
    """The driver's multi-GPU launch uses backend "nccl" (RCCL), which a one-GPU box can only run with one rank: process
    group with a device id, barriers, the max-reduction of the clock and the in-place broadcasts on HBM tensors all
    execute over RCCL; the result must still equal the whole-tree run."""
    env = dict(os.environ, KA_BENCH_FORCE_MULTI="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("KA_BENCH_BACKEND", None)
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--scale-workload", "--nseq", "512", "--len", "300"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert d["config"]["identical_to_a_single_gpu_run"] is True
