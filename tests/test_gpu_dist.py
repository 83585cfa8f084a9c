"""The multi-GPU layer with DEVICE tensors on the one GPU a test box has: two ranks share cuda:0, rendezvous over gloo
(RCCL refuses two ranks on one device), device shards are staged through the host by gpu-ntt_amd/dist.py, and the
per-rank transform is the real library call.  What an 8-GPU lease would otherwise find first -- import, dtype and
device-placement errors in shard_range / scatter_transform_gather / end_to_end_leg / timed_region -- is found here."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, hashlib, importlib
import numpy as np
import torch
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
g = load_pkg()
g.load_library()
dist_mod = importlib.import_module("gpu_ntt_amd.dist")
dev = "cuda:0"
torch.cuda.set_device(0)
dist, rank, world = dist_mod.init_process_group("gloo", dev)
assert world == 2
logn, batch = 13, 16            # C4-shaped: one modulus, strong scaling over the ranks
n = 1 << logn
prm = g.NTTParameters(logn, g.X_N_minus, 64)
lo, hi = g.shard_range(batch, rank, world)
per = hi - lo
table = g.to_device(prm.forward_table_device_order, dev) if rank == 0 else \
        torch.zeros(prm.root_of_unity_size, dtype=torch.int64, device=dev)
cfg = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=g.X_N_minus)
run_shard = lambda a, b: g.GPU_NTT(a, b, table, prm.modulus, cfg, per)
sys.path.insert(0, ROOT)
import bench
x_full = bench.splitmix64_mod(77, batch * n, prm.modulus.value)
d_in = g.to_device(x_full[lo * n:hi * n], dev)
d_out = torch.empty_like(d_in)
# table broadcast (rank 1 starts from zeros), scatter, transform, gather: the bench's end-to-end leg
e2e = dist_mod.end_to_end_leg(dist, rank, world, dev, table, d_in, d_out, run_shard, per)
assert e2e["transform_ms"] > 0 and e2e["bytes_scattered"] == per * n * 8
# ... and with a batch that really lives on rank 0
full = g.to_device(x_full, dev) if rank == 0 else None
shard = torch.empty(per * n, dtype=torch.int64, device=dev)
gathered, t = dist_mod.scatter_transform_gather(dist, rank, world, full, shard, d_out, run_shard, dev)
wall = dist_mod.timed_region(lambda: run_shard(shard, d_out), 3, dist, dev)
assert wall > 0
if rank == 0:
    print("E2E", hashlib.sha256(g.to_host(gathered).tobytes()).hexdigest(), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_end_to_end_leg_with_device_tensors_two_ranks_one_gpu(tmp_path):
    from oracle import oracle as O
    sys.path.insert(0, ROOT)
    import bench
    P = O.Port(64)
    logn, batch = 13, 16
    n = 1 << logn
    oprm = P.merge_params(logn, O.X_N_minus)
    x = bench.splitmix64_mod(77, batch * n, oprm["mod"][0])
    want = np.concatenate([P.merge_ntt(x[p * n:(p + 1) * n], oprm) for p in range(batch)])
    script = tmp_path / "worker.py"
    script.write_text("ROOT = %r\n" % ROOT + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = [l for l in r.stdout.splitlines() if l.startswith("E2E")][0].split()[1]
    assert got == hashlib.sha256(want.tobytes()).hexdigest()


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """`python bench.py --gpus 2` end to end on the one-GPU box: the script starts its own two ranks, both mapped onto
    cuda:0, control-plane collectives over gloo (GPUNTT_BENCH_BACKEND, a diagnostic switch; RCCL is the default) -- the
    N-rank code path of the headline config (shard geometry, barrier + MAX timing, end-to-end leg behind its watchdog)
    and of the sweep, with the real library calls."""
    import json
    env = dict(os.environ, GPUNTT_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "batch-shard x2"
    assert "error" not in line["end_to_end"], line["end_to_end"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sweep", "--sweep-min", "12",
                          "--sweep-max", "12", "--sweep-kinds", "merge"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0
