#!/usr/bin/env python3
"""First contact with RCCL through gpu-ntt_amd/dist.py's OWN helpers (run under torchrun, one rank per GPU):
  1. init_process_group("nccl", device_id = cuda:LOCAL_RANK)      2. 1-element all_reduce (SUM and MAX)
  3. 1 MiB broadcast from rank 0      4. 1 MiB-per-rank scatter from rank 0      5. gather back to rank 0
every payload checked word for word.  Prints one line `RCCL_SMOKE {...json...}` on rank 0; exit code 0 = pass.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rccl_smoke.py
`--backend gloo` runs the same steps on CPU tensors (tests/test_dist_shard.py does that here, without a GPU)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--words", type=int, default=1 << 17)  # 1 MiB of int64
    args = ap.parse_args()
    import torch
    _load_pkg()
    dm = importlib.import_module("gpu_ntt_amd.dist")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = None
    if args.backend == "nccl":
        torch.cuda.set_device(local)
        dev = "cuda:%d" % local
    t0 = time.perf_counter()
    dist, rank, world = dm.init_process_group(args.backend, dev)
    if dist is None:
        print("RCCL_SMOKE " + json.dumps({"ok": False, "error": "WORLD_SIZE is 1: run under torchrun"}))
        return 2
    tdev = dev or "cpu"
    res = {"backend": dist.get_backend(), "world": world, "init_s": time.perf_counter() - t0}
    one = torch.tensor([rank + 1], dtype=torch.int64, device=tdev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    assert int(one[0]) == world * (world + 1) // 2, "all_reduce SUM"
    mx = torch.tensor([float(rank)], dtype=torch.float64, device=tdev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    assert float(mx[0]) == world - 1, "all_reduce MAX"
    w = args.words
    pattern = torch.arange(w, dtype=torch.int64, device=tdev) * 0x9E3779B1 + 7
    table = pattern.clone() if rank == 0 else torch.zeros(w, dtype=torch.int64, device=tdev)
    res["broadcast_s"] = dm._timed_collective(lambda: dm._broadcast(dist, table, 0), dist, dev)
    assert torch.equal(table, pattern), "broadcast payload"
    full = torch.cat([pattern + r for r in range(world)]) if rank == 0 else None
    shard = torch.zeros(w, dtype=torch.int64, device=tdev)
    res["scatter_s"] = dm._timed_collective(lambda: dm._scatter(dist, rank, world, full, shard), dist, dev)
    assert torch.equal(shard, pattern + rank), "scatter payload"
    shard_out = shard * 3 + rank
    gathered = torch.zeros(world * w, dtype=torch.int64, device=tdev) if rank == 0 else None
    res["gather_s"] = dm._timed_collective(lambda: dm._gather(dist, rank, world, shard_out, gathered), dist, dev)
    if rank == 0:
        want = torch.cat([(pattern + r) * 3 + r for r in range(world)])
        assert torch.equal(gathered, want), "gather payload"
    digs = dm.gather_digests(shard_out.cpu().numpy(), dist)
    assert len(digs) == world and len(set(digs)) == world
    dist.barrier()
    res["ok"] = True
    if rank == 0:
        print("RCCL_SMOKE " + json.dumps(res), flush=True)
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except AssertionError as e:
        print("RCCL_SMOKE " + json.dumps({"ok": False, "error": "payload check failed: %s" % e}), flush=True)
        sys.exit(1)
