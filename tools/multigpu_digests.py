#!/usr/bin/env python3
"""Sharded result == single-GPU result, by digest (run under torchrun, one rank per GPU): a seeded batch of Merge
transforms (u64 2^16 x 64 per rank by default, C5's RNS stack with --rns) is transformed shard by shard -- rank r owns
polynomials [lo, hi) of the global batch (gpu_ntt_amd.shard_range) -- and the SHA-256 of every shard is all-gathered;
rank 0 also transforms the WHOLE batch on its own GPU and compares the digests of the same slices.  One line
`MULTIGPU_DIGESTS {...}` on rank 0; exit code 0 = identical."""
import argparse
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402
from bench import splitmix64_mod  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--per-rank", type=int, default=64)
    args = ap.parse_args()
    import torch
    g = _load_pkg()
    g.load_library()
    dm = importlib.import_module("gpu_ntt_amd.dist")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GPUNTT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    dist, rank, world = dm.init_process_group(backend, dev)
    logn, n = args.logn, 1 << args.logn
    batch = args.per_rank * world
    prm = g.NTTParameters(logn, g.X_N_plus, 64)
    table = g.to_device(prm.forward_table_device_order, dev)
    cfg = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=g.X_N_plus)
    lo, hi = g.shard_range(batch, rank, world)
    x = splitmix64_mod(0xD16E57, (hi - lo) * n, prm.modulus.value, offset=lo * n)  # global index decides the values
    d = g.to_device(x, dev)
    g.GPU_NTT_Inplace(d, table, prm.modulus, cfg, hi - lo)
    torch.cuda.synchronize()
    digs = dm.gather_digests(g.to_host(d), dist)
    ok = True
    if rank == 0:
        xa = splitmix64_mod(0xD16E57, batch * n, prm.modulus.value)
        da = g.to_device(xa, dev)
        g.GPU_NTT_Inplace(da, table, prm.modulus, cfg, batch)
        torch.cuda.synchronize()
        ya = g.to_host(da)
        want = []
        for r in range(world):
            a, b = g.shard_range(batch, r, world)
            want.append(hashlib.sha256(np.ascontiguousarray(ya[a * n:b * n]).tobytes()).hexdigest())
        ok = digs == want
        print("MULTIGPU_DIGESTS " + json.dumps({"ok": ok, "world": world, "batch": batch, "log2N": logn,
                                                "shards_equal": [a == b for a, b in zip(digs, want)]}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
