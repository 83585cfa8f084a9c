"""Multi-GPU batch sharding: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on
ROCm, "gloo" for CPU-side tests).

The reference has no distributed code at all; its only parallel axis is the
batch of independent polynomials (gridDim.z, reference ntt.cu:2125).  The MI355X build shards
that axis: rank r of G owns polynomials [lo, hi) and runs the ordinary single-GPU entry points
on its shard.  A transform never crosses a GPU, so there is NO data-path collective; the only
collectives are control-plane: a barrier around timed regions, a MAX-reduction of elapsed
times, and (optionally) a digest all-gather to check results.

When the batch ORIGINATES on one GPU there is data movement around the transform (SURVEY.md 8e ii):
one broadcast of the twiddle table at set-up, a scatter of the shards before and a gather after.
`scatter_transform_gather` / `end_to_end_leg` do exactly that with RCCL (grouped send/recv under
torch.distributed.scatter / gather) and time each part on its own, so a multi-GPU run reports the
resident-shard throughput and the end-to-end figure side by side.
"""
import hashlib
import os
import time

import numpy as np

from . import shard_range  # noqa: F401  (re-exported)


def init_process_group(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (dist | None, rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return None, 0, 1
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl" and device is not None:
        kwargs["device_id"] = torch.device(device)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return dist, rank, world


def _sync(device):
    if device is not None and str(device).startswith("cuda"):
        import torch
        torch.cuda.synchronize()


def timed_region(fn, steps, dist=None, device=None):
    """Run fn() `steps` times bracketed by barrier + device synchronise on both sides and
    return the wall time in seconds, MAX-reduced over ranks (the slowest rank defines the
    job's throughput)."""
    import torch
    _sync(device)
    if dist is not None:
        dist.barrier()
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _sync(device)
    if dist is not None:
        dist.barrier()
    _sync(device)
    wall = time.perf_counter() - t0
    if dist is not None:
        dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    return wall


def gather_digests(local_array, dist=None):
    """SHA-256 of the local shard's bytes from every rank (rank order); a cheap way to compare
    a sharded run with a single-process run without moving the data."""
    h = hashlib.sha256(np.ascontiguousarray(local_array).tobytes()).hexdigest()
    if dist is None:
        return [h]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, h)
    return out


def _timed_collective(fn, dist, device):
    """seconds of fn() bracketed by barrier + device synchronise, MAX over ranks"""
    import torch
    _sync(device)
    dist.barrier()
    _sync(device)
    t0 = time.perf_counter()
    fn()
    _sync(device)
    dist.barrier()
    _sync(device)
    dt = time.perf_counter() - t0
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def _staged(dist, t):
    """gloo moves host memory only (its scatter / gather have no device form): device tensors are staged through
    the host around the collective.  RCCL ("nccl") takes them as they are."""
    return t is not None and t.is_cuda and dist.get_backend() != "nccl"


def _scatter(dist, rank, world, full, shard_in):
    if not _staged(dist, shard_in):
        dist.scatter(shard_in, list(full.chunk(world)) if rank == 0 else None, src=0)
        return
    import torch
    host = torch.empty(shard_in.shape, dtype=shard_in.dtype)
    dist.scatter(host, list(full.cpu().chunk(world)) if rank == 0 else None, src=0)
    shard_in.copy_(host)


def _gather(dist, rank, world, shard_out, gathered):
    if not _staged(dist, shard_out):
        dist.gather(shard_out, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
        return
    import torch
    host = torch.empty(world * shard_out.numel(), dtype=shard_out.dtype) if rank == 0 else None
    dist.gather(shard_out.cpu(), list(host.chunk(world)) if rank == 0 else None, dst=0)
    if rank == 0:
        gathered.copy_(host)


def _broadcast(dist, t, src=0):
    if not _staged(dist, t):
        dist.broadcast(t, src=src)
        return
    host = t.cpu()
    dist.broadcast(host, src=src)
    t.copy_(host)


def scatter_transform_gather(dist, rank, world, full, shard_in, shard_out, run_shard, device=None):
    """The batch starts on rank 0 (`full`: world equal shards back to back, None elsewhere): scatter,
    run_shard(shard_in, shard_out) on every rank, gather on rank 0.  Returns (gathered tensor on rank 0 |
    None, {"scatter_s", "transform_s", "gather_s"}), every part MAX-reduced over the ranks.  Device tensors go
    through RCCL directly; under gloo they are staged through the host (same code path otherwise)."""
    import torch
    t_s = _timed_collective(lambda: _scatter(dist, rank, world, full, shard_in), dist, device)
    t_c = _timed_collective(lambda: run_shard(shard_in, shard_out), dist, device)
    gathered = torch.empty(world * shard_out.numel(), dtype=shard_out.dtype, device=shard_out.device) \
        if rank == 0 else None
    t_g = _timed_collective(lambda: _gather(dist, rank, world, shard_out, gathered), dist, device)
    return gathered, {"scatter_s": t_s, "transform_s": t_c, "gather_s": t_g}


def end_to_end_leg(dist, rank, world, device, table, d_in, d_out, run_shard, batch):
    """SURVEY.md 8e(ii) for bench.py: table broadcast (set-up, once), then scatter + transform + gather of
    a batch that lives on rank 0.  The shard contents are this rank's synthetic input replicated (the
    timing does not depend on the values).  Returns a JSON-ready dict."""
    import torch
    t_b = _timed_collective(lambda: _broadcast(dist, table, 0), dist, device)
    full = torch.cat([d_in] * world) if rank == 0 else None
    shard = torch.empty_like(d_in)
    _, t = scatter_transform_gather(dist, rank, world, full, shard, d_out, run_shard, device)
    total = t["scatter_s"] + t["transform_s"] + t["gather_s"]
    return {"table_broadcast_ms": t_b * 1e3, "scatter_ms": t["scatter_s"] * 1e3,
            "transform_ms": t["transform_s"] * 1e3, "gather_ms": t["gather_s"] * 1e3,
            "ntt_per_s_including_scatter_gather": world * batch / total,
            "bytes_scattered": int(d_in.numel() * d_in.element_size() * (world - 1)),
            "note": "batch resident on rank 0 before and after; one call, not averaged (set-up excluded)"}
