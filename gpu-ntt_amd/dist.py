"""Multi-GPU batch sharding: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on
ROCm, "gloo" for CPU-side tests).

The reference has no distributed code at all; its only parallel axis is the
batch of independent polynomials (gridDim.z, reference ntt.cu:2125).  The MI355X build shards
that axis: rank r of G owns polynomials [lo, hi) and runs the ordinary single-GPU entry points
on its shard.  A transform never crosses a GPU, so there is NO data-path collective; the only
collectives are control-plane: a barrier around timed regions, a MAX-reduction of elapsed
times, and (optionally) a digest all-gather to check results.
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
