#!/usr/bin/env python3
"""bench.py -- headline benchmark: forward Merge NTT, 64-bit, N = 2^16, batch = 1024 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one gpuntt GPU_NTT call (all of its kernel launches) over one batch of synthetic
random polynomials already resident in HBM.  Rank r owns its own batch (weak scaling, no
data-path collective: polynomials are independent); the only collectives are
the timing barrier and the max-reduction of the elapsed time.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline arithmetic).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402

LOGN = 16
BATCH = 1024
BITS = 64
CPU_PASSES = 6  # ~10 s of single-core CPU work at ~580 NTT/s
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "forward-NTTs/sec + achieved HBM GB/s, 64-bit Merge N=2^16 batch=1024"


def splitmix64_mod(seed, count, q):
    """x[k] = splitmix64(seed ^ k) mod q, k = 0..count-1 (the portable synthetic input of
    SURVEY.md 8d), vectorised with numpy's wrapping uint64 arithmetic."""
    k = np.arange(count, dtype=np.uint64)
    x = (np.uint64(seed) ^ k) + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x = x ^ (x >> np.uint64(31))
    return x % np.uint64(q)


def cpu_baseline(modulus_value, x_sample, y_gpu_sample, logn):
    """The CPU leg (the only place bench.py touches oracle/): times the CPU path on a bounded
    sample of the same workload on this host -- oracle/_ref (the reference's own NTTCPU::ntt,
    kind 'reference') when the prebuilt file is present, else this repo's C restatement (kind
    'port') -- and uses its output to check the GPU result of the same polynomials bit for bit."""
    from oracle import oracle as O
    n = 1 << logn
    polys = x_sample.size // n
    if O.have_ref():
        R = O.Ref(BITS)
        prm = R.merge_params(logn, O.X_N_minus)
        assert prm["mod"][0] == modulus_value
        R.merge_ntt(x_sample[:n], prm)  # warm
        t0 = time.perf_counter()
        for _ in range(CPU_PASSES):
            y = R.merge_ntt(x_sample, prm)
        dt = time.perf_counter() - t0
        kind = "reference"
    else:
        P = O.Port(BITS)
        prm = P.merge_params(logn, O.X_N_minus)
        P.merge_ntt(x_sample[:n], prm)
        t0 = time.perf_counter()
        for _ in range(CPU_PASSES):
            y = P.merge_ntt(x_sample, prm)
        dt = time.perf_counter() - t0
        kind = "port"
    if not np.array_equal(y, y_gpu_sample):
        raise SystemExit("bench: GPU result differs from the CPU reference path")
    return {"value": CPU_PASSES * polys / dt, "unit": "NTT/s", "cores": 1, "kind": kind,
            "gpu_output_bit_exact": True,
            "sample": "%d passes over %d of the %d polynomials of one batch (u64, N=2^%d), "
                      "NTTCPU::ntt on one host core, %.1f s of CPU work"
                      % (CPU_PASSES, polys, BATCH, logn, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-polys", type=int, default=BATCH)
    args = ap.parse_args()

    import importlib
    import torch
    g = _load_pkg()
    g.load_library()  # raises if the HIP extension is missing: there is no fallback
    dist_mod = importlib.import_module("gpu_ntt_amd.dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist, rank, world = dist_mod.init_process_group("nccl", dev)

    prm = g.NTTParameters(LOGN, g.X_N_minus, BITS)
    n = 1 << LOGN
    # synthetic input x[p][i] = splitmix64(seed ^ (p*N+i)) mod q, seed per rank
    x = splitmix64_mod(0x5EED0002 + rank, BATCH * n, prm.modulus.value)
    d_in = g.to_device(x, dev)
    d_out = torch.empty_like(d_in)
    table = g.to_device(prm.forward_table_device_order, dev)
    cfg = g.ntt_configuration(n_power=LOGN, ntt_type=g.FORWARD, reduction_poly=g.X_N_minus)

    def step():
        g.GPU_NTT(d_in, d_out, table, prm.modulus, cfg, BATCH)

    step()
    torch.cuda.synchronize()
    y_first = g.to_host(d_out)[:args.cpu_polys * n].copy()  # checked in the cpu_baseline leg

    # clock settle (untimed, before the W warm-up steps): the part needs ~60 ms of load to leave its
    # idle clocks; without this a short --steps/--warmup run reads 10-15 % slow
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.15:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)

    def timed_steps():
        # HIP events on the stream the kernels are launched on bracket exactly the K steps
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()

    # barrier + synchronize on both sides, MAX over ranks (gpu-ntt_amd/dist.py)
    wall = dist_mod.timed_region(timed_steps, 1, dist, dev)
    dev_ms = e0.elapsed_time(e1)
    if dist is not None:
        t = torch.tensor([dev_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dev_ms = float(t[0])

    if rank == 0:
        # HBM bytes per call from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # passes of this same command, summarised by tools/pmc_summary.py with the gfx950 x2 fetch
        # correction); a profile cannot be taken from inside the timed run, so the committed
        # summary of the current round is quoted
        traffic = None
        try:
            pm = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc_traffic.json"))
            if pm:
                traffic = json.load(open(os.path.join(ROOT, "profiles", pm[-1])))["bytes_per_call"]
        except Exception:
            traffic = None
        ms_per_step = wall * 1e3 / args.steps
        call_ms = dev_ms / args.steps
        alg_bytes = 2 * n * (BITS // 8) * BATCH  # every coefficient read once + written once
        achieved = alg_bytes / (call_ms * 1e-3) / 1e9
        line = {
            "metric": METRIC,
            "value": world * BATCH * args.steps / wall,
            "unit": "NTT/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "Merge-NTT Data64 log2N=16 batch=1024 forward (BASELINE configs[1])",
                       "log2N": LOGN, "batch_per_gpu": BATCH, "reduction_poly": "X_N_minus",
                       "modulus": prm.modulus.value, "out_of_place": True,
                       "parallelism": "batch-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_call": alg_bytes,
                         "call_ms_hip_events": call_ms,
                         "note": "one call = every launch of one GPU_NTT (twiddle prep + 6-stage strided "
                                 "pass + 10-stage contiguous pass = 2 HBM sweeps, traffic = PMC bytes per "
                                 "call); the contiguous pass dominates and is VALU-issue bound; per-kernel "
                                 "averages in profiles/"},
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(prm.modulus.value, x[:args.cpu_polys * n], y_first, LOGN)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
