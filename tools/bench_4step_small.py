#!/usr/bin/env python3
"""Cost of the default-on table check of the 4-step entry points on SMALL calls: GPU_4STEP_NTT drop-in with the check
(preparation kernel verifies the tables; generic kernels enqueued behind the fast ones) against option
check_4step_tables = 0 (round-4 behaviour) and FourStepPlan::execute; us per call, forward and inverse.

    python tools/bench_4step_small.py > profiles/r05_4step_small_calls.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

g = load_pkg()
if os.environ.get("GPUNTT_LIB"):
    g.LIB_PATH = os.environ["GPUNTT_LIB"]
g.load_library()


def timed(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print("# u64; us per call; lib = %s" % os.path.basename(g.LIB_PATH))
print("# logN batch dir  checked  unchecked  plan   checked/unchecked")
SHAPES = ((12, 1), (12, 64), (13, 8), (14, 8), (14, 512), (16, 1), (16, 16), (18, 4), (20, 1), (20, 16), (22, 4), (24, 1), (24, 4))
RNS = "--rns" in sys.argv  # the RNS overload with ONE device-side modulus (how the reference's examples call the entry point)
ARGS = [a for a in sys.argv[1:] if a != "--rns"]
if ARGS:  # e.g. 24:1 20:16
    SHAPES = tuple(tuple(int(v) for v in a.split(":")) for a in ARGS)
if RNS:
    print("# RNS overload, one device-side modulus")
for logn, batch in SHAPES:
    p4 = g.NTTParameters4Step(logn, 64)
    for inverse in (False, True):
        tabs = [g.to_device(t) for t in p4.tables["inv" if inverse else "fwd"]]
        cfg = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD,
                                       mod_inverse=p4.n_inv if inverse else 0)
        d_in = g.to_device(np.random.default_rng(logn).integers(0, p4.modulus.value, size=batch * p4.n, dtype=np.uint64))
        d_out = torch.zeros_like(d_in)
        call = lambda: g.GPU_4STEP_NTT(d_in, d_out, *tabs, p4.modulus, cfg, batch)  # noqa: E731
        if RNS:
            mods = g.modulus_array_to_device([p4.modulus], 64)
            ninv = g.to_device(np.array([p4.n_inv], dtype=np.uint64))
            rcfg = g.ntt4step_rns_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD, mod_inverse=ninv)
            call = lambda: g.GPU_4STEP_NTT(d_in, d_out, *tabs, mods, rcfg, batch, 1)  # noqa: E731
        g.set_option("check_4step_tables", "1")
        t_c = timed(call, iters=100 if logn >= 22 else 200)
        g.set_option("check_4step_tables", "0")
        t_u = timed(call, iters=100 if logn >= 22 else 200)
        g.set_option("check_4step_tables", "1")
        plan = g.FourStepPlan(*tabs, p4.modulus, cfg, batch_hint=batch)
        t_p = timed(lambda: plan.execute(d_in, d_out, batch), iters=100 if logn >= 22 else 200)
        plan.close()
        print("%2d %3d %s  %9.1f %9.1f %9.1f   %5.2f" % (logn, batch, "inv" if inverse else "fwd", t_c, t_u, t_p, t_c / t_u), flush=True)
