#!/usr/bin/env python3
"""A/B of option two_sweep_big (round 5 experiment): u64 Merge forward 2^23 / 2^24 in two sweeps on 16384-coefficient tiles
(one strided pass of 9 / 10 stages + the 14-stage contiguous pass) against the three-sweep plan; drop-in and NTTPlan,
alternating in one session; polynomial 0 and the last one checked against the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
from gpu_utils import MergeCase  # noqa: E402
from oracle import oracle as O  # noqa: E402

g = load_pkg()
g.load_library()


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for logn, batch in ((23, 8), (24, 4), (24, 64)):
    c = MergeCase(g, 64, logn, O.X_N_minus)
    n = c.n
    x0 = c.random(1, 7)
    x1 = c.random(1, 8)
    want0, want1 = c.P.merge_ntt(x0, c.oprm), c.P.merge_ntt(x1, c.oprm)
    d = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    for opt in ("0", "1", "0", "1"):
        g.set_option("two_sweep_big", opt)
        d[:n] = g.to_device(x0)
        d[-n:] = g.to_device(x1)
        if batch > 2:
            d[n:-n] = g.to_device(x0).repeat(batch - 2)
        g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus, c.cfg(), batch)
        torch.cuda.synchronize()
        ok = np.array_equal(g.to_host(d[:n]), want0) and np.array_equal(g.to_host(d[-n:]), want1)
        t = timed(lambda: g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus, c.cfg(), batch), 20 if batch < 64 else 5)
        plan = g.NTTPlan(c.fwd_dev, c.prm.modulus, logn, O.X_N_minus, g.FORWARD, batch_hint=batch)
        tp = timed(lambda: plan.execute(d, d, batch), 20 if batch < 64 else 5)
        plan.close()
        print("2^%d x %d two_sweep_big=%s: drop-in %.4f ms, plan %.4f ms, exact %s" % (logn, batch, opt, t, tp, ok), flush=True)
g.set_option("two_sweep_big", "0")
