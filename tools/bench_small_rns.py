#!/usr/bin/env python3
"""RNS stacks of rings below one tile (VERDICT r4 missing #3): drop-in GPU_NTT / GPU_INTT with mod_count primes against the
single-modulus call of the same shape, and against the Barrett kernels (path = generic) that served these calls before
round 5.  N = 2^8 .. 2^11, batch * N = 2^24 coefficients, u64 and u32.

    python tools/bench_small_rns.py > profiles/r05_small_rns.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
from gpu_utils import MergeCase, find_ntt_factors  # noqa: E402
from oracle import oracle as O  # noqa: E402

g = load_pkg()
g.load_library()


def timed(fn, iters=50, warm=5):
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


print("# us per call, batch * N = 2^24 coefficients, in place; rns = drop-in RNS call with mod_count distinct primes")
print("# bits logN mc dir  single  rns  rns/single  rns_generic")
for bits in (64, 32):
    for logn in (8, 9, 10, 11):
        n = 1 << logn
        batch = (1 << 24) >> logn
        for mc in (2, 4, 8):
            seen = {}
            fl = [find_ntt_factors(60 if bits == 64 else 30, logn, skip=i) for i in range(mc)]
            cases = [MergeCase(g, bits, logn, O.X_N_plus, f) for f in fl]
            fwd = np.zeros(mc * n, dtype=cases[0].P.T)
            inv = np.zeros_like(fwd)
            for i, c in enumerate(cases):
                fwd[i * n:(i + 1) * n] = c.prm.forward_table_device_order
                inv[i * n:(i + 1) * n] = c.prm.inverse_table_device_order
            d_fwd, d_inv = g.to_device(fwd), g.to_device(inv)
            mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
            ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=cases[0].P.T))
            d = g.to_device(cases[0].P.splitmix(1, 0, batch * n, min(c.q for c in cases)))
            c0 = cases[0]
            cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus)
            icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=O.X_N_plus, mod_inverse=ninv)
            for direction in ("fwd", "inv"):
                if direction == "fwd":
                    single = lambda: g.GPU_NTT_Inplace(d, c0.fwd_dev, c0.prm.modulus, c0.cfg(), batch)
                    rns = lambda: g.GPU_NTT_Inplace(d, d_fwd, mods, cfg, batch, mc)
                else:
                    single = lambda: g.GPU_INTT_Inplace(d, c0.inv_dev, c0.prm.modulus, c0.cfg(True), batch)
                    rns = lambda: g.GPU_INTT_Inplace(d, d_inv, mods, icfg, batch, mc)
                t_single = timed(single)
                t_rns = timed(rns)
                g.set_option("path", "generic")
                t_gen = timed(rns, iters=10, warm=2)
                g.set_option("path", "default")
                print("%d %2d %d %s  %8.1f %8.1f  %5.2f  %8.1f" % (bits, logn, mc, direction, t_single, t_rns, t_rns / t_single, t_gen), flush=True)
