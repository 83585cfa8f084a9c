#!/usr/bin/env python3
"""VERDICT r3 #2, measured: drop-in RNS calls (moduli in device memory) whose stack holds a 61- / 62-bit prime, and the
PerCoefficient layout with an RNS stack, on the lazy kernels vs the generic Barrett kernels.
    python tools/bench_rns_wide.py          (one JSON-ish line per case; every output compared between the paths)"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import _load_pkg
from bench_configs import time_ms
from gpu_utils import MergeCase, find_ntt_factors
from oracle import oracle as O
g = _load_pkg(); g.load_library()


def stack(bits, logn, poly, widths):
    fl, seen = [], set()
    for w in widths:
        skip = 0
        while True:
            f = find_ntt_factors(w, logn, skip)
            if f[0] not in seen:
                break
            skip += 1
        seen.add(f[0]); fl.append(f)
    cases = [MergeCase(g, bits, logn, poly, f) for f in fl]
    n = 1 << logn
    fwd = np.zeros(len(cases) * n, dtype=cases[0].P.T)
    for i, c in enumerate(cases):
        fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
    return cases, g.to_device(fwd), g.modulus_array_to_device([c.prm.modulus for c in cases], bits)


# C5 shape: u64, 2^16, batch 512, 8 primes, X^N + 1
logn, batch = 16, 512
for name, widths in (("8 x 60 bit", (60,) * 8), ("7 x 60 + 61 bit", (60,) * 7 + (61,)), ("7 x 60 + 62 bit", (60,) * 7 + (62,)),
                     ("8 x 62 bit", (62,) * 8)):
    cases, tab, mods = stack(64, logn, O.X_N_plus, widths)
    n, mc = 1 << logn, len(cases)
    x = np.concatenate([cases[p % mc].P.splitmix(7 + p, 0, n, cases[p % mc].q) for p in range(batch)])
    d_in = g.to_device(x); d_out = torch.empty_like(d_in)
    cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus)
    res = {}
    for path in ("default", "generic", "fast-strict"):
        g.set_option("path", path)
        fn = lambda: g.GPU_NTT(d_in, d_out, tab, mods, cfg, batch, mc)
        fn(); torch.cuda.synchronize()
        res[path] = (time_ms(fn, 20), g.to_host(d_out).copy())
    g.set_option("path", "default")
    plan = g.NTTPlan(tab, [c.prm.modulus for c in cases], logn, O.X_N_plus, g.FORWARD, batch_hint=batch)
    fn = lambda: plan.execute(d_in, d_out, batch)
    fn(); torch.cuda.synchronize()
    t_plan = time_ms(fn, 20)
    same = all(np.array_equal(res["default"][1], res[p][1]) for p in ("generic", "fast-strict"))
    print("C5-shaped drop-in RNS, %-16s: default %.4f ms, lazy only (fast-strict) %.4f, generic %.4f, NTTPlan %.4f; equal %s"
          % (name, res["default"][0], res["fast-strict"][0], res["generic"][0], t_plan, same), flush=True)
    plan.close()

# PerCoefficient layout, single modulus vs RNS stack (per-lane moduli)
for bits, logn, w, widths in ((64, 9, 1 << 17, (60, 60, 60)), (64, 9, 1 << 17, (60, 61, 62)), (64, 8, 1 << 18, (60, 60, 60)),
                              (32, 9, 1 << 18, (30, 30, 30))):
    cases, tab, mods = stack(bits, logn, O.X_N_plus, widths)
    n, mc = 1 << logn, len(cases)
    x = (np.arange(w * n, dtype=np.uint64) * 0x9E3779B97F4A7C15 % min(c.q for c in cases)).astype(g.np_dtype(bits))
    d_in = g.to_device(x); d_out = torch.empty_like(d_in)
    cfg = g.ntt_rns_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=O.X_N_plus)
    res = {}
    for path in ("default", "generic", "fast-strict"):
        g.set_option("path", path)
        fn = lambda: g.GPU_NTT(d_in, d_out, tab, mods, cfg, w, mc)
        fn(); torch.cuda.synchronize()
        res[path] = (time_ms(fn, 10), g.to_host(d_out).copy())
    g.set_option("path", "default")
    c0 = cases[0]
    scfg = g.ntt_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=O.X_N_plus)
    fn = lambda: g.GPU_NTT(d_in, d_out, c0.fwd_dev, c0.prm.modulus, scfg, w)
    fn(); torch.cuda.synchronize()
    t_single = time_ms(fn, 10)
    same = all(np.array_equal(res["default"][1], res[p][1]) for p in ("generic", "fast-strict"))
    print("PerCoefficient u%d 2^%d x %d, stack %s: default %.4f ms, lazy only %.4f, generic %.4f; single modulus %.4f; equal %s"
          % (bits, logn, w, widths, res["default"][0], res["fast-strict"][0], res["generic"][0], t_single, same), flush=True)
