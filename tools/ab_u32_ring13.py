"""A/B of the tile of the 32-bit ring 2^13 (8192 coefficients of its own vs half of a 16384-coefficient tile), forward and
inverse, batch 16 .. 16384:   python tools/ab_u32_ring13.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np, torch
from __graft_entry__ import _load_pkg
from bench_configs import time_ms
g = _load_pkg(); g.load_library()
logn = 13
prm = g.NTTParameters(logn, g.X_N_minus, 32)
tab = g.to_device(prm.forward_table_device_order)
itab = g.to_device(prm.inverse_table_device_order)
cfg = g.ntt_configuration(n_power=logn, reduction_poly=g.X_N_minus)
icfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=g.X_N_minus, mod_inverse=prm.n_inv)
for batch in (16, 64, 256, 1024, 8192, 16384):
    x = (np.arange(batch << logn, dtype=np.uint64) * 2654435761 % prm.modulus.value).astype(np.uint32)
    d = g.to_device(x)
    row = []
    for opt in ("0", "1000000"):
        g.set_option("u32_ring13_batch", opt)
        f = time_ms(lambda: g.GPU_NTT_Inplace(d, tab, prm.modulus, cfg, batch), 50)
        i = time_ms(lambda: g.GPU_INTT_Inplace(d, itab, prm.modulus, icfg, batch), 50)
        row.append((f, i))
    print("u32 2^13 batch %5d: tile 16384 fwd %.4f inv %.4f ms | tile 8192 fwd %.4f inv %.4f ms" % (batch, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
