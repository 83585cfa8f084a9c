"""PerCoefficient layout: the Barrett (generic) kernels against the fast strided kernels on the same call, bit-identical
outputs required:   python tools/bench_percoefficient.py"""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from __graft_entry__ import _load_pkg
from bench_configs import time_ms
g = _load_pkg(); g.load_library()
for bits, logn, batch in ((64, 9, 1 << 17), (64, 8, 1 << 18), (32, 9, 1 << 18), (64, 5, 1 << 21)):
    prm = g.NTTParameters(logn, g.X_N_plus, bits)
    n = 1 << logn
    x = (np.arange(batch * n, dtype=np.uint64) * 0x9E3779B97F4A7C15 % prm.modulus.value).astype(g.np_dtype(bits))
    d_in = g.to_device(x); d_out = torch.empty_like(d_in)
    tab = g.to_device(prm.forward_table_device_order)
    cfg = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, ntt_layout=g.PerCoefficient, reduction_poly=g.X_N_plus)
    res = {}
    for path in ("generic", "fast-strict"):
        g.set_option("path", path)
        fn = lambda: g.GPU_NTT(d_in, d_out, tab, prm.modulus, cfg, batch)
        fn(); torch.cuda.synchronize()
        res[path] = (time_ms(fn, 10), g.to_host(d_out).copy())
    print("PerCoefficient u%d 2^%d x %d: generic %.4f ms, fast %.4f ms, equal %s" % (bits, logn, batch, res["generic"][0], res["fast-strict"][0], np.array_equal(res["generic"][1], res["fast-strict"][1])), flush=True)
