#!/usr/bin/env python3
"""C2 (u64 2^16 x 1024) executed as `chunks` back-to-back plan.execute() calls on batch slices: does keeping the
hand-off between the two passes inside the 256 MiB Infinity Cache (chunk <= ~64 MiB) pay?  in place / out of place."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import _load_pkg
g = _load_pkg(); g.load_library()
logn, batch = 16, 1024
n = 1 << logn
prm = g.NTTParameters(logn, g.X_N_minus, 64)
x = (np.arange(batch * n, dtype=np.uint64) * 0x9E3779B97F4A7C15 % prm.modulus.value)
d_in = g.to_device(x); d_out = torch.empty_like(d_in)
tab = g.to_device(prm.forward_table_device_order)
for inplace in (False, True):
    for chunks in (1, 2, 4, 8, 16, 32):
        per = batch // chunks
        plan = g.NTTPlan(tab, prm.modulus, logn, g.X_N_minus, g.FORWARD, batch_hint=per)
        src = d_out if inplace else d_in
        def step():
            for c in range(chunks):
                plan.execute(src[c * per * n:(c + 1) * per * n], d_out[c * per * n:(c + 1) * per * n], per)
        for _ in range(30): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): step()
        e1.record(); torch.cuda.synchronize()
        print("inplace=%d chunks=%2d (%4d MiB each)  %.4f ms per 1024 transforms" % (inplace, chunks, per * n * 8 >> 20, e0.elapsed_time(e1) / 300), flush=True)
