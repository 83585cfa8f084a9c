#!/usr/bin/env python3
"""C2 (u64 2^16 x 1024, in place) as `chunks` batch slices issued round-robin on `streams` HIP streams (one NTTPlan,
plan.execute per slice): does keeping a slice's hand-off between the two passes inside the 256 MiB Infinity Cache pay
once the launch tails of one stream are covered by the other stream's kernels?  (tools/chunk_probe.py: one stream)"""
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
d = g.to_device(x)
tab = g.to_device(prm.forward_table_device_order)
side = [torch.cuda.Stream() for _ in range(4)]
for streams in (1, 2, 3, 4):
    for chunks in (1, 2, 4, 8, 16, 32):
        if chunks < streams:
            continue
        per = batch // chunks
        plan = g.NTTPlan(tab, prm.modulus, logn, g.X_N_minus, g.FORWARD, batch_hint=per)
        torch.cuda.synchronize()
        def step():
            for c in range(chunks):
                s = side[c % streams]
                sl = d[c * per * n:(c + 1) * per * n]
                plan.execute(sl, sl, per, stream=s)
        for _ in range(20): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 300
        for _ in range(iters): step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / iters
        print("streams=%d chunks=%2d (%4d MiB each)  %.4f ms per 1024 transforms" % (streams, chunks, per * n * 8 >> 20, ms), flush=True)
        plan.close()
