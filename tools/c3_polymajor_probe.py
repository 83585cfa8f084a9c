#!/usr/bin/env python3
"""VERDICT r3 #5a: C3 (u64 4-step 2^24 x 64, reference layout) polynomial-major -- all three sweeps of ONE polynomial
(128 MiB) or two, back to back on two streams through FourStepPlan::execute(batch = 1 | 2 | 4 ...), so that sweeps 2 and
3 find the hand-off of the sweep before them in the 256 MiB Infinity Cache instead of HBM.  tools/chunk_probe2.py did
this for C2 (two sweeps, memory < 1/2 of the energy): -2.7 %.  Prints ms per 64 transforms, forward and inverse.
    python tools/c3_polymajor_probe.py > gpurun_out/c3_polymajor.txt"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import _load_pkg
g = _load_pkg(); g.load_library()
logn, batch = 24, 64
p4 = g.NTTParameters4Step(logn, 64)
n = p4.n
base = (np.arange(4 * n, dtype=np.uint64) * 0x9E3779B97F4A7C15 % p4.modulus.value)
d_in = g.to_device(base).repeat(batch // 4)
d_out = torch.empty_like(d_in)
side = [torch.cuda.Stream() for _ in range(4)]
for inverse in (False, True):
    tabs = [g.to_device(t) for t in p4.tables["inv" if inverse else "fwd"]]
    cfg = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD,
                                   mod_inverse=p4.n_inv if inverse else 0)
    ref = None
    for streams, per in ((1, 64), (1, 32), (2, 16), (2, 8), (2, 4), (1, 2), (2, 2), (3, 2), (1, 1), (2, 1), (3, 1), (4, 1)):
        plan = g.FourStepPlan(*tabs, p4.modulus, cfg, batch_hint=per)
        chunks = batch // per
        def step():
            for c in range(chunks):
                s = side[c % streams] if per < batch else None
                a = d_in[c * per * n:(c + 1) * per * n]
                b = d_out[c * per * n:(c + 1) * per * n]
                plan.execute(a, b, per, stream=s)
        for _ in range(3): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 10
        for _ in range(iters): step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / iters
        if ref is None:
            ref = ms
            want = d_out.clone()
        same = bool(torch.equal(want, d_out))
        print("%s streams=%d batch-per-call=%2d (%5d MiB)  %.3f ms per 64 transforms  (%+.1f %% vs one call)  identical %s"
              % ("inverse" if inverse else "forward", streams, per, per * n * 8 >> 20, ms, (ms / ref - 1) * 100, same), flush=True)
        plan.close()
    del tabs
