#!/usr/bin/env python3
"""What a GPU-NTT user in FHE key-switching issues: many SMALL RNS calls.  Drop-in GPU_NTT / GPU_INTT
(RNS overload, moduli in device memory) against NTTPlan::execute on the same buffers: N = 2^12 .. 2^16, mod_count in
{4, 8, 16, 32}, batch in {mod_count, 2 * mod_count}, forward and inverse; one stream (us per call, HIP events over
back-to-back calls) and four streams driven by four host threads (aggregate calls per second).

    python tools/bench_small_dropin.py > profiles/r05_small_dropin.txt
"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
from gpu_utils import find_ntt_factors  # noqa: E402

g = load_pkg()
g.load_library()
QUICK = "--quick" in sys.argv
MAXMC = 32
BASE = [find_ntt_factors(60, 16, skip=i) for i in range(MAXMC)]  # 60-bit primes with 2^17 | q - 1


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


def setup(logn, mc, batch, inverse, stream=None):
    n = 1 << logn
    prms = []
    for q, _, psi16 in BASE[:mc]:
        psi = pow(psi16, 1 << (16 - logn), q)
        prms.append(g.NTTParameters(logn, g.X_N_plus, 64, (q, psi * psi % q, psi)))
    tab = np.zeros(mc * n, dtype=np.uint64)
    for i, p in enumerate(prms):
        tab[i * n:(i + 1) * n] = p.inverse_table_device_order if inverse else p.forward_table_device_order
    table = g.to_device(tab)
    mods = g.modulus_array_to_device([p.modulus for p in prms], 64)
    ninv = g.to_device(np.array([p.n_inv for p in prms], dtype=np.uint64))
    rng = np.random.default_rng(logn * 100 + mc)
    d = g.to_device(rng.integers(0, min(p.modulus.value for p in prms), size=batch * n, dtype=np.uint64))
    cfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD, reduction_poly=g.X_N_plus,
                                  mod_inverse=ninv if inverse else None, stream=stream)
    plan = g.NTTPlan(table, [p.modulus for p in prms], logn, g.X_N_plus, g.INVERSE if inverse else g.FORWARD,
                     mod_inverse=[p.n_inv for p in prms] if inverse else None, batch_hint=batch)
    fn = g.GPU_INTT_Inplace if inverse else g.GPU_NTT_Inplace
    dropin = lambda: fn(d, table, mods, cfg, batch, mc)  # noqa: E731
    planned = lambda: plan.execute(d, d, batch, stream=stream)  # noqa: E731
    return dropin, planned, (table, mods, ninv, d, plan)


_w = torch.zeros(1 << 24, device="cuda")
_t0 = time.perf_counter()
while time.perf_counter() - _t0 < 0.5:  # leave the idle clocks before the first cell is timed
    _w.add_(1.0)
torch.cuda.synchronize()
print("# u64, X^N+1, 60-bit primes, in place; us per call on ONE stream (HIP events over back-to-back calls)")
print("# logN mc batch dir   dropin_us  plan_us  dropin/plan")
worst = 0.0
cells = []
for logn in ((12, 14, 16) if QUICK else (12, 13, 14, 15, 16)):
    for mc in ((4, 16) if QUICK else (4, 8, 16, 32)):
        for batch in (mc, 2 * mc):
            for inverse in (False, True):
                dropin, planned, keep = setup(logn, mc, batch, inverse)
                t_d, t_p = timed(dropin), timed(planned)
                worst = max(worst, t_d / t_p)
                cells.append((logn, mc, batch, inverse, t_d, t_p))
                print("%2d %2d %3d %s  %8.1f %8.1f  %5.2f" % (logn, mc, batch, "inv" if inverse else "fwd", t_d, t_p, t_d / t_p), flush=True)
                keep[4].close()
print("# worst drop-in / plan ratio over the cells: %.2f" % worst)

print("# four streams, four host threads, each its own stack and buffers: aggregate calls per second (wall clock)")
print("# logN mc batch dir   dropin_calls_per_s  plan_calls_per_s  ratio")
for logn, mc in ((13, 8), (16, 8)) if QUICK else ((12, 4), (13, 8), (14, 16), (16, 8), (16, 32)):
    for inverse in (False, True):
        batch = mc
        res = {}
        for which in ("dropin", "plan"):
            streams = [torch.cuda.Stream() for _ in range(4)]
            jobs = [setup(logn, mc, batch, inverse, s) for s in streams]
            iters = 300

            def work(i):
                fn = jobs[i][0] if which == "dropin" else jobs[i][1]
                with torch.cuda.stream(streams[i]):
                    for _ in range(iters):
                        fn()
            for i in range(4):  # warm every stream (scratch, prediction)
                with torch.cuda.stream(streams[i]):
                    for _ in range(10):
                        (jobs[i][0] if which == "dropin" else jobs[i][1])()
            torch.cuda.synchronize()
            th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            res[which] = 4 * iters / (time.perf_counter() - t0)
            for j in jobs:
                j[2][4].close()
        print("%2d %2d %3d %s  %10.0f %10.0f  %5.2f" % (logn, mc, batch, "inv" if inverse else "fwd", res["dropin"], res["plan"],
                                                      res["plan"] / res["dropin"]), flush=True)
