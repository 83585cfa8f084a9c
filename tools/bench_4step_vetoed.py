#!/usr/bin/env python3
"""What a 4-step call with tables that are NOT those of one root costs (the reference's own timing program passes random
words, benchmark/bench_4step_ntt.cu:80-90): default path (table check vetoes the fast kernels; they run the element-by-element
algorithm themselves where a block can, the generic kernels do the rest) against path = generic (the element-by-element
kernels alone) and against the same call with consistent tables.  u64, forward and inverse; ms per call.
    python tools/bench_4step_vetoed.py [logN:batch ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
g = load_pkg(); g.load_library()


def timed(fn, iters=30, warm=5):
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


SHAPES = ((12, 16384), (13, 8192), (14, 4096), (16, 1024), (17, 512), (20, 64))
if len(sys.argv) > 1:
    SHAPES = tuple(tuple(int(v) for v in a.split(":")) for a in sys.argv[1:])
print("# u64; ms per call; random W (every call vetoed)")
print("# logN batch dir   consistent   vetoed(default)   path=generic   vetoed/generic")
for logn, batch in SHAPES:
    p4 = g.NTTParameters4Step(logn, 64)
    rng = np.random.default_rng(logn)
    for inverse in (False, True):
        good = [g.to_device(t) for t in p4.tables["inv" if inverse else "fwd"]]
        bad = [good[0], good[1], g.to_device(rng.integers(1, p4.modulus.value, size=p4.n, dtype=np.uint64))]
        cfg = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD,
                                       mod_inverse=p4.n_inv if inverse else 0)
        d_in = g.to_device(rng.integers(0, p4.modulus.value, size=batch * p4.n, dtype=np.uint64))
        d_out = torch.zeros_like(d_in)
        t_good = timed(lambda: g.GPU_4STEP_NTT(d_in, d_out, *good, p4.modulus, cfg, batch))
        t_bad = timed(lambda: g.GPU_4STEP_NTT(d_in, d_out, *bad, p4.modulus, cfg, batch))
        g.set_option("path", "generic")
        t_gen = timed(lambda: g.GPU_4STEP_NTT(d_in, d_out, *bad, p4.modulus, cfg, batch))
        g.set_option("path", "default")
        print("%2d %6d %s   %9.4f   %9.4f   %9.4f   %5.2f" % (logn, batch, "inv" if inverse else "fwd", t_good, t_bad, t_gen, t_bad / t_gen), flush=True)
