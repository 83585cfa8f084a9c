#!/usr/bin/env python3
"""Random stress of the 4-step entry points against the oracle: random ring 2^12 .. 2^22, word size, batch (up to several
tiles / tile positions per XCD), overload; forward and inverse, EVERY polynomial compared.
    python tools/stress_fourstep.py [seed] [seconds]      (round 3: 1307 shapes over two seeds, all equal)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import _load_pkg
from oracle import oracle as O
import test_gpu_4step as T
g = _load_pkg(); g.load_library()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t0 = time.time(); cnt = 0
while time.time() - t0 < (float(sys.argv[2]) if len(sys.argv) > 2 else 240.0):
    bits = int(rng.choice([32, 64])); logn = int(rng.integers(12, 23))
    maxb = {12: 300, 13: 300, 14: 300, 15: 70, 16: 40, 17: 24, 18: 12, 19: 9, 20: 9, 21: 5, 22: 3}[logn]
    batch = int(rng.integers(1, maxb + 1))
    P = O.Port(bits); p4 = g.NTTParameters4Step(logn, bits); oprm = P.fourstep_params(logn); n = p4.n
    x = P.splitmix(int(rng.integers(1, 1 << 30)), 0, batch * n, p4.modulus.value)
    want = np.concatenate([P.fourstep_ntt(x[p * n:(p + 1) * n], oprm) for p in range(batch)])
    rns = bool(rng.integers(0, 2))
    got = T.run_fourstep(g, p4, x, batch, inverse=False, rns=rns)
    assert np.array_equal(got, want), ("forward", bits, logn, batch, rns)
    xin = np.concatenate([P.fourstep_intt_first_transpose(want[p * n:(p + 1) * n], oprm) for p in range(batch)])
    back = T.run_fourstep(g, p4, xin, batch, inverse=True, rns=bool(rng.integers(0, 2)))
    assert np.array_equal(back, x), ("inverse", bits, logn, batch)
    cnt += 1
print("stress OK: %d random 4-step shapes (both directions, every polynomial) in %.0f s" % (cnt, time.time() - t0))
