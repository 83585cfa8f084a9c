#!/usr/bin/env python3
"""Random stress of "GPU_4STEP_NTT computes whatever its three tables say" (round 5): random ring 2^12 .. 2^22, word size,
direction, batch (1 .. many tiles, ragged) and overload, tables that are consistent or corrupted at random (one W word, random
W, one n1 / n2 word, n1 / n2 swapped); the DEFAULT call -- table check on the device, fast kernels as their own fall-back
where a block can be, generic kernels behind the rest -- must return bit for bit what the element-by-element kernels
(path = generic) compute from the same tables.        python tools/stress_fourstep_tables.py [seed] [seconds]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import _load_pkg
g = _load_pkg(); g.load_library()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
MAXB = {12: 600, 13: 300, 14: 300, 15: 40, 16: 20, 17: 9, 18: 5, 19: 3, 20: 3, 21: 2, 22: 2}
params = {}


def call(p4, tabs, d_in, batch, inverse, rns):
    d_out = torch.full_like(d_in, -7)
    kind = g.INVERSE if inverse else g.FORWARD
    if rns:
        mods = g.modulus_array_to_device([p4.modulus], p4.bits)
        ninv = g.to_device(np.array([p4.n_inv], dtype=g.np_dtype(p4.bits)))
        g.GPU_4STEP_NTT(d_in, d_out, *tabs, mods, g.ntt4step_rns_configuration(n_power=p4.logn, ntt_type=kind, mod_inverse=ninv), batch, 1)
    else:
        g.GPU_4STEP_NTT(d_in, d_out, *tabs, p4.modulus,
                        g.ntt4step_configuration(n_power=p4.logn, ntt_type=kind, mod_inverse=p4.n_inv if inverse else 0), batch)
    torch.cuda.synchronize()
    return g.to_host(d_out)


t0 = time.time(); cnt = 0; kinds = {}
while time.time() - t0 < budget:
    bits = int(rng.choice([32, 64])); logn = int(rng.integers(12, 23))
    if (bits, logn) not in params:
        params[(bits, logn)] = g.NTTParameters4Step(logn, bits)
    p4 = params[(bits, logn)]
    q, n, n1, n2 = p4.modulus.value, p4.n, p4.n1, p4.n2
    dt = g.np_dtype(bits)
    batch = int(rng.choice([1, 2, 3, MAXB[logn]])) if rng.integers(0, 2) else int(rng.integers(1, MAXB[logn] + 1))
    inverse = bool(rng.integers(0, 2)); rns = bool(rng.integers(0, 4) == 0)
    t1, t2, w = (t.copy() for t in p4.tables["inv" if inverse else "fwd"])
    kind = int(rng.integers(0, 6))
    if kind == 1:
        pos = int(rng.integers(0, n)); w[pos] = (int(w[pos]) + 1) % q
    elif kind == 2:
        w = rng.integers(1, q, size=n, dtype=np.uint64).astype(dt)
    elif kind == 3:
        t2[int(rng.integers(1, t2.size))] ^= dt(1)
    elif kind == 4:
        t1[int(rng.integers(1, t1.size))] ^= dt(1)
    elif kind == 5:
        big = max(t1.size, t2.size)
        pad = lambda t: np.concatenate([t, np.ones(big - t.size, dtype=dt)])
        t1, t2 = pad(t2), pad(t1)
    dev = [g.to_device(np.ascontiguousarray(t)) for t in (t1, t2, w)]
    d_in = g.to_device(rng.integers(0, q, size=batch * n, dtype=np.uint64).astype(dt))
    g.set_option("path", "generic")
    ref = call(p4, dev, d_in, batch, inverse, rns)
    g.set_option("path", "default")
    for rep in range(2):
        got = call(p4, dev, d_in, batch, inverse, rns)
        assert np.array_equal(got, ref), (cnt, bits, logn, batch, inverse, rns, kind, rep)
    kinds[kind] = kinds.get(kind, 0) + 1
    cnt += 1
print("4-step table stress OK: %d random calls x 2 (kinds consistent / one W word / random W / n2 word / n1 word / swapped: %s), rings 2^12 .. 2^22, in %.0f s"
      % (cnt, dict(sorted(kinds.items())), time.time() - t0))
