#!/usr/bin/env python3
"""Random stress of the 32-bit entry points after the round-5 butterfly rewrite (multiply-add chains, one-quotient
normalisation): random ring 2^4 .. 2^20, random primes of 12 .. 30 bits (both lazy families: q < 2^29 and 2^29 <= q < 2^30, and
moduli far below the word size, where floor(2^32 / q) is large), X^N+1 / X^N-1, forward / inverse, single modulus (drop-in and
NTTPlan) and RNS stacks of 2 .. 4 primes, in place / out of place.  EVERY polynomial of every call is compared with the oracle.
    python tools/stress_u32.py [seed] [seconds]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import _load_pkg
from gpu_utils import MergeCase, find_ntt_factors
from oracle import oracle as O
g = _load_pkg(); g.load_library()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
cases = {}


def case(width, logn, poly, skip):
    key = (width, logn, poly, skip)
    if key not in cases:
        q, _, psi = find_ntt_factors(width, logn, skip)
        cases[key] = MergeCase(g, 32, logn, poly, (q, psi * psi % q, psi))
    return cases[key]


t0 = time.time(); cnt = 0; kinds = {}
while time.time() - t0 < budget:
    logn = int(rng.integers(4, 21))
    mc = int(rng.choice([1, 1, 1, 2, 3, 4]))
    poly = O.X_N_plus if rng.integers(0, 2) else O.X_N_minus
    lo = max(12, logn + 3)
    widths = [int(rng.integers(lo, 31)) for _ in range(mc)]
    cs, seen = [], {}
    try:
        for w in widths:
            seen[w] = seen.get(w, -1) + 1
            cs.append(case(w, logn, poly, seen[w]))
    except ValueError:
        continue  # no further prime of that width for this ring
    n = 1 << logn
    maxb = max(1, (1 << 20) >> logn)
    batch = int(rng.integers(1, min(maxb, 4096) + 1))
    x = np.concatenate([cs[p % mc].P.splitmix(int(rng.integers(1, 1 << 30)) + p, 0, n, cs[p % mc].q) for p in range(batch)])
    inverse = bool(rng.integers(0, 2))
    want = np.concatenate([cs[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cs[p % mc].oprm, inverse=inverse) for p in range(batch)])
    d = g.to_device(x)
    inplace = bool(rng.integers(0, 2))
    if mc == 1:
        c = cs[0]
        tab = c.inv_dev if inverse else c.fwd_dev
        style = int(rng.integers(0, 2))
        if style == 1:
            plan = g.NTTPlan(tab, c.prm.modulus, logn, reduction_poly=poly, ntt_type=g.INVERSE if inverse else g.FORWARD,
                             mod_inverse=c.prm.n_inv if inverse else None, batch_hint=batch)
            o = d if inplace else torch.zeros_like(d)
            plan.execute(d, o, batch)
            torch.cuda.synchronize()
            plan.close()
        elif inplace:
            (g.GPU_INTT_Inplace if inverse else g.GPU_NTT_Inplace)(d, tab, c.prm.modulus, c.cfg(inverse), batch)
            o = d
        else:
            o = torch.zeros_like(d)
            (g.GPU_INTT if inverse else g.GPU_NTT)(d, o, tab, c.prm.modulus, c.cfg(inverse), batch)
    else:
        style = 2
        fwd = np.zeros(mc * n, dtype=np.uint32); inv = np.zeros_like(fwd)
        for i, c in enumerate(cs):
            fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
            inv[i * n:i * n + c.prm.root_of_unity_size] = c.prm.inverse_table_device_order
        mods = g.modulus_array_to_device([c.prm.modulus for c in cs], 32)
        if inverse:
            ninv = g.to_device(np.array([c.prm.n_inv for c in cs], dtype=np.uint32))
            cfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
            fn, tab = g.GPU_INTT, g.to_device(inv)
        else:
            cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
            fn, tab = g.GPU_NTT, g.to_device(fwd)
        if inplace:
            (g.GPU_INTT_Inplace if inverse else g.GPU_NTT_Inplace)(d, tab, mods, cfg, batch, mc)
            o = d
        else:
            o = torch.zeros_like(d)
            fn(d, o, tab, mods, cfg, batch, mc)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), want), (cnt, logn, widths, batch, inverse, style, inplace)
    kinds[style] = kinds.get(style, 0) + 1
    cnt += 1
print("u32 stress OK: %d random calls (single drop-in / plan / RNS: %s), widths 12 .. 30 bits, rings 2^4 .. 2^20, every polynomial, in %.0f s"
      % (cnt, kinds, time.time() - t0))
