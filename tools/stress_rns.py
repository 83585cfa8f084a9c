#!/usr/bin/env python3
"""Random stress of the drop-in RNS entry points under the family prediction (host::RnsGuess): random ring 2^12 .. 2^18,
random stacks of 1 .. 6 primes of 58 .. 62 bits, random batch, forward / inverse, in place / out of place -- and only TWO
device buffers of moduli that are rewritten before most calls, so predictions are stale, wrong or right at random.
EVERY polynomial of every call is compared with the oracle.      python tools/stress_rns.py [seed] [seconds]"""
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
        q, _, psi = find_ntt_factors(width, max(logn, 13), skip)
        psi = pow(psi, 1 << (max(logn, 13) - logn), q)
        cases[key] = MergeCase(g, 64, logn, poly, (q, psi * psi % q, psi))
    return cases[key]


bufs = [torch.zeros(3 * 6, dtype=torch.int64, device="cuda") for _ in range(2)]
t0 = time.time(); cnt = 0; kinds = {}
while time.time() - t0 < budget:
    logn = int(rng.choice([12, 13, 13, 14, 14, 15, 16, 16, 17, 18]))
    mc = int(rng.integers(1, 7))
    poly = O.X_N_plus if rng.integers(0, 2) else O.X_N_minus
    style = int(rng.integers(0, 4))  # 0: all <= 60, 1: one 61, 2: one 62, 3: anything
    widths = [int(rng.choice([58, 59, 60])) for _ in range(mc)]
    if style == 1: widths[int(rng.integers(0, mc))] = 61
    if style == 2: widths[int(rng.integers(0, mc))] = 62
    if style == 3: widths = [int(rng.choice([59, 60, 61, 62])) for _ in range(mc)]
    cs, seen = [], {}
    for w in widths:
        seen[w] = seen.get(w, -1) + 1
        cs.append(case(w, logn, poly, seen[w]))
    n = 1 << logn
    maxb = max(1, (1 << 19) >> logn)
    batch = int(rng.integers(1, maxb + 1))
    fwd = np.zeros(mc * n, dtype=np.uint64); inv = np.zeros_like(fwd)
    for i, c in enumerate(cs):
        fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
        inv[i * n:i * n + c.prm.root_of_unity_size] = c.prm.inverse_table_device_order
    buf = bufs[int(rng.integers(0, 2))]
    src = g.modulus_array_to_device([c.prm.modulus for c in cs], 64)
    mods = buf[:src.numel()]
    mods.copy_(src)
    x = np.concatenate([cs[p % mc].P.splitmix(int(rng.integers(1, 1 << 30)) + p, 0, n, cs[p % mc].q) for p in range(batch)])
    inverse = bool(rng.integers(0, 2))
    want = np.concatenate([cs[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cs[p % mc].oprm, inverse=inverse) for p in range(batch)])
    d = g.to_device(x)
    if inverse:
        ninv = g.to_device(np.array([c.prm.n_inv for c in cs], dtype=np.uint64))
        cfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
        fn, tab = g.GPU_INTT, g.to_device(inv)
    else:
        cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
        fn, tab = g.GPU_NTT, g.to_device(fwd)
    if rng.integers(0, 2):
        (g.GPU_INTT_Inplace if inverse else g.GPU_NTT_Inplace)(d, tab, mods, cfg, batch, mc)
        o = d
    else:
        o = torch.zeros_like(d)
        fn(d, o, tab, mods, cfg, batch, mc)
    if rng.integers(0, 3) == 0:
        torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), want), (cnt, logn, widths, batch, inverse)
    kinds[style] = kinds.get(style, 0) + 1
    cnt += 1
print("RNS stress OK: %d random calls (styles %r), every polynomial, in %.0f s" % (cnt, kinds, time.time() - t0))
