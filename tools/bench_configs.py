#!/usr/bin/env python3
"""Secondary benchmark lines (not the driver's bench.py contract): every BASELINE.json config
and the log2N sweep, one JSON object per line, for DESIGN.md / profiles/.

    python tools/bench_configs.py [--what configs|sweep|sweep32|sweep64|4step32|all] [--iters 20] > profiles/configs_rNN.jsonl

Each line: {"name", "dtype", "algo", "log2N", "batch", "ms", "ntt_per_s", "alg_GBps", "frac_of_8TBps",
            "checked"} -- `checked` = a sampled polynomial matched the oracle bit-for-bit.
Algorithmic bytes = 2 * N * sizeof(T) per transform."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402
from oracle import oracle as O  # noqa: E402

PEAK = 8000.0


def time_ms(fn, iters, warm=3):
    """Average ms per call.  Warm-up runs for at least `warm` calls AND ~60 ms so the clocks have
    settled (a 10 ms measurement straight after an idle period reads 15 % slow); the timed loop
    runs at least `iters` calls and ~120 ms."""
    import time
    import torch
    t0 = time.perf_counter()
    k = 0
    while k < warm or time.perf_counter() - t0 < 0.06:
        fn()
        k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    per = max((time.perf_counter() - t0) / max(k, 1), 1e-6)
    iters = max(iters, min(2000, int(0.12 / per)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def emit(name, bits, algo, logn, batch, ms, checked, extra=None):
    n = 1 << logn
    alg = 2 * n * (bits // 8) * batch
    gbps = alg / (ms * 1e-3) / 1e9
    d = {"name": name, "dtype": "u%d" % bits, "algo": algo, "log2N": logn, "batch": batch,
         "ms": round(ms, 5), "ntt_per_s": round(batch / (ms * 1e-3), 1), "alg_GBps": round(gbps, 1),
         "frac_of_8TBps": round(gbps / PEAK, 4), "checked": bool(checked)}
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)


def merge_case(g, bits, logn, batch, poly, iters, name, inverse=False):
    import torch
    P = O.Port(bits)
    prm = g.NTTParameters(logn, poly, bits)
    oprm = P.merge_params(logn, poly)
    n = 1 << logn
    x = P.splitmix(0x5EED0000 + logn, 0, batch * n, prm.modulus.value)
    d_in = g.to_device(x)
    d_out = torch.empty_like(d_in)
    tab = g.to_device(prm.inverse_table_device_order if inverse else prm.forward_table_device_order)
    cfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD,
                              reduction_poly=poly, mod_inverse=prm.n_inv if inverse else 0)
    fn = (lambda: g.GPU_INTT(d_in, d_out, tab, prm.modulus, cfg, batch)) if inverse else \
         (lambda: g.GPU_NTT(d_in, d_out, tab, prm.modulus, cfg, batch))
    fn()
    torch.cuda.synchronize()
    y = g.to_host(d_out)
    p = batch - 1
    ok = np.array_equal(y[p * n:(p + 1) * n], P.merge_ntt(x[p * n:(p + 1) * n], oprm, inverse=inverse))
    emit(name, bits, "merge-inv" if inverse else "merge-fwd", logn, batch, time_ms(fn, iters), ok)


def polymul_case(g, bits, logn, batch, iters, name):
    """extension GPU_PolyMul: out = INTT(NTT(a) (.) NTT(b)), negacyclic; reported per product with the
    algorithmic bytes of its three transforms + the pointwise step (3 x 2N + 3N words)."""
    import torch
    P = O.Port(bits)
    prm = g.NTTParameters(logn, g.X_N_plus, bits)
    oprm = P.merge_params(logn, O.X_N_plus)
    n = 1 << logn
    a = P.splitmix(0x5EED0010, 0, batch * n, prm.modulus.value)
    b = P.splitmix(0x5EED0011, 0, batch * n, prm.modulus.value)
    da, db = g.to_device(a), g.to_device(b)
    out = torch.empty_like(da)
    tf, ti = g.to_device(prm.forward_table_device_order), g.to_device(prm.inverse_table_device_order)
    cfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=g.X_N_plus,
                              mod_inverse=prm.n_inv)
    g.GPU_PolyMul(da, db, out, tf, ti, prm.modulus, cfg, batch)
    torch.cuda.synchronize()
    p = batch - 1
    sl = slice(p * n, (p + 1) * n)
    want = P.merge_ntt(P.pointwise(P.merge_ntt(a[sl], oprm), P.merge_ntt(b[sl], oprm), oprm["mod"]), oprm,
                       inverse=True)
    ok = np.array_equal(g.to_host(out)[sl], want)
    fn = lambda: g.GPU_PolyMul(da, db, out, tf, ti, prm.modulus, cfg, batch)  # noqa: E731
    ms = time_ms(fn, iters)
    d = {"name": name, "dtype": "u%d" % bits, "algo": "polymul-negacyclic", "log2N": logn, "batch": batch,
         "ms": round(ms, 5), "products_per_s": round(batch / (ms * 1e-3), 1),
         "alg_GBps": round(9 * n * (bits // 8) * batch / (ms * 1e-3) / 1e9, 1), "checked": bool(ok)}
    d["frac_of_8TBps"] = round(d["alg_GBps"] / PEAK, 4)
    print(json.dumps(d), flush=True)


def rns_case(g, logn, batch, iters, name, golden_dir):
    import torch
    rns = json.load(open(os.path.join(golden_dir, "rns_c5.json")))
    P = O.Port(64)
    n = 1 << logn
    mc = len(rns["primes"])
    prms = [g.NTTParameters(logn, g.X_N_plus, 64, (e["q"], e["omega"], e["psi"])) for e in rns["primes"]]
    fwd = np.zeros(mc * n, dtype=np.uint64)
    for i, p in enumerate(prms):
        fwd[i * n:i * n + p.root_of_unity_size] = p.forward_table_device_order
    mods = g.modulus_array_to_device([p.modulus for p in prms], 64)
    x = np.concatenate([P.splitmix(900 + p, 0, n, prms[p % mc].modulus.value) for p in range(batch)])
    d_in, tab = g.to_device(x), g.to_device(fwd)
    d_out = torch.empty_like(d_in)
    cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=g.X_N_plus)
    fn = lambda: g.GPU_NTT(d_in, d_out, tab, mods, cfg, batch, mc)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    y = g.to_host(d_out)
    p = batch - 3
    e = rns["primes"][p % mc]
    oprm = P.merge_params(logn, O.X_N_plus, (e["q"], e["omega"], e["psi"]))
    ok = np.array_equal(y[p * n:(p + 1) * n], P.merge_ntt(x[p * n:(p + 1) * n], oprm))
    emit(name, 64, "merge-fwd-rns8", logn, batch, time_ms(fn, iters), ok, {"mod_count": mc})


def fourstep_case(g, bits, logn, batch, iters, name, check=True):
    """GPU_4STEP_NTT forward + inverse on pre-transposed data (the API's own contract), plus the
    full natural-order pipeline including both GPU_Transpose sweeps."""
    import torch
    P = O.Port(bits)
    p4 = g.NTTParameters4Step(logn, bits)
    n = p4.n
    x = P.splitmix(0x5EED0003, 0, batch * n, p4.modulus.value)
    a = g.to_device(x)
    b = torch.empty_like(a)
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    ti = [g.to_device(t) for t in p4.tables["inv"]]
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
    ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
    ok = True
    if check:
        oprm = P.fourstep_params(logn)
        g.GPU_Transpose(a, b, p4.n1, p4.n2, logn, batch)
        torch.cuda.synchronize()
        g.GPU_4STEP_NTT(b, a, *tf, p4.modulus, cf, batch)
        torch.cuda.synchronize()
        g.GPU_Transpose(a, b, p4.n1, p4.n2, logn, batch)
        torch.cuda.synchronize()
        y = g.to_host(b)
        ok = np.array_equal(y[:n], P.fourstep_ntt(x[:n], oprm))
        a.copy_(g.to_device(x))
    fwd = lambda: g.GPU_4STEP_NTT(a, b, *tf, p4.modulus, cf, batch)  # noqa: E731
    inv = lambda: g.GPU_4STEP_NTT(a, b, *ti, p4.modulus, ci, batch)  # noqa: E731
    emit(name + "-fwd", bits, "4step-fwd", logn, batch, time_ms(fwd, iters), ok)
    # the RNS overload with one device-side modulus, as the reference's own example calls it
    mods = g.modulus_array_to_device([p4.modulus], bits)
    crf = g.ntt4step_rns_configuration(n_power=logn, ntt_type=g.FORWARD,
                                       mod_inverse=g.to_device(np.array([p4.n_inv], dtype=g.np_dtype(bits))))
    fwd_rns = lambda: g.GPU_4STEP_NTT(a, b, *tf, mods, crf, batch, 1)  # noqa: E731
    emit(name + "-fwd-rns-overload", bits, "4step-fwd-rns1", logn, batch, time_ms(fwd_rns, iters), ok)
    emit(name + "-inv", bits, "4step-inv", logn, batch, time_ms(inv, iters), ok)

    def full():
        g.GPU_Transpose(a, b, p4.n1, p4.n2, logn, batch)
        g.GPU_4STEP_NTT(b, a, *tf, p4.modulus, cf, batch)
        g.GPU_Transpose(a, b, p4.n1, p4.n2, logn, batch)
    emit(name + "-fwd+2transposes", bits, "4step-fwd-natural", logn, batch, time_ms(full, iters), ok)

    # the same natural-order result from the fused extension entry point (3 sweeps)
    a.copy_(g.to_device(x))
    g.GPU_4STEP_NTT_NaturalOrder(a, b, *tf, p4.modulus, cf, batch)
    torch.cuda.synchronize()
    ok2 = (not check) or np.array_equal(g.to_host(b)[:n], P.fourstep_ntt(x[:n], oprm))
    nat = lambda: g.GPU_4STEP_NTT_NaturalOrder(a, b, *tf, p4.modulus, cf, batch)  # noqa: E731
    emit(name + "-fwd-natural-fused", bits, "4step-fwd-natural-fused", logn, batch, time_ms(nat, iters), ok2)

    # inverse of the natural-order result: three-call form (first transpose on the device) vs the fused entry
    a.copy_(g.to_device(x))
    g.GPU_4STEP_NTT_NaturalOrder(a, b, *tf, p4.modulus, cf, batch)  # b = forward result
    c = torch.empty_like(a)
    g.GPU_4STEP_NTT_NaturalOrder(b, c, *ti, p4.modulus, ci, batch)
    torch.cuda.synchronize()
    ok3 = np.array_equal(g.to_host(c)[:n], x[:n])

    def inv3():
        g.GPU_Transpose(b, a, p4.n2, p4.n1, logn, batch)
        g.GPU_4STEP_NTT(a, c, *ti, p4.modulus, ci, batch)
        g.GPU_Transpose(c, a, p4.n1, p4.n2, logn, batch)
    emit(name + "-inv+2transposes", bits, "4step-inv-natural", logn, batch, time_ms(inv3, iters), ok3)
    invn = lambda: g.GPU_4STEP_NTT_NaturalOrder(b, c, *ti, p4.modulus, ci, batch)  # noqa: E731
    emit(name + "-inv-natural-fused", bits, "4step-inv-natural-fused", logn, batch, time_ms(invn, iters), ok3)


def fourstep_inv_case(g, bits, logn, batch, iters, name):
    """GPU_4STEP_NTT inverse alone, checked through the round trip with the forward call (polynomial 0 and the last one)."""
    import torch
    P = O.Port(bits)
    p4 = g.NTTParameters4Step(logn, bits)
    n = p4.n
    x = P.splitmix(0x5EED0004, 0, batch * n, p4.modulus.value)
    a = g.to_device(x)
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    ti = [g.to_device(t) for t in p4.tables["inv"]]
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
    ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
    g.GPU_4STEP_NTT(a, b, *tf, p4.modulus, cf, batch)   # a = x^T (n2 x n1)  ->  b = spectrum (n1 x n2)
    g.GPU_4STEP_NTT(b, c, *ti, p4.modulus, ci, batch)   # -> c = x as n1 x n2, i.e. the transpose of a's layout
    g.GPU_Transpose(c, a, p4.n1, p4.n2, logn, batch)    # n1 x n2 -> n2 x n1
    torch.cuda.synchronize()
    y = g.to_host(a)
    # the forward call read `a` as the n2 x n1 transpose of the natural polynomial; the inverse returns it n1 x n2
    # (low n1-index on top), which GPU_Transpose turned into natural order
    nat = lambda v: v.reshape(p4.n2, p4.n1).T.reshape(-1)  # noqa: E731
    ok = np.array_equal(y[:n], nat(x[:n])) and np.array_equal(y[-n:], nat(x[-n:]))
    inv = lambda: g.GPU_4STEP_NTT(b, c, *ti, p4.modulus, ci, batch)  # noqa: E731
    emit(name, bits, "4step-inv", logn, batch, time_ms(inv, iters), ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    g = _load_pkg()
    g.load_library()
    golden = os.path.join(ROOT, "tests", "golden")
    if args.what in ("configs", "all"):
        merge_case(g, 64, 16, 1024, g.X_N_minus, args.iters, "C2 Merge u64 2^16 x1024")
        merge_case(g, 64, 16, 1024, g.X_N_minus, args.iters, "C2i Merge-INTT u64 2^16 x1024", inverse=True)
        fourstep_case(g, 64, 24, 64, max(3, args.iters // 4), "C3 4-Step u64 2^24 x64", check=True)
        merge_case(g, 32, 14, 1024, g.X_N_minus, args.iters, "C4 Merge u32 2^14 x1024 (per-GPU shard of 8192)")
        merge_case(g, 32, 14, 8192, g.X_N_minus, args.iters, "C4full Merge u32 2^14 x8192 (whole batch on one GPU)")
        rns_case(g, 16, 512, args.iters, "C5 RNS Merge u64 2^16 x512, 8 primes, X^N+1", golden)
        polymul_case(g, 64, 16, 1024, args.iters, "PolyMul u64 2^16 x1024 (extension)")
    if args.what in ("sweep32", "sweep64"):
        bits = int(args.what[-2:])
        for logn in range(12, 25):
            batch = max(1, 1 << (26 - logn))
            merge_case(g, bits, logn, batch, g.X_N_minus, max(3, args.iters // 2), "sweep-merge")
            merge_case(g, bits, logn, batch, g.X_N_minus, max(3, args.iters // 2), "sweep-merge-inv", inverse=True)
    if args.what == "mergebig":  # three-sweep Merge rings, both directions (A/B of the stage split: GPUNTT_CONTIG_K)
        for logn in range(23, 27):
            batch = max(1, 1 << (27 - logn))
            merge_case(g, 64, logn, batch, g.X_N_minus, max(3, args.iters // 2), "big-merge")
            merge_case(g, 64, logn, batch, g.X_N_minus, max(3, args.iters // 2), "big-merge-inv", inverse=True)
    if args.what == "4step64":  # forward / inverse of the reference layout, every ring size
        for logn in range(12, 25):
            fourstep_case(g, 64, logn, max(1, 1 << (26 - logn)), max(3, args.iters // 2), "sweep-4step", check=(logn <= 20))
    if args.what == "4stepinv":  # inverse of the reference layout above one tile
        for bits in (64, 32):
            for logn in range(14, 25):
                fourstep_inv_case(g, bits, logn, max(1, 1 << (26 - logn)), max(3, args.iters // 2), "4step-inv-u%d" % bits)
        fourstep_inv_case(g, 64, 24, 64, 5, "C3 4-Step u64 2^24 x64-inv")
    if args.what == "4step32":  # the 32-bit one-launch rings and their neighbours
        for logn in range(12, 25):
            fourstep_case(g, 32, logn, max(1, 1 << (26 - logn)), max(3, args.iters // 2), "sweep-4step-u32", check=(logn <= 20))
    if args.what in ("sweep", "all"):
        for bits in (64, 32):
            for logn in range(12, 25):
                batch = max(1, 1 << (26 - logn))
                merge_case(g, bits, logn, batch, g.X_N_minus, max(3, args.iters // 2), "sweep-merge")
        for logn in range(12, 25):
            batch = max(1, 1 << (26 - logn))
            fourstep_case(g, 64, logn, batch, max(3, args.iters // 2), "sweep-4step", check=(logn <= 20))


if __name__ == "__main__":
    main()
