"""-m gpu parity tests: The lazy-range kernel families (31 q / 16 q / 8 q / 4 q; 32-bit 8 q / 4 q) and the family prediction of the RNS overloads, whose moduli live in device memory (reference GPU_NTT RNS overload, src/lib/ntt_merge/ntt.cu:2560-2746): 61- / 62-bit moduli, stacks mixing widths, rings below one tile, moduli rewritten in place, the preparation kernel's own fall-back, the wide net of large rings."""
import os
import time

import numpy as np
import pytest

from gpu_utils import (MergeCase, cpu_class_on_tables, distinct_factors, distinct_factors_scaled, find_ntt_factors,  # noqa: F401
                       oracle_batch, rns_stack)
from oracle import oracle as O
from test_gpu_merge import _rns_setup, _small_prime_factors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    pkg.load_library()
    return pkg

@pytest.mark.parametrize("qbits", [61, 62])
def test_61_and_62_bit_moduli_on_the_fast_kernels(g, qbits):
    """61- and 62-bit moduli (the top of the reference's documented domain, modular_arith.cuh:66-67) run on the
    lazy-residue kernels -- LIMIT = 8 range schedule for 61 bits, LIMIT = 4 with products corrected to [0, 2q) for 62 --
    instead of dropping to the Barrett kernels: every plan shape, both directions, drop-in calls and NTTPlan (which
    reports the path), an RNS plan mixing 59/60/61-bit primes"""
    import torch
    for logn, batch in ((4, 9), (10, 5), (12, 3), (13, 6), (16, 5), (18, 2), (21, 2)):
        for poly in (O.X_N_plus, O.X_N_minus):
            # (skip the primes within 2^-40 of a power of two: the reference's floating-point log2 rounds them up)
            c = MergeCase(g, 64, logn, poly, find_ntt_factors(qbits, logn, skip=400 if logn < 10 else 0))
            assert c.prm.modulus.bit == qbits
            x = c.random(batch, 6100 + logn)
            want = c.P.merge_ntt(x, c.oprm)
            assert np.array_equal(c.gpu_forward(x, inplace=bool(logn & 1)), want), ("fwd", logn, poly)
            assert np.array_equal(c.gpu_inverse(want, inplace=not (logn & 1)), x), ("inv", logn, poly)
            assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True))
            fplan = g.NTTPlan(c.fwd_dev, c.prm.modulus, logn, poly, g.FORWARD, batch_hint=batch)
            assert fplan.fast_path == (logn >= 5)
            d = g.to_device(x)
            fplan.execute(d, d, batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), want)
    if qbits == 62:
        return
    # RNS plan with one 61-bit prime in the stack
    logn, batch = 14, 9
    fl = [find_ntt_factors(59, logn), find_ntt_factors(61, logn), find_ntt_factors(60, logn)]
    cases, fwd, inv, mods, ninv = _rns_setup(g, 64, logn, O.X_N_plus, fl)
    n = 1 << logn
    x = np.concatenate([cases[p % 3].P.splitmix(6600 + p, 0, n, cases[p % 3].q) for p in range(batch)])
    want = np.concatenate([cases[p % 3].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % 3].oprm) for p in range(batch)])
    moduli = [c.prm.modulus for c in cases]
    fplan = g.NTTPlan(fwd, moduli, logn, O.X_N_plus, g.FORWARD, batch_hint=batch)
    iplan = g.NTTPlan(inv, moduli, logn, O.X_N_plus, g.INVERSE, mod_inverse=[c.prm.n_inv for c in cases], batch_hint=batch)
    assert fplan.fast_path and iplan.fast_path
    d = g.to_device(x)
    fplan.execute(d, d, batch)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(d), want)
    iplan.execute(d, d, batch)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(d), x)

def test_moduli_just_below_a_power_of_two(g):
    """Modulus<T>::bit is (T)(log2((double) q) + 1) as in the reference (modular_arith.cuh:44-47): a prime within
    ~2^-48 of a power of two gets its width over-stated by one (2^60 - 107 -> bit 61).  The three words equal the
    oracle's, and every kernel family computes the right transform with them (the over-stated width selects the
    next lazy range: 60-bit -> 8q kernels).  A 61-bit prime that close to 2^61 gets bit = 62 and a mu that no longer
    fits the word (2^125 / q >= 2^64): the reference stores the truncated value and its own Barrett product is wrong
    from then on -- this library refuses to build such a Modulus."""
    fac = find_ntt_factors(61, 3)
    with pytest.raises((ValueError, g.GpuNttError)):
        g.Modulus(fac[0], bits=64)
    seen = set()
    for qbits in (54, 57, 59, 60):
        for logn in (1, 3, 5, 12, 13):
            fac = find_ntt_factors(qbits, logn if logn < 12 else 5)  # the same near-2^k primes serve the big rings
            fac = fac if logn < 12 else find_ntt_factors(qbits, logn)
            c = MergeCase(g, 64, logn, O.X_N_plus, fac)  # asserts {value, bit, mu} == oracle's
            seen.add((qbits, int(c.prm.modulus.bit)))
            x = c.random(5, 9100 + logn + qbits)
            want = c.P.merge_ntt(x, c.oprm)
            assert np.array_equal(c.gpu_forward(x, inplace=bool(logn & 2)), want), (qbits, logn)
            assert np.array_equal(c.gpu_inverse(want, inplace=not (logn & 2)), x), (qbits, logn)
    assert (60, 61) in seen and (59, 60) in seen and (57, 58) in seen  # the over-stated widths were exercised

def test_31q_range_switch(g):
    """Forward transforms of 64-bit moduli with 31 q < 2^64 (every pool prime) take the LIMIT = 31 kernels, 32-bit
    moduli below 2^29 the LIMIT = 8 kernels (both directions).
    All must equal the oracle, and a 60-bit prime above 2^64 / 31 / a 30-bit prime must stay on the default
    kernels (and equal the oracle too)."""
    from test_gpu_merge import _run_in_subprocess
    code = r'''
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.environ["PYTHONPATH"]), ""))
from conftest import load_pkg
from gpu_utils import MergeCase, find_ntt_factors
from oracle import oracle as O
g = load_pkg(); g.load_library()
for logn, batch in ((9, 9), (12, 5), (16, 7), (17, 3), (20, 2)):
    for poly in (O.X_N_minus, O.X_N_plus):
        c = MergeCase(g, 64, logn, poly)
        assert c.q <= (2**64 - 1) // 31
        x = c.random(batch, 77 + logn)
        assert np.array_equal(c.gpu_forward(x, inplace=bool(logn & 1)), c.P.merge_ntt(x, c.oprm)), (logn, poly)
# 32-bit words: pool prime below 2^29 (LIMIT = 8 kernels when switched on), 4096- and 16384-coefficient tiles
for logn, batch in ((9, 9), (12, 5), (14, 6), (16, 4), (20, 2)):
    for poly in (O.X_N_minus, O.X_N_plus):
        c = MergeCase(g, 32, logn, poly)
        assert c.q < 2**29
        x = c.random(batch, 177 + logn)
        want = c.P.merge_ntt(x, c.oprm)
        assert np.array_equal(c.gpu_forward(x, inplace=bool(logn & 1)), want), (32, logn, poly)
        assert np.array_equal(c.gpu_inverse(want, inplace=not (logn & 1)), x), (32, logn, poly)
        assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True)), (32, logn, poly)
f32 = find_ntt_factors(30, 14)  # a 30-bit prime keeps the 4 q kernels
c = MergeCase(g, 32, 14, O.X_N_plus, f32)
x = c.random(3, 6)
assert np.array_equal(c.gpu_forward(x), c.P.merge_ntt(x, c.oprm))
assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True))
f = find_ntt_factors(60, 16)
assert f[0] > (2**64 - 1) // 31
c = MergeCase(g, 64, 16, O.X_N_plus, f)
x = c.random(3, 5)
assert np.array_equal(c.gpu_forward(x), c.P.merge_ntt(x, c.oprm))
print("31q switch OK")
'''
    for env in ({}, {"GPUNTT_PATH": "fast-strict"}):
        assert "31q switch OK" in _run_in_subprocess(code, env)

def test_31q_range_at_its_boundary(g):
    """The largest NTT primes at or below (2^64 - 1) / 31 leave the LIMIT = 31 kernels no slack (31 q is within
    2^17 * 31 of 2^64), the next ones above take the 16 q kernels: worst-case inputs (all q - 1, alternating 0 / q - 1)
    and random ones against the oracle, single pass, two passes and the big tiles."""
    from gpu_utils import _is_probable_prime
    bound = (2**64 - 1) // 31

    def factors(logn, below):
        step = 1 << (logn + 1)
        q = bound // step * step + 1
        while (q > bound) if below else (q <= bound):
            q += -step if below else step
        while not _is_probable_prime(q):
            q += -step if below else step
        assert (q <= bound) == below
        gen = 2
        while True:
            psi = pow(gen, (q - 1) >> (logn + 1), q)
            if pow(psi, 1 << logn, q) == q - 1:
                return q, psi * psi % q, psi
            gen += 1

    for logn in (12, 13, 16):
        for below in (True, False):
            f = factors(logn, below)
            for poly in (O.X_N_minus, O.X_N_plus):
                c = MergeCase(g, 64, logn, poly, f)
                n, q = c.n, c.q
                worst = [np.full(n, q - 1, dtype=object), np.array([0, q - 1] * (n // 2), dtype=object),
                         np.array([q - 1, 0] * (n // 2), dtype=object)]
                x = np.concatenate([np.concatenate(worst).astype(c.P.T), c.random(2, 3 + logn)])
                want = c.P.merge_ntt(x, c.oprm)
                assert np.array_equal(c.gpu_forward(x), want), (logn, below, poly)
                assert np.array_equal(c.gpu_inverse(want, inplace=True), x), (logn, below, poly)

@pytest.mark.parametrize("logn,batch,widths", [(12, 11, (60, 61)), (13, 10, (62, 60, 61)), (14, 260, (60, 60, 61, 60)),
                                               (14, 7, (60, 60)), (16, 9, (61, 60, 60)), (16, 8, (62, 62)), (21, 3, (60, 61)),
                                               (22, 2, (62, 60))])
def test_rns_stacks_with_61_and_62_bit_primes_on_the_lazy_kernels(g, logn, batch, widths):
    """a drop-in RNS call whose moduli live in device memory classifies them in its preparation kernel;
    the go-flag now has four states, so a stack that contains a 61- / 62-bit prime (inside the reference's domain,
    modular_arith.cuh:66-67; RNS indexing ntt.cu:613) runs the 8 q / 4 q lazy family instead of the Barrett kernels.  path =
    fast-strict enqueues NO generic kernels behind an RNS call, so a result can only come from a lazy family.  Every
    polynomial, forward + inverse, rings whose default family uses a bigger tile included (2^13, 2^14 x 260, 2^21, 2^22:
    the table is permuted on the device for the family that runs)."""
    import torch
    poly = O.X_N_plus if logn % 2 else O.X_N_minus
    cases = [MergeCase(g, 64, logn, poly, f) for f in distinct_factors_scaled(widths, logn)]
    mc, n = len(cases), 1 << logn
    fwd = np.zeros(mc * n, dtype=np.uint64)
    inv = np.zeros_like(fwd)
    for i, c in enumerate(cases):
        sz = c.prm.root_of_unity_size
        fwd[i * n:i * n + sz] = c.prm.forward_table_device_order
        inv[i * n:i * n + sz] = c.prm.inverse_table_device_order
    d_fwd, d_inv = g.to_device(fwd), g.to_device(inv)
    mods = g.modulus_array_to_device([c.prm.modulus for c in cases], 64)
    ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=np.uint64))
    x = np.concatenate([cases[p % mc].P.splitmix(91000 + p, 0, n, cases[p % mc].q) for p in range(batch)])
    want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm) for p in range(batch)])
    cf = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
    ci = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
    g.set_option("path", "fast-strict")
    try:
        for rep in range(2):
            d = g.to_device(x)
            o = torch.zeros_like(d)
            g.GPU_NTT(d, o, d_fwd, mods, cf, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), want), ("forward", rep)
            g.GPU_INTT_Inplace(o, d_inv, mods, ci, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), x), ("inverse", rep)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))

def test_rns_family_prediction_survives_moduli_rewritten_in_place(g):
    """Drop-in RNS calls enqueue only the lazy family their stack needed LAST time (host::RnsGuess: the preparation kernel
    reports the classification to a host-mapped word the next call reads without synchronising) plus the generic kernels
    behind "return if the flag names the predicted state".  The stateless contract must survive the worst caller: ONE
    device buffer of moduli rewritten between calls with stacks of different widths (every prediction stale or wrong),
    with and without a synchronisation in between, option rns_predict off, and the 4-step RNS overload the same way."""
    import torch
    logn, batch = 13, 9
    n = 1 << logn
    stacks = {}
    import json
    c5 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rns_c5.json")))["primes"]
    pool = []  # primes of the reference's pools (2^59 + small: 31 q < 2^64 -> the 31 q family on forward calls)
    for e in c5[:3]:
        psi = pow(e["psi"], 1 << (16 - logn), e["q"])
        pool.append((e["q"], psi * psi % e["q"], psi))
    for name, widths in (("w60", (60, 60, 60)), ("w61", (60, 61, 60)), ("w62", (62, 60, 61)), ("pool", None)):
        cases = [MergeCase(g, 64, logn, O.X_N_plus, f) for f in (pool if widths is None else distinct_factors_scaled(widths, logn))]
        fwd = np.zeros(3 * n, dtype=np.uint64)
        for i, c in enumerate(cases):
            fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
        x = np.concatenate([cases[p % 3].P.splitmix(97000 + p, 0, n, cases[p % 3].q) for p in range(batch)])
        want = np.concatenate([cases[p % 3].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % 3].oprm) for p in range(batch)])
        stacks[name] = (g.modulus_array_to_device([c.prm.modulus for c in cases], 64), g.to_device(fwd), x, want)
    mods = torch.zeros_like(stacks["w60"][0])  # THE buffer every call below passes
    cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus)
    order = ["w60", "w60", "pool", "pool", "w61", "w61", "w61", "pool", "w60", "w62", "w62", "w60", "w61", "w62", "pool", "pool"]
    for predict in ("1", "0"):
        g.set_option("rns_predict", predict)
        try:
            for sync in (True, False):
                for i, name in enumerate(order):
                    src, table, x, want = stacks[name]
                    mods.copy_(src)
                    d = g.to_device(x)
                    o = torch.zeros_like(d)
                    g.GPU_NTT(d, o, table, mods, cfg, batch, 3)
                    if sync:
                        torch.cuda.synchronize()
                    assert np.array_equal(g.to_host(o), want), (predict, sync, i, name)
        finally:
            g.set_option("rns_predict", "1")
    # the 4-step RNS overload, one device-side modulus rewritten in place: 60 -> 62 -> 62 -> 61 -> 60 bits
    P = O.Port(64)
    logn, batch = 13, 3
    shape = g.NTTParameters4Step(logn, 64)
    n, n1, n2 = shape.n, shape.n1, shape.n2
    mod_buf = None
    for step, qbits in enumerate((60, 62, 62, 61, 60, 60)):
        q, omega, psi = distinct_factors_scaled([qbits], logn)[0]
        m = g.Modulus(q, bits=64)
        oprm = P.merge_params(logn, O.X_N_minus, (q, omega, psi))
        x = P.splitmix(97500 + step, 0, batch * n, q)
        y = P.merge_ntt(x, oprm)
        w = torch.zeros(n, dtype=torch.int64, device="cuda")
        t1 = torch.zeros(n1 >> 1, dtype=torch.int64, device="cuda")
        t2 = torch.zeros(n2 >> 1, dtype=torch.int64, device="cuda")
        g.GPU_Generate4StepW(w, omega, m, logn, g.FORWARD)
        g.GPU_GeneratePowerTable(t1, pow(omega, n // n1, q), m, int(np.log2(n1)) - 1, True)
        g.GPU_GeneratePowerTable(t2, pow(omega, n // n2, q), m, int(np.log2(n2)) - 1, True)
        src = g.modulus_array_to_device([m], 64)
        if mod_buf is None:
            mod_buf = torch.zeros_like(src)
        mod_buf.copy_(src)
        ninv = g.to_device(np.array([pow(n, -1, q)], dtype=np.uint64))
        cfg4 = g.ntt4step_rns_configuration(n_power=logn, ntt_type=g.FORWARD, mod_inverse=ninv)
        d_in = g.to_device(x.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1).copy())
        d_out = torch.zeros_like(d_in)
        g.GPU_4STEP_NTT(d_in, d_out, t1, t2, w, mod_buf, cfg4, batch, 1)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_out), y), ("4-step", step, qbits)

def test_rns_calls_from_four_host_threads_share_one_prediction_slot(g):
    """four host threads, each on its own stream, call the drop-in RNS entry points with the SAME device moduli (one
    prediction slot, one mutex: host::rns_guess) -- forward, inverse, repeatedly, while a fifth stack is used from the main
    thread; every result equal to the oracle.  (ctypes releases the GIL: the library calls really overlap on the host.)"""
    import threading
    import torch
    logn, batch, mc = 14, 6, 3
    n = 1 << logn
    cases = [MergeCase(g, 64, logn, O.X_N_plus, f) for f in distinct_factors_scaled((60, 61, 60), logn)]
    fwd = np.zeros(mc * n, dtype=np.uint64)
    inv = np.zeros_like(fwd)
    for i, c in enumerate(cases):
        fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
        inv[i * n:i * n + c.prm.root_of_unity_size] = c.prm.inverse_table_device_order
    d_fwd, d_inv = g.to_device(fwd), g.to_device(inv)
    mods = g.modulus_array_to_device([c.prm.modulus for c in cases], 64)
    ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=np.uint64))
    xs = [np.concatenate([cases[p % mc].P.splitmix(98000 + 100 * t + p, 0, n, cases[p % mc].q) for p in range(batch)])
          for t in range(4)]
    wants = [np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm) for p in range(batch)])
             for x in xs]
    errors = []

    def worker(t):
        try:
            s = torch.cuda.Stream()
            cf = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus, stream=s)
            ci = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=O.X_N_plus, mod_inverse=ninv, stream=s)
            with torch.cuda.stream(s):
                d = g.to_device(xs[t])
            s.synchronize()
            for it in range(25):
                g.GPU_NTT_Inplace(d, d_fwd, mods, cf, batch, mc)
                if it % 5 == 0:
                    s.synchronize()
                    if not np.array_equal(g.to_host(d), wants[t]):
                        errors.append(("forward", t, it))
                g.GPU_INTT_Inplace(d, d_inv, mods, ci, batch, mc)
            s.synchronize()
            if not np.array_equal(g.to_host(d), xs[t]):
                errors.append(("round trip", t))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    torch.cuda.synchronize()
    for th in threads:
        th.start()
    # meanwhile, another stack from the main thread (default stream)
    c5 = MergeCase(g, 64, 13, O.X_N_minus)
    x5 = c5.random(4, 98900)
    for _ in range(10):
        assert np.array_equal(c5.gpu_forward(x5), c5.P.merge_ntt(x5, c5.oprm))
    for th in threads:
        th.join()
    assert not errors, errors

@pytest.mark.parametrize("bits", [64, 32])
def test_rns_rings_below_one_tile_on_the_lazy_kernels(g, bits):
    """VERDICT r4 missing #3: RNS stacks of rings smaller than a tile (reference ForwardCoreLowRing / InverseCoreLowRing RNS
    forms, src/lib/ntt_merge/ntt.cu:116-219, 326-433) ran on the Barrett kernels (7 x slower).  A 4096-coefficient tile now
    holds polynomials of different moduli: per wave from 1024 coefficients (scalar modulus), per lane for 16 .. 512
    (kern::merge_pass_lazy_vqc).  path = fast-strict: no generic kernel is enqueued, the lazy families own every call.
    N = 2^4 .. 2^11, mod_count 2 .. 8, stacks of 60-bit primes and stacks with 61- / 62-bit primes (the 8 q / 4 q families),
    ragged batches, both polynomials, in place and out of place, every polynomial against NTTCPU; NTTPlan the same."""
    import torch
    g.set_option("path", "fast-strict")
    try:
        shapes = [(4, 2, 1000), (5, 3, 777), (6, 5, 513), (7, 8, 300), (8, 7, 129), (9, 4, 65), (10, 3, 23), (11, 6, 13),
                  (9, 2, 3), (4, 8, 4096), (10, 8, 64), (11, 2, 2)]
        for idx, (logn, mc, batch) in enumerate(shapes):
            if bits == 64:
                widths = [(60, 60, 60, 60), (60, 61, 60, 61), (62, 60, 61, 60)][idx % 3]
            else:
                widths = (30, 29, 30, 28)
            widths = [widths[i % 4] for i in range(mc)]
            poly = O.X_N_plus if idx % 2 == 0 else O.X_N_minus
            cases, d_fwd, d_inv = rns_stack(g, bits, logn, widths, poly)
            n = 1 << logn
            mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
            ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=cases[0].P.T))
            x = np.concatenate([cases[p % mc].P.splitmix(98000 + 31 * idx + p, 0, n, cases[p % mc].q) for p in range(batch)])
            want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm) for p in range(batch)])
            cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
            icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
            for rep in range(2):  # the second round runs with the stack's family prediction in place
                d = g.to_device(x)
                o = torch.zeros_like(d)
                g.GPU_NTT(d, o, d_fwd, mods, cfg, batch, mc)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(o), want), ("fwd", bits, logn, mc, batch, rep)
                g.GPU_INTT_Inplace(o, d_inv, mods, icfg, batch, mc)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(o), x), ("inv", bits, logn, mc, batch, rep)
            # prepared form
            fplan = g.NTTPlan(d_fwd, [c.prm.modulus for c in cases], logn, poly, g.FORWARD, batch_hint=batch)
            iplan = g.NTTPlan(d_inv, [c.prm.modulus for c in cases], logn, poly, g.INVERSE,
                              mod_inverse=[c.prm.n_inv for c in cases], batch_hint=batch)
            assert fplan.fast_path and iplan.fast_path, (bits, logn, mc)
            d = g.to_device(x)
            fplan.execute(d, d, batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), want), ("plan fwd", bits, logn, mc)
            iplan.execute(d, d, batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), x), ("plan inv", bits, logn, mc)
            fplan.close()
            iplan.close()
        # rings of 2 .. 8 coefficients with mod_count > 1 stay on the generic kernels (a thread's 16 coefficients would span
        # polynomials of different moduli): refused under fast-strict, exact on the default path
        cases, d_fwd, d_inv = rns_stack(g, bits, 3, [60, 59, 60] if bits == 64 else [30, 29, 30], O.X_N_plus)
        mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
        x = np.concatenate([cases[p % 3].P.splitmix(98900 + p, 0, 8, cases[p % 3].q) for p in range(700)])
        want = np.concatenate([cases[p % 3].P.merge_ntt(x[p * 8:(p + 1) * 8], cases[p % 3].oprm) for p in range(700)])
        cfg = g.ntt_rns_configuration(n_power=3, reduction_poly=O.X_N_plus)
        d = g.to_device(x)
        with pytest.raises(ValueError, match="fast path unavailable"):
            g.GPU_NTT_Inplace(d, d_fwd, mods, cfg, 700, 3)
        g.set_option("path", "default")
        g.GPU_NTT_Inplace(d, d_fwd, mods, cfg, 700, 3)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), want)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))

# ---------------------------------------------------------------- the preparation kernel's own fall-back
@pytest.mark.parametrize("bits", [64, 32])
def test_rns_fallback_inside_the_preparation_kernel(g, bits, golden_dir):
    """Drop-in RNS Merge calls have NO generic launch behind them any more (two skipped shadow launches
    cost every call ~13 us): when the stack does not fit the one lazy family the host enqueued, the preparation kernel
    transforms the batch itself (prep.hip: slow_rns_transform).  Option rns_force_fallback makes that path serve EVERY
    drop-in RNS Merge call: the existing RNS, signed / centred, *_Ordered, GPU_PolyMul and small-ring tests are re-run
    through it, and a direct sweep of shapes -- rings 2^4 .. 2^17, both polynomials, ragged batches, out of place --
    against NTTCPU per polynomial."""
    import torch
    import test_gpu_merge as M
    g.set_option("rns_force_fallback", "1")
    try:
        M.test_rns_multi_modulus(g, bits)
        M.test_modulus_ordered_and_poly_ordered(g, bits)
        if bits == 64:
            M.test_polymul_rns(g)
            M.test_rns_c5_against_golden(g, golden_dir)
        for idx, (logn, mc, batch) in enumerate(((4, 3, 1000), (9, 2, 37), (12, 5, 11), (13, 4, 9), (16, 3, 4), (17, 2, 3))):
            widths = ([60, 61, 62, 60, 59] if bits == 64 else [30, 29, 30, 28, 27])[:mc]
            poly = O.X_N_plus if idx % 2 == 0 else O.X_N_minus
            cases, d_fwd, d_inv = rns_stack(g, bits, logn, widths, poly)
            n = 1 << logn
            mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
            ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=cases[0].P.T))
            x = np.concatenate([cases[p % mc].P.splitmix(99000 + 17 * idx + p, 0, n, cases[p % mc].q) for p in range(batch)])
            want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm) for p in range(batch)])
            cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
            icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
            d = g.to_device(x)
            o = torch.zeros_like(d)
            g.GPU_NTT(d, o, d_fwd, mods, cfg, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), x) and np.array_equal(g.to_host(o), want), ("fwd", bits, logn, mc)
            # signed input: x - q on every second coefficient is the same residue
            xs = x.astype(np.int64 if bits == 64 else np.int32)
            qs = np.concatenate([np.full(n, cases[p % mc].q, dtype=np.uint64) for p in range(batch)]).astype(xs.dtype)
            xs[1::2] -= qs[1::2]
            ds = g.to_device(xs)
            g.GPU_NTT(ds, o, d_fwd, mods, cfg, batch, mc, dtype="s%d" % bits)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), want), ("signed fwd", bits, logn, mc)
            g.GPU_INTT_Inplace(o, d_inv, mods, icfg, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), x), ("inv", bits, logn, mc)
            # centred signed output of the inverse
            o.copy_(g.to_device(want))
            c_out = torch.zeros_like(o)
            g.GPU_INTT(o, c_out, d_inv, mods, icfg, batch, mc, dtype="s%d" % bits)
            torch.cuda.synchronize()
            got = g.to_host(c_out, signed=True).astype(object)
            for p in (0, batch - 1):
                q = cases[p % mc].q
                ref = [int(v) - q if int(v) > q // 2 else int(v) for v in x[p * n:(p + 1) * n]]
                assert [int(v) for v in got[p * n:(p + 1) * n]] == ref, ("centred", bits, logn, p)
        # the PerCoefficient layout with a stack: the same fall-back walks COLUMNS (coefficient i of column c at i * w + c)
        for logn, w, mc in ((9, 512, 3), (6, 256, 5), (3, 1024, 2)):
            widths = ([60, 61, 62, 60, 59] if bits == 64 else [30, 29, 30, 28, 27])[:mc]
            cases, d_fwd, d_inv = rns_stack(g, bits, logn, widths, O.X_N_plus)
            n = 1 << logn
            mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
            ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=cases[0].P.T))
            cols = np.stack([cases[p % mc].P.splitmix(99500 + p, 0, n, cases[p % mc].q) for p in range(w)])
            mat = np.ascontiguousarray(cols.T)
            want_f = np.stack([cases[p % mc].P.merge_ntt(cols[p], cases[p % mc].oprm) for p in range(w)]).T
            cfg = g.ntt_rns_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=O.X_N_plus)
            icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, ntt_layout=g.PerCoefficient,
                                           reduction_poly=O.X_N_plus, mod_inverse=ninv)
            d = g.to_device(mat.reshape(-1))
            o = torch.zeros_like(d)
            g.GPU_NTT(d, o, d_fwd, mods, cfg, w, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o).reshape(n, w), want_f), ("percoefficient fwd", bits, logn, w, mc)
            g.GPU_INTT_Inplace(o, d_inv, mods, icfg, w, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o).reshape(n, w), mat), ("percoefficient inv", bits, logn, w, mc)
    finally:
        g.set_option("rns_force_fallback", "0")

@pytest.mark.parametrize("bits", [64, 32])
def test_preparation_reciprocal_is_exact(g, bits):
    """floor(2^(W-1+b) / q) as the preparation kernels derive it for every device-side modulus (round 5: a double-precision
    estimate + exact 128-bit corrections instead of a 64-step restoring division -- it sits on the critical path of every
    drop-in RNS call).  Every Shoup quotient of an RNS call is built on it: it must be EXACT.  Edge moduli (just above /
    below powers of two, the pool primes, 3, the largest words) and 200 000 random ones against Python integers."""
    import torch
    W = bits
    rng = np.random.default_rng(bits)
    qs = [3, 5, 6, 7, 10000, 469762049, (1 << (W - 2)) - 1, (1 << (W - 2)) + 1, (1 << (W - 1)) - 1, (1 << (W - 1)) + 1,
          (1 << W) - 1, (1 << W) - 2, 4, 8, 1 << (W - 1), 2, 1, 0]
    if bits == 64:
        qs += [576460756061519873, 576460752308273153, (1 << 61) - 1, (1 << 62) - 57, (1 << 60) + 33, (1 << 59) + 1,
               (1 << 62) + 1, (1 << 61) + 1, (1 << 33) + 1, (1 << 32) - 1, (1 << 32) + 1, (1 << 53) + 1, (1 << 53) - 1]
    for b in range(2, W + 1):
        qs += [(1 << (b - 1)) + 1, (1 << b) - 1, (1 << (b - 1)) + (1 << (b - 2)) if b > 2 else 3]
    widths = rng.integers(2, W + 1, size=200000)
    rnd = [int(rng.integers(1 << (int(b) - 1), 1 << int(b), dtype=np.uint64)) if b < 64 else
           int(rng.integers(1 << 63, (1 << 64) - 1, dtype=np.uint64, endpoint=True)) for b in widths]
    qs = np.array(qs + rnd, dtype=np.uint64).astype(g.np_dtype(bits))
    got = g.to_host(g.debug_recip_norm(g.to_device(qs)))
    torch.cuda.synchronize()
    for q, r in zip(qs.tolist(), got.tolist()):
        q = int(q)
        want = 0 if (q < 3 or q & (q - 1) == 0) else (1 << (W - 1 + q.bit_length())) // q
        assert int(r) == want, (bits, q, int(r), want)

def _timed_call(fn):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

@pytest.mark.parametrize("logn,batch", [(17, 6), (20, 3)])
def test_first_call_of_a_wide_stack_on_a_large_ring_is_not_the_slow_path(g, logn, batch):
    """61- / 62-bit stacks in moduli buffers the library has never seen, rings of 2^17 and 2^20: the first call (and every
    call, when the caller uploads its stack to a fresh buffer each time) is exact and takes milliseconds, not the
    in-preparation fall-back's tens of milliseconds per polynomial; both directions, the *_Ordered entry point too."""
    import torch
    poly = O.X_N_plus
    for widths in ([60, 61, 60], [62, 60], [60, 60]):
        mc = len(widths)
        cases, d_fwd, d_inv = rns_stack(g, 64, logn, widths, poly)
        n = 1 << logn
        x = np.concatenate([cases[p % mc].P.splitmix(123 + p, 0, n, cases[p % mc].q) for p in range(batch)])
        want = oracle_batch(cases, x)
        ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=np.uint64))
        cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
        icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
        worst, keep = 0.0, []
        for rep in range(3):  # a FRESH moduli buffer per call: the pointer-keyed prediction never settles
            mods = g.modulus_array_to_device([c.prm.modulus for c in cases], 64)
            d = g.to_device(x)
            worst = max(worst, _timed_call(lambda: g.GPU_NTT_Inplace(d, d_fwd, mods, cfg, batch, mc)))
            assert np.array_equal(g.to_host(d), want), (widths, rep, "fwd")
            worst = max(worst, _timed_call(lambda: g.GPU_INTT_Inplace(d, d_inv, mods, icfg, batch, mc)))
            assert np.array_equal(g.to_host(d), x), (widths, rep, "inv")
            keep.append(mods)  # (the next buffer must not reuse this address)
        # generic kernels on 3 polynomials of 2^20: ~2 ms; the in-preparation fall-back: > 100 ms
        assert worst < 40.0, (widths, logn, "a first call took %.1f ms" % worst)
        # ordered entry point on a fresh buffer
        mods = g.modulus_array_to_device([c.prm.modulus for c in cases], 64)
        order = torch.tensor(list(range(mc))[::-1], dtype=torch.int32, device="cuda")
        xo = np.concatenate([cases[(mc - 1 - p % mc)].P.splitmix(777 + p, 0, n, cases[(mc - 1 - p % mc)].q) for p in range(batch)])
        wo = np.concatenate([cases[(mc - 1 - p % mc)].P.merge_ntt(xo[p * n:(p + 1) * n], cases[(mc - 1 - p % mc)].oprm)
                             for p in range(batch)])
        d, o = g.to_device(xo), torch.zeros(batch * n, dtype=torch.int64, device="cuda")
        ms = _timed_call(lambda: g.GPU_NTT_Modulus_Ordered(d, o, d_fwd, mods, cfg, batch, mc, order))
        assert np.array_equal(g.to_host(o), wo) and ms < 40.0, (widths, "ordered", ms)

def test_fresh_buffer_per_call_keeps_the_family_of_its_shape(g):
    """Small rings keep the in-preparation fall-back for a stack the enqueued family cannot serve (milliseconds there).  A
    caller that uploads the same 61-bit stack to a fresh buffer per call must meet it ONCE: the prediction for a buffer
    never seen before starts from what stacks of the same shape needed last (prep.hip: g_shape_hint)."""
    logn, batch, widths = 13, 64, [61, 60, 60, 60, 60]  # five primes: a shape no other test of this module uses
    mc = len(widths)
    cases, d_fwd, _ = rns_stack(g, 64, logn, widths, O.X_N_plus)
    n = 1 << logn
    x = np.concatenate([cases[p % mc].P.splitmix(5 + p, 0, n, cases[p % mc].q) for p in range(batch)])
    want = oracle_batch(cases, x)
    cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus)
    times, keep = [], []
    for rep in range(6):
        mods = g.modulus_array_to_device([c.prm.modulus for c in cases], 64)
        keep.append(mods)
        d = g.to_device(x)
        if rep == 1:  # the first call's state must have reached the host-mapped word before the second prediction
            import torch
            torch.cuda.synchronize()
        times.append(_timed_call(lambda: g.GPU_NTT_Inplace(d, d_fwd, mods, cfg, batch, mc)))
        assert np.array_equal(g.to_host(d), want), rep
    # calls 3 .. 6 run on the predicted 8 q family: as fast as each other and several times faster than a fall-back call
    assert max(times[2:]) < 2.0, times

def test_out_of_domain_stack_on_a_large_ring(g):
    """A stack with a modulus the lazy families cannot take (here: 2, below the domain's minimum of 3 -- the classification
    is what matters, not the arithmetic) on a ring of 2^17: the generic kernels behind the call serve it, call after call,
    in milliseconds; their result is whatever the public Barrett operators give, identical to path = generic."""
    import torch
    logn, batch = 17, 4
    cases, d_fwd, _ = rns_stack(g, 64, logn, [60, 60], O.X_N_plus)
    n = 1 << logn
    mods_h = [cases[0].prm.modulus, g.Modulus(value=2, bit=2, mu=16, bits=64)]  # (bit and mu given: no validation)
    x = np.concatenate([cases[0].P.splitmix(31 + p, 0, n, 2) for p in range(batch)])  # residues below every modulus
    cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus)
    mods = g.modulus_array_to_device(mods_h, 64)
    g.set_option("path", "generic")
    try:
        d = g.to_device(x)
        g.GPU_NTT_Inplace(d, d_fwd, mods, cfg, batch, 2)
        torch.cuda.synchronize()
        ref = g.to_host(d)
    finally:
        g.set_option("path", "default")
    for rep in range(4):
        d = g.to_device(x)
        ms = _timed_call(lambda: g.GPU_NTT_Inplace(d, d_fwd, mods, cfg, batch, 2))
        assert np.array_equal(g.to_host(d), ref), rep
        assert ms < 40.0, (rep, ms)
    # polynomial 0 uses the 60-bit prime: the oracle's result
    assert np.array_equal(ref[:n], cases[0].P.merge_ntt(x[:n], cases[0].oprm))
