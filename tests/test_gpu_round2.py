"""-m gpu parity tests of the round-2 additions: NTTPlan (prepared transforms), PerCoefficient layout
with mod_count > 1, GPU_PolyMul squaring (a == b), the public device class OPERATOR_GPU<T>,
GPU_Transpose past the 65535-slice limit, workspace release."""
import os

import numpy as np
import pytest

from gpu_utils import MergeCase, find_ntt_factors
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


@pytest.mark.parametrize("bits", [32, 64])
def test_plan_single_modulus(g, bits):
    """NTTPlan == GPU_NTT / GPU_INTT == oracle for every plan shape: one tile, two passes (big and small
    batch hints pick different tiles at 2^14), three passes; in place and out of place; signed I/O"""
    import torch
    for logn, batch, poly in ((3, 9, O.X_N_plus), (10, 5, O.X_N_minus), (12, 3, O.X_N_plus), (13, 6, O.X_N_minus),
                              (14, 300, O.X_N_plus), (14, 4, O.X_N_minus), (16, 5, O.X_N_plus),
                              (18, 2, O.X_N_minus), (21, 2, O.X_N_plus)):
        c = MergeCase(g, bits, logn, poly)
        x = c.random(batch, 6100 + logn)
        want = c.P.merge_ntt(x, c.oprm)
        fplan = g.NTTPlan(c.fwd_dev, c.prm.modulus, logn, poly, g.FORWARD, batch_hint=batch)
        iplan = g.NTTPlan(c.inv_dev, c.prm.modulus, logn, poly, g.INVERSE, mod_inverse=c.prm.n_inv,
                          batch_hint=batch)
        # tiny rings stay on the generic kernels (one launch is the whole job), like the drop-in calls
        assert fplan.fast_path == iplan.fast_path == (logn >= (5 if bits == 64 else 11))
        d = g.to_device(x)
        o = torch.zeros_like(d)
        fplan.execute(d, o, batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(o), want), ("fwd", bits, logn)
        iplan.execute(o, o, batch)  # in place
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(o), x), ("inv", bits, logn)
        # a plan runs any batch size, not only its hint
        fplan.execute(d, d, 1)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d)[:c.n], want[:c.n])
        fplan.close()
        iplan.close()
    # signed input / centred output (the Data64s / Data32s instantiations)
    c = MergeCase(g, bits, 13, O.X_N_plus)
    q = c.q
    xs = (c.random(3, 77).astype(np.int64) - q // 2).astype(np.int32 if bits == 32 else np.int64)
    xr = np.where(xs < 0, xs.astype(object) + q, xs.astype(object)).astype(c.P.T)
    fplan = g.NTTPlan(c.fwd_dev, c.prm.modulus, 13, O.X_N_plus, g.FORWARD)
    iplan = g.NTTPlan(c.inv_dev, c.prm.modulus, 13, O.X_N_plus, g.INVERSE, mod_inverse=c.prm.n_inv)
    d = g.to_device(xs)
    o = torch.zeros_like(d)
    fplan.execute(d, o, 3, io_signed=True)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), c.P.merge_ntt(xr, c.oprm))
    iplan.execute(o, o, 3, io_signed=True)
    torch.cuda.synchronize()
    back = g.to_host(o, signed=True)
    assert np.array_equal(np.where(back < 0, back.astype(object) + q, back.astype(object)).astype(c.P.T), xr)
    assert int(np.abs(back.astype(object)).max()) <= q // 2


def test_plan_rns_and_wide_moduli(g):
    """RNS plans: 6 primes at 2^16 (config 5's shape) on the fast kernels, and a stack containing a
    62-bit prime, which the plan classifies at construction (LIMIT = 4 kernels for the whole stack)"""
    import torch
    P = O.Port(64)
    logn, batch, mc = 16, 18, 6
    fl = _small_prime_factors(P, logn, mc)
    cases, fwd, inv, mods, ninv = _rns_setup(g, 64, logn, O.X_N_plus, fl)
    n = 1 << logn
    x = np.concatenate([cases[p % mc].P.splitmix(500 + p, 0, n, cases[p % mc].q) for p in range(batch)])
    want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm) for p in range(batch)])
    moduli = [c.prm.modulus for c in cases]
    ws = torch.zeros(g.NTTPlan.workspace_bytes(logn, mc, 64), dtype=torch.uint8, device="cuda:0")
    fplan = g.NTTPlan(fwd, moduli, logn, O.X_N_plus, g.FORWARD, batch_hint=batch, workspace=ws)
    iplan = g.NTTPlan(inv, moduli, logn, O.X_N_plus, g.INVERSE, mod_inverse=[c.prm.n_inv for c in cases],
                      batch_hint=batch)
    assert fplan.fast_path and iplan.fast_path
    d = g.to_device(x)
    fplan.execute(d, d, batch)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(d), want)
    iplan.execute(d, d, batch)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(d), x)
    # a 62-bit modulus in the stack -> the whole stack runs on the LIMIT = 4 kernels, same results
    logn = 13
    fl = [find_ntt_factors(58, logn), find_ntt_factors(62, logn), find_ntt_factors(60, logn)]
    cases, fwd, inv, mods, ninv = _rns_setup(g, 64, logn, O.X_N_minus, fl)
    n = 1 << logn
    x = np.concatenate([cases[p % 3].P.splitmix(600 + p, 0, n, cases[p % 3].q) for p in range(7)])
    want = np.concatenate([cases[p % 3].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % 3].oprm) for p in range(7)])
    moduli = [c.prm.modulus for c in cases]
    fplan = g.NTTPlan(fwd, moduli, logn, O.X_N_minus, g.FORWARD)
    iplan = g.NTTPlan(inv, moduli, logn, O.X_N_minus, g.INVERSE, mod_inverse=[c.prm.n_inv for c in cases])
    assert fplan.fast_path and iplan.fast_path
    d = g.to_device(x)
    o = torch.zeros_like(d)
    fplan.execute(d, o, 7)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), want)
    iplan.execute(o, o, 7)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), x)


def test_plan_graph_capture_without_warm_up(g):
    """execute() allocates nothing and never synchronises: the very first execution of a fresh plan on a
    fresh stream can be captured into a hipGraph and replayed (the drop-in calls need one eager call
    first because their scratch buffer is created lazily)"""
    import torch
    c = MergeCase(g, 64, 16, O.X_N_minus)
    batch = 8
    x = c.random(batch, 4711)
    want = c.P.merge_ntt(x, c.oprm)
    s = torch.cuda.Stream()
    fplan = g.NTTPlan(c.fwd_dev, c.prm.modulus, 16, O.X_N_minus, g.FORWARD, batch_hint=batch, stream=s)
    iplan = g.NTTPlan(c.inv_dev, c.prm.modulus, 16, O.X_N_minus, g.INVERSE, mod_inverse=c.prm.n_inv,
                      batch_hint=batch, stream=s)
    s.synchronize()
    d = g.to_device(x)
    o = torch.zeros_like(d)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        fplan.execute(d, o, batch, stream=s)
    gr.replay()
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), want)
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2, stream=s):
        iplan.execute(o, o, batch, stream=s)
    gr2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), x)


def test_plan_argument_errors(g):
    c = MergeCase(g, 64, 8, O.X_N_minus)
    with pytest.raises(ValueError, match="Invalid n_power range!"):
        g.NTTPlan(c.fwd_dev, c.prm.modulus, 0)
    with pytest.raises(ValueError, match="Invalid mod_inverse!"):
        g.NTTPlan(c.inv_dev, c.prm.modulus, 8, O.X_N_minus, g.INVERSE)
    with pytest.raises(ValueError):
        g.NTTPlan.workspace_bytes(29, 1, 64)


@pytest.mark.parametrize("bits", [32, 64])
def test_percoefficient_rns(g, bits):
    """PerCoefficient layout with mod_count > 1 (reference ForwardCoreTranspose / InverseCoreTranspose RNS
    overloads, ntt.cu:1693-1835, 1957-2074): column c of the N x batch matrix is a polynomial of modulus
    c % mod_count, transformed with that modulus' table slot; both the tile-pass path (wide matrices) and
    the small-matrix kernel"""
    import torch
    P = O.Port(bits)
    for logn, w, mc, poly in ((9, 1024, 3, O.X_N_plus), (7, 256, 2, O.X_N_minus), (5, 16, 3, O.X_N_plus),
                              (9, 8, 2, O.X_N_minus), (8, 64, 3, O.X_N_plus)):
        fl = _small_prime_factors(P, logn, mc)
        cases, fwd, inv, mods, ninv = _rns_setup(g, bits, logn, poly, fl)
        n = 1 << logn
        cols = np.stack([cases[p % mc].P.splitmix(800 + p, 0, n, cases[p % mc].q) for p in range(w)])  # w x n
        mat = np.ascontiguousarray(cols.T)
        want_f = np.stack([cases[p % mc].P.merge_ntt(cols[p], cases[p % mc].oprm) for p in range(w)]).T
        want_i = np.stack([cases[p % mc].P.merge_ntt(cols[p], cases[p % mc].oprm, inverse=True) for p in range(w)]).T
        cfg = g.ntt_rns_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=poly)
        d = g.to_device(mat.reshape(-1))
        o = torch.zeros_like(d)
        g.GPU_NTT(d, o, fwd, mods, cfg, w, mc)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(o).reshape(n, w), want_f), ("fwd", bits, logn, w, mc)
        icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, ntt_layout=g.PerCoefficient,
                                       reduction_poly=poly, mod_inverse=ninv)
        g.GPU_INTT_Inplace(d, inv, mods, icfg, w, mc)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d).reshape(n, w), want_i), ("inv", bits, logn, w, mc)


@pytest.mark.parametrize("bits", [32, 64])
def test_polymul_squaring(g, bits):
    """device_a == device_b: the square, not INTT(NTT(NTT(a)) . NTT(a)) (one transform, pointwise square)"""
    import torch
    for logn, batch, poly in ((6, 4, O.X_N_plus), (12, 3, O.X_N_minus), (14, 2, O.X_N_plus)):
        c = MergeCase(g, bits, logn, poly)
        a = c.random(batch, 1200 + logn)
        n = c.n
        if logn <= 9:
            want = np.concatenate([c.P.schoolbook(a[i * n:(i + 1) * n], a[i * n:(i + 1) * n], poly, c.oprm["mod"])
                                   for i in range(batch)])
        else:
            fa = c.P.merge_ntt(a, c.oprm)
            want = c.P.merge_ntt(c.P.pointwise(fa, fa, c.oprm["mod"]), c.oprm, inverse=True)
        da = g.to_device(a)
        out = torch.zeros_like(da)
        g.GPU_PolyMul(da, da, out, c.fwd_dev, c.inv_dev, c.prm.modulus, c.cfg(inverse=True), batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(out), want), (bits, logn, poly)
        da = g.to_device(a)
        g.GPU_PolyMul(da, da, da, c.fwd_dev, c.inv_dev, c.prm.modulus, c.cfg(inverse=True), batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(da), want)
    # RNS form
    logn, batch = 13, 6
    fl = [find_ntt_factors(58, logn), find_ntt_factors(60, logn), find_ntt_factors(59, logn)]
    if bits == 64:
        cases, fwd, inv, mods, ninv = _rns_setup(g, 64, logn, O.X_N_plus, fl)
        n = 1 << logn
        a = np.concatenate([cases[p % 3].P.splitmix(1300 + p, 0, n, cases[p % 3].q) for p in range(batch)])
        want = []
        for p in range(batch):
            c = cases[p % 3]
            fa = c.P.merge_ntt(a[p * n:(p + 1) * n], c.oprm)
            want.append(c.P.merge_ntt(c.P.pointwise(fa, fa, c.oprm["mod"]), c.oprm, inverse=True))
        da = g.to_device(a)
        out = torch.zeros_like(da)
        cfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=O.X_N_plus, mod_inverse=ninv)
        g.GPU_PolyMul(da, da, out, fwd, inv, mods, cfg, batch, 3)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(out), np.concatenate(want))


@pytest.mark.parametrize("bits", [32, 64])
def test_operator_gpu_device_class(g, bits):
    """the public OPERATOR_GPU<T> (include/gpuntt/common/modular_arith.cuh; reference modular_arith.cuh:
    174-454) on the device against exact integer arithmetic: add, sub, mult, reduce, reduce(signed),
    centered_reduction -- every modulus width of the documented domain, edge operands included"""
    rng = np.random.default_rng(99)
    T = np.uint32 if bits == 32 else np.uint64
    widths = (2, 3, 14, 20, 29, 30) if bits == 32 else (2, 3, 20, 31, 32, 33, 50, 59, 60, 61, 62)
    for wbits in widths:
        # widest odd value of that width; from 50 bits on, one that the reference's double log2 does not round up to
        # the next width (2^61 - 1 would get bit = 62 and a mu past the word: outside the reference's own domain)
        q = (1 << wbits) - 1 if wbits < 50 else (1 << wbits) - (1 << (wbits - 30)) - 1
        if wbits == 2:
            q = 3
        assert q % 2 == 1 and q.bit_length() == wbits
        m = g.Modulus(q, bits=bits)
        cnt = 4096
        a = rng.integers(0, q, size=cnt, dtype=np.uint64).astype(T)
        b = rng.integers(0, q, size=cnt, dtype=np.uint64).astype(T)
        a[:4] = [0, q - 1, q - 1, 1]
        b[:4] = [0, q - 1, 1, q - 1]
        da, db = g.to_device(a), g.to_device(b)
        ao, bo = a.astype(object), b.astype(object)
        want = {0: (ao + bo) % q, 1: (ao - bo) % q, 2: (ao * bo) % q, 3: ao % q}
        for op, w in want.items():
            got = g.to_host(g.operator_gpu(op, da, db, m))
            assert np.array_equal(got.astype(object), w), (bits, wbits, op)
        # reduce(signed): inputs in (-q, q)
        s = (rng.integers(0, 2 * q - 1, size=cnt, dtype=np.uint64).astype(object) - (q - 1))
        sd = g.to_device(np.array(s, dtype=np.int32 if bits == 32 else np.int64))
        got = g.to_host(g.operator_gpu(4, sd, None, m))
        assert np.array_equal(got.astype(object), np.array([int(v) % q for v in s], dtype=object)), (bits, wbits)
        # centered_reduction: [0, q) -> (-q/2, q/2]
        got = g.to_host(g.operator_gpu(5, da, None, m), signed=True)
        wantc = np.array([int(v) - q if int(v) > (q >> 1) else int(v) for v in a], dtype=object)
        assert np.array_equal(got.astype(object), wantc), (bits, wbits)


def test_transpose_more_than_65535_slices(g):
    """GPU_Transpose puts the batch in gridDim.z (as the reference does, ntt_4step.cu:36-66); batches past
    the 65535 limit are launched in slices instead of failing"""
    import torch
    row, col, batch = 4, 8, 70000
    x = np.arange(batch * row * col, dtype=np.uint32)
    d = g.to_device(x)
    o = torch.zeros_like(d)
    g.GPU_Transpose(d, o, row, col, 5, batch)
    torch.cuda.synchronize()
    want = x.reshape(batch, row, col).transpose(0, 2, 1).reshape(-1)
    assert np.array_equal(g.to_host(o), want)


def test_release_workspaces(g):
    """the drop-in calls keep a twiddle scratch per stream; releasing it is safe and the next call
    simply allocates again"""
    c = MergeCase(g, 64, 13, O.X_N_minus)
    x = c.random(2, 5)
    want = c.P.merge_ntt(x, c.oprm)
    assert np.array_equal(c.gpu_forward(x), want)
    g.release_workspaces()
    assert np.array_equal(c.gpu_forward(x), want)


def test_fast_kernels_above_2_24(g):
    """rings 2^25 / 2^26 run on the fast (lazy-residue) kernels, not the generic fallback of round 1:
    GPUNTT_PATH=fast-strict makes any call the fast kernels cannot take throw (reference: the grid-swapped
    ForwardCore_ / InverseCore_ rows of ntt.cuh:669-697)"""
    from test_gpu_merge import _run_in_subprocess
    code = """
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.environ["PYTHONPATH"]), ""))
from conftest import load_pkg
from gpu_utils import MergeCase
from oracle import oracle as O
g = load_pkg(); g.load_library()
for bits, logn, poly in ((64, 25, O.X_N_minus), (32, 25, O.X_N_plus), (64, 26, O.X_N_plus)):
    c = MergeCase(g, bits, logn, poly)
    x = c.random(2, 2500 + logn)
    y = c.gpu_forward(x, inplace=True)
    n = c.n
    assert np.array_equal(y[n:], c.P.merge_ntt(x[n:], c.oprm)), (bits, logn)
    assert np.array_equal(c.gpu_inverse(y, inplace=True), x), (bits, logn)
# the hook itself: a 62-bit modulus has no fast path -> the single-modulus entry point falls back silently
# (host-side check), an RNS stack of tiny rings (tile would mix moduli) must throw under fast-strict
print("OK")
"""
    assert "OK" in _run_in_subprocess(code, {"GPUNTT_PATH": "fast-strict"})


@pytest.mark.parametrize("bits,logn,poly", [(64, 28, O.X_N_plus), (64, 27, O.X_N_minus), (32, 25, O.X_N_minus)])
def test_largest_rings_sparse_known_answer(g, bits, logn, poly):
    """the top of the documented range (n_power <= 28, reference ntt.cu:2088-2091) without a CPU transform of
    that size: a polynomial with a handful of non-zero coefficients has the closed-form spectrum
    out[bitrev(k)] = sum_j a_j w^(j k) (X^N-1) / sum_j a_j psi^((2k+1) j) (X^N+1), checked
    at sampled k with Python integers; plus the exact forward -> inverse round trip of random data."""
    import torch
    prm = g.NTTParameters(logn, poly, bits)
    q, n = prm.modulus.value, 1 << logn
    fwd = g.to_device(prm.forward_table_device_order)
    rng = np.random.default_rng(logn)
    js = [0, 1, 5, n // 3, n - 1]
    av = [int(v) for v in rng.integers(1, q, size=len(js), dtype=np.uint64)]
    x = np.zeros(n, dtype=g.np_dtype(bits))
    for j, a in zip(js, av):
        x[j] = a
    d = g.to_device(x)
    cfg = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=poly)
    g.GPU_NTT_Inplace(d, fwd, prm.modulus, cfg, 1)
    torch.cuda.synchronize()
    y = g.to_host(d)

    def brev(v):
        return int(format(v, "0%db" % logn)[::-1], 2)
    for k in [0, 1, 2, 3, n // 2, n // 2 + 1, n - 1] + [int(v) for v in rng.integers(0, n, size=24)]:
        if poly == O.X_N_minus:
            want = sum(a * pow(prm.omega, (j * k) % n, q) for j, a in zip(js, av)) % q
        else:
            want = sum(a * pow(prm.psi, ((2 * k + 1) * j) % (2 * n), q) for j, a in zip(js, av)) % q
        assert int(y[brev(k)]) == want, (bits, logn, k)
    del y
    # round trip of random data, in place
    inv = g.to_device(prm.inverse_table_device_order)
    r = (rng.integers(0, q, size=n, dtype=np.uint64)).astype(g.np_dtype(bits))
    d = g.to_device(r)
    g.GPU_NTT_Inplace(d, fwd, prm.modulus, cfg, 1)
    icfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=prm.n_inv)
    g.GPU_INTT_Inplace(d, inv, prm.modulus, icfg, 1)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(d), r)


@pytest.mark.parametrize("logn", [27, 28])
def test_largest_rings_reference_digest(g, golden_dir, logn):
    """One whole forward transform at n_power 27 and 28 against the digest of NTTCPU<Data64>::ntt from the reference
    build (tests/golden/digests_large.json, tools/make_golden_large.py; the input is the seeded splitmix stream)."""
    import hashlib
    import json
    import torch
    rec = next(r for r in json.load(open(os.path.join(golden_dir, "digests_large.json")))["cases"] if r["logn"] == logn)
    poly, bits, n = rec["poly"], rec["bits"], 1 << logn
    prm = g.NTTParameters(logn, poly, bits)
    assert (prm.modulus.value, prm.omega, prm.psi) == (rec["q"], rec["omega"], rec["psi"])
    x = O.Port(bits).splitmix(rec["seed"], 0, n, rec["q"])
    assert hashlib.sha256(x.tobytes()).hexdigest() == rec["sha256_input"]
    d = g.to_device(x)
    del x
    fwd = g.to_device(prm.forward_table_device_order)
    g.GPU_NTT_Inplace(d, fwd, prm.modulus, g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=poly), 1)
    torch.cuda.synchronize()
    y = g.to_host(d)
    assert [int(v) for v in y[:4]] == rec["first_words"] and [int(v) for v in y[-4:]] == rec["last_words"]
    assert hashlib.sha256(y.tobytes()).hexdigest() == rec["sha256_forward"]


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


# ------------------------------------------------------------------ FourStepPlan (prepared 4-step transforms)
@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_plan_vs_oracle(g, bits):
    """FourStepPlan.execute == the oracle's 4-step, every n1 x n2 shape class, both layouts (reference
    n2 x n1 -> n1 x n2 and natural order), both directions, batch sizes other than the hint, plan reused
    across calls, preparation in a caller-owned workspace; no scratch of the drop-in calls is touched."""
    import torch
    P = O.Port(bits)
    for logn in (12, 13, 15, 16, 17, 19, 20):
        p4 = g.NTTParameters4Step(logn, bits)
        oprm = P.fourstep_params(logn)
        tf = [g.to_device(t) for t in p4.tables["fwd"]]
        ti = [g.to_device(t) for t in p4.tables["inv"]]
        cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
        ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
        ws = torch.zeros(g.FourStepPlan.workspace_bytes(logn, bits), dtype=torch.uint8, device="cuda")
        g.release_workspaces()
        hint = 8 if logn == 19 else 2
        pf = g.FourStepPlan(*tf, p4.modulus, cf, natural_order=False, batch_hint=hint, workspace=ws)
        pi = g.FourStepPlan(*ti, p4.modulus, ci, natural_order=False, batch_hint=hint)
        nf = g.FourStepPlan(*tf, p4.modulus, cf, natural_order=True, batch_hint=hint)
        ni = g.FourStepPlan(*ti, p4.modulus, ci, natural_order=True, batch_hint=hint)
        assert pf.fast_path and pi.fast_path and nf.fast_path and ni.fast_path
        for batch in ((hint, 1, 3) if logn <= 17 else (hint,)):
            x = P.splitmix(900 + logn + batch, 0, batch * p4.n, p4.modulus.value)
            want = np.concatenate([P.fourstep_ntt(x[i * p4.n:(i + 1) * p4.n], oprm) for i in range(batch)])
            # reference layout: transposed input, transposed output
            xt = x.reshape(batch, p4.n1, p4.n2).transpose(0, 2, 1).reshape(-1).copy()
            d_in = g.to_device(xt)
            d_out = torch.zeros_like(d_in)
            pf.execute(d_in, d_out, batch)
            torch.cuda.synchronize()
            got = g.to_host(d_out).reshape(batch, p4.n1, p4.n2).transpose(0, 2, 1).reshape(-1)
            assert np.array_equal(got, want), ("plan fwd", bits, logn, batch)
            xin = np.concatenate([P.fourstep_intt_first_transpose(want[i * p4.n:(i + 1) * p4.n], oprm)
                                  for i in range(batch)])
            d_in = g.to_device(xin)
            d_out = torch.zeros_like(d_in)
            pi.execute(d_in, d_out, batch)
            torch.cuda.synchronize()
            back = g.to_host(d_out).reshape(batch, p4.n1, p4.n2).transpose(0, 2, 1).reshape(-1)
            assert np.array_equal(back, x), ("plan inv", bits, logn, batch)
            # natural order
            d_in = g.to_device(x)
            d_out = torch.zeros_like(d_in)
            nf.execute(d_in, d_out, batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d_out), want), ("plan natural fwd", bits, logn, batch)
            d_back = torch.zeros_like(d_out)
            ni.execute(d_out, d_back, batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d_back), x), ("plan natural inv", bits, logn, batch)
        for p in (pf, pi, nf, ni):
            p.close()


def test_fourstep_plan_golden_2_24(g):
    """C3's ring through a plan: the forward 2^24 result equals the reference build's digest
    (tests/golden/digests.json, three-call pipeline) and the natural-order plan's result"""
    import json
    import torch
    from gpu_utils import sha
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "digests.json")))
    rec = [r for r in gold["fourstep"] if r["logn"] == 24 and r["bits"] == 64][0]
    P = O.Port(64)
    p4 = g.NTTParameters4Step(24, 64)
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    cf = g.ntt4step_configuration(n_power=24, ntt_type=g.FORWARD)
    x = P.splitmix(rec["seed"], 0, p4.n, rec["q"])
    assert sha(x) == rec["sha_in"]
    nf = g.FourStepPlan(*tf, p4.modulus, cf, natural_order=True, batch_hint=1)
    pf = g.FourStepPlan(*tf, p4.modulus, cf, natural_order=False, batch_hint=1)
    d_in = g.to_device(x)
    d_t = torch.zeros_like(d_in)
    d_out = torch.zeros_like(d_in)
    g.GPU_Transpose(d_in, d_t, p4.n1, p4.n2, 24, 1)
    torch.cuda.synchronize()
    pf.execute(d_t, d_out, 1)
    torch.cuda.synchronize()
    g.GPU_Transpose(d_out, d_t, p4.n1, p4.n2, 24, 1)
    torch.cuda.synchronize()
    assert sha(g.to_host(d_t)) == rec["sha_fwd"]
    nf.execute(d_in, d_out, 1)
    torch.cuda.synchronize()
    assert sha(g.to_host(d_out)) == rec["sha_fwd"]


def test_fourstep_plan_slow_modulus_and_errors(g):
    """a 62-bit modulus has no fast 4-step kernels: the plan reports it and execute() runs the generic
    path with the same result as GPU_4STEP_NTT; argument errors throw like the drop-in calls"""
    import torch
    P = O.Port(64)
    logn = 13
    p4 = g.NTTParameters4Step(logn, 64)
    with pytest.raises(ValueError):
        g.FourStepPlan.workspace_bytes(11, 64)
    with pytest.raises(ValueError):
        g.FourStepPlan.workspace_bytes(25, 64)
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
    pf = g.FourStepPlan(*tf, p4.modulus, cf)
    d = g.to_device(np.zeros(p4.n, dtype=np.uint64))
    with pytest.raises(ValueError):
        pf.execute(d, d, 1)  # in place is not supported by the 4-step
    pf.execute(d, torch.zeros_like(d), 0)  # empty batch: no-op
    # generic path under option path = generic: plan creation sees it and falls back
    g.set_option("path", "generic")
    try:
        ps = g.FourStepPlan(*tf, p4.modulus, cf)
        assert not ps.fast_path
        x = P.splitmix(77, 0, 2 * p4.n, p4.modulus.value)
        d_in = g.to_device(x)
        d_a = torch.zeros_like(d_in)
        d_b = torch.zeros_like(d_in)
        ps.execute(d_in, d_a, 2)
        g.GPU_4STEP_NTT(d_in, d_b, *tf, p4.modulus, cf, 2)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_a), g.to_host(d_b))
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
    pf.execute(d, torch.zeros_like(d), 1)  # the fast plan made before still runs its prepared path
    torch.cuda.synchronize()


# ------------------------------------------------------------------ seeded random sweep of the Merge entry points
def test_random_merge_cases_vs_oracle(g):
    """70 seeded random (word size, modulus width, ring, batch, reduction polynomial, call form) cases: forward
    result and raw-input inverse result equal the oracle's bit for bit.  Moduli are searched primes of the drawn
    width (not the reference's pools), so odd widths hit every lazy range (16q / 8q / 4q) and the Barrett kernels."""
    import torch
    rng = np.random.default_rng(20260929)
    for case_no in range(70):
        bits = int(rng.choice([32, 64]))
        logn = int(rng.choice([1, 2, 3, 5, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18]))
        qbits = int(rng.choice([20, 27, 29] if bits == 32 else [24, 33, 47, 55, 59, 60, 61, 62]))
        if qbits < logn + 12:  # enough candidates k * 2^(logn+1) + 1 of that width
            qbits = logn + 12 if bits == 64 else 29
        poly = O.X_N_plus if rng.integers(2) else O.X_N_minus
        n = 1 << logn
        batch = int(rng.integers(1, max(2, min(70, (1 << 19) // n) + 1)))
        form = int(rng.integers(3))  # 0 out of place, 1 in place, 2 plan
        # (62-bit primes a few hundred below 2^62 get bit = 63 from the reference's double log2: outside its domain)
        fac = find_ntt_factors(qbits, logn, skip=400 if (qbits == 62 and logn < 10) else int(rng.integers(3)))
        c = MergeCase(g, bits, logn, poly, fac)
        tag = (case_no, bits, qbits, logn, batch, poly, form)
        x = c.random(batch, 7000 + case_no)
        want = c.P.merge_ntt(x, c.oprm)
        y = c.random(batch, 8000 + case_no)  # raw input of the inverse (not a forward result)
        want_inv = c.P.merge_ntt(y, c.oprm, inverse=True)
        if form == 2:
            fp = g.NTTPlan(c.fwd_dev, c.prm.modulus, logn, poly, g.FORWARD, batch_hint=int(rng.integers(1, 2000)))
            ip = g.NTTPlan(c.inv_dev, c.prm.modulus, logn, poly, g.INVERSE, mod_inverse=c.prm.n_inv,
                           batch_hint=batch)
            d = g.to_device(x)
            o = torch.zeros_like(d)
            fp.execute(d, o, batch)
            e = g.to_device(y)
            ip.execute(e, e, batch)
            torch.cuda.synchronize()
            got, got_inv = g.to_host(o), g.to_host(e)
            fp.close()
            ip.close()
        else:
            got = c.gpu_forward(x, inplace=(form == 1))
            got_inv = c.gpu_inverse(y, inplace=(form == 1))
        assert np.array_equal(got, want), ("forward",) + tag
        assert np.array_equal(got_inv, want_inv), ("inverse",) + tag


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


# ------------------------------------------------------------------ tables built on the device
@pytest.mark.parametrize("bits", [32, 64])
def test_device_generated_tables_equal_host_tables(g, bits):
    """GPU_GeneratePowerTable / GPU_Generate4StepW write the very words NTTParameters<T> / NTTParameters4Step<T>
    build on the host (pinned to the reference build through tests/golden): Merge forward and inverse device-order
    tables for both reduction polynomials, the 4-step n1 / n2 tables and both W matrices for every n1 x n2 shape."""
    import torch
    dt = torch.int32 if bits == 32 else torch.int64
    for logn in (1, 2, 5, 11, 12, 16, 20):
        for poly in (O.X_N_plus, O.X_N_minus):
            prm = g.NTTParameters(logn, poly, bits)
            q = prm.modulus.value
            root = prm.psi if poly == O.X_N_plus else prm.omega
            lg = logn if poly == O.X_N_plus else logn - 1
            assert prm.root_of_unity_size == 1 << lg
            for base, host in ((root, prm.forward_table_device_order), (pow(root, -1, q), prm.inverse_table_device_order)):
                d = torch.zeros(1 << lg, dtype=dt, device="cuda")
                g.GPU_GeneratePowerTable(d, base, prm.modulus, lg, True)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(d).view(g.np_dtype(bits)), host), (bits, logn, poly)
    for logn in range(12, 23):
        p4 = g.NTTParameters4Step(logn, bits)
        q = p4.modulus.value
        root = p4.omega  # cyclic: root_of_unity = omega
        for tag, r, kind in (("fwd", root, g.FORWARD), ("inv", pow(root, -1, q), g.INVERSE)):
            t1, t2, w = p4.tables[tag]
            d = torch.zeros(p4.n, dtype=dt, device="cuda")
            g.GPU_Generate4StepW(d, r, p4.modulus, logn, kind)
            d1 = torch.zeros(p4.n1 >> 1, dtype=dt, device="cuda")
            d2 = torch.zeros(p4.n2 >> 1, dtype=dt, device="cuda")
            g.GPU_GeneratePowerTable(d1, pow(r, p4.n // p4.n1, q), p4.modulus, int(np.log2(p4.n1)) - 1, True)
            g.GPU_GeneratePowerTable(d2, pow(r, p4.n // p4.n2, q), p4.modulus, int(np.log2(p4.n2)) - 1, True)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d).view(g.np_dtype(bits)), w), (bits, logn, tag, "W")
            assert np.array_equal(g.to_host(d1).view(g.np_dtype(bits)), t1), (bits, logn, tag, "n1")
            assert np.array_equal(g.to_host(d2).view(g.np_dtype(bits)), t2), (bits, logn, tag, "n2")
    # natural (not bit-reversed) order and argument errors
    prm = g.NTTParameters(8, O.X_N_minus, bits)
    d = torch.zeros(64, dtype=dt, device="cuda")
    g.GPU_GeneratePowerTable(d, prm.omega, prm.modulus, 6, False)
    torch.cuda.synchronize()
    assert [int(v) for v in g.to_host(d).view(g.np_dtype(bits))] == [pow(prm.omega, k, prm.modulus.value) for k in range(64)]
    with pytest.raises(ValueError):
        g.GPU_GeneratePowerTable(d, prm.modulus.value, prm.modulus, 6, True)  # base not reduced
    with pytest.raises(ValueError):
        g.GPU_Generate4StepW(d, prm.omega, prm.modulus, 11, g.FORWARD)


def test_fourstep_2_24_from_device_generated_tables(g):
    """C3's ring with no host-built table at all: W and the n1 / n2 tables generated on the device (digest of W equals
    the reference build's), a FourStepPlan prepared from them, forward result equal to the reference digest"""
    import json
    import torch
    from gpu_utils import sha
    rec = [r for r in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "digests.json")))["fourstep"]
           if r["logn"] == 24 and r["bits"] == 64][0]
    q, root = rec["q"], rec["omega"]
    m = g.Modulus(q, bits=64)
    n1, n2, n = 256, 65536, 1 << 24
    w = torch.zeros(n, dtype=torch.int64, device="cuda")
    t1 = torch.zeros(n1 >> 1, dtype=torch.int64, device="cuda")
    t2 = torch.zeros(n2 >> 1, dtype=torch.int64, device="cuda")
    g.GPU_Generate4StepW(w, root, m, 24, g.FORWARD)
    g.GPU_GeneratePowerTable(t1, pow(root, n // n1, q), m, 7, True)
    g.GPU_GeneratePowerTable(t2, pow(root, n // n2, q), m, 15, True)
    torch.cuda.synchronize()
    assert sha(g.to_host(w).view(np.uint64)) == rec["sha_W_fwd"]
    assert sha(g.to_host(t2).view(np.uint64)) == rec["sha_n2_fwd_gpu"]
    plan = g.FourStepPlan(t1, t2, w, m, g.ntt4step_configuration(n_power=24, ntt_type=g.FORWARD), natural_order=True)
    torch.cuda.synchronize()
    del w  # the prepared pairs are all the plan needs
    x = O.Port(64).splitmix(rec["seed"], 0, n, q)
    d_in = g.to_device(x)
    d_out = torch.zeros_like(d_in)
    plan.execute(d_in, d_out, 1)
    torch.cuda.synchronize()
    assert sha(g.to_host(d_out)) == rec["sha_fwd"]


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


def test_scratch_exhaustion_falls_back_to_the_generic_kernels(g):
    """a drop-in call whose per-(device, stream) twiddle scratch cannot be allocated (several streams, nearly full HBM)
    runs on the generic kernels, which need none, instead of failing; option no_scratch = 1 simulates the
    failed hipMalloc.  Merge single / RNS, 4-step both directions; under fast-strict the same situation throws."""
    import torch
    g.set_option("no_scratch", 1)
    try:
        for bits, logn, batch in ((64, 13, 5), (32, 14, 3), (64, 16, 2)):
            c = MergeCase(g, bits, logn, O.X_N_plus)
            x = c.random(batch, 8800 + logn)
            want = c.P.merge_ntt(x, c.oprm)
            assert np.array_equal(c.gpu_forward(x), want)
            assert np.array_equal(c.gpu_inverse(want, inplace=True), x)
        P = O.Port(64)
        fl = _small_prime_factors(P, 12, 3)
        cases, fwd, inv, mods, ninv = _rns_setup(g, 64, 12, O.X_N_minus, fl)
        x = np.concatenate([cases[p % 3].random(1, 8900 + p) for p in range(6)])
        d = g.to_device(x)
        g.GPU_NTT_Inplace(d, fwd, mods, g.ntt_rns_configuration(n_power=12, reduction_poly=O.X_N_minus), 6, 3)
        torch.cuda.synchronize()
        y = g.to_host(d)
        for p in range(6):
            assert np.array_equal(y[p * 4096:(p + 1) * 4096], P.merge_ntt(x[p * 4096:(p + 1) * 4096], cases[p % 3].oprm))
        from test_gpu_4step import run_fourstep
        p4 = g.NTTParameters4Step(13, 64)
        oprm = P.fourstep_params(13)
        x4 = P.splitmix(8950, 0, 2 * p4.n, p4.modulus.value)
        want4 = P.fourstep_ntt(x4, oprm)
        assert np.array_equal(run_fourstep(g, p4, x4, 2, inverse=False), want4)
        assert np.array_equal(run_fourstep(g, p4, P.fourstep_intt_first_transpose(want4, oprm), 2, inverse=True), x4)
        g.set_option("path", "fast-strict")
        c = MergeCase(g, 64, 13, O.X_N_plus)
        with pytest.raises((ValueError, g.GpuNttError)):
            c.gpu_forward(c.random(1, 1))
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
        g.set_option("no_scratch", 0)


@pytest.mark.parametrize("bits", [32, 64])
def test_percoefficient_on_the_fast_kernels(g, bits):
    """PerCoefficient layout (reference ForwardCoreTranspose / InverseCoreTranspose, ntt.cu:1554-2074), single modulus:
    since round 3 the strided lazy-residue kernels run it from the prepared table of the N-ring (4-7x the Barrett
    kernels: 2^9 x 2^17 u64 1.59 -> 0.37 ms).  Under path = fast-strict a call that fell back would throw.  Column x of
    the output == NTTCPU::ntt(column x of the input), both polynomials, both directions, one and two strided passes."""
    import torch
    g.set_option("path", "fast-strict")
    try:
        for logn, w, poly in ((9, 4096, O.X_N_plus), (9, 512, O.X_N_minus), (8, 8192, O.X_N_plus), (5, 65536, O.X_N_minus),
                              (6, 1024, O.X_N_plus), (3, 16384, O.X_N_minus)):
            c = MergeCase(g, bits, logn, poly)
            n = c.n
            cols = c.random(w, 9900 + logn + w).reshape(w, n)
            mat = np.ascontiguousarray(cols.T)
            want_f = np.ascontiguousarray(c.P.merge_ntt(cols.reshape(-1), c.oprm).reshape(w, n).T)
            cfg = g.ntt_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=poly)
            d = g.to_device(mat.reshape(-1))
            o = torch.zeros_like(d)
            g.GPU_NTT(d, o, c.fwd_dev, c.prm.modulus, cfg, w)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o).reshape(n, w), want_f), ("fwd", bits, logn, w)
            icfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE, ntt_layout=g.PerCoefficient,
                                       reduction_poly=poly, mod_inverse=c.prm.n_inv)
            g.GPU_INTT_Inplace(o, c.inv_dev, c.prm.modulus, icfg, w)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o).reshape(n, w), mat), ("inv", bits, logn, w)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
