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
        q = (1 << wbits) - 1
        while q % 2 == 0 or q < 3:
            q -= 1
        if wbits == 2:
            q = 3
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


@pytest.mark.parametrize("bits,logn,poly", [(64, 28, O.X_N_plus), (32, 25, O.X_N_minus)])
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
