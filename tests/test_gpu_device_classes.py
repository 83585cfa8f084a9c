"""-m gpu parity tests: Public device-side pieces and small entry points: OPERATOR_GPU<T>, CooleyTukeyUnit / GentlemanSandeUnit (reference ntt.cuh:69-92), GPU_Transpose beyond 65535 slices, GPU_PolyMul squaring."""
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

# ---------------------------------------------------------------- public device butterflies
@pytest.mark.parametrize("bits", [64, 32])
def test_public_device_butterfly_units(g, bits):
    """CooleyTukeyUnit / GentlemanSandeUnit (reference src/include/gpuntt/ntt_merge/ntt.cuh:69-92) are part of the public
    header: a caller kernel built on them compiles and computes U' = U + V*w, V' = U - V*w / U' = U + V, V' = (U - V)*w
    (mod q) -- checked on the device against Python integers, pool prime and a 62- / 30-bit prime."""
    import torch
    rng = np.random.default_rng(bits)
    for q in ((576460756061519873, find_ntt_factors(62, 10)[0]) if bits == 64 else (469762049, find_ntt_factors(30, 10)[0])):
        m = g.Modulus(q, bits=bits)
        cnt = 5000
        U, V, W = (rng.integers(0, q, size=cnt, dtype=np.uint64) for _ in range(3))
        U[:3], V[:3], W[:3] = (0, q - 1, q - 1), (q - 1, q - 1, 0), (q - 1, 1, q - 1)
        dt = g.np_dtype(bits)
        for gs in (False, True):
            du, dv, dw = (g.to_device(a.astype(dt)) for a in (U, V, W))
            g.butterfly_unit(du, dv, dw, m, gentleman_sande=gs)
            torch.cuda.synchronize()
            u, v, w = ([int(t) for t in a] for a in (U, V, W))
            if gs:
                wu = [(a + b) % q for a, b in zip(u, v)]
                wv = [((a - b) % q) * c % q for a, b, c in zip(u, v, w)]
            else:
                wu = [(a + b * c) % q for a, b, c in zip(u, v, w)]
                wv = [(a - b * c) % q for a, b, c in zip(u, v, w)]
            assert [int(t) for t in g.to_host(du)] == wu, (bits, q, gs)
            assert [int(t) for t in g.to_host(dv)] == wv, (bits, q, gs)

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
