"""-m gpu parity tests: The PerCoefficient layout (reference ForwardCoreTranspose / InverseCoreTranspose, src/lib/ntt_merge/ntt.cu:1554-2074): single modulus and RNS stacks, generic and fast kernels, per-lane moduli, tiny rings."""
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

@pytest.mark.parametrize("bits", [32, 64])
def test_percoefficient_rns_on_the_lazy_kernels_with_per_lane_moduli(g, bits):
    """PerCoefficient layout with mod_count > 1 (reference ForwardCoreTranspose / InverseCoreTranspose RNS overloads,
    src/lib/ntt_merge/ntt.cu:1693-1835, 1957-2074): column c is a polynomial of modulus c % mod_count.  The lanes of a
    wave hold different columns, so the strided lazy kernels run with PER-LANE moduli (kern::merge_pass_lazy_vq; 64-bit
    words: the 4 q range, 61- / 62-bit primes included).  path = fast-strict: no generic kernels behind the call.  Every
    column against NTTCPU, both directions, one- and two-pass shapes, mod_count that does not divide the row, in place
    and out of place, signed input / centred output."""
    import torch
    wide = (60, 61, 62) if bits == 64 else (30, 29, 30)
    g.set_option("path", "fast-strict")
    try:
        for logn, w, mc, poly in ((9, 1024, 3, O.X_N_plus), (9, 256, 2, O.X_N_minus), (8, 256, 3, O.X_N_plus),
                                  (7, 128, 5, O.X_N_minus), (4, 4096, 3, O.X_N_plus), (6, 64, 1, O.X_N_minus),
                                  (9, 8192, 7, O.X_N_plus)):
            fl = distinct_factors_scaled([wide[i % 3] for i in range(mc)], logn)
            cases = [MergeCase(g, bits, logn, poly, f) for f in fl]
            n = 1 << logn
            fwd = np.zeros(mc * n, dtype=cases[0].P.T)
            inv = np.zeros_like(fwd)
            for i, c in enumerate(cases):
                sz = c.prm.root_of_unity_size
                fwd[i * n:i * n + sz] = c.prm.forward_table_device_order
                inv[i * n:i * n + sz] = c.prm.inverse_table_device_order
            d_fwd, d_inv = g.to_device(fwd), g.to_device(inv)
            mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
            ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=cases[0].P.T))
            cols = np.stack([cases[p % mc].P.splitmix(95000 + p, 0, n, cases[p % mc].q) for p in range(w)])  # w x n
            mat = np.ascontiguousarray(cols.T)
            want_f = np.stack([cases[p % mc].P.merge_ntt(cols[p], cases[p % mc].oprm) for p in range(w)]).T
            cfg = g.ntt_rns_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=poly)
            icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, ntt_layout=g.PerCoefficient,
                                           reduction_poly=poly, mod_inverse=ninv)
            for rep in range(2):
                d = g.to_device(mat.reshape(-1))
                o = torch.zeros_like(d)
                g.GPU_NTT(d, o, d_fwd, mods, cfg, w, mc)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(o).reshape(n, w), want_f), ("fwd", bits, logn, w, mc, rep)
                g.GPU_INTT_Inplace(o, d_inv, mods, icfg, w, mc)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(o).reshape(n, w), mat), ("inv", bits, logn, w, mc, rep)
            # inverse of raw data against the oracle, centred signed output
            want_i = np.stack([cases[p % mc].P.merge_ntt(cols[p], cases[p % mc].oprm, inverse=True) for p in range(w)]).T
            d = g.to_device(mat.reshape(-1))
            o = torch.zeros_like(d)
            g.GPU_INTT(d, o, d_inv, mods, icfg, w, mc, dtype="s%d" % bits)
            torch.cuda.synchronize()
            got = g.to_host(o, signed=True).reshape(n, w).astype(object)
            for p in (0, 1, w // 2, w - 1):
                q = cases[p % mc].q
                col = np.array([int(v) for v in want_i[:, p]], dtype=object)
                centred = np.array([v - q if v > q // 2 else v for v in col], dtype=object)
                assert all(int(a) == int(b) for a, b in zip(got[:, p], centred)), ("centred", bits, logn, p)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))

# ---------------------------------------------------------------- per-lane moduli, passes of fewer than 4 stages
@pytest.mark.parametrize("bits", [64, 32])
def test_percoefficient_rns_tiny_rings_keep_the_right_modulus_per_column(g, bits):
    """a per-lane-modulus pass of K < 4 stages holds columns c, c + 256, ... in ONE thread and picks one
    modulus for all of them -- right only when mod_count divides 256.  (logn, columns, mod_count) = (3, 512, 3),
    (2, 1024, 5), (1, 2048, 3) returned wrong residues on the default path in round 4; they now take the generic
    kernels (fast-strict refuses them), mod_count 2 / 4 stay on the per-lane kernels.  Every column against NTTCPU."""
    import torch
    wide = (60, 61, 62) if bits == 64 else (30, 29, 30)
    for logn, w, mc, lazy_ok in ((3, 512, 3, False), (2, 1024, 5, False), (1, 2048, 3, False), (3, 512, 4, True), (2, 2048, 2, True)):
        fl = distinct_factors([wide[i % 3] for i in range(mc)], logn)
        cases = [MergeCase(g, bits, logn, O.X_N_plus, f) for f in fl]
        n = 1 << logn
        fwd = np.zeros(mc * n, dtype=cases[0].P.T)
        inv = np.zeros_like(fwd)
        for i, c in enumerate(cases):
            fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
            inv[i * n:i * n + c.prm.root_of_unity_size] = c.prm.inverse_table_device_order
        d_fwd, d_inv = g.to_device(fwd), g.to_device(inv)
        mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
        ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=cases[0].P.T))
        cols = np.stack([cases[p % mc].P.splitmix(96000 + p, 0, n, cases[p % mc].q) for p in range(w)])
        mat = np.ascontiguousarray(cols.T)
        want_f = np.stack([cases[p % mc].P.merge_ntt(cols[p], cases[p % mc].oprm) for p in range(w)]).T
        cfg = g.ntt_rns_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=O.X_N_plus)
        icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, ntt_layout=g.PerCoefficient,
                                       reduction_poly=O.X_N_plus, mod_inverse=ninv)
        for path in ("default", "fast-strict"):
            g.set_option("path", path)
            try:
                d = g.to_device(mat.reshape(-1))
                o = torch.zeros_like(d)
                if path == "fast-strict" and not lazy_ok:
                    with pytest.raises(ValueError, match="fast path unavailable"):
                        g.GPU_NTT(d, o, d_fwd, mods, cfg, w, mc)
                    continue
                g.GPU_NTT(d, o, d_fwd, mods, cfg, w, mc)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(o).reshape(n, w), want_f), ("fwd", bits, logn, w, mc, path)
                g.GPU_INTT_Inplace(o, d_inv, mods, icfg, w, mc)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(o).reshape(n, w), mat), ("inv", bits, logn, w, mc, path)
            finally:
                g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
