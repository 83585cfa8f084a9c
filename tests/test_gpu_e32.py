"""-m gpu parity tests of the 32-coefficients-per-lane geometry of the 32-bit single-sweep rings
(gpu-ntt_amd/csrc/merge_e32_kernels.hpp; replaces reference ForwardCore / InverseCore for the rings 2^12 .. 2^15,
src/lib/ntt_merge/ntt.cu:596-761, 1086-1318): every entry point that can reach those kernels, against the oracle,
with the geometry switched on for all four rings and -- same inputs -- switched off (option u32_e32)."""
import numpy as np
import pytest

from oracle import oracle as O
from gpu_utils import MergeCase, find_ntt_factors, oracle_batch
from test_gpu_merge import _rns_setup, _small_prime_factors

pytestmark = pytest.mark.gpu

ALL_RINGS = 0x1F000  # bits 12 .. 15: the one-tile rings; bit 16: the full-tile contiguous pass of larger rings
DEFAULT = 0x1F000


@pytest.fixture(scope="module")
def g(pkg):
    pkg.load_library()
    return pkg


@pytest.fixture(params=[ALL_RINGS, 0], ids=["e32-on", "e32-off"])
def mask(g, request):
    g.set_option("u32_e32", request.param)
    yield request.param
    g.set_option("u32_e32", DEFAULT)


@pytest.mark.parametrize("poly", [O.X_N_minus, O.X_N_plus], ids=["cyclic", "negacyclic"])
def test_every_ring_both_directions(g, mask, poly):
    # pool prime (q < 2^29: the 8 q lazy range) and a searched 30-bit prime (the 4 q range)
    for logn in (12, 13, 14, 15):
        for factors in (None, find_ntt_factors(30, logn)):
            c = MergeCase(g, 32, logn, poly, factors)
            for batch, inplace in ((1, False), (5, True)):
                x = c.random(batch, 1000 * logn + batch)
                y = c.gpu_forward(x, inplace=inplace)
                assert np.array_equal(y, oracle_batch([c], x)), (logn, factors, batch, "fwd")
                assert np.array_equal(c.gpu_inverse(y, inplace=inplace), x), (logn, factors, batch, "inv")
                z = c.random(batch, 77 + logn)
                assert np.array_equal(c.gpu_inverse(z, inplace=not inplace), oracle_batch([c], z, inverse=True))


@pytest.mark.parametrize("logn,batch", [(16, 3), (18, 3), (19, 2), (20, 3), (21, 2), (22, 2), (23, 3), (24, 1)])
def test_contiguous_pass_of_larger_rings(g, mask, logn, batch):
    """Rings above one tile: the full-tile contiguous pass of the plan (forward: the last pass, on lazy input from the
    strided passes; inverse: the first, handing lazy values on) runs on the 32-coefficients-per-lane geometry where the
    plan's contiguous pass fills its tile (4096-coefficient tiles: 2^18, 2^19, 2^24; 16384: 2^20 .. 2^22; 32768: 2^23, which
    becomes TWO sweeps; 2^16 keeps a 9-stage pass on the 16-coefficient kernels) -- both lazy families, a stack of three
    primes, both directions."""
    import torch
    # (no 30-bit prime is 1 mod 2^25: the 4 q family stops at 2^23)
    for factors in ((None, find_ntt_factors(30, logn)) if logn <= 23 else (None,)):
        c = MergeCase(g, 32, logn, O.X_N_plus, factors)
        x = c.random(batch, 31 * logn + batch)
        y = c.gpu_forward(x, inplace=True)
        assert np.array_equal(y, oracle_batch([c], x)), (logn, factors, "fwd")
        assert np.array_equal(c.gpu_inverse(y, inplace=False), x), (logn, factors, "inv")
    if logn <= 20:
        P = O.Port(32)
        mc = 3
        cases, fwd, inv, mods, ninv = _rns_setup(g, 32, logn, O.X_N_minus, _small_prime_factors(P, logn, mc))
        n = 1 << logn
        b = 4
        x = np.concatenate([cases[p % mc].P.splitmix(70 + p, 0, n, cases[p % mc].q) for p in range(b)])
        d = g.to_device(x)
        g.GPU_NTT_Inplace(d, fwd, mods, g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_minus), b, mc)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), oracle_batch(cases, x)), (logn, "rns fwd")
        g.GPU_INTT_Inplace(d, inv, mods, g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=O.X_N_minus,
                                                                   mod_inverse=ninv), b, mc)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), x), (logn, "rns inv")


def test_edge_values(g, mask):
    for logn in (12, 14, 15):
        c = MergeCase(g, 32, logn, O.X_N_plus)
        n, q = c.n, c.q
        rows = [np.zeros(n), np.full(n, q - 1), np.eye(1, n, 0)[0], np.eye(1, n, n - 1)[0] * (q - 1), np.ones(n)]
        x = np.concatenate([np.asarray(r, dtype=object) for r in rows]).astype(c.P.T)
        assert np.array_equal(c.gpu_forward(x), oracle_batch([c], x))
        assert np.array_equal(c.gpu_inverse(x), oracle_batch([c], x, inverse=True))


def test_signed_input_and_centred_output(g, mask):
    import torch
    for logn in (13, 14, 15):
        c = MergeCase(g, 32, logn, O.X_N_minus)
        q = c.q
        x = c.random(3, 31 + logn)
        xs = np.where(x > q // 2, x.astype(object) - q, x.astype(object)).astype(np.int32)
        d_in = g.to_device(xs)
        d_out = torch.zeros_like(d_in)
        g.GPU_NTT(d_in, d_out, c.fwd_dev, c.prm.modulus, c.cfg(), 3, dtype="s32")
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_out), oracle_batch([c], x))
        d_back = torch.zeros_like(d_out)
        g.GPU_INTT(d_out, d_back, c.inv_dev, c.prm.modulus, c.cfg(True), 3, dtype="s32")
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_back, signed=True), xs)


def test_rns_stack_and_ordered_entry_points(g, mask):
    import torch
    P = O.Port(32)
    for logn, batch, mc in ((12, 7, 3), (14, 9, 4), (15, 5, 2)):
        for poly in (O.X_N_plus, O.X_N_minus):
            cases, fwd, inv, mods, ninv = _rns_setup(g, 32, logn, poly, _small_prime_factors(P, logn, mc))
            n = 1 << logn
            x = np.concatenate([cases[p % mc].P.splitmix(50 + p, 0, n, cases[p % mc].q) for p in range(batch)])
            want = oracle_batch(cases, x)
            cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
            icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
            d = g.to_device(x)
            g.GPU_NTT_Inplace(d, fwd, mods, cfg, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), want), (logn, poly)
            g.GPU_INTT_Inplace(d, inv, mods, icfg, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), x), (logn, poly)
            # poly-ordered: the polynomials live in permuted slots of a larger buffer
            slots = [(p * 5 + 1) % (batch + 2) for p in range(batch)]  # batch + 2 is coprime to 5 for 7, 9, 5
            assert len(set(slots)) == batch
            buf = np.full((batch + 2) * n, 12345, dtype=cases[0].P.T)
            for p, sl in enumerate(slots):
                buf[sl * n:(sl + 1) * n] = x[p * n:(p + 1) * n]
            wbuf = buf.copy()
            for p, sl in enumerate(slots):
                wbuf[sl * n:(sl + 1) * n] = want[p * n:(p + 1) * n]
            d = g.to_device(buf)
            d_slots = torch.tensor(slots, dtype=torch.int32, device="cuda")
            g.GPU_NTT_Poly_Ordered(d, d, fwd, mods, cfg, batch, mc, d_slots)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), wbuf), ("poly-ordered", logn, poly)
            g.GPU_NTT_Poly_Ordered(d, d, inv, mods, icfg, batch, mc, d_slots)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), buf), ("poly-ordered inv", logn, poly)
            # modulus-ordered: a sub-stack
            order = list(range(mc))[::-1][:max(1, mc - 1)]
            d_order = torch.tensor(order, dtype=torch.int32, device="cuda")
            omc = len(order)
            xo = np.concatenate([cases[order[p % omc]].P.splitmix(900 + p, 0, n, cases[order[p % omc]].q)
                                 for p in range(batch)])
            wo = np.concatenate([cases[order[p % omc]].P.merge_ntt(xo[p * n:(p + 1) * n], cases[order[p % omc]].oprm)
                                 for p in range(batch)])
            d = g.to_device(xo)
            o = torch.zeros_like(d)
            g.GPU_NTT_Modulus_Ordered(d, o, fwd, mods, cfg, batch, omc, d_order)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), wo), ("mod-ordered", logn, poly)


def test_polymul_product_rides_on_the_final_store(g, mask):
    # GPU_PolyMul fuses the pointwise product into the forward transform's final store: one-tile rings, a ring whose last
    # pass is the full-tile contiguous pass (2^20), and the two-sweep ring 2^23 on the 32768-coefficient tile
    import torch
    for logn, batch in ((14, 2), (15, 2), (20, 2), (23, 1)):
        c = MergeCase(g, 32, logn, O.X_N_plus)
        q = c.q
        a, b = c.random(batch, 5 + logn), c.random(batch, 6 + logn)
        fa, fb = oracle_batch([c], a), oracle_batch([c], b)
        prod = (fa.astype(np.uint64) * fb.astype(np.uint64) % np.uint64(q)).astype(c.P.T)
        want = oracle_batch([c], prod, inverse=True)
        da, db = g.to_device(a), g.to_device(b)
        out = torch.zeros_like(da)
        g.GPU_PolyMul(da, db, out, c.fwd_dev, c.inv_dev, c.prm.modulus, c.cfg(True), batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(out), want), logn


def test_plan_keeps_its_tile_when_the_option_changes(g):
    # an NTTPlan prepared for the 32768-coefficient tile must run on it whatever the option says at execute() time
    import torch
    g.set_option("u32_e32", ALL_RINGS)
    try:
        c = MergeCase(g, 32, 15, O.X_N_plus)
        x = c.random(4, 9)
        want = oracle_batch([c], x)
        plan = g.NTTPlan(c.fwd_dev, c.prm.modulus, 15, O.X_N_plus, batch_hint=4)
        iplan = g.NTTPlan(c.inv_dev, c.prm.modulus, 15, O.X_N_plus, ntt_type=g.INVERSE, mod_inverse=c.prm.n_inv,
                          batch_hint=4)
        for m in (ALL_RINGS, 0):
            g.set_option("u32_e32", m)
            d = g.to_device(x)
            o = torch.zeros_like(d)
            plan.execute(d, o, 4)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), want), m
            iplan.execute(o, o, 4)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), x), m
    finally:
        g.set_option("u32_e32", DEFAULT)


def test_every_polynomial_of_a_full_chip_batch(g):
    # 2^15 x 2048 and 2^14 x 4096 (2^26 coefficients each): every polynomial against the oracle, both directions, and the
    # two geometries against each other
    for logn, batch in ((15, 2048), (14, 4096)):
        c = MergeCase(g, 32, logn, O.X_N_plus)
        x = c.random(batch, 4242 + logn)
        want = oracle_batch([c], x)
        got = {}
        for m in (ALL_RINGS, 0):
            g.set_option("u32_e32", m)
            try:
                got[m] = c.gpu_forward(x, inplace=True)
                assert np.array_equal(c.gpu_inverse(got[m], inplace=True), x), (logn, m)
            finally:
                g.set_option("u32_e32", DEFAULT)
        assert np.array_equal(got[ALL_RINGS], want), logn
        assert np.array_equal(got[0], want), logn


def test_option_refuses_rings_outside_the_geometry(g):
    with pytest.raises(Exception):
        g.set_option("u32_e32", 1 << 17)
    with pytest.raises(Exception):
        g.set_option("u32_e32", "abc")
    g.set_option("u32_e32", DEFAULT)
