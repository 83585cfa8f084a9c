"""-m gpu tests added in round 4 (the 4-step table contract of that round became a default-on check in round 5: tests/test_gpu_round5.py): the two plan / option
interactions ADVICE r3 found, and the round's kernel variants (A/B switches must not change a single bit)."""
import os

import numpy as np
import pytest

from gpu_utils import MergeCase, find_ntt_factors
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(pkg):
    import torch
    assert torch.cuda.is_available()
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    pkg.load_library()
    return pkg


def _fourstep_forward(g, p4, tables, x, batch):
    """GPU_Transpose -> GPU_4STEP_NTT -> GPU_Transpose (example/ntt_4step/test_4step_ntt.cu:147-178)"""
    import torch
    d_a = g.to_device(x)
    d_b = torch.zeros_like(d_a)
    g.GPU_Transpose(d_a, d_b, p4.n1, p4.n2, p4.logn, batch)
    g.GPU_4STEP_NTT(d_b, d_a, *tables, p4.modulus, g.ntt4step_configuration(n_power=p4.logn, ntt_type=g.FORWARD), batch)
    g.GPU_Transpose(d_a, d_b, p4.n1, p4.n2, p4.logn, batch)
    torch.cuda.synchronize()
    return g.to_host(d_b)


def _distinct_factors(widths, logn):
    """distinct NTT primes of the given widths with (q, omega, psi) for a ring of 2^logn.  Searched for 2^max(logn, 12) and
    brought down by squaring psi: the topmost primes = 1 mod 2^(logn + 1) of a small ring lie within double rounding of
    the power of two, get an over-stated `bit` and (61 -> 62) a mu that does not fit the word (Modulus refuses them)."""
    out, seen = [], set()
    lg = max(logn, 12)
    for w in widths:
        skip = 0
        while True:
            q, _, psi = find_ntt_factors(w, lg, skip)
            if q not in seen:
                break
            skip += 1
        seen.add(q)
        psi = pow(psi, 1 << (lg - logn), q)
        assert pow(psi, 1 << logn, q) == q - 1
        out.append((q, psi * psi % q, psi))
    return out


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
    cases = [MergeCase(g, 64, logn, poly, f) for f in _distinct_factors(widths, logn)]
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


@pytest.mark.parametrize("qbits", [60, 61, 62])
def test_fourstep_rns_overload_with_a_wide_device_side_modulus_on_the_lazy_kernels(g, qbits):
    """the RNS overload of GPU_4STEP_NTT with ONE device-side modulus (how the reference's example calls it,
    example/ntt_4step/test_4step_ntt.cu:126-146) of 61 / 62 bits: the 4 q family behind the three-state go-flag, no
    generic kernels (path = fast-strict).  Rings whose default family runs one launch on a bigger tile (2^13, 2^14 x 256)
    and the 8192-tile inverse (2^21) included.  Device-generated tables; expected values from the Merge oracle through
    GPU_4STEP_NTT(transpose(x)) == MergeNTT(x) and transpose(GPU_4STEP_NTT(y, INVERSE)) == x."""
    import torch
    P = O.Port(64)
    g.set_option("path", "fast-strict")
    try:
        for logn, batch in ((12, 3), (13, 5), (14, 256), (16, 3), (18, 2), (21, 2)):
            q, omega, psi = find_ntt_factors(qbits, logn)
            m = g.Modulus(q, bits=64)
            assert m.bit == qbits
            shape = g.NTTParameters4Step(logn, 64)
            n, n1, n2 = shape.n, shape.n1, shape.n2
            oprm = P.merge_params(logn, O.X_N_minus, (q, omega, psi))
            x = P.splitmix(4700 + logn + qbits, 0, batch * n, q)
            y = P.merge_ntt(x, oprm)
            mods = g.modulus_array_to_device([m], 64)
            ninv = g.to_device(np.array([pow(n, -1, q)], dtype=np.uint64))
            for inverse in (False, True):
                r = pow(omega, -1, q) if inverse else omega
                kind = g.INVERSE if inverse else g.FORWARD
                w = torch.zeros(n, dtype=torch.int64, device="cuda")
                t1 = torch.zeros(n1 >> 1, dtype=torch.int64, device="cuda")
                t2 = torch.zeros(n2 >> 1, dtype=torch.int64, device="cuda")
                g.GPU_Generate4StepW(w, r, m, logn, kind)
                g.GPU_GeneratePowerTable(t1, pow(r, n // n1, q), m, int(np.log2(n1)) - 1, True)
                g.GPU_GeneratePowerTable(t2, pow(r, n // n2, q), m, int(np.log2(n2)) - 1, True)
                cfg = g.ntt4step_rns_configuration(n_power=logn, ntt_type=kind, mod_inverse=ninv)
                src = x.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1).copy() if not inverse else y
                d_in = g.to_device(src)
                d_out = torch.zeros_like(d_in)
                g.GPU_4STEP_NTT(d_in, d_out, t1, t2, w, mods, cfg, batch, 1)
                torch.cuda.synchronize()
                got = g.to_host(d_out)
                if not inverse:
                    assert np.array_equal(got, y), ("forward", qbits, logn)
                else:
                    assert np.array_equal(got.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1), x), ("inverse", qbits, logn)
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
            fl = _distinct_factors([wide[i % 3] for i in range(mc)], logn)
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


@pytest.mark.parametrize("qbits", [61, 62])
def test_natural_order_fourstep_and_percoefficient_with_wide_single_modulus(g, qbits):
    """the last two Barrett fall-backs inside the reference's domain (modular_arith.cuh:66-67) for a HOST-side modulus:
    the natural-order 4-step extension and the single-modulus PerCoefficient layout with a 61- / 62-bit prime now run the
    4 q lazy kernels (path = fast-strict throws if a call would need the generic kernels)."""
    import torch
    P = O.Port(64)
    g.set_option("path", "fast-strict")
    try:
        # natural-order 4-step: 2^12 (one launch), 2^13 / 2^16 (strided + transposing row pass), 2^18 (two strided passes)
        for logn, batch in ((12, 3), (13, 2), (16, 2), (18, 1)):
            q, omega, psi = find_ntt_factors(qbits, logn)
            m = g.Modulus(q, bits=64)
            assert m.bit == qbits
            shape = g.NTTParameters4Step(logn, 64)
            n, n1, n2 = shape.n, shape.n1, shape.n2
            oprm = P.merge_params(logn, O.X_N_minus, (q, omega, psi))
            x = P.splitmix(5100 + logn + qbits, 0, batch * n, q)
            y = P.merge_ntt(x, oprm)  # bit-reversed Merge spectrum; NTT_4STEP_CPU::ntt order is its n1 x n2 transpose
            want = y.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1)
            tabs = {}
            for inverse in (False, True):
                r = pow(omega, -1, q) if inverse else omega
                kind = g.INVERSE if inverse else g.FORWARD
                w = torch.zeros(n, dtype=torch.int64, device="cuda")
                t1 = torch.zeros(n1 >> 1, dtype=torch.int64, device="cuda")
                t2 = torch.zeros(n2 >> 1, dtype=torch.int64, device="cuda")
                g.GPU_Generate4StepW(w, r, m, logn, kind)
                g.GPU_GeneratePowerTable(t1, pow(r, n // n1, q), m, int(np.log2(n1)) - 1, True)
                g.GPU_GeneratePowerTable(t2, pow(r, n // n2, q), m, int(np.log2(n2)) - 1, True)
                tabs[inverse] = (t1, t2, w)
            d_in = g.to_device(x)
            d_out = torch.zeros_like(d_in)
            g.GPU_4STEP_NTT_NaturalOrder(d_in, d_out, *tabs[False], m, g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD), batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d_out), want), ("natural forward", qbits, logn)
            d_back = torch.zeros_like(d_out)
            ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=pow(n, -1, q))
            g.GPU_4STEP_NTT_NaturalOrder(d_out, d_back, *tabs[True], m, ci, batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d_back), x), ("natural inverse", qbits, logn)
        # PerCoefficient layout, single wide modulus: one- and two-pass shapes
        for logn, w, poly in ((9, 1024, O.X_N_plus), (8, 256, O.X_N_minus), (5, 4096, O.X_N_plus)):
            f = _distinct_factors([qbits], logn)[0]
            c = MergeCase(g, 64, logn, poly, f)
            assert c.prm.modulus.bit == qbits
            n = c.n
            cols = c.random(w, 5300 + logn + w).reshape(w, n)
            mat = np.ascontiguousarray(cols.T)
            want_f = np.ascontiguousarray(c.P.merge_ntt(cols.reshape(-1), c.oprm).reshape(w, n).T)
            cfg = g.ntt_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=poly)
            d = g.to_device(mat.reshape(-1))
            o = torch.zeros_like(d)
            g.GPU_NTT(d, o, c.fwd_dev, c.prm.modulus, cfg, w)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o).reshape(n, w), want_f), ("percoefficient fwd", qbits, logn, w)
            icfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE, ntt_layout=g.PerCoefficient, reduction_poly=poly,
                                       mod_inverse=c.prm.n_inv)
            g.GPU_INTT_Inplace(o, c.inv_dev, c.prm.modulus, icfg, w)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o).reshape(n, w), mat), ("percoefficient inv", qbits, logn, w)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))


def test_u32_ring_2_13_on_its_own_tile(g):
    """32-bit ring 2^13 runs on a 8192-coefficient tile of its own instead of sharing a 16384-coefficient one (VERDICT r3
    weak #8: batch 1 of 2^13 was slower than batch 1 of 2^14; tools/ab_u32_ring13.py: equal or faster at every batch size).
    Both lazy ranges (29-bit pool prime: 8 q; 30-bit prime: 4 q), drop-in and NTTPlan, every polynomial.  (The option that
    chose between the two tiles is retired: the A/B is closed.)"""
    import torch
    try:
        for opt in ("own tile",):
            for factors in (None, find_ntt_factors(30, 13)):
                for poly in (O.X_N_plus, O.X_N_minus):
                    c = MergeCase(g, 32, 13, poly, factors)
                    for batch in (1, 3, 16, 17, 40):
                        x = c.random(batch, 6100 + batch)
                        want = c.P.merge_ntt(x, c.oprm)
                        assert np.array_equal(c.gpu_forward(x, inplace=bool(batch & 1)), want), ("fwd", opt, poly, batch)
                        assert np.array_equal(c.gpu_inverse(want, inplace=not (batch & 1)), x), ("inv", opt, poly, batch)
                        plan = g.NTTPlan(c.fwd_dev, c.prm.modulus, 13, poly, g.FORWARD, batch_hint=batch)
                        d = g.to_device(x)
                        o = torch.zeros_like(d)
                        plan.execute(d, o, batch)
                        torch.cuda.synchronize()
                        assert plan.fast_path and np.array_equal(g.to_host(o), want), ("plan", opt, poly, batch)
                        plan.close()
    finally:
        pass


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
        cases = [MergeCase(g, 64, logn, O.X_N_plus, f) for f in (pool if widths is None else _distinct_factors(widths, logn))]
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
        q, omega, psi = _distinct_factors([qbits], logn)[0]
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
    cases = [MergeCase(g, 64, logn, O.X_N_plus, f) for f in _distinct_factors((60, 61, 60), logn)]
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
