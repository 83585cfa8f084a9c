"""-m gpu parity tests for the 4-Step NTT and GPU_Transpose, mirroring
example/ntt_4step/test_4step_ntt.cu:147-178 and test_4step_intt.cu:81-179:
  forward:  GPU_Transpose -> GPU_4STEP_NTT(FORWARD) -> GPU_Transpose == NTT_4STEP_CPU::ntt
  inverse:  intt_first_transpose -> GPU_4STEP_NTT(INVERSE) -> GPU_Transpose == NTT_4STEP_CPU::intt"""
import json
import os

import numpy as np
import pytest

from gpu_utils import sha
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(pkg):
    import torch
    assert torch.cuda.is_available()
    if not os.path.exists(pkg.LIB_PATH):  # fresh checkout: hipcc is part of the image
        pkg.build_library()
    pkg.load_library()
    return pkg


def run_fourstep(g, p4, x, batch, inverse, rns=False):
    """full natural-order pipeline on the GPU; returns the flat result"""
    import torch
    t1, t2, w = (g.to_device(t) for t in p4.tables["inv" if inverse else "fwd"])
    d_a = g.to_device(x)
    d_b = torch.zeros_like(d_a)
    if rns:
        mods = g.modulus_array_to_device([p4.modulus], p4.bits)
        ninv = g.to_device(np.array([p4.n_inv], dtype=g.np_dtype(p4.bits)))
        cfg = g.ntt4step_rns_configuration(n_power=p4.logn,
                                           ntt_type=g.INVERSE if inverse else g.FORWARD,
                                           mod_inverse=ninv)
        args = (mods, cfg, batch, 1)
    else:
        cfg = g.ntt4step_configuration(n_power=p4.logn,
                                       ntt_type=g.INVERSE if inverse else g.FORWARD,
                                       mod_inverse=p4.n_inv if inverse else 0)
        args = (p4.modulus, cfg, batch)
    torch.cuda.synchronize()
    if not inverse:
        g.GPU_Transpose(d_a, d_b, p4.n1, p4.n2, p4.logn, batch)
        torch.cuda.synchronize()
        g.GPU_4STEP_NTT(d_b, d_a, t1, t2, w, *args)
        torch.cuda.synchronize()
        g.GPU_Transpose(d_a, d_b, p4.n1, p4.n2, p4.logn, batch)
        torch.cuda.synchronize()
        return g.to_host(d_b)
    g.GPU_4STEP_NTT(d_a, d_b, t1, t2, w, *args)
    torch.cuda.synchronize()
    g.GPU_Transpose(d_b, d_a, p4.n1, p4.n2, p4.logn, batch)
    torch.cuda.synchronize()
    return g.to_host(d_a)


@pytest.mark.parametrize("bits", [32, 64])
def test_transpose(g, bits):
    import torch
    rng = np.random.default_rng(3)
    for row, col, batch in ((32, 128, 3), (256, 64, 2), (128, 512, 1)):
        x = rng.integers(0, 2**31, size=batch * row * col).astype(g.np_dtype(bits))
        d = g.to_device(x)
        o = torch.zeros_like(d)
        g.GPU_Transpose(d, o, row, col, int(np.log2(row * col)), batch)
        torch.cuda.synchronize()
        want = x.reshape(batch, row, col).transpose(0, 2, 1).reshape(-1)
        assert np.array_equal(g.to_host(o), want)


@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_vs_oracle(g, bits):
    # every n1 x n2 shape class: n1 in {32, 64, 128}, n2 from 128 (one tile) to 32768 (strided)
    P = O.Port(bits)
    # 2^19 runs with batch 8: n2 = 16384 rows x (8 x 32) transforms reaches the 16384-coefficient
    # tile of the 64-bit row pass (lazy_tile_log: >= 256 transforms), which batch 1 never does
    for logn in (12, 13, 14, 15, 16, 17, 18, 19, 20):
        p4 = g.NTTParameters4Step(logn, bits)
        oprm = P.fourstep_params(logn)
        assert np.array_equal(p4.tables["fwd"][2], oprm["W_fwd"])
        batch = 2 if logn <= 16 else (8 if logn == 19 else 1)
        x = P.splitmix(400 + logn, 0, batch * p4.n, p4.modulus.value)
        want = P.fourstep_ntt(x, oprm)
        got = run_fourstep(g, p4, x, batch, inverse=False, rns=(logn % 2 == 0))
        assert np.array_equal(got, want), ("forward", bits, logn)
        xin = P.fourstep_intt_first_transpose(want, oprm)
        back = run_fourstep(g, p4, xin, batch, inverse=True, rns=(logn % 2 == 1))
        assert np.array_equal(back, x), ("inverse", bits, logn)


@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_natural_order_vs_oracle(g, bits):
    """extension GPU_4STEP_NTT_NaturalOrder == the three-call pipeline == NTT_4STEP_CPU::ntt / intt,
    every shape class (single transposing row pass n2 <= 512, strided + transposing pass above)"""
    import torch
    P = O.Port(bits)
    for logn in (12, 13, 14, 15, 16, 17, 18, 19, 20):
        p4 = g.NTTParameters4Step(logn, bits)
        oprm = P.fourstep_params(logn)
        batch = 3 if logn <= 16 else 1
        x = P.splitmix(700 + logn, 0, batch * p4.n, p4.modulus.value)
        want = np.concatenate([P.fourstep_ntt(x[i * p4.n:(i + 1) * p4.n], oprm) for i in range(batch)])
        tf = [g.to_device(t) for t in p4.tables["fwd"]]
        ti = [g.to_device(t) for t in p4.tables["inv"]]
        d_in = g.to_device(x)
        d_out = torch.zeros_like(d_in)
        cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
        g.GPU_4STEP_NTT_NaturalOrder(d_in, d_out, *tf, p4.modulus, cf, batch)
        torch.cuda.synchronize()
        got = g.to_host(d_out)
        assert np.array_equal(got, want), ("forward", bits, logn)
        ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
        d_back = torch.zeros_like(d_out)
        g.GPU_4STEP_NTT_NaturalOrder(d_out, d_back, *ti, p4.modulus, ci, batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_back), x), ("inverse", bits, logn)
    # same buffer twice is rejected (the input doubles as scratch)
    with pytest.raises(ValueError):
        g.GPU_4STEP_NTT_NaturalOrder(d_in, d_in, *tf, p4.modulus, cf, batch)


def test_fourstep_rns_overload_generic_kernels_on_capped_grid(g):
    """the RNS overload with one device-side modulus enqueues the generic kernels behind the fast
    ones; when they own the call (61/62-bit modulus) they run on a capped grid that walks the tiles.
    option path = generic-capped forces exactly that state for a pool prime."""
    import torch
    P = O.Port(64)
    logn, batch = 20, 8  # 2048 tiles > 1024 blocks
    p4 = g.NTTParameters4Step(logn, 64)
    oprm = P.fourstep_params(logn)
    x = P.splitmix(901, 0, batch * p4.n, p4.modulus.value)
    want = P.fourstep_ntt(x, oprm)
    g.set_option("path", "generic-capped")
    try:
        got = run_fourstep(g, p4, x, batch, inverse=False, rns=True)
        back = run_fourstep(g, p4, P.fourstep_intt_first_transpose(want, oprm), batch, inverse=True, rns=True)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
    assert np.array_equal(got, want)
    assert np.array_equal(back, x)


def test_fourstep_natural_order_generic_fallback(g):
    """no fast path (forced here; also what a 61/62-bit modulus gets): generic 4-step kernels
    between two transposes, same result"""
    import torch
    P = O.Port(64)
    logn = 13
    p4 = g.NTTParameters4Step(logn, 64)
    oprm = P.fourstep_params(logn)
    x = P.splitmix(801, 0, p4.n, p4.modulus.value)
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    d_in = g.to_device(x)
    d_out = torch.zeros_like(d_in)
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
    g.set_option("path", "generic")
    try:
        g.GPU_4STEP_NTT_NaturalOrder(d_in, d_out, *tf, p4.modulus, cf, 1)
        torch.cuda.synchronize()
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
    assert np.array_equal(g.to_host(d_out), P.fourstep_ntt(x, oprm))


@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_golden(g, bits, golden_dir):
    P = O.Port(bits)
    gold = np.load(os.path.join(golden_dir, "fourstep_u%d.npz" % bits))
    recs = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["fourstep"]
            if r["bits"] == bits]
    for r in recs:
        p4 = g.NTTParameters4Step(r["logn"], bits)
        assert sha(p4.tables["fwd"][2]) == r["sha_W_fwd"] and sha(p4.tables["inv"][2]) == r["sha_W_inv"]
        assert sha(p4.tables["fwd"][1]) == r["sha_n2_fwd_gpu"]
        x = P.splitmix(r["seed"], 0, p4.n, r["q"])
        assert sha(x) == r["sha_in"]
        fwd = run_fourstep(g, p4, x, 1, inverse=False)
        assert sha(fwd) == r["sha_fwd"], ("fwd", r["logn"])
        # intt_first_transpose: flat[i*n2+j] = x[i + j*n1]
        xin = x.reshape(p4.n2, p4.n1).T.reshape(-1).copy()
        assert sha(xin) == r["sha_first_transpose"]
        inv = run_fourstep(g, p4, xin, 1, inverse=True)
        assert sha(inv) == r["sha_inv"], ("inv", r["logn"])
        if "l%d_fwd" % r["logn"] in gold:
            assert np.array_equal(fwd, gold["l%d_fwd" % r["logn"]])
            assert np.array_equal(inv, gold["l%d_inv" % r["logn"]])


def test_fourstep_natural_order_2_24_by_digest(g, golden_dir):
    """BASELINE config 3's own ring (u64, 2^24 = 256 x 65536) through the one-call natural-order
    form, forward and inverse, against the reference build's digests (tests/golden/digests.json)"""
    import torch
    P = O.Port(64)
    r = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["fourstep"]
         if r["bits"] == 64 and r["logn"] == 24][0]
    p4 = g.NTTParameters4Step(24, 64)
    x = P.splitmix(r["seed"], 0, p4.n, r["q"])
    assert sha(x) == r["sha_in"]
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    d_in = g.to_device(x)
    d_out = torch.zeros_like(d_in)
    g.GPU_4STEP_NTT_NaturalOrder(d_in, d_out, *tf, p4.modulus, g.ntt4step_configuration(n_power=24, ntt_type=g.FORWARD), 1)
    torch.cuda.synchronize()
    assert sha(g.to_host(d_out)) == r["sha_fwd"]
    del tf
    ti = [g.to_device(t) for t in p4.tables["inv"]]
    d_x = g.to_device(x)  # the forward call used its input as scratch
    d_back = torch.zeros_like(d_x)
    ci = g.ntt4step_configuration(n_power=24, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
    g.GPU_4STEP_NTT_NaturalOrder(d_x, d_back, *ti, p4.modulus, ci, 1)
    torch.cuda.synchronize()
    assert sha(g.to_host(d_back)) == r["sha_inv"]


def test_full_size_c3_properties(g, golden_dir):
    """BASELINE config 3 at full size (u64, 2^24, batch 64, 8 GiB in + 8 GiB out): 4 distinct
    polynomials repeated 16 times; the distinct ones are pinned by the reference digest (polynomial 0
    is the golden input) and by the forward -> inverse round trip, the repeats must be identical copies."""
    import torch
    P = O.Port(64)
    r = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["fourstep"]
         if r["bits"] == 64 and r["logn"] == 24][0]
    p4 = g.NTTParameters4Step(24, 64)
    n, batch, distinct = p4.n, 64, 4
    base = np.concatenate([P.splitmix(r["seed"] + 7 * i, 0, n, r["q"]) for i in range(distinct)])
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    d_base = g.to_device(base)
    d_in = d_base.repeat(batch // distinct)
    d_mid = torch.zeros_like(d_in)
    cf = g.ntt4step_configuration(n_power=24, ntt_type=g.FORWARD)
    g.GPU_4STEP_NTT_NaturalOrder(d_in, d_mid, *tf, p4.modulus, cf, batch)
    torch.cuda.synchronize()
    y = d_mid.view(batch // distinct, distinct * n)
    assert bool((y == y[0:1]).all()), "copies of the same polynomial transformed differently"
    y0 = g.to_host(d_mid[: distinct * n].clone())
    assert sha(y0[:n]) == r["sha_fwd"]
    assert int(y0.max()) < r["q"]
    del tf
    ti = [g.to_device(t) for t in p4.tables["inv"]]
    d_back = d_in  # reuse
    ci = g.ntt4step_configuration(n_power=24, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
    g.GPU_4STEP_NTT_NaturalOrder(d_mid, d_back, *ti, p4.modulus, ci, batch)
    torch.cuda.synchronize()
    z = d_back.view(batch // distinct, distinct * n)
    assert bool((z == d_base.view(1, -1)).all()), "forward -> inverse is not the identity at batch 64"


def test_full_size_c3_reference_layout(g, golden_dir):
    """BASELINE config 3's OWN call at its own shape: GPU_4STEP_NTT in the reference layout (n2 x n1 in, n1 x n2 out;
    example/ntt_4step/test_4step_ntt.cu:147-178, test_4step_intt.cu:81-179), u64, 2^24, batch 64, forward then inverse,
    through the plain overload, the RNS overload (one device-side modulus) and FourStepPlan -- 8 distinct polynomials x
    8 copies, EVERY one of the 64 outputs compared on the device, three times over.  The expected spectra are the
    reference build's: NTT_4STEP_CPU::ntt of each polynomial, pinned by tests/golden/c3_polys.json
    (tools/make_golden.py --add-c3), uploaded and transposed on the device into the call's n1 x n2 layout."""
    import torch
    P = O.Port(64)
    rec = json.load(open(os.path.join(golden_dir, "c3_polys.json")))
    logn, batch, distinct = 24, 64, len(rec["polys"])
    assert distinct == 8
    p4 = g.NTTParameters4Step(logn, 64)
    assert (p4.modulus.value, p4.n1, p4.n2) == (rec["q"], rec["n1"], rec["n2"])
    n, n1, n2 = p4.n, p4.n1, p4.n2
    oprm = P.fourstep_params(logn)
    x_nat = np.concatenate([P.splitmix(r["seed"], 0, n, rec["q"]) for r in rec["polys"]])
    want = np.empty_like(x_nat)
    for i, r in enumerate(rec["polys"]):
        assert sha(x_nat[i * n:(i + 1) * n]) == r["sha_in"]
        want[i * n:(i + 1) * n] = P.fourstep_ntt(x_nat[i * n:(i + 1) * n], oprm)
        assert sha(want[i * n:(i + 1) * n]) == r["sha_fwd"], ("oracle != reference build digest", i)
    d_x = g.to_device(x_nat)
    # the forward call reads x^T (n2 x n1); its output is the spectrum as n1 x n2 = NTT_4STEP_CPU::ntt's order transposed;
    # the inverse call returns x so that one GPU_Transpose(n1, n2) restores the natural order
    exp_in = d_x.view(distinct, n1, n2).transpose(1, 2).contiguous().view(-1)
    exp_fwd = g.to_device(want).view(distinct, n2, n1).transpose(1, 2).contiguous().view(1, distinct * n)
    exp_back = d_x.view(distinct, n2, n1).transpose(1, 2).contiguous().view(1, distinct * n)
    del d_x
    d_in = exp_in.repeat(batch // distinct)
    in_digest = int(d_in.view(torch.int64).sum())
    d_mid = torch.zeros_like(d_in)
    d_back = torch.zeros_like(d_in)
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    ti = [g.to_device(t) for t in p4.tables["inv"]]
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
    ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
    mods = g.modulus_array_to_device([p4.modulus], 64)
    ninv = g.to_device(np.array([p4.n_inv], dtype=np.uint64))
    rf = g.ntt4step_rns_configuration(n_power=logn, ntt_type=g.FORWARD, mod_inverse=ninv)
    ri = g.ntt4step_rns_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=ninv)
    plan_f = g.FourStepPlan(*tf, p4.modulus, cf, batch_hint=batch)
    plan_i = g.FourStepPlan(*ti, p4.modulus, ci, batch_hint=batch)
    assert plan_f.fast_path and plan_i.fast_path
    forms = {
        "plain": (lambda: g.GPU_4STEP_NTT(d_in, d_mid, *tf, p4.modulus, cf, batch),
                  lambda: g.GPU_4STEP_NTT(d_mid, d_back, *ti, p4.modulus, ci, batch)),
        "rns": (lambda: g.GPU_4STEP_NTT(d_in, d_mid, *tf, mods, rf, batch, 1),
                lambda: g.GPU_4STEP_NTT(d_mid, d_back, *ti, mods, ri, batch, 1)),
        "plan": (lambda: plan_f.execute(d_in, d_mid, batch), lambda: plan_i.execute(d_mid, d_back, batch)),
    }
    g.set_option("path", "fast-strict")
    try:
        for it in range(3):
            for name, (fwd, inv) in forms.items():
                d_mid.zero_()
                d_back.zero_()
                fwd()
                torch.cuda.synchronize()
                bad = (d_mid.view(batch // distinct, distinct * n) != exp_fwd).view(batch, n).any(dim=1)
                assert not bool(bad.any()), ("forward", name, it, torch.nonzero(bad).view(-1).tolist())
                inv()
                torch.cuda.synchronize()
                bad = (d_back.view(batch // distinct, distinct * n) != exp_back).view(batch, n).any(dim=1)
                assert not bool(bad.any()), ("inverse", name, it, torch.nonzero(bad).view(-1).tolist())
                assert int(d_in.view(torch.int64).sum()) == in_digest, "the out-of-place call modified its input"
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
        plan_f.close()
        plan_i.close()


def test_fourstep_unsupported_size_is_silent(g, capfd):
    # reference ntt_4step.cu:2529-2532: message on stdout, no exception
    import torch
    p4 = g.NTTParameters4Step(12, 64)
    t1, t2, w = (g.to_device(t) for t in p4.tables["fwd"])
    d = g.to_device(np.zeros(4096, dtype=np.uint64))
    o = torch.zeros_like(d)
    cfg = g.ntt4step_configuration(n_power=11)
    g.GPU_4STEP_NTT(d, o, t1, t2, w, p4.modulus, cfg, 1)
    torch.cuda.synchronize()


@pytest.mark.parametrize("bits,logn,batch", [(64, 12, 5), (64, 13, 3), (64, 14, 256), (64, 14, 3), (32, 12, 3),
                                             (32, 13, 3), (32, 13, 8), (32, 14, 1), (32, 14, 5)])
def test_fourstep_rings_that_fit_one_tile(g, bits, logn, batch):
    """2^12 .. 2^14: GPU_4STEP_NTT is the Merge transform of the ring with the natural-order side transposed, and
    runs as ONE launch (kern::fourstep_small_lazy: Merge table rebuilt from the 4-step tables, transposition in LDS)
    where the ring fits a tile -- 64-bit 2^12 / 2^13, 2^14 forward from 256 polynomials (smaller batches and the
    inverse keep the two-phase path), 32-bit 2^12 .. 2^14 (2^13 on a 8192-coefficient tile no Merge plan uses) incl.
    a ragged last tile.  Both overloads, both directions, against NTT_4STEP_CPU."""
    P = O.Port(bits)
    p4 = g.NTTParameters4Step(logn, bits)
    oprm = P.fourstep_params(logn)
    x = P.splitmix(1200 + logn + batch, 0, batch * p4.n, p4.modulus.value)
    sample = sorted({0, batch // 2, batch - 1})
    want = {p: P.fourstep_ntt(x[p * p4.n:(p + 1) * p4.n], oprm) for p in sample}
    for rns in (False, True):
        got = run_fourstep(g, p4, x, batch, inverse=False, rns=rns)
        for p in sample:
            assert np.array_equal(got[p * p4.n:(p + 1) * p4.n], want[p]), ("forward", bits, logn, batch, rns, p)
        xin = np.concatenate([P.fourstep_intt_first_transpose(got[p * p4.n:(p + 1) * p4.n], oprm) for p in range(batch)])
        back = run_fourstep(g, p4, xin, batch, inverse=True, rns=rns)
        assert np.array_equal(back, x), ("inverse", bits, logn, batch, rns)


@pytest.mark.parametrize("qbits", [61, 62])
def test_fourstep_61_and_62_bit_moduli_on_the_fast_kernels(g, qbits):
    """GPU_4STEP_NTT with a 61- / 62-bit prime (inside the reference's documented domain,
    src/include/gpuntt/common/modular_arith.cuh:66-67) runs on the LIMIT = 8 / 4 fast kernels under path = fast-strict:
    the one-launch 2^12 ring, the Merge-form forward plans and the two-phase inverse.  NTTParameters4Step has no
    custom-prime constructor (neither has the reference), so the tables are generated on the device; the expected
    values come from the Merge oracle through the identity GPU_4STEP_NTT(transpose(x)) == MergeNTT(x) (forward) and
    transpose(GPU_4STEP_NTT(y, INVERSE)) == x (inverse), with the 4-step root as the ring's root."""
    import torch
    from gpu_utils import find_ntt_factors
    P = O.Port(64)
    g.set_option("path", "fast-strict")
    try:
        for logn, batch in ((12, 3), (13, 2), (16, 2), (18, 1)):
            q, omega, psi = find_ntt_factors(qbits, logn)
            m = g.Modulus(q, bits=64)
            assert m.bit == qbits
            shape = g.NTTParameters4Step(logn, 64)  # n1 x n2 of this ring
            n, n1, n2 = shape.n, shape.n1, shape.n2
            oprm = P.merge_params(logn, O.X_N_minus, (q, omega, psi))
            x = P.splitmix(4100 + logn + qbits, 0, batch * n, q)
            y = np.concatenate([P.merge_ntt(x[p * n:(p + 1) * n], oprm) for p in range(batch)])
            for inverse in (False, True):
                r = pow(omega, -1, q) if inverse else omega
                kind = g.INVERSE if inverse else g.FORWARD
                w = torch.zeros(n, dtype=torch.int64, device="cuda")
                t1 = torch.zeros(n1 >> 1, dtype=torch.int64, device="cuda")
                t2 = torch.zeros(n2 >> 1, dtype=torch.int64, device="cuda")
                g.GPU_Generate4StepW(w, r, m, logn, kind)
                g.GPU_GeneratePowerTable(t1, pow(r, n // n1, q), m, int(np.log2(n1)) - 1, True)
                g.GPU_GeneratePowerTable(t2, pow(r, n // n2, q), m, int(np.log2(n2)) - 1, True)
                cfg = g.ntt4step_configuration(n_power=logn, ntt_type=kind, mod_inverse=pow(n, -1, q) if inverse else 0)
                if not inverse:
                    src = x.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1).copy()
                else:
                    src = y
                d_in = g.to_device(src)
                d_out = torch.zeros_like(d_in)
                g.GPU_4STEP_NTT(d_in, d_out, t1, t2, w, m, cfg, batch)
                torch.cuda.synchronize()
                got = g.to_host(d_out)
                if not inverse:
                    assert np.array_equal(got, y), ("forward", qbits, logn)
                else:
                    back = got.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1)
                    assert np.array_equal(back, x), ("inverse", qbits, logn)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))


@pytest.mark.parametrize("bits,logn,batch", [(64, 14, 256), (32, 14, 96), (64, 13, 64), (32, 13, 96)])
def test_one_launch_rings_every_polynomial_repeatedly(g, bits, logn, batch):
    """stress form of the test above: EVERY polynomial of the batch is compared, six times over.  Round 3 found the
    one-launch kernel on the 16384-coefficient tile returning one wrong polynomial in about a thousand: the compiler had
    sunk LDS reads below the barrier that protects the buffer from the next round's writes (kern::pin_loaded); a sampled
    check passes that almost every time."""
    P = O.Port(bits)
    p4 = g.NTTParameters4Step(logn, bits)
    oprm = P.fourstep_params(logn)
    x = P.splitmix(1900 + logn + batch, 0, batch * p4.n, p4.modulus.value)
    want = np.concatenate([P.fourstep_ntt(x[p * p4.n:(p + 1) * p4.n], oprm) for p in range(batch)])
    xin = np.concatenate([P.fourstep_intt_first_transpose(want[p * p4.n:(p + 1) * p4.n], oprm) for p in range(batch)])
    import torch
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    ti = [g.to_device(t) for t in p4.tables["inv"]]
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
    ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
    for it in range(6):
        got = run_fourstep(g, p4, x, batch, inverse=False, rns=bool(it & 1))
        assert np.array_equal(got, want), ("forward", bits, logn, it)
        back = run_fourstep(g, p4, xin, batch, inverse=True, rns=bool(it & 1))
        assert np.array_equal(back, x), ("inverse", bits, logn, it)
        # the natural-order extension takes the one-launch path for the same rings (transposition on the spectrum side)
        d_in = g.to_device(x)
        d_out = torch.zeros_like(d_in)
        g.GPU_4STEP_NTT_NaturalOrder(d_in, d_out, *tf, p4.modulus, cf, batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_out), want), ("natural forward", bits, logn, it)
        d_back = torch.zeros_like(d_out)
        g.GPU_4STEP_NTT_NaturalOrder(d_out, d_back, *ti, p4.modulus, ci, batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_back), x), ("natural inverse", bits, logn, it)


@pytest.mark.parametrize("bits,logn,batch", [(64, 13, 5), (64, 14, 7), (64, 15, 40), (32, 16, 24), (64, 16, 9), (64, 17, 12),
                                             (32, 20, 5), (64, 21, 3), (32, 22, 2), (32, 21, 3), (64, 22, 3)])
def test_inverse_in_merge_form_every_polynomial_repeatedly(g, bits, logn, batch):
    """The inverse 4-step above one tile is the ring's inverse Merge plan with the transposition on its FIRST pass
    (kern::fourstep_inv_first_lazy) and the remaining stages inside the n2-long rows: one partial contiguous pass for
    n2 = 256 / 512 (2^13 .. 2^16), one or two strided passes from 2^17.  Every polynomial of multi-tile batches, three
    times over, both overloads; FourStepPlan must agree with the drop-in call bit for bit."""
    import torch
    P = O.Port(bits)
    p4 = g.NTTParameters4Step(logn, bits)
    oprm = P.fourstep_params(logn)
    n = p4.n
    x = P.splitmix(2300 + logn + batch, 0, batch * n, p4.modulus.value)
    want = np.concatenate([P.fourstep_ntt(x[p * n:(p + 1) * n], oprm) for p in range(batch)])
    xin = np.concatenate([P.fourstep_intt_first_transpose(want[p * n:(p + 1) * n], oprm) for p in range(batch)])
    g.set_option("path", "fast-strict")
    try:
        for it in range(3):
            back = run_fourstep(g, p4, xin, batch, inverse=True, rns=bool(it & 1))
            assert np.array_equal(back, x), ("inverse", bits, logn, it)
        ti = [g.to_device(t) for t in p4.tables["inv"]]
        ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
        d_in = g.to_device(xin)
        d_a, d_b = torch.zeros_like(d_in), torch.zeros_like(d_in)
        g.GPU_4STEP_NTT(d_in, d_a, *ti, p4.modulus, ci, batch)
        plan = g.FourStepPlan(*ti, p4.modulus, ci, batch_hint=batch)
        assert plan.fast_path
        plan.execute(d_in, d_b, batch)
        torch.cuda.synchronize()
        assert torch.equal(d_a, d_b), ("plan == drop-in", bits, logn)
        plan.close()
    finally:
        g.set_option("path", "default")


def test_fourstep_u32_30_bit_modulus_on_the_big_tiles(g):
    """32-bit rings 2^20 .. 2^22 with a 30-bit prime: the pool primes of those rings lie below 2^29 and take the
    LIMIT = 8 family, so the default 32-bit family on the 16384-coefficient tiles (forward Merge plan behind the
    transposed gather; inverse: transposing first pass on the big tile + one strided pass, host::fourstep_inv_tile) is
    reached with a prime of our own.  Tables generated on the device; expected values from the Merge oracle through
    GPU_4STEP_NTT(transpose(x)) == MergeNTT(x) and transpose(GPU_4STEP_NTT(y, INVERSE)) == x."""
    import torch
    from gpu_utils import find_ntt_factors
    P = O.Port(32)
    g.set_option("path", "fast-strict")
    try:
        for logn, batch in ((20, 3), (21, 2), (22, 3)):
            q, omega, psi = find_ntt_factors(30, logn)
            m = g.Modulus(q, bits=32)
            assert m.bit == 30
            shape = g.NTTParameters4Step(logn, 32)
            n, n1, n2 = shape.n, shape.n1, shape.n2
            oprm = P.merge_params(logn, O.X_N_minus, (q, omega, psi))
            x = P.splitmix(5100 + logn, 0, batch * n, q)
            y = np.concatenate([P.merge_ntt(x[p * n:(p + 1) * n], oprm) for p in range(batch)])
            for inverse in (False, True):
                r = pow(omega, -1, q) if inverse else omega
                kind = g.INVERSE if inverse else g.FORWARD
                w = torch.zeros(n, dtype=torch.int32, device="cuda")
                t1 = torch.zeros(n1 >> 1, dtype=torch.int32, device="cuda")
                t2 = torch.zeros(n2 >> 1, dtype=torch.int32, device="cuda")
                g.GPU_Generate4StepW(w, r, m, logn, kind)
                g.GPU_GeneratePowerTable(t1, pow(r, n // n1, q), m, int(np.log2(n1)) - 1, True)
                g.GPU_GeneratePowerTable(t2, pow(r, n // n2, q), m, int(np.log2(n2)) - 1, True)
                cfg = g.ntt4step_configuration(n_power=logn, ntt_type=kind, mod_inverse=pow(n, -1, q) if inverse else 0)
                src = x.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1).copy() if not inverse else y
                d_in = g.to_device(src)
                d_out = torch.zeros_like(d_in)
                for it in range(2):
                    d_out.zero_()
                    g.GPU_4STEP_NTT(d_in, d_out, t1, t2, w, m, cfg, batch)
                    torch.cuda.synchronize()
                    got = g.to_host(d_out)
                    if not inverse:
                        assert np.array_equal(got, y), ("forward", logn, it)
                    else:
                        back = got.reshape(batch, n1, n2).transpose(0, 2, 1).reshape(-1)
                        assert np.array_equal(back, x), ("inverse", logn, it)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))


def test_fourstep_seeded_random_shapes_both_directions(g):
    """seeded random sweep of the 4-step entry points in Merge form: ring 2^12 .. 2^23, odd batch sizes (poly-minor block
    order from 2^20, reversed sweeps, ragged one-tile batches), u32 / u64, plain and RNS(1) overloads, forward and inverse,
    every polynomial against the oracle"""
    rng = np.random.default_rng(20260930)
    cases = [(int(rng.choice([32, 64])), int(logn), int(rng.integers(1, 8 if logn < 20 else 4)))
             for logn in list(range(12, 24)) + [14, 16, 18, 20, 21, 22]]
    for bits, logn, batch in cases:
        P = O.Port(bits)
        p4 = g.NTTParameters4Step(logn, bits)
        oprm = P.fourstep_params(logn)
        n = p4.n
        x = P.splitmix(9000 + 31 * logn + batch + bits, 0, batch * n, p4.modulus.value)
        want = np.concatenate([P.fourstep_ntt(x[p * n:(p + 1) * n], oprm) for p in range(batch)])
        rns = bool(rng.integers(0, 2))
        got = run_fourstep(g, p4, x, batch, inverse=False, rns=rns)
        assert np.array_equal(got, want), ("forward", bits, logn, batch, rns)
        xin = np.concatenate([P.fourstep_intt_first_transpose(want[p * n:(p + 1) * n], oprm) for p in range(batch)])
        back = run_fourstep(g, p4, xin, batch, inverse=True, rns=not rns)
        assert np.array_equal(back, x), ("inverse", bits, logn, batch, rns)
