"""-m gpu parity tests: NTTPlan / FourStepPlan (extension: include/gpuntt/ntt_merge/ntt.cuh, ntt_4step/ntt_4step.cuh): prepared transforms against the oracle, graph capture without warm-up, argument errors, the 4-step plan at 2^24 by reference-build digest."""
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
