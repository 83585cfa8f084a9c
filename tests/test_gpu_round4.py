"""-m gpu tests added in round 4: the 4-step table contract and its validation option, the two plan / option
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


@pytest.mark.parametrize("bits,logn", [(64, 12), (64, 16), (32, 18)])
def test_fourstep_table_contract(g, bits, logn):
    """include/gpuntt/ntt_4step/ntt_4step.cuh, TABLE CONTRACT: the fast path derives its twiddles from n1_table and one
    row of W (row n1/2 forward) and never reads n2_table or the rest of W -- where the reference multiplies by
    W[address] element by element (src/lib/ntt_4step/ntt_4step.cu:1049-1058).
      * option off (default): a W matrix whose OTHER rows are garbage and a garbage n2_table give the same result as the
        consistent tables (the documented narrowing);
      * option validate_4step_tables on: consistent tables pass; random tables (what benchmark/bench_4step_ntt.cu:80-90
        passes) and the garbage-rows W are refused with std::invalid_argument -> ValueError;
      * the generic kernels (path = generic) read every table like the reference: there the garbage W changes the result."""
    P = O.Port(bits)
    p4 = g.NTTParameters4Step(logn, bits)
    oprm = P.fourstep_params(logn)
    batch = 2
    x = P.splitmix(31000 + logn, 0, batch * p4.n, p4.modulus.value)
    want = P.fourstep_ntt(x, oprm)
    t1, t2, w = p4.tables["fwd"]
    rng = np.random.default_rng(logn)
    w_bad = rng.integers(1, p4.modulus.value, size=w.size, dtype=np.uint64).astype(w.dtype)
    row = (p4.n1 // 2) * p4.n2
    w_bad[row:row + p4.n2] = w[row:row + p4.n2]  # the one row the fast path reads stays
    t2_bad = rng.integers(1, p4.modulus.value, size=t2.size, dtype=np.uint64).astype(t2.dtype)
    good = [g.to_device(t) for t in (t1, t2, w)]
    narrowed = [g.to_device(t) for t in (t1, t2_bad, w_bad)]
    random_w = [g.to_device(t) for t in (t1, t2, rng.integers(1, p4.modulus.value, size=w.size, dtype=np.uint64).astype(w.dtype))]
    try:
        g.set_option("validate_4step_tables", "0")
        assert np.array_equal(_fourstep_forward(g, p4, good, x, batch), want)
        assert np.array_equal(_fourstep_forward(g, p4, narrowed, x, batch), want), "fast path read more than the contract says"
        g.set_option("validate_4step_tables", "1")
        assert np.array_equal(_fourstep_forward(g, p4, good, x, batch), want)
        for name, tabs in (("garbage rows", narrowed), ("random W", random_w)):
            with pytest.raises(ValueError, match="4-step tables"):
                _fourstep_forward(g, p4, tabs, x, batch)
        # plans check once, in their constructor
        cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
        with pytest.raises(ValueError, match="4-step tables"):
            g.FourStepPlan(*random_w, p4.modulus, cf, batch_hint=batch)
        g.FourStepPlan(*good, p4.modulus, cf, batch_hint=batch).close()
        # inverse tables, consistent: accepted
        ti = [g.to_device(t) for t in p4.tables["inv"]]
        g.FourStepPlan(*ti, p4.modulus, g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv),
                       batch_hint=batch).close()
        g.set_option("validate_4step_tables", "0")
        g.set_option("path", "generic")
        assert np.array_equal(_fourstep_forward(g, p4, good, x, batch), want)
        assert not np.array_equal(_fourstep_forward(g, p4, narrowed, x, batch), want), \
            "the generic kernels are documented to read W element by element"
    finally:
        g.set_option("validate_4step_tables", "0")
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))


def test_u32_tile_option_with_fourstep_rings_below_2_18(g):
    """option u32_tile = 14 sent the forward 4-step of the 32-bit rings 2^15 .. 2^17 to a 16384-coefficient
    tile whose remaining low stages (9 .. 12) have no lazy-input kernel -- GPU_4STEP_NTT threw after its first pass had
    written `out`.  Those rings keep the 4096-coefficient tile now; drop-in and plan, every polynomial."""
    import torch
    import test_gpu_4step as T
    P = O.Port(32)
    g.set_option("u32_tile", "14")
    g.set_option("path", "fast-strict")
    try:
        for logn, batch in ((15, 9), (16, 5), (17, 3), (18, 3), (19, 2)):
            p4 = g.NTTParameters4Step(logn, 32)
            oprm = P.fourstep_params(logn)
            x = P.splitmix(32000 + logn, 0, batch * p4.n, p4.modulus.value)
            want = P.fourstep_ntt(x, oprm)
            assert np.array_equal(T.run_fourstep(g, p4, x, batch, inverse=False), want), logn
            tf = [g.to_device(t) for t in p4.tables["fwd"]]
            plan = g.FourStepPlan(*tf, p4.modulus, g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD), batch_hint=batch)
            assert plan.fast_path
            d_a = g.to_device(x)
            d_b = torch.zeros_like(d_a)
            g.GPU_Transpose(d_a, d_b, p4.n1, p4.n2, logn, batch)
            plan.execute(d_b, d_a, batch)
            g.GPU_Transpose(d_a, d_b, p4.n1, p4.n2, logn, batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d_b), want), ("plan", logn)
            plan.close()
            xin = P.fourstep_intt_first_transpose(want, oprm)
            assert np.array_equal(T.run_fourstep(g, p4, xin, batch, inverse=True), x), ("inverse", logn)
    finally:
        g.set_option("u32_tile", "0")
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))


def test_inverse_fourstep_plan_keeps_its_tile_when_the_option_changes(g):
    """an inverse FourStepPlan of the 64-bit ring 2^21 permutes its table for the 8192-coefficient tile;
    execute() re-read the live u64_big_tiles option and ran the 4096-tile kernels on that table after the option
    changed -- silently wrong.  The plan stores its tile now ("plans keep the choice made when they were created")."""
    import torch
    P = O.Port(64)
    logn, batch = 21, 2
    p4 = g.NTTParameters4Step(logn, 64)
    oprm = P.fourstep_params(logn)
    x = P.splitmix(33021, 0, batch * p4.n, p4.modulus.value)
    y = P.fourstep_ntt(x, oprm)
    xin = P.fourstep_intt_first_transpose(y, oprm)
    ti = [g.to_device(t) for t in p4.tables["inv"]]
    ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
    try:
        g.set_option("u64_big_tiles", "14")
        plan_big = g.FourStepPlan(*ti, p4.modulus, ci, batch_hint=batch)
        g.set_option("u64_big_tiles", "0")
        plan_small = g.FourStepPlan(*ti, p4.modulus, ci, batch_hint=batch)
        for opt in ("0", "13", "14"):
            g.set_option("u64_big_tiles", opt)
            for plan in (plan_big, plan_small):
                d_in = g.to_device(xin)
                d_out = torch.zeros_like(d_in)
                d_nat = torch.zeros_like(d_in)
                plan.execute(d_in, d_out, batch)
                g.GPU_Transpose(d_out, d_nat, p4.n1, p4.n2, logn, batch)
                torch.cuda.synchronize()
                assert np.array_equal(g.to_host(d_nat), x), (opt, plan is plan_big)
        plan_big.close()
        plan_small.close()
    finally:
        g.set_option("u64_big_tiles", "14")
