"""-m gpu tests of the family prediction of drop-in RNS Merge calls (gpu-ntt_amd/csrc/prep.hip: rns_guess,
merge_ntt.hip: cast_wide_net).  The moduli of an RNS call live in device memory (reference GPU_NTT RNS overload,
src/lib/ntt_merge/ntt.cu:2560-2746), so the host predicts the lazy family from what the same stack needed before.
ADVICE r5: a stack nothing is known about must not fall into the preparation kernel's own fall-back (one polynomial per
block, stage by stage through global memory) when the ring is large -- that path takes ~40 ms per polynomial of 2^20."""
import time

import numpy as np
import pytest

from oracle import oracle as O
from gpu_utils import oracle_batch
from test_gpu_round5 import _rns_stack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(pkg):
    pkg.load_library()
    return pkg


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
        cases, d_fwd, d_inv = _rns_stack(g, 64, logn, widths, poly)
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
    cases, d_fwd, _ = _rns_stack(g, 64, logn, widths, O.X_N_plus)
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
    cases, d_fwd, _ = _rns_stack(g, 64, logn, [60, 60], O.X_N_plus)
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
