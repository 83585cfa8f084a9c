"""-m gpu parity tests for the Merge NTT: HIP path (through the C ABI) vs the oracle, the
golden fixtures and size-independent properties.  Mirrors the reference's GPU examples
(example/ntt_merge/test_merge_ntt.cu:46-341, test_merge_intt.cu:46-379): bit-exact equality."""
import json
import os

import numpy as np
import pytest

from gpu_utils import MergeCase, oracle_batch, sha
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(pkg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    if not os.path.exists(pkg.LIB_PATH):  # fresh checkout: hipcc is part of the image
        pkg.build_library()
    pkg.load_library()
    return pkg


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("poly", [O.X_N_plus, O.X_N_minus])
def test_forward_inverse_all_sizes(g, bits, poly):
    # every pass shape: single tile (1..12), strided + contiguous (13..20)
    for logn in range(1, 21):
        c = MergeCase(g, bits, logn, poly)
        batch = 5 if logn <= 14 else (3 if logn <= 17 else 1)
        x = c.random(batch, 1000 + logn)
        want = c.P.merge_ntt(x, c.oprm)
        got = c.gpu_forward(x, inplace=(logn % 2 == 0))
        assert np.array_equal(got, want), ("forward", bits, poly, logn)
        back = c.gpu_inverse(got, inplace=(logn % 2 == 1))
        assert np.array_equal(back, x), ("inverse", bits, poly, logn)
        # inverse of raw data against the oracle (not only as a round trip)
        assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True))


def test_u64_ring_2_14_in_one_big_tile(g):
    """forward 64-bit calls of >= 256 transforms of length 2^14 run in one 16384-coefficient tile
    per polynomial (one sweep); smaller batches and the inverse keep the two-pass plan"""
    for poly in (O.X_N_plus, O.X_N_minus):
        c = MergeCase(g, 64, 14, poly)
        for batch in (256, 300, 255):
            x = c.random(batch, 14000 + batch)
            y = c.gpu_forward(x, inplace=(batch == 300))
            for p in (0, 1, batch // 2, batch - 1):
                sl = slice(p * c.n, (p + 1) * c.n)
                assert np.array_equal(y[sl], c.P.merge_ntt(x[sl], c.oprm)), (poly, batch, p)
            assert np.array_equal(c.gpu_inverse(y, inplace=True), x)


def test_random_configurations(g):
    """seeded sweep over (word size, ring size, polynomial, batch, direction, in place, custom prime):
    the pass planner, tile choice and path heuristics see combinations the structured tests do not"""
    from gpu_utils import find_ntt_factors
    rng = np.random.default_rng(20260929)
    cache = {}
    for it in range(48):
        bits = int(rng.choice([32, 64]))
        logn = int(rng.integers(1, 19))
        poly = O.X_N_plus if rng.integers(0, 2) else O.X_N_minus
        batch = int(rng.integers(1, 12)) if logn <= 15 else int(rng.integers(1, 4))
        custom = bool(rng.integers(0, 3) == 0)
        key = (bits, logn, poly, custom)
        if key not in cache:
            factors = None
            if custom:
                qbits = int(rng.integers(max(logn + 3, 20), 31 if bits == 32 else 61))
                factors = find_ntt_factors(qbits, logn)
            cache[key] = MergeCase(g, bits, logn, poly, factors)
        c = cache[key]
        x = c.random(batch, 5000 + it)
        inverse, inplace = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        want = c.P.merge_ntt(x, c.oprm, inverse=inverse)
        got = c.gpu_inverse(x, inplace=inplace) if inverse else c.gpu_forward(x, inplace=inplace)
        assert np.array_equal(got, want), (it, bits, logn, poly, batch, inverse, inplace, custom, c.q)


@pytest.mark.parametrize("bits", [32, 64])
def test_ragged_batches_small_rings(g, bits):
    # tiles that hold several polynomials and a ragged tail (reference LowRing tail guard,
    # ntt.cu:27,107)
    for logn in (1, 2, 3, 4, 7, 9, 11):
        c = MergeCase(g, bits, logn, O.X_N_minus)
        for batch in (1, 2, 3, 17, (4096 >> logn) + 1):
            x = c.random(batch, 7 * logn + batch)
            assert np.array_equal(c.gpu_forward(x), c.P.merge_ntt(x, c.oprm))
            assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True))


@pytest.mark.parametrize("bits", [32, 64])
def test_edge_values(g, bits):
    # all-zero, all-(q-1), delta and constant polynomials
    c = MergeCase(g, bits, 13, O.X_N_plus)
    n, q = c.n, c.q
    rows = [np.zeros(n), np.full(n, q - 1), np.eye(1, n, 0)[0], np.eye(1, n, n - 1)[0] * (q - 1),
            np.ones(n)]
    x = np.concatenate([np.asarray(r, dtype=object) for r in rows]).astype(c.P.T)
    assert np.array_equal(c.gpu_forward(x), c.P.merge_ntt(x, c.oprm))
    assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True))


@pytest.mark.parametrize("bits", [32, 64])
def test_signed_input_and_centered_output(g, bits):
    # GPU_NTT<Data64s>: signed input in (-q/2, q/2] gives the same result as its residue
    # (test_merge_ntt.cu:185-341); GPU_INTT<Data64s>: centred output (test_merge_intt.cu:206-379)
    import torch
    for logn in (5, 12, 13, 14):  # 13: the 64-bit 8192-coefficient tile
        c = MergeCase(g, bits, logn, O.X_N_minus)
        q, n = c.q, c.n
        x = c.random(2, 31 + logn)
        sdt = np.int32 if bits == 32 else np.int64
        xs = np.where(x > q // 2, x.astype(object) - q, x.astype(object)).astype(sdt)
        d_in = g.to_device(xs)
        d_out = torch.zeros_like(d_in)
        g.GPU_NTT(d_in, d_out, c.fwd_dev, c.prm.modulus, c.cfg(), 2, dtype="s%d" % bits)
        torch.cuda.synchronize()
        want = c.P.merge_ntt(x, c.oprm)
        assert np.array_equal(g.to_host(d_out), want)
        d_back = torch.zeros_like(d_out)
        g.GPU_INTT(d_out, d_back, c.inv_dev, c.prm.modulus, c.cfg(True), 2, dtype="s%d" % bits)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d_back, signed=True), xs)


def _rns_setup(g, bits, logn, poly, factor_list):
    """RNS stack: table of modulus i at i << n_power (ntt.cu:672-678)."""
    cases = [MergeCase(g, bits, logn, poly, f) for f in factor_list]
    n = 1 << logn
    fwd = np.zeros(len(cases) * n, dtype=cases[0].P.T)
    inv = np.zeros_like(fwd)
    for i, c in enumerate(cases):
        sz = c.prm.root_of_unity_size
        fwd[i * n:i * n + sz] = c.prm.forward_table_device_order
        inv[i * n:i * n + sz] = c.prm.inverse_table_device_order
    mods = g.modulus_array_to_device([c.prm.modulus for c in cases], bits)
    ninv = g.to_device(np.array([c.prm.n_inv for c in cases], dtype=cases[0].P.T))
    return cases, g.to_device(fwd), g.to_device(inv), mods, ninv


def _small_prime_factors(P, logn, count):
    """`count` distinct NTT primes for any logn from the reference pools (4-step pool primes
    all have 2^13 | q-1) with omega/psi derived from the pool psi."""
    out = []
    for lg in (12, 13, 14, 15, 17, 19, 20, 21, 22, 23, 24):
        if lg < logn:
            continue
        prm = P.fourstep_params(lg, with_W=False)
        q = prm["mod"][0]
        if any(q == f[0] for f in out):
            continue
        psi = pow(prm["psi"], 1 << (lg - logn), q)
        out.append((q, psi * psi % q, psi))
        if len(out) == count:
            break
    assert len(out) == count
    return out


@pytest.mark.parametrize("bits", [32, 64])
def test_rns_multi_modulus(g, bits):
    # mod_count > 1 is never exercised by the reference's own examples; pinned via the oracle
    import torch
    P = O.Port(bits)
    for logn, batch, mc in ((3, 10, 3), (8, 21, 3), (11, 7, 2), (12, 6, 3), (14, 6, 3)):
        for poly in (O.X_N_plus, O.X_N_minus):
            cases, fwd, inv, mods, ninv = _rns_setup(g, bits, logn, poly,
                                                     _small_prime_factors(P, logn, mc))
            n = 1 << logn
            x = np.concatenate([cases[p % mc].P.splitmix(50 + p, 0, n, cases[p % mc].q)
                                for p in range(batch)])
            want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm)
                                   for p in range(batch)])
            cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
            d = g.to_device(x)
            g.GPU_NTT_Inplace(d, fwd, mods, cfg, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), want), (bits, logn, poly)
            icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly,
                                           mod_inverse=ninv)
            g.GPU_INTT_Inplace(d, inv, mods, icfg, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d), x), (bits, logn, poly)


def test_rns_c5_against_golden(g, golden_dir):
    # BASELINE config 5: 8 distinct 60-bit primes, log2N = 16; digests from the reference build
    import torch
    rns = json.load(open(os.path.join(golden_dir, "rns_c5.json")))
    fl = [(e["q"], e["omega"], e["psi"]) for e in rns["primes"]]
    for poly, tag in ((O.X_N_plus, "plus"), (O.X_N_minus, "minus")):
        cases, fwd, inv, mods, ninv = _rns_setup(g, 64, 16, poly, fl)
        n = 1 << 16
        x = np.concatenate([c.P.splitmix(e["seed_" + tag], 0, n, c.q)
                            for c, e in zip(cases, rns["primes"])] * 2)  # batch 16
        d = g.to_device(x)
        cfg = g.ntt_rns_configuration(n_power=16, reduction_poly=poly)
        out = torch.zeros_like(d)
        g.GPU_NTT(d, out, fwd, mods, cfg, 16, 8)
        torch.cuda.synchronize()
        y = g.to_host(out)
        for p in range(16):
            assert sha(y[p * n:(p + 1) * n]) == rns["primes"][p % 8]["sha_fwd_" + tag]
        icfg = g.ntt_rns_configuration(n_power=16, ntt_type=g.INVERSE, reduction_poly=poly,
                                       mod_inverse=ninv)
        g.GPU_INTT(d, out, inv, mods, icfg, 16, 8)
        torch.cuda.synchronize()
        z = g.to_host(out)
        for p in range(16):
            assert sha(z[p * n:(p + 1) * n]) == rns["primes"][p % 8]["sha_inv_" + tag]


def test_full_size_c5_properties(g, golden_dir):
    """BASELINE config 5 at its real batch (512 polynomials over 8 primes, N = 2^16, X^N+1), drop-in call and
    NTTPlan: the first 8 polynomials by the reference build's digests, sampled polynomials of every prime against
    the oracle, range per prime, the exact round trip, and linearity of the first half against the second."""
    import torch
    rns = json.load(open(os.path.join(golden_dir, "rns_c5.json")))
    fl = [(e["q"], e["omega"], e["psi"]) for e in rns["primes"]]
    cases, fwd, inv, mods, ninv = _rns_setup(g, 64, 16, O.X_N_plus, fl)
    n, batch, mc = 1 << 16, 512, 8
    P = cases[0].P
    x = np.concatenate([P.splitmix(rns["primes"][p]["seed_plus"] if p < mc else 7000 + p, 0, n, cases[p % mc].q)
                        for p in range(batch)])
    d = g.to_device(x)
    out = torch.zeros_like(d)
    cfg = g.ntt_rns_configuration(n_power=16, reduction_poly=O.X_N_plus)
    g.GPU_NTT(d, out, fwd, mods, cfg, batch, mc)
    torch.cuda.synchronize()
    y = g.to_host(out)
    for p in range(mc):
        assert sha(y[p * n:(p + 1) * n]) == rns["primes"][p]["sha_fwd_plus"], p
    # EVERY polynomial against NTTCPU::ntt, and GPU_INTT of the raw data against NTTCPU::intt (a
    # position-dependent wrong-twiddle bug is linear and invertible by the matching inverse plan)
    assert np.array_equal(y, oracle_batch(cases, x)), "C5 forward: some polynomial differs from the oracle"
    yr = y.reshape(batch, n)
    for i, c in enumerate(cases):
        assert int(yr[i::mc].max()) < c.q
    plan = g.NTTPlan(fwd, [c.prm.modulus for c in cases], 16, O.X_N_plus, g.FORWARD, batch_hint=batch)
    out2 = torch.zeros_like(d)
    plan.execute(d, out2, batch)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(out2), y)
    plan.close()
    # linearity: polynomials p and p + 256 share a prime
    h = batch // 2
    qs = np.tile(np.array([c.q for c in cases], dtype=np.uint64), h // mc)  # 60-bit primes: a + b < 2^61, no wrap
    a, b = x[:h * n].reshape(h, n), x[h * n:].reshape(h, n)
    s = ((a + b) % qs[:, None]).reshape(-1)
    ds = g.to_device(s)
    g.GPU_NTT_Inplace(ds, fwd, mods, cfg, h, mc)
    torch.cuda.synchronize()
    ys = g.to_host(ds).reshape(h, n)
    for p in (0, 1, 7, 100, 255):
        want = (yr[p] + yr[p + h]) % qs[p]
        assert np.array_equal(ys[p], want), p
    icfg = g.ntt_rns_configuration(n_power=16, ntt_type=g.INVERSE, reduction_poly=O.X_N_plus, mod_inverse=ninv)
    g.GPU_INTT_Inplace(out, inv, mods, icfg, batch, mc)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(out), x)
    g.GPU_INTT(d, out, inv, mods, icfg, batch, mc)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(out), oracle_batch(cases, x, inverse=True)), "C5 inverse: some polynomial differs"


@pytest.mark.parametrize("bits", [32, 64])
def test_golden_vectors(g, bits, golden_dir):
    gold = np.load(os.path.join(golden_dir, "merge_u%d.npz" % bits))
    recs = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["merge"]
            if r["bits"] == bits]
    assert len(recs) == 26  # 2^1 .. 2^20 (round 1) + the Merge 2^24 records SURVEY.md 8c asks for (round 4)
    assert any(r["logn"] == 24 for r in recs)
    for r in recs:
        c = MergeCase(g, bits, r["logn"], r["poly"])
        x = c.P.splitmix(r["seed"], 0, r["batch"] * c.n, r["q"])
        assert sha(c.prm.forward_table_device_order) == r["sha_fwd_gpu_table"]
        fwd, inv = c.gpu_forward(x), c.gpu_inverse(x)
        assert sha(fwd) == r["sha_fwd"] and sha(inv) == r["sha_inv"], r
        key = "p%d_l%d" % (r["poly"], r["logn"])
        if key + "_fwd" in gold:
            assert np.array_equal(fwd, gold[key + "_fwd"]) and np.array_equal(inv, gold[key + "_inv"])


def test_reference_example_stream_known_answer(g, golden_dir):
    # the mt19937(0) input of example/ntt_merge/test_merge_ntt.cu:70-96 (needs the libstdc++
    # stream from oracle/_ref; digests were produced by the reference's NTTCPU)
    if not O.have_ref():
        pytest.skip("oracle/_ref not present")
    R = O.Ref(64)
    for r in json.load(open(os.path.join(golden_dir, "digests.json")))["mt19937"]:
        c = MergeCase(g, 64, r["logn"], O.X_N_minus)
        x = R.mt19937_uniform(0, r["q"], c.n)
        y = c.gpu_forward(x, inplace=True)
        assert [int(v) for v in y[:4]] == r["first_out"] and sha(y) == r["sha_out"]


def test_full_size_c2_properties(g):
    """BASELINE config 2 (u64, N = 2^16, batch = 1024) at full size: sampled polynomials
    against the oracle, the exact round trip, and linearity NTT(a)+NTT(b) == NTT(a+b)."""
    import torch
    c = MergeCase(g, 64, 16, O.X_N_minus)
    batch, n, q = 1024, c.n, c.q
    x = c.random(batch, 0x5EED0002)
    d = g.to_device(x)
    out = torch.empty_like(d)
    g.GPU_NTT(d, out, c.fwd_dev, c.prm.modulus, c.cfg(), batch)
    torch.cuda.synchronize()
    y = g.to_host(out)
    # all 1024 polynomials against NTTCPU::ntt
    assert np.array_equal(y, oracle_batch([c], x)), "C2 forward: some polynomial differs from the oracle"
    assert int(y.max()) < q
    # linearity over the first 512 vs the last 512 polynomials
    a, b = x[:512 * n], x[512 * n:]
    s = ((a.astype(object) + b.astype(object)) % q).astype(np.uint64)
    ys = c.gpu_forward(s)
    want = ((y[:512 * n].astype(object) + y[512 * n:].astype(object)) % q).astype(np.uint64)
    assert np.array_equal(ys, want)
    g.GPU_INTT_Inplace(out, c.inv_dev, c.prm.modulus, c.cfg(True), batch)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(out), x)
    g.GPU_INTT(d, out, c.inv_dev, c.prm.modulus, c.cfg(True), batch)  # the raw data as a spectrum: all 1024 against NTTCPU::intt
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(out), oracle_batch([c], x, inverse=True)), "C2 inverse: some polynomial differs"


def test_full_size_c4_properties(g):
    """BASELINE config 4 (u32, N = 2^14): one GPU's shard (batch 1024) and the whole batch (8192) at
    full size -- sampled polynomials against the oracle, range, exact round trip, linearity."""
    import torch
    c = MergeCase(g, 32, 14, O.X_N_minus)
    n, q = c.n, c.q
    for batch in (1024, 8192):
        x = c.random(batch, 0x5EED0004 + batch)
        d = g.to_device(x)
        out = torch.empty_like(d)
        g.GPU_NTT(d, out, c.fwd_dev, c.prm.modulus, c.cfg(), batch)
        torch.cuda.synchronize()
        y = g.to_host(out)
        assert np.array_equal(y, oracle_batch([c], x)), ("C4 forward: some polynomial differs from the oracle", batch)
        assert int(y.max()) < q
        h = batch // 2
        s = ((x[:h * n].astype(np.uint64) + x[h * n:].astype(np.uint64)) % np.uint64(q)).astype(np.uint32)
        ys = c.gpu_forward(s)
        want = ((y[:h * n].astype(np.uint64) + y[h * n:].astype(np.uint64)) % np.uint64(q)).astype(np.uint32)
        assert np.array_equal(ys, want)
        g.GPU_INTT_Inplace(out, c.inv_dev, c.prm.modulus, c.cfg(True), batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(out), x)
        g.GPU_INTT(d, out, c.inv_dev, c.prm.modulus, c.cfg(True), batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(out), oracle_batch([c], x, inverse=True)), ("C4 inverse", batch)


def test_streams_are_honoured(g):
    import torch
    c = MergeCase(g, 64, 14, O.X_N_minus)
    x = c.random(8, 77)
    s = torch.cuda.Stream()
    d = g.to_device(x)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus, c.cfg(stream=s), 8)
    s.synchronize()
    assert np.array_equal(g.to_host(d), c.P.merge_ntt(x, c.oprm))


def test_error_behaviour(g):
    # std::invalid_argument("Invalid n_power range!") / ("Invalid ntt_layout!")
    # (reference ntt.cu:2088-2091, 2252-2254) surface as ValueError through the C ABI
    c = MergeCase(g, 64, 4, O.X_N_minus)
    d = g.to_device(c.random(1, 1))
    for bad in (0, 29, -3):
        cfg = g.ntt_configuration(n_power=bad)
        with pytest.raises(ValueError, match="Invalid n_power range!"):
            g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus, cfg, 1)
    cfg = g.ntt_configuration(n_power=4, ntt_layout=7)
    with pytest.raises(ValueError, match="Invalid ntt_layout!"):
        g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus, cfg, 1)
    # batch 0 is a no-op
    g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus, c.cfg(), 0)


def _run_in_subprocess(code, env_path):
    """the path override is read once per process, so A/B runs need their own interpreter"""
    import subprocess
    import sys
    extra = env_path if isinstance(env_path, dict) else {"GPUNTT_PATH": env_path}
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.abspath(__file__)), **extra)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


_SWEEP = r'''
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.environ["PYTHONPATH"]), ""))
from conftest import load_pkg
from gpu_utils import MergeCase
from oracle import oracle as O
g = load_pkg(); g.load_library()
for bits, poly in ((64, O.X_N_plus), (64, O.X_N_minus), (32, O.X_N_plus), (32, O.X_N_minus)):
    for logn in range(1, 21):
        c = MergeCase(g, bits, logn, poly)
        batch = 3 if logn <= 17 else 2
        x = c.random(batch, 4242 + logn)
        want = c.P.merge_ntt(x, c.oprm)
        got = c.gpu_forward(x, inplace=bool(logn & 1))
        assert np.array_equal(got, want), ("fwd", bits, poly, logn)
        assert np.array_equal(c.gpu_inverse(got, inplace=not (logn & 1)), x), ("inv", bits, poly, logn)
        assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True)), ("inv raw", bits, poly, logn)
# user prime with other limb structure (4-step pool prime, 60 bit) and a small 31-bit prime in u64
for f, logn in (((576460752303415297, 288482366111684746, 238394956950829), 12),):
    c = MergeCase(g, 64, logn, O.X_N_plus, f)
    x = c.random(4, 9)
    assert np.array_equal(c.gpu_forward(x), c.P.merge_ntt(x, c.oprm))
    assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True))
print("SWEEP-OK")
'''


@pytest.mark.parametrize("path", ["fast", "generic"])
def test_both_kernel_paths_all_sizes(g, path):
    """The 64-bit fast path (lazy residues, prepared Shoup twiddles) and the generic Barrett
    path must each be bit-exact for every ring size, regardless of the size heuristic."""
    assert "SWEEP-OK" in _run_in_subprocess(_SWEEP, path)


def test_moduli_without_lazy_headroom(g):
    """61- and 62-bit primes (the reference's documented Data64 limit, modular_arith.cuh:66-67)
    run the exact-Barrett branch of the fast kernels; an RNS stack mixing 60/61/62-bit moduli
    exercises the per-block switch."""
    import torch
    from gpu_utils import find_ntt_factors
    for bits in (61, 62):
        for logn in (12, 14):
            f = find_ntt_factors(bits, logn)
            assert f[0].bit_length() == bits
            for poly in (O.X_N_plus, O.X_N_minus):
                c = MergeCase(g, 64, logn, poly, f)
                assert c.prm.modulus.bit == bits
                x = c.random(4, bits + logn)
                want = c.P.merge_ntt(x, c.oprm)
                assert np.array_equal(c.gpu_forward(x), want)
                assert np.array_equal(c.gpu_inverse(want, inplace=True), x)
    logn, batch = 13, 9
    fl = [find_ntt_factors(62, logn), find_ntt_factors(60, logn), find_ntt_factors(61, logn)]
    cases, fwd, inv, mods, ninv = _rns_setup(g, 64, logn, O.X_N_plus, fl)
    n = 1 << logn
    x = np.concatenate([cases[p % 3].P.splitmix(70 + p, 0, n, cases[p % 3].q) for p in range(batch)])
    want = np.concatenate([cases[p % 3].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % 3].oprm)
                           for p in range(batch)])
    d = g.to_device(x)
    g.GPU_NTT_Inplace(d, fwd, mods, g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus),
                      batch, 3)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(d), want)
    g.GPU_INTT_Inplace(d, inv, mods,
                       g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE,
                                               reduction_poly=O.X_N_plus, mod_inverse=ninv), batch, 3)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(d), x)


def test_rns_generic_kernels_walk_tiles_with_capped_grid(g):
    """An RNS call whose moduli lack the lazy headroom is served by the generic kernels, which
    for RNS calls are launched with at most 1024 blocks that walk the tiles (so that their usual
    role -- a skipped shadow launch -- is cheap): more tiles than blocks, both directions."""
    import torch
    from gpu_utils import find_ntt_factors
    logn, batch = 13, 1200  # 2400 tiles
    fl = [find_ntt_factors(62, logn), find_ntt_factors(60, logn), find_ntt_factors(61, logn)]
    cases, fwd, inv, mods, ninv = _rns_setup(g, 64, logn, O.X_N_plus, fl)
    n = 1 << logn
    x = np.concatenate([cases[p % 3].P.splitmix(170 + p, 0, n, cases[p % 3].q) for p in range(batch)])
    d = g.to_device(x)
    # round 4: 61- / 62-bit primes run the 4 q lazy family now, so no modulus of the documented domain reaches the generic
    # kernels through the go-flag; option path = generic-capped puts the call into exactly that state
    g.set_option("path", "generic-capped")
    try:
        g.GPU_NTT_Inplace(d, fwd, mods, g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus),
                          batch, 3)
        torch.cuda.synchronize()
        y = g.to_host(d)
        for p in list(range(0, batch, 97)) + [batch - 1]:
            c = cases[p % 3]
            assert np.array_equal(y[p * n:(p + 1) * n], c.P.merge_ntt(x[p * n:(p + 1) * n], c.oprm)), p
        g.GPU_INTT_Inplace(d, inv, mods,
                           g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE,
                                                   reduction_poly=O.X_N_plus, mod_inverse=ninv), batch, 3)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), x)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))


@pytest.mark.parametrize("bits", [32, 64])
def test_polymul_vs_schoolbook_and_oracle(g, bits):
    """extension GPU_PolyMul = INTT(NTT(a) (.) NTT(b)): against schoolbook multiplication in the
    ring (what example/ntt_merge/test_cpu_merge_ntt.cu:69-101 checks on the CPU) for small rings,
    against the oracle's ntt / pointwise / intt composition for larger ones; cyclic and negacyclic"""
    import torch
    for logn, batch in ((4, 5), (9, 3), (12, 2), (14, 3)):
        for poly in (O.X_N_plus, O.X_N_minus):
            c = MergeCase(g, bits, logn, poly)
            a = c.random(batch, 900 + logn)
            b = c.random(batch, 950 + logn)
            n = c.n
            if logn <= 9:
                want = np.concatenate([c.P.schoolbook(a[i * n:(i + 1) * n], b[i * n:(i + 1) * n], poly, c.oprm["mod"])
                                       for i in range(batch)])
            else:
                want = c.P.merge_ntt(c.P.pointwise(c.P.merge_ntt(a, c.oprm), c.P.merge_ntt(b, c.oprm),
                                                   c.oprm["mod"]), c.oprm, inverse=True)
            da, db = g.to_device(a), g.to_device(b)
            out = torch.zeros_like(da)
            g.GPU_PolyMul(da, db, out, c.fwd_dev, c.inv_dev, c.prm.modulus, c.cfg(inverse=True), batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(out), want), (bits, logn, poly)
            # in place on a
            da, db = g.to_device(a), g.to_device(b)
            g.GPU_PolyMul(da, db, da, c.fwd_dev, c.inv_dev, c.prm.modulus, c.cfg(inverse=True), batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(da), want)
            # in place on b
            da, db = g.to_device(a), g.to_device(b)
            g.GPU_PolyMul(da, db, db, c.fwd_dev, c.inv_dev, c.prm.modulus, c.cfg(inverse=True), batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(db), want)


def test_polymul_rns(g):
    import torch
    logn, batch = 13, 7
    from gpu_utils import find_ntt_factors
    fl = [find_ntt_factors(58, logn), find_ntt_factors(60, logn), find_ntt_factors(62, logn)]
    cases, fwd, inv, mods, ninv = _rns_setup(g, 64, logn, O.X_N_plus, fl)
    n = 1 << logn
    a = np.concatenate([cases[p % 3].P.splitmix(270 + p, 0, n, cases[p % 3].q) for p in range(batch)])
    b = np.concatenate([cases[p % 3].P.splitmix(370 + p, 0, n, cases[p % 3].q) for p in range(batch)])
    want = []
    for p in range(batch):
        c = cases[p % 3]
        fa, fb = c.P.merge_ntt(a[p * n:(p + 1) * n], c.oprm), c.P.merge_ntt(b[p * n:(p + 1) * n], c.oprm)
        want.append(c.P.merge_ntt(c.P.pointwise(fa, fb, c.oprm["mod"]), c.oprm, inverse=True))
    da, db = g.to_device(a), g.to_device(b)
    out = torch.zeros_like(da)
    cfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=O.X_N_plus, mod_inverse=ninv)
    g.GPU_PolyMul(da, db, out, fwd, inv, mods, cfg, batch, 3)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(out), np.concatenate(want))


def test_small_and_odd_sized_moduli_u64(g):
    """64-bit words holding small primes (14..50 bit): exercises the shift-free branch of the
    one-multiply final normalisation and the lazy bounds far from the 60-bit case."""
    from gpu_utils import find_ntt_factors
    for bits, logn in ((14, 10), (20, 13), (27, 13), (28, 14), (31, 12), (40, 13), (50, 16), (59, 13)):
        f = find_ntt_factors(bits, logn)
        for poly in (O.X_N_plus, O.X_N_minus):
            c = MergeCase(g, 64, logn, poly, f)
            x = c.random(8, bits * 3 + logn)
            want = c.P.merge_ntt(x, c.oprm)
            assert np.array_equal(c.gpu_forward(x), want), (bits, logn, poly)
            assert np.array_equal(c.gpu_inverse(want), x), (bits, logn, poly)


@pytest.mark.parametrize("bits", [32, 64])
def test_percoefficient_layout(g, bits):
    """Column-wise transforms (cfg.ntt_layout = PerCoefficient): the N x batch matrix is transformed
    along its columns in place.  The reference only checks this GPU-vs-GPU against the transposed
    PerPolynomial result (test_merge_ntt.cu:343-474, test_merge_intt.cu:381-515); here it is pinned
    to the oracle: column x of the output == NTTCPU::ntt(column x of the input)."""
    import torch
    for logn, w, poly in ((9, 1024, O.X_N_plus), (7, 128, O.X_N_minus), (4, 256, O.X_N_plus),
                          (9, 256, O.X_N_minus), (5, 16, O.X_N_plus), (3, 2, O.X_N_minus), (8, 64, O.X_N_plus),
                          (1, 1, O.X_N_minus), (9, 8, O.X_N_plus)):
        c = MergeCase(g, bits, logn, poly)
        n = c.n
        cols = c.random(w, 900 + logn + w).reshape(w, n)            # row p = polynomial p
        mat = np.ascontiguousarray(cols.T)                          # N x W, element (coef, poly)
        want_f = np.ascontiguousarray(c.P.merge_ntt(cols.reshape(-1), c.oprm).reshape(w, n).T)
        want_i = np.ascontiguousarray(c.P.merge_ntt(cols.reshape(-1), c.oprm, inverse=True).reshape(w, n).T)
        cfg = g.ntt_configuration(n_power=logn, ntt_layout=g.PerCoefficient, reduction_poly=poly)
        d = g.to_device(mat.reshape(-1))
        o = torch.zeros_like(d)
        g.GPU_NTT(d, o, c.fwd_dev, c.prm.modulus, cfg, w)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(o).reshape(n, w), want_f), ("fwd", bits, logn, w)
        icfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE, ntt_layout=g.PerCoefficient,
                                   reduction_poly=poly, mod_inverse=c.prm.n_inv)
        g.GPU_INTT_Inplace(d, c.inv_dev, c.prm.modulus, icfg, w)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d).reshape(n, w), want_i), ("inv", bits, logn, w)
    c = MergeCase(g, bits, 4, O.X_N_minus)
    d = g.to_device(c.random(3, 1))
    with pytest.raises(ValueError, match="power of two"):
        g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus,
                          g.ntt_configuration(n_power=4, ntt_layout=g.PerCoefficient), 3)
    with pytest.raises(ValueError, match="Invalid n_power range!"):
        g.GPU_NTT_Inplace(d, c.fwd_dev, c.prm.modulus,
                          g.ntt_configuration(n_power=10, ntt_layout=g.PerCoefficient), 4)


def _prime_stack(g, bits, logn, poly, count):
    """`count` primes with their tables stacked at i << n_power (+ device moduli / n^-1 arrays)."""
    from gpu_utils import find_ntt_factors
    fl = [find_ntt_factors(60 if bits == 64 else 30, logn, skip=i) for i in range(count)]
    return _rns_setup(g, bits, logn, poly, fl)


@pytest.mark.parametrize("bits", [32, 64])
def test_modulus_ordered_and_poly_ordered(g, bits):
    """GPU_NTT_Modulus_Ordered / GPU_NTT_Poly_Ordered (reference ntt.cuh:495-603; no reference test
    exists, pinned per polynomial through the oracle)."""
    import torch
    for logn, poly in ((10, O.X_N_plus), (12, O.X_N_minus), (13, O.X_N_plus), (15, O.X_N_minus)):
        n = 1 << logn
        cases, fwd, inv, mods, ninv = _prime_stack(g, bits, logn, poly, 6)
        # ---- modulus ordered: primes 1, 3, 4 of the 6-prime stack, batch 7
        order = [1, 3, 4]
        d_order = torch.tensor(order, dtype=torch.int32, device="cuda")
        batch, mc = 7, 3
        prime_of = [order[p % mc] for p in range(batch)]
        x = np.concatenate([cases[pr].P.splitmix(300 + p, 0, n, cases[pr].q) for p, pr in enumerate(prime_of)])
        want = np.concatenate([cases[pr].P.merge_ntt(x[p * n:(p + 1) * n], cases[pr].oprm)
                               for p, pr in enumerate(prime_of)])
        d = g.to_device(x)
        o = torch.zeros_like(d)
        cfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=poly)
        g.GPU_NTT_Modulus_Ordered(d, o, fwd, mods, cfg, batch, mc, d_order)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(o), want), ("mod-ordered fwd", bits, logn)
        icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
        g.GPU_NTT_Modulus_Ordered(o, o, inv, mods, icfg, batch, mc, d_order)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(o), x), ("mod-ordered inv", bits, logn)

        # ---- poly ordered: 8 of 10 stored polynomials, 4 moduli (reference example 3 pattern)
        slots = [9, 4, 8, 3, 0, 6, 1, 2]
        d_slots = torch.tensor(slots, dtype=torch.int32, device="cuda")
        batch, mc, stored = 8, 4, 10
        buf = np.zeros(stored * n, dtype=cases[0].P.T)
        for p, sl in enumerate(slots):
            buf[sl * n:(sl + 1) * n] = cases[p % mc].P.splitmix(500 + p, 0, n, cases[p % mc].q)
        for sl in (5, 7):  # untouched slots keep arbitrary content
            buf[sl * n:(sl + 1) * n] = 12345
        want = buf.copy()
        for p, sl in enumerate(slots):
            want[sl * n:(sl + 1) * n] = cases[p % mc].P.merge_ntt(buf[sl * n:(sl + 1) * n], cases[p % mc].oprm)
        d = g.to_device(buf)
        g.GPU_NTT_Poly_Ordered(d, d, fwd, mods, cfg, batch, mc, d_slots)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), want), ("poly-ordered fwd", bits, logn)
        g.GPU_NTT_Poly_Ordered(d, d, inv, mods, icfg, batch, mc, d_slots)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), buf), ("poly-ordered inv", bits, logn)
    with pytest.raises(ValueError, match="Invalid n_power range!"):
        g.GPU_NTT_Modulus_Ordered(d, d, fwd, mods, g.ntt_rns_configuration(n_power=9), 1, 1, d_order)


@pytest.mark.parametrize("bits,logn", [(64, 21), (64, 22), (64, 24), (32, 21), (32, 23), (64, 25), (32, 25), (64, 26)])
def test_large_rings(g, bits, logn):
    """Two-sweep plans with a big contiguous tile (u64 2^21 both directions, 2^22 forward), three-sweep
    plans (up to 2^24, fast path) and rings above the fast path's table limit
    (2^25+, generic kernels; the reference needs its grid-swapped ForwardCore_ there,
    ntt.cu:763-1084).  Batch 2 so the fast path is eligible where it applies."""
    c = MergeCase(g, bits, logn, O.X_N_minus if logn % 2 else O.X_N_plus)
    x = c.random(2, 77 + logn)
    y = c.gpu_forward(x, inplace=True)
    n = c.n
    assert np.array_equal(y[n:], c.P.merge_ntt(x[n:], c.oprm))
    assert np.array_equal(c.gpu_inverse(y, inplace=True), x)


def test_hip_graph_capture_and_replay(g):
    """A whole GPU_NTT + GPU_INTT pair (prep kernel + tile passes) captured into a hipGraph on a
    side stream and replayed; the twiddle workspace of that stream is created by one eager call
    first (the only allocation the library ever makes)."""
    import torch
    c = MergeCase(g, 64, 14, O.X_N_plus)
    batch = 16
    x = c.random(batch, 4321)
    s = torch.cuda.Stream()
    d = g.to_device(x)
    o = torch.zeros_like(d)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        g.GPU_NTT(d, o, c.fwd_dev, c.prm.modulus, c.cfg(stream=s), batch)  # warms the workspace
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        cs = torch.cuda.current_stream()
        g.GPU_NTT(d, o, c.fwd_dev, c.prm.modulus, c.cfg(stream=cs), batch)
        g.GPU_INTT_Inplace(o, c.inv_dev, c.prm.modulus, c.cfg(True, stream=cs), batch)
    o.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(o), x)


def test_two_streams_concurrently(g):
    """Independent calls on two streams (each with its own twiddle workspace) interleaved from one
    host thread, different rings and moduli."""
    import torch
    ca, cb = MergeCase(g, 64, 16, O.X_N_minus), MergeCase(g, 32, 15, O.X_N_plus)
    xa, xb = ca.random(32, 1), cb.random(64, 2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    da, db = g.to_device(xa), g.to_device(xb)
    torch.cuda.synchronize()
    for _ in range(4):
        g.GPU_NTT_Inplace(da, ca.fwd_dev, ca.prm.modulus, ca.cfg(stream=sa), 32)
        g.GPU_NTT_Inplace(db, cb.fwd_dev, cb.prm.modulus, cb.cfg(stream=sb), 64)
        g.GPU_INTT_Inplace(da, ca.inv_dev, ca.prm.modulus, ca.cfg(True, stream=sa), 32)
        g.GPU_INTT_Inplace(db, cb.inv_dev, cb.prm.modulus, cb.cfg(True, stream=sb), 64)
    g.GPU_NTT_Inplace(da, ca.fwd_dev, ca.prm.modulus, ca.cfg(stream=sa), 32)
    g.GPU_NTT_Inplace(db, cb.fwd_dev, cb.prm.modulus, cb.cfg(stream=sb), 64)
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(da), ca.P.merge_ntt(xa, ca.oprm))
    assert np.array_equal(g.to_host(db), cb.P.merge_ntt(xb, cb.oprm))
