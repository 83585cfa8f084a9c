"""Pins oracle/ntt_oracle.c (the C restatement) to the reference.

(1) bit-exact against oracle/_ref/libgpuntt_ref.so -- the reference's own CPU classes
    compiled from /root/reference (skipped only if that prebuilt file is absent);
(2) bit-exact against the committed fixtures in tests/golden/ (generated from that build
    by tools/make_golden.py), which always run.
Also restates the reference's CPU self-consistency examples
(example/ntt_merge/test_cpu_merge_ntt.cu:69-101, example/ntt_4step/test_cpu_4step_ntt.cu:40-79):
INTT(NTT(a) .* NTT(b)) == schoolbook(a*b).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@needs_ref
@pytest.mark.parametrize("bits", [32, 64])
def test_barrett_scalar_ops(bits):
    P, R = O.Port(bits), O.Ref(bits)
    rng = np.random.default_rng(1)
    moduli = [469762049, 268460033, 12289, 7681] if bits == 32 else \
        [576460756061519873, 576460752303415297, 288230377292562433, (1 << 61) + 20 * (1 << 32) + 1,
         469762049, 4611686018326724609]
    for q in moduli:
        mod = P.modulus(q)
        assert mod == R.modulus(q)
        vals = [0, 1, 2, q - 1, q - 2, q >> 1] + [int(v) % q for v in rng.integers(0, 2**63, 200)]
        for a in vals[:40]:
            for b in vals[::7]:
                assert P.mult(a, b, mod) == R.mult(a, b, q) == (a * b) % q
        for a in vals[2:12]:
            assert P.modinv(a, mod) == R.modinv(a, q)
            assert P.exp(a, 12345, mod) == R.exp(a, 12345, q) == pow(a, 12345, q)


@needs_ref
@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("poly", [O.X_N_plus, O.X_N_minus])
def test_merge_matches_reference_build(bits, poly):
    P, R = O.Port(bits), O.Ref(bits)
    for logn in list(range(1, 15)) + [16]:
        pp, rp = P.merge_params(logn, poly), R.merge_params(logn, poly)
        for k in ("mod", "omega", "psi", "n_inv", "root", "inv_root", "root_size"):
            assert pp[k] == rp[k], (k, logn)
        assert np.array_equal(pp["fwd"], rp["fwd"]) and np.array_equal(pp["inv"], rp["inv"])
        assert np.array_equal(P.bitrev_table(pp["fwd"]), rp["fwd_gpu"])
        assert np.array_equal(P.bitrev_table(pp["inv"]), rp["inv_gpu"])
        x = P.splitmix(99 + logn, 0, 2 * pp["n"], pp["mod"][0])
        y = P.merge_ntt(x, pp)
        assert np.array_equal(y, R.merge_ntt(x, rp))
        z = P.merge_ntt(y, pp, inverse=True)
        assert np.array_equal(z, R.merge_ntt(y, rp, True)) and np.array_equal(z, x)
        R.merge_free(rp)


@needs_ref
def test_merge_custom_factors_match_reference_build():
    # the user-prime constructor NTTParameters(LOGN, NTTFactors, poly), nttparameters.cu:51-82,
    # with the factors the reference examples use (test_cpu_merge_ntt.cu:45-47)
    P, R = O.Port(64), O.Ref(64)
    f = (576460752303415297, 288482366111684746, 238394956950829)
    for poly in (O.X_N_plus, O.X_N_minus):
        pp, rp = P.merge_params(12, poly, f), R.merge_params(12, poly, f)
        assert pp["mod"] == rp["mod"] and pp["n_inv"] == rp["n_inv"]
        assert np.array_equal(pp["fwd"], rp["fwd"])
        x = P.splitmix(5, 0, pp["n"], f[0])
        assert np.array_equal(P.merge_ntt(x, pp), R.merge_ntt(x, rp))
        R.merge_free(rp)


@needs_ref
@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_matches_reference_build(bits):
    P, R = O.Port(bits), O.Ref(bits)
    for logn in (12, 13, 14, 15, 16, 17):
        pp, rp = P.fourstep_params(logn), R.fourstep_params(logn)
        for k in ("mod", "omega", "psi", "n_inv", "n1", "n2"):
            assert pp[k] == rp[k], (k, logn)
        for k in ("n1_fwd", "n2_fwd", "W_fwd", "n1_inv", "n2_inv", "W_inv"):
            assert np.array_equal(pp[k], rp[k]), (k, logn)
        x = P.splitmix(3 + logn, 0, pp["n"], pp["mod"][0])
        y = P.fourstep_ntt(x, pp)
        assert np.array_equal(y, R.fourstep_run(x, rp, 0))
        z = P.fourstep_ntt(y, pp, inverse=True)
        assert np.array_equal(z, R.fourstep_run(y, rp, 1)) and np.array_equal(z, x)
        assert np.array_equal(P.fourstep_intt_first_transpose(x, pp), R.fourstep_run(x, rp, 2))
        R.fourstep_free(rp)


@needs_ref
@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_on_caller_supplied_tables_matches_reference_build(bits):
    """NTT_4STEP_CPU computes whatever its parameter set's public tables say (ntt_4step_cpu.cu:33-111) -- the CPU-side
    meaning of a GPU_4STEP_NTT call with arbitrary tables (ntt_4step.cu:1049-1058, 776-779).  The restatement on
    caller-supplied tables (Port.fourstep_ntt_tables) against the reference's own class with its six public table
    vectors overwritten (oracle/ref_driver.cpp: fourstep_run_tables): consistent tables, single corrupted words,
    random tables, a word that is not a residue, another modulus."""
    P, R = O.Port(bits), O.Ref(bits)
    dt = P.T
    for logn in (12, 13, 15):
        pp, rp = P.fourstep_params(logn), R.fourstep_params(logn)
        q, n, n1, n2 = pp["mod"][0], pp["n"], pp["n1"], pp["n2"]
        rng = np.random.default_rng(900 + logn + bits)
        x = P.splitmix(4400 + logn, 0, 2 * n, q)
        for inverse in (False, True):
            tag = "inv" if inverse else "fwd"
            t1, t2, w = pp["n1_" + tag], pp["n2_" + tag], pp["W_" + tag]
            assert np.array_equal(t1, rp["n1_" + tag]) and np.array_equal(w, rp["W_" + tag])
            cases = {"own tables": (t1, t2, w)}
            wb = w.copy()
            wb[int(rng.integers(0, n))] = dt(int(rng.integers(0, q)))
            cases["one W word"] = (t1, t2, wb)
            cases["random W"] = (t1, t2, rng.integers(0, q, size=n, dtype=np.uint64).astype(dt))
            t1b, t2b = t1.copy(), t2.copy()
            t1b[int(rng.integers(0, t1.size))] = dt(int(rng.integers(0, q)))
            t2b[int(rng.integers(0, t2.size))] = dt(int(rng.integers(0, q)))
            cases["one word of each small table"] = (t1b, t2b, w)
            cases["everything random"] = tuple(rng.integers(0, q, size=t.size, dtype=np.uint64).astype(dt)
                                               for t in (t1, t2, w))
            wq = w.copy()
            wq[5] = dt(q)
            cases["a word equal to q"] = (t1, t2, wq)
            for name, (a, b, c) in cases.items():
                got = P.fourstep_ntt_tables(x, pp, a, b, c, inverse)
                want = R.fourstep_run_tables(x, rp, a, b, c, inverse)
                assert np.array_equal(got, want), (bits, logn, tag, name)
                if name == "own tables":
                    assert np.array_equal(got, P.fourstep_ntt(x, pp, inverse))
                    assert np.array_equal(want, R.fourstep_run(x, rp, 1 if inverse else 0))
            # another modulus with the same (now meaningless) tables reduced below it, n^-1 replaced as well
            q2 = 10007 if bits == 32 else 1000003
            x2 = (x % dt(q2)).astype(dt)
            tabs = tuple((t % dt(q2)).astype(dt) for t in (t1, t2, w))
            got = P.fourstep_ntt_tables(x2, pp, *tabs, inverse, q=q2, n_inv=77)
            want = R.fourstep_run_tables(x2, rp, *tabs, inverse, q=q2, n_inv=77)
            assert np.array_equal(got, want), (bits, logn, tag, "modulus %d" % q2)
        R.fourstep_free(rp)


@pytest.mark.parametrize("bits", [32, 64])
def test_merge_golden_full_vectors(bits, golden_dir):
    P = O.Port(bits)
    g = np.load(os.path.join(golden_dir, "merge_u%d.npz" % bits))
    recs = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["merge"]
            if r["bits"] == bits]
    seen = 0
    for r in recs:
        if r["logn"] > 16:
            continue
        prm = P.merge_params(r["logn"], r["poly"])
        assert prm["mod"] == (r["q"], r["bit"], r["mu"])
        assert (prm["omega"], prm["psi"], prm["n_inv"]) == (r["omega"], r["psi"], r["n_inv"])
        x = P.splitmix(r["seed"], 0, r["batch"] * prm["n"], r["q"])
        assert sha(x) == r["sha_in"]
        fg, ig = P.bitrev_table(prm["fwd"]), P.bitrev_table(prm["inv"])
        assert sha(fg) == r["sha_fwd_gpu_table"] and sha(ig) == r["sha_inv_gpu_table"]
        fwd, inv = P.merge_ntt(x, prm), P.merge_ntt(x, prm, inverse=True)
        assert sha(fwd) == r["sha_fwd"] and sha(inv) == r["sha_inv"]
        key = "p%d_l%d" % (r["poly"], r["logn"])
        if key + "_fwd" in g:
            assert np.array_equal(fwd, g[key + "_fwd"]) and np.array_equal(inv, g[key + "_inv"])
            assert np.array_equal(fg[:64], g[key + "_tabf"][:fg.size])
            seen += 1
    assert seen == 16


@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_golden(bits, golden_dir):
    P = O.Port(bits)
    g = np.load(os.path.join(golden_dir, "fourstep_u%d.npz" % bits))
    recs = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["fourstep"]
            if r["bits"] == bits and r["logn"] <= 17]
    assert len(recs) >= 6
    for r in recs:
        prm = P.fourstep_params(r["logn"])
        assert prm["mod"] == (r["q"], r["bit"], r["mu"]) and prm["n_inv"] == r["n_inv"]
        assert (prm["n1"], prm["n2"]) == (r["n1"], r["n2"])
        assert sha(prm["W_fwd"]) == r["sha_W_fwd"] and sha(prm["W_inv"]) == r["sha_W_inv"]
        assert sha(P.bitrev_table(prm["n2_fwd"])) == r["sha_n2_fwd_gpu"]
        x = P.splitmix(r["seed"], 0, prm["n"], r["q"])
        fwd, inv = P.fourstep_ntt(x, prm), P.fourstep_ntt(x, prm, inverse=True)
        assert sha(fwd) == r["sha_fwd"] and sha(inv) == r["sha_inv"]
        assert sha(P.fourstep_intt_first_transpose(x, prm)) == r["sha_first_transpose"]
        if "l%d_fwd" % r["logn"] in g:
            assert np.array_equal(fwd, g["l%d_fwd" % r["logn"]])
            assert np.array_equal(inv, g["l%d_inv" % r["logn"]])


def test_rns_c5_golden(golden_dir):
    P = O.Port(64)
    rns = json.load(open(os.path.join(golden_dir, "rns_c5.json")))
    assert len({e["q"] for e in rns["primes"]}) == 8
    for e in rns["primes"][::3]:
        for poly, tag in ((O.X_N_plus, "plus"), (O.X_N_minus, "minus")):
            prm = P.merge_params(16, poly, (e["q"], e["omega"], e["psi"]))
            x = P.splitmix(e["seed_" + tag], 0, prm["n"], e["q"])
            assert sha(P.bitrev_table(prm["fwd"])) == e["sha_tab_" + tag]
            assert sha(P.merge_ntt(x, prm)) == e["sha_fwd_" + tag]
            assert sha(P.merge_ntt(x, prm, True)) == e["sha_inv_" + tag]


@needs_ref
def test_reference_example_known_answers(golden_dir):
    # mt19937(0) stream of example/ntt_merge/test_merge_ntt.cu:70-96, values from the reference build
    R, P = O.Ref(64), O.Port(64)
    for r in json.load(open(os.path.join(golden_dir, "digests.json")))["mt19937"]:
        prm = P.merge_params(r["logn"], O.X_N_minus)
        x = R.mt19937_uniform(0, r["q"], prm["n"])
        assert [int(v) for v in x[:4]] == r["first_in"] and sha(x) == r["sha_in"]
        y = P.merge_ntt(x, prm)
        assert [int(v) for v in y[:4]] == r["first_out"] and sha(y) == r["sha_out"]


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("poly", [O.X_N_plus, O.X_N_minus])
def test_cpu_merge_polymul_vs_schoolbook(bits, poly):
    # example/ntt_merge/test_cpu_merge_ntt.cu:69-101 (the reference's own CPU example)
    P = O.Port(bits)
    prm = P.merge_params(9, poly)
    q = prm["mod"][0]
    a, b = P.splitmix(1, 0, prm["n"], q), P.splitmix(2, 0, prm["n"], q)
    c = P.merge_ntt(P.pointwise(P.merge_ntt(a, prm), P.merge_ntt(b, prm), prm["mod"]), prm, True)
    assert np.array_equal(c, P.schoolbook(a, b, poly, prm["mod"]))


def test_cpu_4step_polymul_vs_schoolbook():
    # example/ntt_4step/test_cpu_4step_ntt.cu:40-79, default logN = 12
    P = O.Port(64)
    prm = P.fourstep_params(12)
    q = prm["mod"][0]
    a, b = P.splitmix(11, 0, prm["n"], q), P.splitmix(12, 0, prm["n"], q)
    c = P.fourstep_ntt(P.pointwise(P.fourstep_ntt(a, prm), P.fourstep_ntt(b, prm), prm["mod"]),
                       prm, True)
    assert np.array_equal(c, P.schoolbook(a, b, O.X_N_minus, prm["mod"]))


def test_fourstep_output_order_matches_merge():
    # 4-step output order out[a*n1+b] = X[bitrev(a,log n2)*n1 + bitrev(b,log n1)]
    # where X is the natural-order cyclic DFT; checked against the Merge oracle (bit-reversed out)
    P = O.Port(64)
    prm = P.fourstep_params(12)
    q, n1, n2 = prm["mod"][0], prm["n1"], prm["n2"]
    mp = P.merge_params(12, O.X_N_minus, (q, prm["omega"], prm["psi"]))
    x = P.splitmix(21, 0, prm["n"], q)
    ym = P.merge_ntt(x, mp)  # ym[bitrev(k,12)] = X[k]
    X = np.empty_like(ym)
    idx = np.array([P.lib.ora_bitreverse(k, 12) for k in range(prm["n"])])
    X[np.arange(prm["n"])] = ym[idx]
    y4 = P.fourstep_ntt(x, prm)
    l1, l2 = n1.bit_length() - 1, n2.bit_length() - 1
    for a in (0, 1, 5, n2 - 1):
        for b in (0, 3, n1 - 1):
            k = P.lib.ora_bitreverse(a, l2) * n1 + P.lib.ora_bitreverse(b, l1)
            assert y4[a * n1 + b] == X[k]


def test_c3_polynomial_fixture_is_anchored(golden_dir):
    """tests/golden/c3_polys.json (8 polynomials of BASELINE config 3, reference-build digests; consumed by the -m gpu
    test test_full_size_c3_reference_layout, which also re-derives every spectrum with the C port): polynomial 0 IS the
    2^24 record of digests.json, the seeds follow the stated stride, inputs are regenerated from the portable stream."""
    c3 = json.load(open(os.path.join(golden_dir, "c3_polys.json")))
    rec = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["fourstep"]
           if r["bits"] == 64 and r["logn"] == 24][0]
    assert (c3["q"], c3["n1"], c3["n2"], c3["seed0"]) == (rec["q"], rec["n1"], rec["n2"], rec["seed"])
    p0 = c3["polys"][0]
    assert (p0["sha_in"], p0["sha_fwd"], p0["sha_inv"]) == (rec["sha_in"], rec["sha_fwd"], rec["sha_inv"])
    assert [p["seed"] for p in c3["polys"]] == [c3["seed0"] + c3["stride"] * i for i in range(8)]
    assert len({p["sha_fwd"] for p in c3["polys"]}) == 8
    P = O.Port(64)
    x = P.splitmix(c3["polys"][3]["seed"], 0, 1 << 24, c3["q"])
    assert sha(x) == c3["polys"][3]["sha_in"]
