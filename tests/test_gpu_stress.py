"""-m gpu stress tests: EVERY polynomial of every call compared with the oracle, over the plan shapes of both entry-point
families (a bounded slice of tools/stress_fourstep.py plus its Merge twin).  Round 3 found a kernel that returned one
wrong polynomial in about a thousand; sampled checks pass that almost every time, so these tests compare all of them
and run every Merge shape twice.  Reference flows: example/ntt_merge/test_merge_ntt.cu:46-341,
example/ntt_4step/test_4step_ntt.cu:147-178, test_4step_intt.cu:81-179."""
import os
import time

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


# ---------------------------------------------------------------------------------------------- 4-step
FOURSTEP_SHAPES = 160
# batch ranges chosen so that the number of tiles crosses the thresholds the kernels care about: several tiles per XCD,
# ragged one-tile batches, the 256-polynomial switch to the 16384-coefficient tile (64-bit 2^14), poly-minor block
# order from 2^20 (batch >= 2), reversed sweeps (every multi-pass plan)
_FS_MAX_BATCH = {12: 300, 13: 300, 14: 300, 15: 40, 16: 20, 17: 9, 18: 5, 19: 3, 20: 3, 21: 2, 22: 2}
_FS_WEIGHT = {12: 6, 13: 6, 14: 8, 15: 6, 16: 6, 17: 4, 18: 3, 19: 2, 20: 2, 21: 1, 22: 1}


def fourstep_shape_list(seed, count):
    rng = np.random.default_rng(seed)
    logns = [l for l, w in _FS_WEIGHT.items() for _ in range(w)]
    out = []
    for _ in range(count):
        bits = int(rng.choice([32, 64]))
        logn = int(rng.choice(logns))
        mb = _FS_MAX_BATCH[logn]
        # half of the small-ring shapes sit on the interesting batch sizes, the rest anywhere (bounded work per shape)
        if logn <= 14 and rng.integers(0, 2):
            batch = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 33, 64, 255, 256, 257, mb]))
        else:
            batch = int(rng.integers(1, min(mb, max(2, (1 << 18) >> logn if logn <= 14 else mb)) + 1))
        out.append((bits, logn, batch, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(1, 1 << 30))))
    return out


def test_fourstep_stress_every_polynomial(g):
    """160 seeded random 4-step shapes (ring 2^12 .. 2^22, both word sizes, both overloads, both directions), EVERY
    polynomial against NTT_4STEP_CPU -- the in-suite slice of tools/stress_fourstep.py (1307 shapes there in round 3)."""
    import test_gpu_4step as T
    cache = {}
    t0 = time.time()
    shapes = fourstep_shape_list(20261001, FOURSTEP_SHAPES)
    for idx, (bits, logn, batch, rns_f, rns_i, seed) in enumerate(shapes):
        key = (bits, logn)
        if key not in cache:
            P = O.Port(bits)
            cache[key] = (P, g.NTTParameters4Step(logn, bits), P.fourstep_params(logn))
        P, p4, oprm = cache[key]
        n = p4.n
        x = P.splitmix(seed, 0, batch * n, p4.modulus.value)
        want = P.fourstep_ntt(x, oprm)
        got = T.run_fourstep(g, p4, x, batch, inverse=False, rns=rns_f)
        assert np.array_equal(got, want), ("forward", idx, bits, logn, batch, rns_f)
        xin = P.fourstep_intt_first_transpose(want, oprm)
        back = T.run_fourstep(g, p4, xin, batch, inverse=True, rns=rns_i)
        assert np.array_equal(back, x), ("inverse", idx, bits, logn, batch, rns_i)
    print("4-step stress: %d shapes in %.0f s" % (len(shapes), time.time() - t0))


# ----------------------------------------------------------------------------------------------- Merge
def _factors_above_31q_limit(logn):
    """a 60-bit NTT prime ABOVE 2^64 / 31: 16 q < 2^64 but 31 q is not, so forward calls keep the default 16 q range"""
    skip = 0
    while True:
        f = find_ntt_factors(60, logn, skip)
        if f[0] > 0xFFFFFFFFFFFFFFFF // 31:
            return f
        skip += 1


# (bits, modulus kind, logn, batch): every plan shape of lazy_tile_log / make_plan_tl --
#   64-bit: single pass (2^11 shares a tile, 2^12), the 8192-coefficient tile (2^13), the 16384-coefficient tile (2^14
#   forward from 256 polynomials, 2^22 forward), two passes 4 + 10 .. 8 + 12, 8 + 13 (2^21), three passes (2^22 inverse,
#   2^23, 2^24), poly-minor block order (>= 2^20 with batch >= 2); lazy ranges 31 q (pool prime, forward), 16 q (pool
#   prime inverse; a 60-bit prime above 2^64 / 31 forward), 8 q (61 bit), 4 q (62 bit);
#   32-bit: 4096 tile (<= 2^12, 2^15 .. 2^19, 2^23), 16384 tile (2^13, 2^14, 2^20 .. 2^22); 8 q (pool, 29 bit), 4 q (30 bit)
MERGE_SHAPES = [
    (64, "pool", 11, 37), (64, "pool", 12, 9), (64, "pool", 13, 33), (64, "pool", 14, 256), (64, "pool", 14, 5),
    (64, "pool", 15, 17), (64, "pool", 16, 24), (64, "pool", 17, 9), (64, "pool", 18, 5), (64, "pool", 19, 3),
    (64, "pool", 20, 3), (64, "pool", 21, 2), (64, "pool", 22, 2), (64, "pool", 23, 1), (64, "pool", 24, 2),
    (64, "q60hi", 13, 9), (64, "q60hi", 14, 256), (64, "q60hi", 16, 8), (64, "q60hi", 21, 2),
    (64, "q61", 12, 5), (64, "q61", 13, 9), (64, "q61", 14, 7), (64, "q61", 16, 6), (64, "q61", 20, 2), (64, "q61", 22, 1),
    (64, "q62", 12, 5), (64, "q62", 13, 9), (64, "q62", 14, 7), (64, "q62", 16, 6), (64, "q62", 20, 2), (64, "q62", 22, 1),
    (32, "pool", 12, 9), (32, "pool", 13, 40), (32, "pool", 14, 33), (32, "pool", 15, 9), (32, "pool", 16, 12),
    (32, "pool", 20, 3), (32, "pool", 21, 2), (32, "pool", 22, 2), (32, "pool", 23, 1),
    (32, "q30", 13, 7), (32, "q30", 14, 9), (32, "q30", 16, 5), (32, "q30", 20, 2),
]


def _merge_factors(kind, logn):
    if kind == "pool":
        return None
    if kind == "q60hi":
        return _factors_above_31q_limit(logn)
    return find_ntt_factors({"q61": 61, "q62": 62, "q30": 30}[kind], logn)


@pytest.mark.parametrize("bits,kind", [(64, "pool"), (64, "q60hi"), (64, "q61"), (64, "q62"), (32, "pool"), (32, "q30")])
def test_merge_stress_every_polynomial_every_plan_shape_twice(g, bits, kind):
    """GPU_NTT / GPU_INTT over every plan shape (tile 4096 / 8192 / 16384, one to three sweeps, reversed sweeps,
    poly-minor order) and every lazy range (4 q / 8 q / 16 q / 31 q), every polynomial against NTTCPU, each shape twice,
    under path = fast-strict (a shape the fast kernels cannot take would throw instead of passing on the Barrett kernels)."""
    g.set_option("path", "fast-strict")
    try:
        for b, k, logn, batch in MERGE_SHAPES:
            if (b, k) != (bits, kind):
                continue
            poly = O.X_N_plus if (logn + batch) % 2 else O.X_N_minus
            c = MergeCase(g, bits, logn, poly, _merge_factors(kind, logn))
            x = c.random(batch, 77000 + 131 * logn + batch)
            want = c.P.merge_ntt(x, c.oprm)
            for rep in range(2):
                got = c.gpu_forward(x, inplace=bool(rep))
                assert np.array_equal(got, want), ("forward", bits, kind, logn, batch, rep)
                back = c.gpu_inverse(got, inplace=not rep)
                assert np.array_equal(back, x), ("inverse", bits, kind, logn, batch, rep)
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))


def test_merge_stress_rns_stacks_every_polynomial_twice(g):
    """the RNS overloads (device-side moduli, go-flag) on stacks of 60-bit, 61-bit and 62-bit primes mixed: every
    polynomial of batches that are not a multiple of the stack, forward and inverse, twice, plus NTTPlan == drop-in"""
    import torch
    for logn, batch, widths in ((12, 11, (60, 60, 60)), (13, 10, (62, 60, 61)), (16, 7, (60, 61, 60)), (16, 9, (60, 60)),
                                (17, 5, (62, 62)), (14, 260, (60, 60, 60, 60))):
        fl, seen = [], set()
        for w in widths:
            skip = 0
            while True:
                f = find_ntt_factors(w, logn, skip)
                if f[0] not in seen:
                    break
                skip += 1
            seen.add(f[0])
            fl.append(f)
        poly = O.X_N_plus
        cases = [MergeCase(g, 64, logn, poly, f) for f in fl]
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
        x = np.concatenate([cases[p % mc].P.splitmix(88000 + p, 0, n, cases[p % mc].q) for p in range(batch)])
        want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm) for p in range(batch)])
        cf = g.ntt_rns_configuration(n_power=logn, reduction_poly=poly)
        ci = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
        for rep in range(2):
            d = g.to_device(x)
            o = torch.zeros_like(d)
            g.GPU_NTT(d, o, d_fwd, mods, cf, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), want), ("rns forward", logn, batch, widths, rep)
            g.GPU_INTT_Inplace(o, d_inv, mods, ci, batch, mc)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(o), x), ("rns inverse", logn, batch, widths, rep)
        plan = g.NTTPlan(d_fwd, [c.prm.modulus for c in cases], logn, poly, g.FORWARD, batch_hint=batch)
        o2 = torch.zeros_like(d)
        plan.execute(d, o2, batch)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(o2), want), ("plan forward", logn, batch, widths)
        plan.close()


def test_u32_random_modulus_widths_every_polynomial():
    """The in-suite slice of tools/stress_u32.py (round 5: the 32-bit butterflies became multiply-add chains and the final
    normalisation one quotient estimate by floor(2^32 / q)): 20 s of seeded random 32-bit calls -- primes of 12 .. 30 bits
    (both lazy families and moduli far below the word size), rings 2^4 .. 2^20, X^N+1 / X^N-1, both directions, single
    modulus (drop-in, NTTPlan) and RNS stacks of 2 .. 4 primes -- EVERY polynomial against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_u32.py"), "20261002", "20"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "u32 stress OK" in r.stdout, r.stdout[-2000:]
