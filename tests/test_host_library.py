"""CPU-side tests of the product library: the C-ABI library loads and exports every symbol
include/gpuntt_c.h declares, the host parameter generators agree with the oracle and the
golden digests, argument checking matches the reference's exception behaviour, the pass
planner covers every n_power, and the batch-shard arithmetic is consistent.  No compute
calls (there is no GPU here and the product has no CPU path)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from gpu_utils import sha
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    pkg.load_library()
    return pkg


def test_c_abi_exports_every_declared_symbol(g):
    hdr = open(os.path.join(ROOT, "include", "gpuntt_c.h")).read()
    declared = set(re.findall(r"\b(gpuntt_[a-z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.startswith("gpuntt_modulus3") and d != "gpuntt_modulus6"}
    assert len(declared) >= 22
    lib = g.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert set(g.EXPORTED_SYMBOLS) == declared


def test_modulus_struct_matches_reference_contract(g):
    P = O.Port(64)
    for q in (576460756061519873, 576460752303415297, 288230377292562433, 12289, 469762049):
        m = g.Modulus(q)
        assert (m.value, m.bit, m.mu) == P.modulus(q)
    # just below a power of two the reference's double log2 over-states the width by one (modular_arith.cuh:44-47)
    for q, bit in ((2**60 - 107, 61), (2**60 - 2559, 61), (2**59 - 55, 60), (2**57 - 111, 58),
                   (2**54 - 131, 54), (2**60 - 2**20 + 1, 60)):
        m = g.Modulus(q)
        assert (m.value, m.bit, m.mu) == P.modulus(q) and m.bit == bit
    # ... and at the top of the domain the over-stated width makes mu pass the word (2^125 / q >= 2^64 for a 61-bit q
    # with bit = 62): the reference stores the truncated value (its Barrett product is wrong from then on), this
    # library refuses the modulus
    assert P.modulus(2**61 - 31)[1] == 62
    with pytest.raises(ValueError):
        g.Modulus(2**61 - 31)
    P32 = O.Port(32)
    for q in (469762049, 268460033, 12289):
        m = g.Modulus(q, bits=32)
        assert (m.value, m.bit, m.mu) == P32.modulus(q)


@pytest.mark.parametrize("bits", [32, 64])
def test_merge_parameter_generator(g, bits, golden_dir):
    P = O.Port(bits)
    recs = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["merge"]
            if r["bits"] == bits]
    for r in recs:
        p = g.NTTParameters(r["logn"], r["poly"], bits)
        assert (p.modulus.value, p.modulus.bit, p.modulus.mu) == (r["q"], r["bit"], r["mu"])
        assert (p.omega, p.psi, p.n_inv) == (r["omega"], r["psi"], r["n_inv"])
        assert sha(p.forward_table_device_order) == r["sha_fwd_gpu_table"]
        assert sha(p.inverse_table_device_order) == r["sha_inv_gpu_table"]
    # custom factors (NTTFactors constructor)
    f = (576460752303415297, 288482366111684746, 238394956950829) if bits == 64 else \
        (268460033, 36747374, 77090)
    p = g.NTTParameters(12, O.X_N_plus, bits, f)
    pp = P.merge_params(12, O.X_N_plus, f)
    assert np.array_equal(p.forward_table_device_order, P.bitrev_table(pp["fwd"]))
    assert p.n_inv == pp["n_inv"]


@pytest.mark.parametrize("bits", [32, 64])
def test_fourstep_parameter_generator(g, bits, golden_dir):
    recs = [r for r in json.load(open(os.path.join(golden_dir, "digests.json")))["fourstep"]
            if r["bits"] == bits and r["logn"] <= 20]
    for r in recs:
        p = g.NTTParameters4Step(r["logn"], bits)
        assert (p.n1, p.n2, p.n_inv) == (r["n1"], r["n2"], r["n_inv"])
        assert sha(p.tables["fwd"][2]) == r["sha_W_fwd"] and sha(p.tables["inv"][2]) == r["sha_W_inv"]
        assert sha(p.tables["fwd"][0]) == r["sha_n1_fwd_gpu"]
        assert sha(p.tables["inv"][1]) == r["sha_n2_inv_gpu"]
    with pytest.raises(ValueError):
        g.NTTParameters4Step(11, bits)


def test_argument_errors_without_gpu(g):
    lib = g.load_library()
    m = g.Modulus(576460756061519873).c()
    d = ctypes.c_void_p(8)  # never dereferenced: the checks below fail before any launch
    for bad in (0, 29):
        assert lib.gpuntt_ntt_u64(d, d, d, m, bad, 0, 1, 0, None, 1) == -1
        assert lib.gpuntt_last_error() == b"Invalid n_power range!"
        assert lib.gpuntt_intt_u64(d, d, d, m, bad, 0, 1, ctypes.c_uint64(1), 0, None, 1) == -1
    assert lib.gpuntt_ntt_u64(d, d, d, m, 12, 9, 1, 0, None, 1) == -1
    assert lib.gpuntt_last_error() == b"Invalid ntt_layout!"
    assert lib.gpuntt_ntt_rns_u64(d, d, d, d, 12, 0, 1, 0, None, 1, 0) == -1
    assert lib.gpuntt_last_error() == b"Invalid mod_count!"
    # ordered entry points: n_power in [10, 28] (reference ntt.cu:3607-3610, 4288-4291)
    for fn in (lib.gpuntt_ntt_modulus_ordered_u64, lib.gpuntt_ntt_poly_ordered_u64,
               lib.gpuntt_ntt_modulus_ordered_u32, lib.gpuntt_ntt_poly_ordered_u32):
        for bad in (9, 29):
            assert fn(d, d, d, d, bad, 0, 1, None, None, 1, 1, d) == -1
            assert lib.gpuntt_last_error() == b"Invalid n_power range!"
    # PerCoefficient range check (reference ntt.cu:2230-2233)
    assert lib.gpuntt_ntt_u64(d, d, d, m, 10, 1, 1, 0, None, 1) == -1
    assert lib.gpuntt_last_error() == b"Invalid n_power range!"
    # NULL data / table / modulus pointers are rejected at the C boundary
    assert lib.gpuntt_ntt_u64(None, d, d, m, 12, 0, 1, 0, None, 1) == -1
    assert lib.gpuntt_last_error() == b"null pointer argument"
    assert lib.gpuntt_intt_u64(d, d, None, m, 12, 0, 1, ctypes.c_uint64(1), 0, None, 1) == -1
    assert lib.gpuntt_ntt_rns_u64(d, d, d, None, 12, 0, 1, 0, None, 1, 1) == -1
    assert lib.gpuntt_4step_u64(d, d, d, None, d, m, 12, 0, ctypes.c_uint64(0), None, 1) == -1
    assert lib.gpuntt_4step_natural_u32(d, None, d, d, d, g.Modulus(469762049, bits=32).c(), 12, 0,
                                        ctypes.c_uint32(0), None, 1) == -1
    assert lib.gpuntt_polymul_u64(d, d, d, d, None, m, 12, 1, ctypes.c_uint64(1), None, 1) == -1
    assert lib.gpuntt_last_error() == b"null pointer argument"


def test_no_cpu_fallback(g):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less hosts")
    x = torch.zeros(16, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU path"):
        g.GPU_NTT_Inplace(x, x, g.Modulus(576460756061519873), g.ntt_configuration(n_power=4), 1)


def test_shard_range_partitions_batch(g):
    for batch, mc in ((1024, 1), (512, 8), (8192, 1), (24, 3)):
        for world in (1, 2, 4, 8):
            spans = [g.shard_range(batch, r, world, mc) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            assert all(lo % mc == 0 and hi % mc == 0 for lo, hi in spans)
    with pytest.raises(ValueError):
        g.shard_range(10, 0, 2, 4)


def test_options_replace_environment_switches(g):
    """the C++ library reads no environment variable: tuning / test switches go through gpuntt_set_option
    (GPU_NTT_SetOption); unknown names and values are refused"""
    lib = g.load_library()
    for name, value in (("path", "generic"), ("path", "fast-strict"), ("path", "generic-capped"), ("path", "fast"), ("path", "default"),
                        ("no_scratch", "1"), ("no_scratch", "0"), ("check_4step_tables", "0"), ("check_4step_tables", "1"),
                        ("rns_force_fallback", "1"), ("rns_force_fallback", "0"), ("rns_predict", "0"), ("rns_predict", "1")):
        g.set_option(name, value)
    # unknown names, and values outside the documented sets, are refused -- never silently mapped to a default;
    # the A/B switches of rounds 1-4 are retired (their questions are closed): their names are unknown now
    for name, value in (("path", "sideways"), ("no_such_option", "1"), ("rns_predict", "yes"), ("rns_predict", "2"),
                        ("no_scratch", ""), ("check_4step_tables", "1x"), ("q59", "1"), ("contig_k", "11"), ("xcd_order", "0"),
                        ("lim31", "0"), ("reverse", "0"), ("u64_big_tiles", "13"), ("u32_tile", "12"), ("u32_ring13_batch", "0"),
                        ("validate_4step_tables", "1"),
                        # the test hooks are NOT options of the public interface: gpuntt_set_option
                        # refuses them, gpuntt_test_set_hook (csrc/test_hooks.h) takes them
                        ("path", "fast-strict"), ("path", "generic-capped"), ("no_scratch", "1"), ("rns_force_fallback", "1"),
                        ("u32_e32", "0")):
        assert lib.gpuntt_set_option(name.encode(), value.encode()) != 0, (name, value)
    assert lib.gpuntt_set_option(None, None) != 0
    for name, value in (("path", "fast-strict"), ("path", "default"), ("no_scratch", "1"), ("no_scratch", "0"),
                        ("rns_force_fallback", "0"), ("u32_e32", "0xf000"), ("u32_e32", "61440"), ("check_4step_tables", "1")):
        assert lib.gpuntt_test_set_hook(name.encode(), value.encode()) == 0, (name, value)
    for name, value in (("u32_e32", "0x20000"), ("u32_e32", "abc"), ("no_scratch", "2"), ("path", "sideways")):
        assert lib.gpuntt_test_set_hook(name.encode(), value.encode()) != 0, (name, value)
    # the public headers do not advertise the hooks
    import glob
    for h in glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "**", "*"), recursive=True):
        if os.path.isfile(h):
            assert "gpuntt_test_" not in open(h).read(), h
    # the launch log: nothing launched, nothing logged
    with g.launch_log() as log:
        pass
    assert log.kernels == []
    import subprocess
    out = subprocess.run("strings -a %s | grep -c '^GPUNTT_[A-Z0-9_]*$'" % g.LIB_PATH, shell=True, capture_output=True,
                         text=True).stdout.strip()
    assert out in ("", "0"), "the library still carries GPUNTT_* environment variable names"


def test_dispatch_table_in_design_md_is_the_tested_table():
    """DESIGN.md 3.8 carries the table `python tests/dispatch_rows.py --markdown` prints (tests/test_gpu_dispatch_table.py
    walks the same rows on the GPU)"""
    from dispatch_rows import ROWS, markdown
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "DESIGN.md")).read()
    for line in markdown().splitlines():
        assert line in text, "DESIGN.md 3.8 is out of date: run python tests/dispatch_rows.py --markdown\n" + line
    assert len({r["id"] for r in ROWS}) == len(ROWS) and all(r["launches"] and r["serves"] for r in ROWS)
