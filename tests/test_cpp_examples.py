"""The reference's example programs are its only tests; tests/cpp holds the same
programs written against this library's drop-in C++ headers.  They prove that caller code in
the reference's style (templates, designated-initialiser configs, NTTParameters / NTTCPU
helpers, CudaException / GPUNTT_CUDA_CHECK names) compiles and links against libgpuntt.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "_bin")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "-j3"])


def _run(name, *args):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        _build()
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_cpu_examples_build_and_pass(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    assert "All Correct." in _run("example_cpu_polymul", 12)


@pytest.mark.gpu
@pytest.mark.parametrize("logn,batch,dt", [(12, 1, "u64"), (5, 3, "u64"), (16, 8, "u64"), (13, 4, "u32"),
                                           (20, 2, "u64")])
def test_gpu_merge_example(logn, batch, dt):
    out = _run("example_merge_ntt", logn, batch, dt)
    assert "All Correct for PerPolynomial NTT." in out and "All Correct for PerPolynomial INTT." in out
    assert "All Correct for GPU_PolyMul." in out


@pytest.mark.gpu
@pytest.mark.parametrize("logn,batch", [(12, 1), (16, 2), (17, 1)])
def test_gpu_4step_example(logn, batch):
    out = _run("example_4step_ntt", logn, batch)
    assert "All Correct." in out and "All Correct (inverse)." in out
    assert "All Correct (natural order, one call)." in out


@pytest.mark.gpu
@pytest.mark.parametrize("logn,batch,devices,mc", [(13, 64, 0, 2), (16, 24, 8, 3), (10, 400, 2, 4)])
def test_multi_device_example(logn, batch, devices, mc):
    """tests/cpp/example_multi_device.cpp: the batch shard from C++ -- one host thread per device (hipSetDevice), drop-in
    call and NTTPlan on a stream per device, every polynomial against NTTCPU.  Asking for more devices than the box has
    degrades to what exists (this box: one), the same code path with one worker."""
    out = _run("example_multi_device", logn, batch, devices, mc)
    assert "All Correct on" in out and "WRONG" not in out


@pytest.mark.gpu
def test_native_benchmark_program_runs():
    """tests/cpp/bench_ntt.cpp (the reference benchmark's axes timed from C++: drop-in call, plan, plan replayed
    from a hipGraph) -- a short run of each algorithm; one JSON object per ring size"""
    import json
    for algo, dt, first, last in (("merge", "u64", 12, 14), ("merge", "u32", 16, 16), ("4step", "u64", 12, 13)):
        lines = [json.loads(l) for l in _run("bench_ntt", algo, dt, first, last, 1, 32).splitlines() if l.startswith("{")]
        assert [d["log2N"] for d in lines] == list(range(first, last + 1))
        assert all(d["us"] > 0 and d["plan_us"] > 0 and d["graph_us"] > 0 for d in lines)


@pytest.mark.gpu
def test_multi_device_benchmark_program_agrees_with_bench_py(pkg):
    """tests/cpp/bench_multi_device.cpp: resident shards, one host thread per device, HIP events, max over the devices --
    the 1 / 2 / 4 / 8-GPU table of the product from C++ (tools/first_multigpu_lease.sh runs it on the first multi-GPU
    node).  On this one-GPU box: the N = 1 lines are bit-exact and agree with the same calls timed through the Python
    harness (what bench.py times) within a few per cent."""
    import json
    import time
    import numpy as np
    import torch
    g = pkg
    g.load_library()
    for cfg, bits, logn, polys, steps in (("c2", 64, 16, 1024, 20), ("c4", 32, 14, 8192, 50)):
        lines = [json.loads(l) for l in _run("bench_multi_device", cfg, steps, 5, "1,2").splitlines() if l.startswith("{")]
        assert [d["n_gpus"] for d in lines] == list(range(1, min(2, torch.cuda.device_count()) + 1))
        d = lines[0]
        assert d["bit_exact_vs_NTTCPU"] is True and d["polys_per_gpu"] == polys and d["ms_per_step"] > 0
        # the same call through the C ABI / Python harness
        prm = g.NTTParameters(logn, g.X_N_minus, bits)
        # random residues like the program's shard: at the socket power cap the time depends on the data (an all-zero
        # batch runs C2 8 % faster)
        one = np.random.default_rng(2).integers(0, prm.modulus.value, size=1 << logn, dtype=np.uint64).astype(g.np_dtype(bits))
        x = g.to_device(one).repeat(polys)
        table = g.to_device(prm.forward_table_device_order)
        c = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=g.X_N_minus)
        step = lambda: g.GPU_NTT_Inplace(x, table, prm.modulus, c, polys)  # noqa: E731
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            step()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        py_ms = e0.elapsed_time(e1) / steps
        assert abs(d["ms_per_step"] - py_ms) / py_ms < 0.06, (cfg, d["ms_per_step"], py_ms)
