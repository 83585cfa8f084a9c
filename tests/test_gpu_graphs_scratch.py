"""-m gpu parity tests: hipGraph capture / replay of drop-in calls and the lifetime of their library-owned twiddle scratch (reference calls are pure launches, src/lib/ntt_merge/ntt.cu:2076-2256): growth under a captured graph, moduli and tables rewritten between capture and replay, release, scratch exhaustion, host threads sharing and owning streams."""
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

def test_release_workspaces(g):
    """the drop-in calls keep a twiddle scratch per stream; releasing it is safe and the next call
    simply allocates again"""
    c = MergeCase(g, 64, 13, O.X_N_minus)
    x = c.random(2, 5)
    want = c.P.merge_ntt(x, c.oprm)
    assert np.array_equal(c.gpu_forward(x), want)
    g.release_workspaces()
    assert np.array_equal(c.gpu_forward(x), want)

def test_scratch_exhaustion_falls_back_to_the_generic_kernels(g):
    """a drop-in call whose per-(device, stream) twiddle scratch cannot be allocated (several streams, nearly full HBM)
    runs on the generic kernels, which need none, instead of failing; option no_scratch = 1 simulates the
    failed hipMalloc.  Merge single / RNS, 4-step both directions; under fast-strict the same situation throws."""
    import torch
    g.set_option("no_scratch", 1)
    try:
        for bits, logn, batch in ((64, 13, 5), (32, 14, 3), (64, 16, 2)):
            c = MergeCase(g, bits, logn, O.X_N_plus)
            x = c.random(batch, 8800 + logn)
            want = c.P.merge_ntt(x, c.oprm)
            assert np.array_equal(c.gpu_forward(x), want)
            assert np.array_equal(c.gpu_inverse(want, inplace=True), x)
        P = O.Port(64)
        fl = _small_prime_factors(P, 12, 3)
        cases, fwd, inv, mods, ninv = _rns_setup(g, 64, 12, O.X_N_minus, fl)
        x = np.concatenate([cases[p % 3].random(1, 8900 + p) for p in range(6)])
        d = g.to_device(x)
        g.GPU_NTT_Inplace(d, fwd, mods, g.ntt_rns_configuration(n_power=12, reduction_poly=O.X_N_minus), 6, 3)
        torch.cuda.synchronize()
        y = g.to_host(d)
        for p in range(6):
            assert np.array_equal(y[p * 4096:(p + 1) * 4096], P.merge_ntt(x[p * 4096:(p + 1) * 4096], cases[p % 3].oprm))
        from test_gpu_4step import run_fourstep
        p4 = g.NTTParameters4Step(13, 64)
        oprm = P.fourstep_params(13)
        x4 = P.splitmix(8950, 0, 2 * p4.n, p4.modulus.value)
        want4 = P.fourstep_ntt(x4, oprm)
        assert np.array_equal(run_fourstep(g, p4, x4, 2, inverse=False), want4)
        assert np.array_equal(run_fourstep(g, p4, P.fourstep_intt_first_transpose(want4, oprm), 2, inverse=True), x4)
        g.set_option("path", "fast-strict")
        c = MergeCase(g, 64, 13, O.X_N_plus)
        with pytest.raises((ValueError, g.GpuNttError)):
            c.gpu_forward(c.random(1, 1))
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
        g.set_option("no_scratch", 0)

# ---------------------------------------------------------------- scratch lifetime under captured graphs
def test_graph_replay_survives_scratch_growth_on_the_capture_stream(g):
    """a GPU_NTT + GPU_INTT pair of a 2^14 ring is captured into a hipGraph on a side stream; an
    eager 2^18 call on the SAME stream then needs a larger twiddle scratch.  The library must not free (or reuse) the
    buffer the graph's kernel arguments point at: the graph is replayed three times afterwards and compared with the
    oracle, the eager call too.  (Reference calls are pure launches, src/lib/ntt_merge/ntt.cu:2076-2256.)"""
    import torch
    c = MergeCase(g, 64, 14, O.X_N_plus)
    big = MergeCase(g, 64, 18, O.X_N_minus)
    batch = 16
    x = c.random(batch, 50001)
    want = c.P.merge_ntt(x, c.oprm)
    xb = big.random(2, 50002)
    s = torch.cuda.Stream()
    d = g.to_device(x)
    f = torch.zeros_like(d)   # forward result
    o = torch.zeros_like(d)   # round trip
    db = g.to_device(xb)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        g.GPU_NTT(d, f, c.fwd_dev, c.prm.modulus, c.cfg(stream=s), batch)  # eager warm-up on the capture stream
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        cs = torch.cuda.current_stream()
        g.GPU_NTT(d, f, c.fwd_dev, c.prm.modulus, c.cfg(stream=cs), batch)
        g.GPU_INTT(f, o, c.inv_dev, c.prm.modulus, c.cfg(True, stream=cs), batch)
    # grow the eager scratch of the capture stream well past what the graph was captured with, and overwrite it
    with torch.cuda.stream(s):
        g.GPU_NTT_Inplace(db, big.fwd_dev, big.prm.modulus, big.cfg(stream=s), 2)
        filler = torch.full((1 << 22,), -1, dtype=torch.int64, device="cuda")  # lands in freed memory, if any was freed
    s.synchronize()
    assert np.array_equal(g.to_host(db), big.P.merge_ntt(xb, big.oprm))
    for rep in range(3):
        f.zero_()
        o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(f), want), ("forward", rep)
        assert np.array_equal(g.to_host(o), x), ("round trip", rep)
        with torch.cuda.stream(s):  # eager calls of yet another size between the replays
            g.GPU_NTT_Inplace(db, big.fwd_dev, big.prm.modulus, big.cfg(stream=s), 1)
        s.synchronize()
    del filler

def test_graph_replay_rns_with_moduli_rewritten_between_capture_and_replay(g):
    """(b) the RNS form: the graph holds a drop-in RNS call whose kernel family was predicted from the stack the buffer
    held at capture time (60-bit primes).  The moduli buffer, the table and the input are then rewritten in place with a
    stack that needs another family (a 62-bit prime): the replay must still be exact (the preparation kernel inside the
    graph re-classifies and, since the family baked into the graph cannot serve the stack, transforms the batch itself),
    eager calls on the capture
    stream in between grow and overwrite that stream's own scratch, and the host-mapped prediction word the graph writes
    to stays alive."""
    import torch
    logn, batch, mc = 13, 6, 3
    n = 1 << logn
    stacks = {}
    for name, widths in (("w60", (60, 60, 60)), ("w62", (60, 62, 61))):
        cases = [MergeCase(g, 64, logn, O.X_N_plus, f) for f in distinct_factors(widths, logn)]
        fwd = np.zeros(mc * n, dtype=np.uint64)
        inv = np.zeros(mc * n, dtype=np.uint64)
        for i, c in enumerate(cases):
            fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
            inv[i * n:i * n + c.prm.root_of_unity_size] = c.prm.inverse_table_device_order
        x = np.concatenate([cases[p % mc].P.splitmix(51000 + p, 0, n, cases[p % mc].q) for p in range(batch)])
        want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm) for p in range(batch)])
        ninv = np.array([c.prm.n_inv for c in cases], dtype=np.uint64)
        stacks[name] = (g.modulus_array_to_device([c.prm.modulus for c in cases], 64), g.to_device(fwd), g.to_device(inv),
                        g.to_device(ninv), x, want)
    s = torch.cuda.Stream()
    mods = stacks["w60"][0].clone()
    fwd_t, inv_t, ninv_t = stacks["w60"][1].clone(), stacks["w60"][2].clone(), stacks["w60"][3].clone()
    d = g.to_device(stacks["w60"][4])
    f = torch.zeros_like(d)
    o = torch.zeros_like(d)
    cfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus, stream=s)
    icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=O.X_N_plus, mod_inverse=ninv_t, stream=s)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(2):  # the second call runs with the stack's prediction in place
            g.GPU_NTT(d, f, fwd_t, mods, cfg, batch, mc)
            g.GPU_INTT(f, o, inv_t, mods, icfg, batch, mc)
            s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        cs = torch.cuda.current_stream()
        ccfg = g.ntt_rns_configuration(n_power=logn, reduction_poly=O.X_N_plus, stream=cs)
        cicfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=O.X_N_plus, mod_inverse=ninv_t, stream=cs)
        g.GPU_NTT(d, f, fwd_t, mods, ccfg, batch, mc)
        g.GPU_INTT(f, o, inv_t, mods, cicfg, batch, mc)
    big = MergeCase(g, 64, 17, O.X_N_minus)
    xb = big.random(1, 51999)
    db = g.to_device(xb)
    for rep, name in enumerate(("w60", "w62", "w62", "w60", "w62")):
        src_m, src_f, src_i, src_n, x, want = stacks[name]
        mods.copy_(src_m)
        fwd_t.copy_(src_f)
        inv_t.copy_(src_i)
        ninv_t.copy_(src_n)
        d.copy_(g.to_device(x))
        f.zero_()
        o.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(f), want), ("forward", rep, name)
        assert np.array_equal(g.to_host(o), x), ("round trip", rep, name)
        with torch.cuda.stream(s):  # eager traffic on the capture stream: its scratch grows (2^17) and is rewritten
            db.copy_(g.to_device(xb))
            g.GPU_NTT_Inplace(db, big.fwd_dev, big.prm.modulus, big.cfg(stream=s), 1)
            g.GPU_NTT(d, f, fwd_t, mods, cfg, batch, mc)  # and an eager call of the same stack as the graph's
        s.synchronize()
        assert np.array_equal(g.to_host(f), want), ("eager", rep, name)
    assert np.array_equal(g.to_host(db), big.P.merge_ntt(xb, big.oprm))

def test_graph_replay_on_another_stream_while_eager_calls_use_the_capture_stream(g):
    """(c) a graph captured on stream A is replayed on stream B while eager drop-in calls of OTHER rings and moduli keep
    stream A busy.  Calls made during a capture take their scratch from a chain of their own (keyed by the capture), so
    the replay and the eager calls never share a buffer: both results are exact, every repetition."""
    import torch
    c = MergeCase(g, 64, 15, O.X_N_minus)
    e = MergeCase(g, 32, 15, O.X_N_plus)
    e2 = MergeCase(g, 64, 15, O.X_N_plus)
    batch = 8
    x, xe, xe2 = c.random(batch, 52001), e.random(batch, 52002), e2.random(batch, 52003)
    want, wante, wante2 = c.P.merge_ntt(x, c.oprm), e.P.merge_ntt(xe, e.oprm), e2.P.merge_ntt(xe2, e2.oprm)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    d = g.to_device(x)
    f = torch.zeros_like(d)
    de, de2 = g.to_device(xe), g.to_device(xe2)
    fe, fe2 = torch.zeros_like(de), torch.zeros_like(de2)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=sa):  # no warm-up: the first call of the capture allocates its own chain
        cs = torch.cuda.current_stream()
        g.GPU_NTT(d, f, c.fwd_dev, c.prm.modulus, c.cfg(stream=cs), batch)
    for rep in range(20):
        f.zero_()
        fe.zero_()
        fe2.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            graph.replay()
        for _ in range(3):
            g.GPU_NTT(de, fe, e.fwd_dev, e.prm.modulus, e.cfg(stream=sa), batch)
            g.GPU_NTT(de2, fe2, e2.fwd_dev, e2.prm.modulus, e2.cfg(stream=sa), batch)
        with torch.cuda.stream(sb):
            graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(f), want), rep
        assert np.array_equal(g.to_host(fe), wante), rep
        assert np.array_equal(g.to_host(fe2), wante2), rep

def test_capture_scratch_belongs_to_its_graph_and_is_reused_when_the_graph_dies(g):
    """a program that re-captures periodically must not grow without bound.  The twiddle scratch of a captured
    call is retained by the graph being captured (hipGraphRetainUserObject, prep.hip: give_to_graph); when the graph and
    its executable are destroyed the buffer goes to a pool and the next capture takes it from there, and the map node of
    the dead chain is erased.  24 capture / replay / destroy rounds of two ring sizes: allocations stop after the first
    rounds; a graph kept alive from round 0 still replays exactly at the end (its buffer was never handed on); results of
    every round equal the oracle."""
    import gc
    import torch
    cases = [MergeCase(g, 64, 13, O.X_N_plus), MergeCase(g, 32, 15, O.X_N_minus)]
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    before = g.scratch_stats()
    keep = None
    rounds = 24
    for r in range(rounds):
        c = cases[r % 2]
        batch = 3
        x = c.random(batch, 54000 + r)
        d = g.to_device(x)
        f = torch.zeros_like(d)
        o = torch.zeros_like(d)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            cs = torch.cuda.current_stream()
            g.GPU_NTT(d, f, c.fwd_dev, c.prm.modulus, c.cfg(stream=cs), batch)
            g.GPU_INTT(f, o, c.inv_dev, c.prm.modulus, c.cfg(True, stream=cs), batch)
        for _ in range(2):
            f.zero_()
            o.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(f), c.P.merge_ntt(x, c.oprm)), r
            assert np.array_equal(g.to_host(o), x), r
        if r == 0:
            keep = (graph, c, x, d, f, o)
        else:
            del graph
            gc.collect()
    after = g.scratch_stats()
    owned = after["graph_owned"] - before["graph_owned"]
    died = after["died"] - before["died"]
    reused = after["reused"] - before["reused"]
    assert owned >= rounds, (before, after)                    # every capture chain's buffer went to its graph
    assert died >= rounds - 2, (before, after)                 # ... and came back when the graph was destroyed
    assert reused >= rounds - 4, (before, after)               # later captures allocate nothing
    assert after["pooled"] <= before["pooled"] + 3, (before, after)
    assert after["chains"] <= before["chains"] + 4, (before, after)  # dead chains leave the map
    graph, c, x, d, f, o = keep
    f.zero_()
    o.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(g.to_host(f), c.P.merge_ntt(x, c.oprm))
    assert np.array_equal(g.to_host(o), x)
    # releasing everything while the kept graph is alive, then destroying it: its buffer is reported dead but belongs to
    # nobody any more -- nothing is pooled twice and calls afterwards stay exact
    g.release_workspaces()
    del graph, keep
    gc.collect()
    st = g.scratch_stats()
    assert st["pooled"] == 0, st
    c = cases[0]
    x = c.random(2, 54999)
    assert np.array_equal(c.gpu_forward(x), c.P.merge_ntt(x, c.oprm))

def test_release_workspaces_after_growth(g):
    """GPU_NTT_ReleaseWorkspaces() frees live and retired buffers; calls afterwards allocate afresh and stay exact."""
    import torch
    small, big = MergeCase(g, 64, 12, O.X_N_plus), MergeCase(g, 64, 19, O.X_N_plus)
    xs, xb = small.random(3, 53001), big.random(1, 53002)
    for _ in range(2):
        assert np.array_equal(small.gpu_forward(xs, inplace=True), small.P.merge_ntt(xs, small.oprm))
        assert np.array_equal(big.gpu_forward(xb, inplace=True), big.P.merge_ntt(xb, big.oprm))
        torch.cuda.synchronize()
        g.release_workspaces()

# ---------------------------------------------------------------- scratch chains under host threads
def test_threads_sharing_and_owning_streams_with_growing_scratch(g):
    """Four host threads issue drop-in calls of GROWING ring sizes -- two of them on ONE shared stream, two on streams of
    their own -- while a fifth replays a graph captured from drop-in calls.  Every chain retires buffers as it grows
    (nothing is freed or synchronised on), a thread holds its chain's lock from the preparation launch to the last kernel
    launch of a call, and calls made during the capture have a chain of their own: every result equals the oracle."""
    import threading
    import torch
    shared = torch.cuda.Stream()
    own = [torch.cuda.Stream(), torch.cuda.Stream()]
    sizes = (10, 12, 13, 14, 15, 16, 17, 18)
    cases = {(bits, n): MergeCase(g, bits, n, O.X_N_plus if n % 2 else O.X_N_minus) for bits in (64, 32) for n in sizes}
    errors = []

    def work(tid, stream, bits):
        try:
            for rep in range(3):
                for n in sizes:
                    c = cases[bits, n]
                    batch = 1 + (tid + rep) % 3
                    x = c.random(batch, 60000 + 100 * tid + n)
                    want = c.P.merge_ntt(x, c.oprm)
                    with torch.cuda.stream(stream):
                        d = g.to_device(x)
                        o = torch.zeros_like(d)
                        g.GPU_NTT(d, o, c.fwd_dev, c.prm.modulus, c.cfg(stream=stream), batch)
                        g.GPU_INTT(o, d, c.inv_dev, c.prm.modulus, c.cfg(True, stream=stream), batch)
                        g.GPU_NTT(d, o, c.fwd_dev, c.prm.modulus, c.cfg(stream=stream), batch)
                        stream.synchronize()
                    if not np.array_equal(g.to_host(o), want):
                        errors.append((tid, bits, n, rep))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    cg = cases[64, 13]
    xg = cg.random(4, 61234)
    dg = g.to_device(xg)
    og = torch.zeros_like(dg)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        g.GPU_NTT(dg, og, cg.fwd_dev, cg.prm.modulus, cg.cfg(stream=torch.cuda.current_stream()), 4)
    want_g = cg.P.merge_ntt(xg, cg.oprm)

    def replay():
        try:
            for _ in range(40):
                og.zero_()
                graph.replay()
                torch.cuda.synchronize()
                if not np.array_equal(g.to_host(og), want_g):
                    errors.append(("graph",))
        except Exception as e:  # noqa: BLE001
            errors.append(("graph", repr(e)))

    threads = [threading.Thread(target=work, args=(0, shared, 64)), threading.Thread(target=work, args=(1, shared, 32)),
               threading.Thread(target=work, args=(2, own[0], 64)), threading.Thread(target=work, args=(3, own[1], 32)),
               threading.Thread(target=replay)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors[:5]

def test_fourstep_table_check_inside_a_captured_graph(g):
    """The veto word of a captured 4-step call lives in the capture's own scratch chain (its reset is a node of the graph,
    the epoch is baked into the kernel arguments): the graph is replayed with consistent tables, after ONE word of W was
    rewritten in place, and after the word was restored -- oracle, the generic kernels' result for the corrupted table,
    oracle again -- while eager 4-step calls of another ring run on the capture stream in between."""
    import torch
    P = O.Port(64)
    logn, batch = 17, 3
    p4 = g.NTTParameters4Step(logn, 64)
    oprm = P.fourstep_params(logn)
    x = P.splitmix(62017, 0, batch * p4.n, p4.modulus.value)
    want = P.fourstep_ntt(x, oprm)  # natural-order result
    t1, t2, w = (g.to_device(t) for t in p4.tables["fwd"])
    d_nat = g.to_device(x)
    d_in = torch.zeros_like(d_nat)
    d_out = torch.zeros_like(d_nat)
    d_res = torch.zeros_like(d_nat)
    g.GPU_Transpose(d_nat, d_in, p4.n1, p4.n2, logn, batch)
    torch.cuda.synchronize()
    cfg_stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cfg_stream):
        g.GPU_4STEP_NTT(d_in, d_out, t1, t2, w, p4.modulus,
                        g.ntt4step_configuration(n_power=logn, stream=torch.cuda.current_stream()), batch)

    def result():
        d_out.fill_(-5)
        graph.replay()
        torch.cuda.synchronize()
        g.GPU_Transpose(d_out, d_res, p4.n1, p4.n2, logn, batch)
        torch.cuda.synchronize()
        return g.to_host(d_res)

    # what the generic kernels make of the corrupted table
    pos = 5 * p4.n2 + 77
    good_word = int(w[pos])
    w[pos] = (good_word + 1) % p4.modulus.value if (good_word + 1) % p4.modulus.value < 2**63 else 1
    g.set_option("path", "generic")
    try:
        d_gen = torch.zeros_like(d_nat)
        g.GPU_4STEP_NTT(d_in, d_gen, t1, t2, w, p4.modulus, g.ntt4step_configuration(n_power=logn), batch)
        g.GPU_Transpose(d_gen, d_res, p4.n1, p4.n2, logn, batch)
        torch.cuda.synchronize()
        want_bad = g.to_host(d_res).copy()
    finally:
        g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))
    w[pos] = good_word
    torch.cuda.synchronize()
    other = g.NTTParameters4Step(14, 64)
    ot = [g.to_device(t) for t in other.tables["fwd"]]
    oin = g.to_device(P.splitmix(62014, 0, other.n, other.modulus.value))
    oout = torch.zeros_like(oin)
    for rep in range(2):
        assert np.array_equal(result(), want), ("consistent tables", rep)
        w[pos] = (good_word + 1) % p4.modulus.value if (good_word + 1) % p4.modulus.value < 2**63 else 1
        torch.cuda.synchronize()
        got = result()
        assert np.array_equal(got, want_bad) and not np.array_equal(got, want), ("corrupted W", rep)
        assert np.array_equal(result(), want_bad), ("corrupted W, second replay", rep)
        w[pos] = good_word
        with torch.cuda.stream(cfg_stream):  # eager calls on the capture stream: another chain, another veto word
            g.GPU_4STEP_NTT(oin, oout, *ot, other.modulus, g.ntt4step_configuration(n_power=14, stream=cfg_stream), 1)
        cfg_stream.synchronize()
    assert np.array_equal(result(), want)

@pytest.mark.parametrize("logn,inverse", [(12, False), (13, True), (16, False), (16, True), (18, True)])
def test_fourstep_rns_overload_captured_then_modulus_and_tables_rewritten(g, logn, inverse):
    """GPU_4STEP_NTT through the RNS overload with ONE device-side modulus (the reference examples' calling style),
    captured into a hipGraph while the host predicts the default kernel family: only that family is baked into the graph
    and its kernels are their own fall-back (kern::F_SELF_FALLBACK).  The graph is replayed (a) as captured, (b) after the
    device buffers -- modulus, n^-1 and all three tables -- were rewritten IN PLACE with those of a 61-bit prime (another
    kernel family: nothing enqueued for it), (c) after one W word of those was corrupted, (d) with the original contents
    again.  Every replay must return what the element-by-element kernels (path = generic, eager) compute from the same
    buffers."""
    import torch
    from gpu_utils import find_ntt_factors
    shape = g.NTTParameters4Step(logn, 64)
    n, n1, n2, batch = shape.n, shape.n1, shape.n2, 3
    kind = g.INVERSE if inverse else g.FORWARD

    def tables_for(q, omega):
        m = g.Modulus(q, bits=64)
        r = pow(omega, -1, q) if inverse else omega
        w = torch.zeros(n, dtype=torch.int64, device="cuda")
        t1 = torch.zeros(n1 >> 1, dtype=torch.int64, device="cuda")
        t2 = torch.zeros(n2 >> 1, dtype=torch.int64, device="cuda")
        g.GPU_Generate4StepW(w, r, m, logn, kind)
        g.GPU_GeneratePowerTable(t1, pow(r, n // n1, q), m, int(np.log2(n1)) - 1, True)
        g.GPU_GeneratePowerTable(t2, pow(r, n // n2, q), m, int(np.log2(n2)) - 1, True)
        torch.cuda.synchronize()
        return m, t1, t2, w

    qa, wa, _ = find_ntt_factors(60, logn)
    qb, wb, _ = find_ntt_factors(61, logn)
    ma, a1, a2, aw = tables_for(qa, wa)
    mb, b1, b2, bw = tables_for(qb, wb)
    t1, t2, w = a1.clone(), a2.clone(), aw.clone()
    mods = g.modulus_array_to_device([ma], 64)
    mods_b = g.modulus_array_to_device([mb], 64)
    mods_a = mods.clone()
    ninv = g.to_device(np.array([pow(n, -1, qa)], dtype=np.uint64))
    ninv_a, ninv_b = ninv.clone(), g.to_device(np.array([pow(n, -1, qb)], dtype=np.uint64))
    d_in = g.to_device(np.random.default_rng(700 + logn).integers(0, min(qa, qb), size=batch * n, dtype=np.uint64))
    d_out = torch.zeros_like(d_in)

    def cfg(stream=None):
        return g.ntt4step_rns_configuration(n_power=logn, ntt_type=kind, mod_inverse=ninv, stream=stream)

    for _ in range(3):  # the prediction settles on the default family
        g.GPU_4STEP_NTT(d_in, d_out, t1, t2, w, mods, cfg(), batch, 1)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        g.GPU_4STEP_NTT(d_in, d_out, t1, t2, w, mods, cfg(torch.cuda.current_stream()), batch, 1)

    def replay():
        d_out.fill_(-9)
        graph.replay()
        torch.cuda.synchronize()
        return g.to_host(d_out).copy()

    def generic():
        g.set_option("path", "generic")
        try:
            ref = torch.zeros_like(d_in)
            g.GPU_4STEP_NTT(d_in, ref, t1, t2, w, mods, cfg(), batch, 1)
            torch.cuda.synchronize()
            return g.to_host(ref).copy()
        finally:
            g.set_option("path", os.environ.get("GPUNTT_PATH", "default"))

    def install(m_dev, nv, x1, x2, xw):
        mods.copy_(m_dev); ninv.copy_(nv); t1.copy_(x1); t2.copy_(x2); w.copy_(xw)
        torch.cuda.synchronize()

    want_a = generic()
    assert np.array_equal(replay(), want_a), "as captured"
    install(mods_b, ninv_b, b1, b2, bw)
    want_b = generic()
    assert not np.array_equal(want_b, want_a)
    assert np.array_equal(replay(), want_b), "61-bit modulus written over the captured buffers"
    w[n // 2 + 3] ^= 1
    torch.cuda.synchronize()
    want_c = generic()
    assert np.array_equal(replay(), want_c), "61-bit modulus, one W word corrupted"
    install(mods_a, ninv_a, a1, a2, aw)
    assert np.array_equal(replay(), want_a), "original contents again"
