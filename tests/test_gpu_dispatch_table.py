"""-m gpu: walks tests/dispatch_rows.py -- the dispatch table of the drop-in entry points -- row by row:
sets up the row's state (modulus class, tables, what the family prediction knows), makes the call under the library's
launch log (gpu-ntt_amd/csrc/test_hooks.h) and checks (1) the kernels enqueued, in order, against the row and (2) the
result against the oracle wherever the reference's CPU classes define one.

    DISPATCH_DISCOVER=1 python -m pytest tests/test_gpu_dispatch_table.py -m gpu -s     prints the observed launches
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from dispatch_rows import ROWS
from gpu_utils import MergeCase, find_ntt_factors, oracle_batch

pytestmark = pytest.mark.gpu
DISCOVER = os.environ.get("DISPATCH_DISCOVER") == "1"


@pytest.fixture(scope="module")
def g(pkg):
    pkg.load_library()
    return pkg


def _tiny(g, bits):
    return g.Modulus(value=2, bit=2, mu=16, bits=bits)  # outside the fast kernels' domain (q < 3); bit / mu given: no validation


def _host_case(g, row, poly):
    bits, logn, cls = row["bits"], row["logn"], row.get("modulus", "pool")
    factors = {"pool": None, "b61": lambda: find_ntt_factors(61, logn, clear_of_top=True),
               "b62": lambda: find_ntt_factors(62, logn, clear_of_top=True),
               "b30": lambda: find_ntt_factors(30, logn), "tiny": None}[cls]
    return MergeCase(g, bits, logn, poly, factors() if callable(factors) else None)


def _stack(g, row, poly):
    """cases (None for the out-of-domain slot), device tables, host Modulus list"""
    from gpu_utils import distinct_factors as _distinct_factors
    bits, logn = row["bits"], row["logn"]
    widths = [w for w in row["stack"] if w != "tiny"]
    real = [MergeCase(g, bits, logn, poly, f) for f in _distinct_factors(widths, logn)]
    cases, it = [], iter(real)
    for w in row["stack"]:
        cases.append(None if w == "tiny" else next(it))
    n = 1 << logn
    dt = real[0].P.T
    fwd, inv = np.zeros(len(cases) * n, dtype=dt), np.zeros(len(cases) * n, dtype=dt)
    for i, c in enumerate(cases):
        src = c if c is not None else real[0]  # the out-of-domain slot gets SOME table: its result is not checked
        sz = src.prm.root_of_unity_size
        fwd[i * n:i * n + sz] = src.prm.forward_table_device_order
        inv[i * n:i * n + sz] = src.prm.inverse_table_device_order
    mods = [c.prm.modulus if c is not None else _tiny(g, bits) for c in cases]
    return cases, real, g.to_device(fwd), g.to_device(inv), mods


def _set_hooks(g, hooks):
    for k, v in hooks.items():
        g.set_option(k, v)


def _unset_hooks(g, hooks):
    defaults = {"path": "default", "no_scratch": "0", "rns_force_fallback": "0", "u32_e32": "0x1f000", "rns_predict": "1",
                "check_4step_tables": "1"}
    for k in hooks:
        g.set_option(k, defaults[k])


def _run_merge(g, row):
    import torch
    poly = O.X_N_plus
    inverse = bool(row.get("inverse"))
    c = _host_case(g, row, poly)
    n, batch = c.n, row["batch"]
    tiny = row.get("modulus") == "tiny"
    m = _tiny(g, row["bits"]) if tiny else c.prm.modulus
    x = c.P.splitmix(17, 0, batch * n, 2 if tiny else c.q)
    d = g.to_device(x)
    cfg = g.ntt_configuration(n_power=row["logn"], ntt_type=g.INVERSE if inverse else g.FORWARD, reduction_poly=poly,
                              mod_inverse=c.prm.n_inv if inverse else 0)
    table = c.inv_dev if inverse else c.fwd_dev
    fn = g.GPU_INTT_Inplace if inverse else g.GPU_NTT_Inplace
    with g.launch_log() as log:
        fn(d, table, m, cfg, batch)
    torch.cuda.synchronize()
    want = None if tiny else oracle_batch([c], x, inverse=inverse)
    return log.kernels, g.to_host(d), want


def _run_merge_rns(g, row, ordered=False):
    import torch
    poly = O.X_N_plus
    inverse = bool(row.get("inverse"))
    bits, logn, batch = row["bits"], row["logn"], row["batch"]
    n = 1 << logn
    cases, real, d_fwd, d_inv, mods_h = _stack(g, row, poly)
    mc = len(cases)
    dt = real[0].P.T
    x = np.concatenate([(cases[p % mc] or real[0]).P.splitmix(40 + p, 0, n, cases[p % mc].q if cases[p % mc] else 2)
                        for p in range(batch)])
    ninv = g.to_device(np.array([(c.prm.n_inv if c else 1) for c in cases], dtype=dt))
    cfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD, reduction_poly=poly,
                                  mod_inverse=ninv if inverse else None)
    table = d_inv if inverse else d_fwd
    g.set_option("reset_predictions", "1")
    predict = row["predict"]
    order = torch.tensor(list(range(mc)), dtype=torch.int32, device="cuda") if ordered else None

    def call(d, mods):
        if ordered:
            g.GPU_NTT_Modulus_Ordered(d, d, table, mods, cfg, batch, mc, order)
        elif inverse:
            g.GPU_INTT_Inplace(d, table, mods, cfg, batch, mc)
        else:
            g.GPU_NTT_Inplace(d, table, mods, cfg, batch, mc)
        torch.cuda.synchronize()  # the state the preparation kernel found reaches the host-mapped word

    if predict == "wrong":
        # three calls of a [60, 60, ...] stack in this buffer, then the buffer is rewritten IN PLACE with the row's stack
        from gpu_utils import distinct_factors as _distinct_factors
        narrow = [MergeCase(g, bits, logn, poly, f) for f in _distinct_factors([60] * mc, logn)]
        mods = g.modulus_array_to_device([c.prm.modulus for c in narrow], bits)
        tab = np.zeros(mc * n, dtype=dt)
        for i, c in enumerate(narrow):
            tab[i * n:i * n + c.prm.root_of_unity_size] = c.prm.inverse_table_device_order if inverse else c.prm.forward_table_device_order
        tsave = table.clone()
        table.copy_(g.to_device(tab))
        xn = np.concatenate([narrow[p % mc].P.splitmix(90 + p, 0, n, narrow[p % mc].q) for p in range(batch)])
        for _ in range(3):
            call(g.to_device(xn), mods)
        table.copy_(tsave)
        mods.copy_(g.modulus_array_to_device(mods_h, bits))
    else:
        mods = g.modulus_array_to_device(mods_h, bits)
        if predict == "right":
            for _ in range(2):
                call(g.to_device(x), mods)
    d = g.to_device(x)
    with g.launch_log() as log:
        call(d, mods)
    got = g.to_host(d)
    want = got.copy()
    for p in range(batch):  # polynomials of the out-of-domain slot are whatever the Barrett operators give: not checked
        c = cases[p % mc]
        if c is not None:
            want[p * n:(p + 1) * n] = c.P.merge_ntt(x[p * n:(p + 1) * n], c.oprm, inverse=inverse)
    return log.kernels, got, want


def _run_4step(g, row):
    import torch
    from gpu_utils import cpu_class_on_tables as _cpu_class_on_tables
    bits, logn, batch = row["bits"], row["logn"], row["batch"]
    inverse = bool(row.get("inverse"))
    P = O.Port(bits)
    cls = row.get("modulus", "pool")
    p4 = g.NTTParameters4Step(logn, bits)
    oprm = P.fourstep_params(logn)
    q, n, n1, n2 = p4.modulus.value, p4.n, p4.n1, p4.n2
    t1, t2, w = [t.copy() for t in p4.tables["inv" if inverse else "fwd"]]
    modulus, n_inv, oracle_q = p4.modulus, p4.n_inv, None
    if cls == "b61":
        # a 61-bit modulus under the pool's (then meaningless) tables: the table check vetoes nothing it can test with a
        # consistent table, so the row uses tables REBUILT for a 61-bit prime's root of order N on the device
        qq, omega, _psi = find_ntt_factors(61, logn, clear_of_top=True)
        modulus = g.Modulus(qq, bits=bits)
        root = pow(omega, pow(2, 0), qq) if not inverse else pow(omega, qq - 2, qq)
        dw = torch.zeros(n, dtype=torch.int64, device="cuda")
        d1 = torch.zeros(n1 >> 1, dtype=torch.int64, device="cuda")
        d2 = torch.zeros(n2 >> 1, dtype=torch.int64, device="cuda")
        g.GPU_Generate4StepW(dw, root, modulus, logn, g.INVERSE if inverse else g.FORWARD)
        g.GPU_GeneratePowerTable(d1, pow(root, n2, qq), modulus, int(np.log2(n1)) - 1, True)
        g.GPU_GeneratePowerTable(d2, pow(root, n1, qq), modulus, int(np.log2(n2)) - 1, True)
        torch.cuda.synchronize()
        t1, t2, w = g.to_host(d1), g.to_host(d2), g.to_host(dw)
        q, n_inv, oracle_q = qq, pow(n, qq - 2, qq), qq
    if row.get("tables") == "vetoed":
        pos = 3 * n2 + 7
        w[pos] = (int(w[pos]) + 1) % q
    x = P.splitmix(77, 0, batch * n, q)
    d_in = g.to_device(x)
    d_out = torch.zeros_like(d_in)
    tabs = [g.to_device(t) for t in (t1, t2, w)]
    ntt_type = g.INVERSE if inverse else g.FORWARD
    if row["entry"] == "4step_rns":
        stack = row["stack"]
        mc = len(stack)
        mods = g.modulus_array_to_device([modulus] * mc, bits)
        ninv = g.to_device(np.array([n_inv] * mc, dtype=g.np_dtype(bits)))
        cfg = g.ntt4step_rns_configuration(n_power=logn, ntt_type=ntt_type, mod_inverse=ninv)
        g.set_option("reset_predictions", "1")
        call = lambda: g.GPU_4STEP_NTT(d_in, d_out, *tabs, mods, cfg, batch, mc)  # noqa: E731
        if row.get("predict") == "right":
            for _ in range(2):
                call()
                torch.cuda.synchronize()
    else:
        cfg = g.ntt4step_configuration(n_power=logn, ntt_type=ntt_type, mod_inverse=n_inv if inverse else 0)
        call = lambda: g.GPU_4STEP_NTT(d_in, d_out, *tabs, modulus, cfg, batch)  # noqa: E731
    d_out.fill_(-7)
    with g.launch_log() as log:
        call()
    torch.cuda.synchronize()
    if oracle_q is None:
        want = _cpu_class_on_tables(P, oprm, (t1, t2, w), x, batch, inverse, n_inv)
    else:
        t1n, t2n = P.bitrev_table(np.ascontiguousarray(t1)), P.bitrev_table(np.ascontiguousarray(t2))
        want = np.empty_like(x)
        for p in range(batch):
            a = x[p * n:(p + 1) * n]
            nat = np.ascontiguousarray(a.reshape(n1, n2).T if inverse else a.reshape(n2, n1).T).reshape(-1)
            r = P.fourstep_ntt_tables(nat, oprm, t1n, t2n, w, inverse, q=oracle_q, n_inv=n_inv)
            want[p * n:(p + 1) * n] = np.ascontiguousarray(r.reshape(n2, n1).T).reshape(-1)
    return log.kernels, g.to_host(d_out), want


@pytest.mark.parametrize("row", ROWS, ids=[r["id"] for r in ROWS])
def test_dispatch_row(g, row):
    hooks = row.get("hooks", {})
    _set_hooks(g, hooks)
    try:
        if row["entry"] == "merge":
            kernels, got, want = _run_merge(g, row)
        elif row["entry"] in ("merge_rns", "merge_ordered"):
            kernels, got, want = _run_merge_rns(g, row, ordered=row["entry"] == "merge_ordered")
        else:
            kernels, got, want = _run_4step(g, row)
    finally:
        _unset_hooks(g, hooks)
    if DISCOVER:
        print("\nDISCOVER %r -> %r%s" % (row["id"], kernels, "" if want is None or np.array_equal(got, want) else "   RESULT DIFFERS"))
        return
    if want is not None:
        assert np.array_equal(got, want), (row["id"], "result differs from the oracle")
    assert kernels == row["launches"], (row["id"], kernels)
