#!/usr/bin/env python3
"""Latency of single-polynomial transforms (the reference's own benchmark axis: batch = 1,
logN 12..24, benchmark/bench_merge_ntt.cu:57-75).  Run once per kernel family:
    GPUNTT_PATH=generic python tools/bench_batch1.py ; GPUNTT_PATH=fast python tools/bench_batch1.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import time_ms  # noqa: E402
from __graft_entry__ import _load_pkg  # noqa: E402
import numpy as np  # noqa: E402

g = _load_pkg()
g.load_library()
import torch  # noqa: E402

for bits in (() if os.environ.get('SKIP_MERGE') else (64, 32)):
    for logn in range(int(os.environ.get('LOGN_MIN', '12')), 25):
        prm = g.NTTParameters(logn, g.X_N_minus, bits)
        n = 1 << logn
        x = (np.arange(n, dtype=np.uint64) * 2654435761 % prm.modulus.value).astype(g.np_dtype(bits))
        d = g.to_device(x)
        tab = g.to_device(prm.forward_table_device_order)
        cfg = g.ntt_configuration(n_power=logn, reduction_poly=g.X_N_minus)
        fn = lambda: g.GPU_NTT_Inplace(d, tab, prm.modulus, cfg, 1)  # noqa: E731
        ms = time_ms(fn, 200 if logn < 20 else 50, warm=10)
        plan = g.NTTPlan(tab, prm.modulus, logn, g.X_N_minus, g.FORWARD, batch_hint=1)
        msp = time_ms(lambda: plan.execute(d, d, 1), 200 if logn < 20 else 50, warm=10)
        plan.close()
        print(json.dumps({"path": os.environ.get("GPUNTT_PATH", "auto"), "dtype": "u%d" % bits, "log2N": logn,
                          "batch": 1, "us": round(ms * 1e3, 2), "plan_us": round(msp * 1e3, 2)}), flush=True)

# 4-step, single polynomial (benchmark/bench_4step_ntt.cu:96-100 sweeps the same axis), pre-transposed input
for logn in range(12, 25):
    p4 = g.NTTParameters4Step(logn, 64)
    x = (np.arange(p4.n, dtype=np.uint64) * 2654435761 % p4.modulus.value).astype(np.uint64)
    a = g.to_device(x)
    b = torch.empty_like(a)
    tf = [g.to_device(t) for t in p4.tables["fwd"]]
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
    fn = lambda: g.GPU_4STEP_NTT(a, b, *tf, p4.modulus, cf, 1)  # noqa: E731
    ms = time_ms(fn, 200 if logn < 20 else 50, warm=10)
    fn2 = lambda: g.GPU_4STEP_NTT_NaturalOrder(a, b, *tf, p4.modulus, cf, 1)  # noqa: E731
    ms2 = time_ms(fn2, 200 if logn < 20 else 50, warm=10)
    pl = g.FourStepPlan(*tf, p4.modulus, cf, natural_order=False, batch_hint=1)
    ms3 = time_ms(lambda: pl.execute(a, b, 1), 200 if logn < 20 else 50, warm=10)
    pl.close()
    pl = g.FourStepPlan(*tf, p4.modulus, cf, natural_order=True, batch_hint=1)
    ms4 = time_ms(lambda: pl.execute(a, b, 1), 200 if logn < 20 else 50, warm=10)
    pl.close()
    print(json.dumps({"path": os.environ.get("GPUNTT_PATH", "auto"), "dtype": "u64", "log2N": logn, "batch": 1,
                      "algo": "4step-fwd", "us": round(ms * 1e3, 2), "natural_order_us": round(ms2 * 1e3, 2),
                      "plan_us": round(ms3 * 1e3, 2), "plan_natural_order_us": round(ms4 * 1e3, 2)}),
          flush=True)
