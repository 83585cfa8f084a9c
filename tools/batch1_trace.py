#!/usr/bin/env python3
"""Where does a single-polynomial transform's time go?  
    python tools/batch1_trace.py run LOGN [BITS]          the workload: 300 x NTTPlan::execute(batch = 1), in place
    python tools/batch1_trace.py summarise DIR            per-kernel average duration, the gap between the two kernels of a
                                                          call (kernel boundary) and the gap between calls, from a
                                                          rocprofv3 --kernel-trace results.db of the line above
  e.g.  (cd /tmp && rocprofv3 --kernel-trace -d $O/kt16 -o kt -- python $R/tools/batch1_trace.py run 16)"""
import glob
import os
import sqlite3
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(logn, bits):
    import numpy as np
    import torch
    from __graft_entry__ import _load_pkg
    g = _load_pkg()
    g.load_library()
    prm = g.NTTParameters(logn, g.X_N_minus, bits)
    n = 1 << logn
    x = (np.arange(n, dtype=np.uint64) * 2654435761 % prm.modulus.value).astype(g.np_dtype(bits))
    d = g.to_device(x)
    tab = g.to_device(prm.forward_table_device_order)
    plan = g.NTTPlan(tab, prm.modulus, logn, g.X_N_minus, g.FORWARD, batch_hint=1)
    for _ in range(300):
        plan.execute(d, d, 1)
    torch.cuda.synchronize()


def summarise(path):
    for db in glob.glob(path + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        sym = [n for n in names if n.startswith("rocpd_info_kernel_symbol")][0]
        dis = [n for n in names if n.startswith("rocpd_kernel_dispatch")][0]
        rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {dis} d join {sym} s on d.kernel_id=s.id "
                              f"where s.kernel_name like '%merge_pass_lazy%' order by d.start"))
        rows = rows[len(rows) // 3:]  # steady state
        kinds = sorted({r[0] for r in rows})
        per_call = len(kinds)
        dur = {k: [] for k in kinds}
        for k, s, e in rows:
            dur[k].append((e - s) / 1e3)
        inner, outer = [], []
        for a, b in zip(rows, rows[1:]):
            (inner if a[0] != b[0] or per_call == 1 and False else outer).append((b[1] - a[2]) / 1e3)
        if per_call == 1:
            inner, outer = [], [(b[1] - a[2]) / 1e3 for a, b in zip(rows, rows[1:])]
        else:
            first = rows[0][0]
            inner = [(b[1] - a[2]) / 1e3 for a, b in zip(rows, rows[1:]) if b[0] != first]
            outer = [(b[1] - a[2]) / 1e3 for a, b in zip(rows, rows[1:]) if b[0] == first]
        med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")  # noqa: E731
        print("# %s" % db)
        for k in kinds:
            print("kernel %-100s median %.2f us (n=%d)" % (k[:100], med(dur[k]), len(dur[k])))
        print("kernels per call %d; sum of kernel medians %.2f us; gap between the kernels of a call (boundary) median %.2f us; "
              "gap between calls median %.2f us; call period median %.2f us"
              % (per_call, sum(med(dur[k]) for k in kinds), med(inner), med(outer),
                 med([(b[1] - a[1]) / 1e3 for a, b in zip(rows[::per_call], rows[per_call::per_call])])))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 64)
    else:
        summarise(sys.argv[2])
