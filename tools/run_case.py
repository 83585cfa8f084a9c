#!/usr/bin/env python3
"""Run ONE configuration a few times (for `rocprofv3 --pmc ...` / `--kernel-trace` passes):
    python tools/run_case.py c2|c2i|c3|c4|c4full|c5|merge:BITS:LOGN:BATCH[:inv] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc  # noqa: E402
from __graft_entry__ import _load_pkg  # noqa: E402

case = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = _load_pkg()
if os.environ.get("GPUNTT_LIB"):  # experiment builds (may be older than the current C ABI)
    g.LIB_PATH = os.environ["GPUNTT_LIB"]
    g.EXPORTED_SYMBOLS = ["gpuntt_last_error"]
g.load_library()
if case == "c2":
    bc.merge_case(g, 64, 16, 1024, g.X_N_minus, iters, "C2")
elif case == "c2i":
    bc.merge_case(g, 64, 16, 1024, g.X_N_minus, iters, "C2i", inverse=True)
elif case == "c3":
    bc.fourstep_case(g, 64, 24, 16, iters, "C3q 4-Step u64 2^24 x16", check=False)
elif case == "c4":
    bc.merge_case(g, 32, 14, 1024, g.X_N_minus, iters, "C4")
elif case == "c4full":
    bc.merge_case(g, 32, 14, 8192, g.X_N_minus, iters, "C4full")
elif case == "c5":
    bc.rns_case(g, 16, 512, iters, "C5", os.path.join(ROOT, "tests", "golden"))
elif case.startswith("merge:"):  # merge:BITS:LOGN:BATCH[:inv]
    f = case.split(":")
    bc.merge_case(g, int(f[1]), int(f[2]), int(f[3]), g.X_N_minus, iters, case, inverse=len(f) > 4)
else:
    raise SystemExit("unknown case " + case)
