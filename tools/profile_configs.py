#!/usr/bin/env python3
"""Short run of the non-headline configs (C3 at batch 16, C4, C5) for `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg -- python tools/profile_configs.py
    python tools/rocprof_summary.py gpurun_out/prof_cfg > profiles/rNN_c3_c4_c5_kernel_stats.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc  # noqa: E402
from __graft_entry__ import _load_pkg  # noqa: E402

g = _load_pkg()
g.load_library()
bc.fourstep_case(g, 64, 24, 16, 4, "C3q 4-Step u64 2^24 x16", check=False)
bc.merge_case(g, 32, 14, 1024, g.X_N_minus, 10, "C4")
bc.rns_case(g, 16, 512, 10, "C5", os.path.join(ROOT, "tests", "golden"))
