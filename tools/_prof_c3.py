import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_configs as bc
from __graft_entry__ import _load_pkg
g = _load_pkg(); g.load_library()
bc.fourstep_case(g, 64, 24, 16, 4, "C3q 4-Step u64 2^24 x16", check=False)
bc.merge_case(g, 32, 14, 1024, g.X_N_minus, 10, "C4")
bc.rns_case(g, 16, 512, 10, "C5", '/root/repo/tests/golden')
