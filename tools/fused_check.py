#!/usr/bin/env python3
"""Quick parity probe of the single-sweep kernel (GPUNTT_FUSED=1): sizes 2^13..2^18, both directions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_pkg  # noqa: E402
from gpu_utils import MergeCase  # noqa: E402
from oracle import oracle as O  # noqa: E402

g = _load_pkg()
g.load_library()
bad = 0
sizes = [int(a) for a in sys.argv[1:]] or [13, 14, 15, 16, 17, 18]
for logn in sizes:
    for poly in (O.X_N_minus, O.X_N_plus):
        c = MergeCase(g, 64, logn, poly)
        for batch in (1, 3, 17):
            x = c.random(batch, 77 + logn)
            want = c.P.merge_ntt(x, c.oprm)
            for inplace in (False, True):
                got = c.gpu_forward(x, inplace=inplace)
                ok = np.array_equal(got, want)
                if not ok:
                    bad += 1
                    d = np.nonzero(got != want)[0]
                    print("FWD MISMATCH logn", logn, "poly", poly, "batch", batch, "inplace", inplace,
                          "count", d.size, "first", d[:8], "polys", sorted(set((d >> logn).tolist()))[:8])
                back = c.gpu_inverse(want, inplace=inplace)
                if not np.array_equal(back, x):
                    bad += 1
                    d = np.nonzero(back != x)[0]
                    print("INV MISMATCH logn", logn, "poly", poly, "batch", batch, "inplace", inplace,
                          "count", d.size, "first", d[:8])
        print("logn", logn, "poly", poly, "done", flush=True)
print("bad", bad)
