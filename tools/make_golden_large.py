#!/usr/bin/env python3
"""Reference-build digests of ONE forward Merge transform at the top of the documented range (n_power 27 and 28,
reference ntt.cu:2088-2091) -> tests/golden/digests_large.json.

Run in the build container (where /root/reference is mounted), ~10 minutes and ~25 GiB of host memory:
    make -C oracle ref && python tools/make_golden_large.py

The input is the portable stream x[k] = splitmix64(seed ^ k) mod q (oracle/ntt_oracle.c), so only the seed is stored;
the output digest is SHA-256 of NTTCPU<Data64>::ntt(x) from the reference's own class (oracle/_ref)."""
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = ((64, 27, O.X_N_minus), (64, 28, O.X_N_plus))
SEED = 0x5EED2700


def main():
    out = {"generator": "tools/make_golden_large.py", "source": "NTTCPU<T>::ntt of the reference build (oracle/_ref)", "cases": []}
    for bits, logn, poly in CASES:
        R, P = O.Ref(bits), O.Port(bits)
        t0 = time.time()
        h = ctypes.c_void_p(R.f("merge_create")(logn, poly, 0, R.c(0), R.c(0), R.c(0)))
        info = (ctypes.c_uint64 * 10)()
        R.f("merge_info")(h, info)
        q, n = int(info[0]), 1 << logn
        x = P.splitmix(SEED + logn, 0, n, q)
        y = np.empty_like(x)
        R.f("merge_run")(h, 0, O._ptr(x), O._ptr(y), 1)
        R.f("merge_destroy")(h)
        rec = {"bits": bits, "logn": logn, "poly": int(poly), "seed": SEED + logn, "q": q, "omega": int(info[3]), "psi": int(info[4]),
               "sha256_input": hashlib.sha256(x.tobytes()).hexdigest(),
               "sha256_forward": hashlib.sha256(y.tobytes()).hexdigest(),
               "first_words": [int(v) for v in y[:4]], "last_words": [int(v) for v in y[-4:]]}
        out["cases"].append(rec)
        print(rec, "%.0f s" % (time.time() - t0), flush=True)
        del x, y
    with open(os.path.join(ROOT, "tests", "golden", "digests_large.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
