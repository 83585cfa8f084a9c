import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
from oracle import oracle as O
import torch
g = load_pkg(); g.load_library()
from test_gpu_4step import run_fourstep
P = O.Port(64)
for (logn, batch) in ((14, 256), (14, 300), (13, 64), (12, 128)):
    p4 = g.NTTParameters4Step(logn, 64)
    oprm = P.fourstep_params(logn)
    x = P.splitmix(1200 + logn + batch, 0, batch * p4.n, p4.modulus.value)
    want = np.concatenate([P.fourstep_ntt(x[p * p4.n:(p + 1) * p4.n], oprm) for p in range(batch)])
    xin = np.concatenate([P.fourstep_intt_first_transpose(want[p * p4.n:(p + 1) * p4.n], oprm) for p in range(batch)])
    for it in range(6):
        for rns in (False, True):
            got = run_fourstep(g, p4, x, batch, inverse=False, rns=rns)
            bad = [p for p in range(batch) if not np.array_equal(got[p * p4.n:(p + 1) * p4.n], want[p * p4.n:(p + 1) * p4.n])]
            back = run_fourstep(g, p4, xin, batch, inverse=True, rns=rns)
            badi = [p for p in range(batch) if not np.array_equal(back[p * p4.n:(p + 1) * p4.n], x[p * p4.n:(p + 1) * p4.n])]
            if bad or badi:
                print("logn", logn, "batch", batch, "iter", it, "rns", rns, "fwd bad polys", bad[:10], len(bad), "inv bad polys", badi[:10], len(badi), flush=True)
                if badi:
                    p = badi[0]
                    d = np.nonzero(back[p * p4.n:(p + 1) * p4.n] != x[p * p4.n:(p + 1) * p4.n])[0]
                    print("   inv first bad poly", p, "mismatches", d.size, "first idx", d[:8], "last", d[-4:], flush=True)
                if bad:
                    p = bad[0]
                    d = np.nonzero(got[p * p4.n:(p + 1) * p4.n] != want[p * p4.n:(p + 1) * p4.n])[0]
                    print("   fwd first bad poly", p, "mismatches", d.size, "first idx", d[:8], "last", d[-4:], flush=True)
    print("done", logn, batch, flush=True)
