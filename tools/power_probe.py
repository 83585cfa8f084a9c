#!/usr/bin/env python3
"""Socket power / shader clock (rocm-smi) while one workload loops for ~6 s:
    python tools/power_probe.py copy|c2|c2i|n12|n12u32|c4
copy = device-to-device copy of 1 GiB (pure HBM streaming), n12 = single-sweep 2^12 u64 transform of 2^26 coefficients."""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from __graft_entry__ import _load_pkg  # noqa: E402

g = _load_pkg()
g.load_library()
what = sys.argv[1]


def smi():
    out = subprocess.run("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Socket' | sed 's/.*: //' | tr '\\n' ' '",
                         shell=True, capture_output=True, text=True).stdout.strip()
    return out


def merge_step(bits, logn, batch, inverse=False):
    prm = g.NTTParameters(logn, g.X_N_minus, bits)
    n = 1 << logn
    x = (np.arange(batch * n, dtype=np.uint64) * 0x9E3779B97F4A7C15 % prm.modulus.value).astype(g.np_dtype(bits))
    d_in = g.to_device(x)
    d_out = torch.empty_like(d_in)
    tab = g.to_device(prm.inverse_table_device_order if inverse else prm.forward_table_device_order)
    cfg = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD, reduction_poly=g.X_N_minus,
                              mod_inverse=prm.n_inv if inverse else 0)
    plan = g.NTTPlan(tab, prm.modulus, logn, g.X_N_minus, g.INVERSE if inverse else g.FORWARD,
                     mod_inverse=prm.n_inv if inverse else None, batch_hint=batch)
    return lambda: plan.execute(d_in, d_out, batch), 2 * n * (bits // 8) * batch


if what == "copy":
    a = torch.empty(1 << 27, dtype=torch.int64, device="cuda:0")
    b = torch.empty_like(a)
    step, nbytes = (lambda: b.copy_(a)), 2 * a.numel() * 8
elif what == "c2":
    step, nbytes = merge_step(64, 16, 1024)
elif what == "c2i":
    step, nbytes = merge_step(64, 16, 1024, True)
elif what == "n12":
    step, nbytes = merge_step(64, 12, 16384)
elif what == "n12u32":
    step, nbytes = merge_step(32, 12, 32768)
elif what == "c4":
    step, nbytes = merge_step(32, 14, 8192)
else:
    raise SystemExit("unknown workload")

samples = []
stop = False


def sampler():
    time.sleep(2.5)
    for _ in range(4):
        samples.append(smi())


th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
calls = 0
while th.is_alive():
    for _ in range(64):
        step()
    calls += 64
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
th.join()
print("%-8s %.4f ms/call  %.0f GB/s (algorithmic)   clock / socket W: %s" % (what, dt / calls * 1e3, nbytes * calls / dt / 1e9,
                                                                            " | ".join(samples)), flush=True)
