#!/usr/bin/env python3
"""Knock-out builds of the 32-bit single-sweep kernel (merge_e32_kernels.hpp): where do the 0.25 ms of C4 go?

    python tools/ko_e32.py build      (CPU: patched copies of the kernel header under /tmp/ko/<variant>/, one library each
                                       as tools/_exp_ko_<variant>.so -- the product sources are not touched)
    python tools/ko_e32.py time       (GPU: C4-shaped calls, u32 2^14 x 8192, through every variant; TIMES ONLY -- every
                                       variant but `full` computes garbage on purpose)
Variants: full; nonorm (no final normalisation); nocorr (no range corrections); notwc / notwb (round C / round B twiddles
from a register constant instead of memory); nobfly (no butterflies: loads, the two exchanges, the transposition, stores);
direct (no final transposition: 16-byte stores straight from the 32 contiguous coefficients of a lane)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu-ntt_amd", "csrc")
VARIANTS = ["full", "nonorm", "nocorr", "notwc", "notwb", "nobfly", "direct"]


def patch(src, v):
    if v == "nonorm":
        src = src.replace("v[j] = lazy::normalize<SCH::d.final_bound>(m, v[j]);", "")
    if v == "nocorr":
        src = src.replace("""                    if constexpr (ku > 0)
                        U = m.template csub<ku>(U);
                    if constexpr (ku < 0)
                        U = m.reduce_2q(U);
                    const T nu = unit""", "                    const T nu = unit")
    if v == "notwc":
        src = src.replace("""                load_tw_c(twv);
                // ---- exchange B -> C""", """                for (int i = 0; i < E32 - 1; i++) twv[i] = TW{a.ninv.w, a.ninv.wp};
                // ---- exchange B -> C""")
    if v == "notwb":
        src = src.replace("load_tw_b(twv, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});\n                if (CANON_IN",
                          "for (int i = 0; i < E32 - 1; i++) twv[i] = TW{a.ninv.w, a.ninv.wp};\n                if (CANON_IN")
        src = src.replace("                load_tw_b(twv, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});\n", "")
    if v == "nobfly":
        src = src.replace("""                    T U = u;
                    if constexpr (ku > 0)""", """                    return;
                    T U = u;
                    if constexpr (ku > 0)""")
    if v == "direct":
        a = src.index("                // ---- 32 contiguous coefficients per lane -> 1 KiB runs per store instruction")
        b = src.index("            else\n            {\n                using SCH = EInvSched")
        src = src[:a] + """                {
#pragma unroll
                    for (int k = 0; k < E32 / 4; k++)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]}, rdst,
                                                               static_cast<int>(tu << 7), 16 * k, POL_OUT);
                }
            }
""" + src[b:]
    return src


def build():
    src = open(os.path.join(CSRC, "merge_e32_kernels.hpp")).read()
    objs = [os.path.join(CSRC, "_obj", f) for f in os.listdir(os.path.join(CSRC, "_obj")) if f.endswith(".o") and f != "lazy_e32.o"]
    for v in VARIANTS:
        d = "/tmp/ko/" + v
        os.makedirs(d, exist_ok=True)
        out = patch(src, v)
        assert v == "full" or out != src, v
        open(os.path.join(d, "merge_e32_kernels.hpp"), "w").write(out)
        o = os.path.join(d, "lazy_e32.o")
        # the patched header shadows the product's: -I <variant dir> first, and lazy_e32.hip is compiled from a copy next to it
        open(os.path.join(d, "lazy_e32.hip"), "w").write(open(os.path.join(CSRC, "lazy_e32.hip")).read())
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-I" + d, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                               "--offload-arch=gfx950", "-Wno-unused-result", "-ffp-contract=off", "-c", os.path.join(d, "lazy_e32.hip"), "-o", o])
        so = os.path.join(ROOT, "tools", "_exp_ko_%s.so" % v)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o, "-Wl,-rpath,/opt/rocm/lib"])
        print("built", so, flush=True)


def time_all():
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from __graft_entry__ import _load_pkg
g = _load_pkg(); g.load_library()
logn, batch = 14, 8192
prm = g.NTTParameters(logn, g.X_N_minus, 32)
x = np.random.default_rng(1).integers(0, prm.modulus.value, size=batch << logn, dtype=np.uint64).astype(np.uint32)
d = g.to_device(x); tab = g.to_device(prm.forward_table_device_order)
cfg = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=g.X_N_minus)
plan = g.NTTPlan(tab, prm.modulus, logn, g.X_N_minus, g.FORWARD, batch_hint=batch)
f = lambda: plan.execute(d, d, batch)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    f(); torch.cuda.synchronize()
    d.copy_(g.to_device(x)) if False else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = []
for rep in range(3):
    d.copy_(torch.from_numpy(x.view(np.int32)).cuda())
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50)
print("%%.4f %%.4f %%.4f" %% tuple(res))
''' % ROOT
    for rnd in range(2):
        for v in VARIANTS:
            so = os.path.join(ROOT, "tools", "_exp_ko_%s.so" % v)
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPUNTT_LIB=so), capture_output=True, text=True)
            print("%-8s %s" % (v, (r.stdout.strip() or r.stderr[-300:])), flush=True)


if __name__ == "__main__":
    {"build": build, "time": time_all}[sys.argv[1]]()
