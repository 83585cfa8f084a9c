#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X NTT library.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--api dropin|plan]
                    [--no-cpu-baseline] [--no-traffic] [--no-e2e]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default = BASELINE.json's metric configuration (configs[1], "c2"): forward Merge NTT, 64-bit, N = 2^16,
batch = 1024 per GPU, through the drop-in GPU_NTT call, IN PLACE like the reference's own benchmark
(benchmark/bench_merge_ntt.cu:62 times GPU_NTT_Inplace; --out-of-place times GPU_NTT(in, out)).  One "step" = one
library call (all of its kernel launches) over one batch of synthetic random polynomials already resident in HBM.

    c2  Merge u64 2^16 x 1024 per GPU, forward, X^N-1                    (weak scaling)
    c3  4-Step u64 2^24 x 64 per GPU, forward + inverse pair              (weak scaling)
    c4  Merge u32 2^14, 8192 polynomials in total sharded over the ranks  (strong scaling: the config is
        DEFINED as 8-way sharded; N = 1 runs the whole batch on one GPU)
    c5  RNS Merge u64 2^16 x 512 per GPU, 8 primes, X^N+1                 (weak scaling)

Multi-GPU: rank r owns its own shard (polynomials are independent): the timed region has
no data-path collective, only the barrier and the MAX-reduction of the elapsed time.  With N > 1 the run
also reports (outside `value`) the end-to-end leg of SURVEY.md 8e(ii): RCCL broadcast of the twiddle
table, scatter of a batch that starts on rank 0, transform, gather back -- each part timed on its own.

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" has the roofline arithmetic).
"""
import os
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # cpu_baseline's OpenMP leg: idle threads must not spin away a container quota
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
REDUCE_DEV = None   # device of the control-plane reductions: the rank's GPU under RCCL, the host under gloo
E2E_TIMEOUT_S = 180  # multi-GPU runs: watchdog around the informational RCCL scatter / gather leg
CPU_SECONDS = 5.0      # per CPU-baseline leg (1 thread, then all threads)

CONFIGS = {
    "c2": dict(kind="merge", bits=64, logn=16, batch=1024, poly="minus", scaling="weak",
               metric="forward-NTTs/sec + achieved HBM GB/s, 64-bit Merge N=2^16 batch=1024",
               workload="Merge-NTT Data64 log2N=16 batch=1024 forward (BASELINE configs[1])"),
    "c3": dict(kind="4step", bits=64, logn=24, batch=64, poly="minus", scaling="weak",
               metric="4-Step transforms/sec (forward + inverse pairs, each counted), 64-bit N=2^24 batch=64",
               workload="4-Step NTT+INTT Data64 log2N=24 batch=64 (BASELINE configs[2])"),
    "c4": dict(kind="merge", bits=32, logn=14, batch=8192, poly="minus", scaling="strong",
               metric="forward-NTTs/sec + achieved HBM GB/s, 32-bit Merge N=2^14, 8192 polynomials sharded",
               workload="Merge-NTT Data32 log2N=14 batch=8192 sharded over the ranks (BASELINE configs[3])"),
    "c5": dict(kind="rns", bits=64, logn=16, batch=512, poly="plus", scaling="weak", mod_count=8,
               metric="forward-NTTs/sec + achieved HBM GB/s, RNS 8 x 60-bit primes, N=2^16 batch=512",
               workload="RNS Merge-NTT 8 primes Data64 log2N=16 batch=512 forward (BASELINE configs[4])"),
}


def splitmix64_mod(seed, count, q, offset=0):
    """x[k] = splitmix64(seed ^ (offset + k)) mod q (the portable synthetic input of SURVEY.md 8d),
    vectorised with numpy's wrapping uint64 arithmetic."""
    k = np.arange(offset, offset + count, dtype=np.uint64)
    x = (np.uint64(seed) ^ k) + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x = x ^ (x >> np.uint64(31))
    return x % np.uint64(q)


# ------------------------------------------------------------------------------ CPU baseline
def _timed_cpu(run_one, polys_total, threads, seconds):
    """run_one(lo, hi) transforms polynomials [lo, hi); repeat over the sample until `seconds` passed."""
    from concurrent.futures import ThreadPoolExecutor
    done = 0
    t0 = time.perf_counter()
    if threads == 1:
        while time.perf_counter() - t0 < seconds:
            run_one(0, polys_total)
            done += polys_total
    else:
        per = max(1, polys_total // threads)
        spans = [(i, min(i + per, polys_total)) for i in range(0, polys_total, per)]
        with ThreadPoolExecutor(max_workers=threads) as ex:
            while time.perf_counter() - t0 < seconds:
                list(ex.map(lambda s: run_one(*s), spans))  # ctypes releases the GIL inside the C call
                done += polys_total
    return done / (time.perf_counter() - t0)


def cpu_baseline(cfg, case, y_gpu_sample, check_only=False):
    """The CPU leg (the only place bench.py touches oracle/): the reference's own CPU transform
    (oracle/_ref, kind 'reference'; this repo's C restatement, kind 'port', when the prebuilt file is absent)
    timed on a bounded sample of the same workload on this host, once on one thread and once on every host
    thread; its output also checks the GPU result of the same polynomials bit for bit."""
    from oracle import oracle as O
    bits, logn = cfg["bits"], cfg["logn"]
    n = 1 << logn
    kind = "reference" if O.have_ref() else "port"
    B = O.Ref(bits) if O.have_ref() else O.Port(bits)
    threads = os.cpu_count() or 1
    x = case["x_sample"]
    polys = x.size // n
    if cfg["kind"] == "4step":
        prm = B.fourstep_params(logn)
        fwd = (lambda a: B.fourstep_run(a, prm, 0)) if O.have_ref() else (lambda a: B.fourstep_ntt(a, prm))
        t0 = time.perf_counter()
        y = fwd(x[:n])
        dt = time.perf_counter() - t0
        ok = np.array_equal(y, y_gpu_sample[:n])
        if not ok:
            raise SystemExit("bench: GPU result differs from the CPU reference path")
        if check_only:
            return {"gpu_output_bit_exact": True, "kind": kind, "polynomials": 1}
        return {"value": 1.0 / dt, "unit": "NTT/s", "cores": 1, "kind": kind, "gpu_output_bit_exact": True,
                "sample": "one forward NTT_4STEP_CPU::ntt of N=2^%d on one host core, %.1f s" % (logn, dt)}
    poly = O.X_N_minus if cfg["poly"] == "minus" else O.X_N_plus
    inverse = cfg.get("direction", "fwd") == "inv"
    if cfg["kind"] == "rns":
        prms = [B.merge_params(logn, poly, f) for f in case["factors"]]
        mc = len(prms)

        def run_one(lo, hi):
            return [B.merge_ntt(x[p * n:(p + 1) * n], prms[p % mc], inverse) for p in range(lo, hi)]
    else:
        prm = B.merge_params(logn, poly)

        def run_one(lo, hi):
            return [B.merge_ntt(x[lo * n:hi * n], prm, inverse)]
    y = np.concatenate(run_one(0, polys))
    if not np.array_equal(y, y_gpu_sample):
        raise SystemExit("bench: GPU result differs from the CPU reference path")
    if check_only:
        return {"gpu_output_bit_exact": True, "kind": kind, "polynomials": polys}
    v1 = _timed_cpu(run_one, polys, 1, CPU_SECONDS)
    how = "batch-parallel ctypes calls from a thread pool"
    if threads > 1 and cfg["kind"] != "rns" and hasattr(B, "merge_ntt_mt"):
        # all host threads: ONE call into the reference build, OpenMP loop over the polynomials around NTTCPU::ntt
        # (oracle/ref_driver.cpp) -- what the host can really do, without a Python task per polynomial
        reps = max(1, (4 * threads + polys - 1) // polys)
        xs = np.tile(x, reps)
        assert np.array_equal(B.merge_ntt_mt(xs, prm, inverse, threads)[:y.size], y)
        # the box may hand this process fewer cores than it lists (container quota): every thread count T, T/2, T/4, T/8
        # gets its share of the time budget, the best one is reported with the threads it used
        cands = sorted({max(1, threads >> k) for k in range(4)}, reverse=True)
        vt, used = 0.0, threads
        for tc in cands:
            done, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < CPU_SECONDS / len(cands):
                B.merge_ntt_mt(xs, prm, inverse, tc)
                done += polys * reps
            rate = done / (time.perf_counter() - t0)
            if rate > vt:
                vt, used = rate, tc
        how = "one OpenMP loop over %d polynomials inside the reference build; best of %s threads: %d" % (polys * reps, cands, used)
        threads = used
    else:
        vt = _timed_cpu(run_one, polys, threads, CPU_SECONDS) if threads > 1 else v1
    return {"value": vt, "unit": "NTT/s", "cores": threads, "kind": kind, "value_1_core": v1,
            "gpu_output_bit_exact": True,
            "sample": "%d polynomials of one batch (u%d, N=2^%d) repeated for %.0f s on 1 host thread and "
                      "%.0f s on %d host threads (%s), NTTCPU::%s"
                      % (polys, bits, logn, CPU_SECONDS, CPU_SECONDS, threads, how, "intt" if inverse else "ntt")}


# --------------------------------------------------------------------------- HBM traffic (PMC)
def measure_traffic(config, api, out_of_place=False, direction="fwd"):
    """HBM bytes per call from the PMC counters, measured in THIS run: two short rocprofv3 --pmc child
    passes of this script (FETCH_SIZE, WRITE_SIZE -- separate passes, MI355X_MICROARCH.md), gfx950 x2
    correction on the fetch side.  Returns (bytes_per_call, detail) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    steps = 5
    totals = {}
    tmp = tempfile.mkdtemp(prefix="gpuntt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--config", config, "--api", api, "--steps", str(steps),
                   "--warmup", "0", "--child", "--direction", direction] + (["--out-of-place"] if out_of_place else [])
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=180)
            dbs = glob.glob(out + "/**/*.db", recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed: %s" % (counter, (r.stderr or r.stdout)[-300:])
            c = sqlite3.connect(dbs[0])
            names = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
            t = lambda p: [n for n in names if n.startswith(p)][0]  # noqa: E731
            q = (f"select s.kernel_name, count(distinct d.id), sum(p.value) from {t('rocpd_pmc_event')} p "
                 f"join {t('rocpd_info_pmc')} i on p.pmc_id=i.id "
                 f"join {t('rocpd_kernel_dispatch')} d on p.event_id=d.event_id "
                 f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id=s.id "
                 f"where i.name='{counter}' and s.kernel_name like '%gpuntt%' group by s.kernel_name")
            totals[counter] = {k: (nl, v) for k, nl, v in c.execute(q)}
        calls = None
        bytes_per_call = 0.0
        detail = {}
        for counter, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            for k, (launches, kb) in totals[counter].items():
                # child run: 1 check call + `steps` timed calls, every call launches each kernel once
                calls = steps + 1
                b = kb * 1024.0 * corr / calls
                bytes_per_call += b
                detail.setdefault(k[:96], {})[counter] = b
        return bytes_per_call, {"per_kernel_bytes_per_call": detail, "fetch_correction": 2.0,
                                "calls_profiled": calls, "method": "rocprofv3 --pmc, two passes, this run"}
    except Exception as e:  # profiling must never take the bench line down
        return None, "traffic measurement failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class PowerReader:
    """Socket power / shader clock / power cap read IN PROCESS (no fork): amdsmi if importable, else the amdgpu hwmon
    files in sysfs.  read() -> (watts | None, sclk_mhz | None); cap_w is the socket power limit."""

    def __init__(self, index=0):
        self.kind, self.cap_w = None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self._smi = amdsmi
            self._h = amdsmi.amdsmi_get_processor_handles()[index]
            info = amdsmi.amdsmi_get_power_info(self._h)
            cap = info.get("power_limit")
            if isinstance(cap, (int, float)) and cap > 0:
                self.cap_w = float(cap) / (1e6 if cap > 100000 else 1.0)
            self.kind = "amdsmi"
            if self.read()[0] is None:
                self.kind = None
        except Exception:
            self.kind = None
        if self.kind is None:
            import glob
            for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
                for f in ("power1_input", "power1_average"):
                    if os.path.exists(os.path.join(hw, f)):
                        self._pw, self._hw, self.kind = os.path.join(hw, f), hw, "sysfs"
                        break
                if self.kind:
                    try:
                        self.cap_w = int(open(os.path.join(hw, "power1_cap")).read()) / 1e6
                    except Exception:
                        pass
                    break

    def read(self):
        try:
            if self.kind == "amdsmi":
                info = self._smi.amdsmi_get_power_info(self._h)
                w = None
                for k in ("current_socket_power", "average_socket_power", "socket_power"):
                    v = info.get(k)
                    if isinstance(v, (int, float)) and v > 0:
                        w = float(v)
                        break
                try:
                    clk = self._smi.amdsmi_get_clock_info(self._h, self._smi.AmdSmiClkType.GFX).get("clk")
                    clk = int(clk) if isinstance(clk, (int, float)) else None
                except Exception:
                    clk = None
                return w, clk
            if self.kind == "sysfs":
                w = int(open(self._pw).read()) / 1e6
                clk = None
                try:
                    clk = int(open(os.path.join(self._hw, "freq1_input")).read()) // 1000000
                except Exception:
                    pass
                return w, clk
        except Exception:
            pass
        return None, None


def measure_power(step, ms_ref, seconds=3.0):
    """Socket power and shader clock sampled while the step loops for `seconds` -- after the timed region, never part of
    `value`.  The launch loop keeps two ~30 ms chunks of calls queued ahead of the GPU and the sensors are read in
    this process between chunks (no fork, no thread), so the GPU never waits for the host; `ms_per_step_while_sampling`
    (HIP events around the whole loop) must agree with the timed `ms_per_step` for the samples to describe the same
    regime -- main() withdraws the energy model otherwise.  Returns None without an in-process sensor."""
    import torch
    rd = PowerReader(int(os.environ.get("LOCAL_RANK", "0")))
    if rd.kind is None:
        return None
    chunk = max(8, min(4096, int(30.0 / max(ms_ref, 1e-3))))
    pending, samples, calls = [], [], 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    while True:
        for _ in range(chunk):
            step()
        calls += chunk
        ev = torch.cuda.Event()
        ev.record()
        pending.append(ev)
        if len(pending) > 2:
            pending.pop(0).synchronize()
        now = time.perf_counter() - t0
        if now > 0.8:  # clocks and the power average have settled under the load
            samples.append(rd.read())
        if now > seconds:
            break
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / calls
    ws = [w for w, _ in samples if w is not None]
    if not ws:
        return None
    clks = [c for _, c in samples if c is not None]
    mean_w = sum(ws) / len(ws)
    return {"sensor": rd.kind, "samples": len(ws), "socket_w_mean": mean_w, "socket_w_min": min(ws), "socket_w_max": max(ws),
            "sclk_mhz_mean": (sum(clks) / len(clks)) if clks else None, "cap_w": rd.cap_w,
            "calls": calls, "ms_per_step_while_sampling": ms, "energy_per_step_j": mean_w * ms * 1e-3,
            "note": "sensors read in-process while the step loops after the timed region (two chunks of calls always "
                    "queued); at the cap the energy per step bounds the time (DESIGN.md 3.4)"}


def quoted_traffic():
    try:
        pm = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc_traffic.json"))
        if pm:
            return json.load(open(os.path.join(ROOT, "profiles", pm[-1])))["bytes_per_call"], pm[-1]
    except Exception:
        pass
    return None, None


# ------------------------------------------------------------------------------- workloads
def build_case(g, cfg, rank, world, dev, api, inplace=True):
    """Returns a dict with step(), the shard geometry, a sample for the CPU check and the pieces the
    end-to-end leg needs."""
    import torch
    bits, logn = cfg["bits"], cfg["logn"]
    n = 1 << logn
    poly = g.X_N_minus if cfg["poly"] == "minus" else g.X_N_plus
    batch = cfg["batch"] // world if cfg["scaling"] == "strong" else cfg["batch"]
    if cfg["scaling"] == "strong" and cfg["batch"] % world:
        raise SystemExit("config batch must divide over the ranks")
    case = {"batch": batch, "n": n, "transforms_per_step": batch}
    seed = 0x5EED0000 + {"c2": 2, "c3": 3, "c4": 4, "c5": 5}.get(cfg.get("name", ""), 9) + 16 * rank
    if cfg["kind"] == "merge":
        prm = g.NTTParameters(logn, poly, bits)
        x = splitmix64_mod(seed, batch * n, prm.modulus.value).astype(g.np_dtype(bits))
        d_in = g.to_device(x, dev)
        d_out = torch.empty_like(d_in)
        inv = cfg.get("direction", "fwd") == "inv"
        table = g.to_device(prm.inverse_table_device_order if inv else prm.forward_table_device_order, dev)
        c = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE if inv else g.FORWARD, reduction_poly=poly,
                                mod_inverse=prm.n_inv if inv else 0)
        xf, xf_in = (g.GPU_INTT, g.GPU_INTT_Inplace) if inv else (g.GPU_NTT, g.GPU_NTT_Inplace)
        # the first call goes d_in -> d_out (the output the CPU leg checks); with `inplace` the timed calls are
        # GPU_NTT_Inplace on d_out (its contents stay residues below q), as the reference's benchmark times it
        src = d_out if inplace else d_in
        mk_plan = lambda: g.NTTPlan(table, prm.modulus, logn, poly, g.INVERSE if inv else g.FORWARD,  # noqa: E731
                                    mod_inverse=prm.n_inv if inv else None, batch_hint=batch)
        if api == "plan":
            plan = mk_plan()
            case["first"] = lambda: plan.execute(d_in, d_out, batch)
            case["step"] = lambda: plan.execute(src, d_out, batch)
            case["plan"] = plan
        else:
            case["first"] = lambda: xf(d_in, d_out, table, prm.modulus, c, batch)
            case["step"] = (lambda: xf_in(d_out, table, prm.modulus, c, batch)) if inplace else \
                           (lambda: xf(d_in, d_out, table, prm.modulus, c, batch))
        def other_step():
            if api == "plan":
                return (lambda: xf_in(d_out, table, prm.modulus, c, batch)) if inplace else \
                       (lambda: xf(d_in, d_out, table, prm.modulus, c, batch))
            p2 = mk_plan()
            case["plan"] = p2
            return lambda: p2.execute(src, d_out, batch)
        case.update(x=x, d_in=d_in, d_out=d_out, table=table, modulus=prm.modulus.value, other_step=other_step,
                    run_shard=lambda a, b: xf(a, b, table, prm.modulus, c, batch))
    elif cfg["kind"] == "rns":
        rns = json.load(open(os.path.join(ROOT, "tests", "golden", "rns_c5.json")))
        mc = cfg["mod_count"]
        factors = [(e["q"], e["omega"], e["psi"]) for e in rns["primes"][:mc]]
        prms = [g.NTTParameters(logn, poly, bits, f) for f in factors]
        inv = cfg.get("direction", "fwd") == "inv"
        tab = np.zeros(mc * n, dtype=np.uint64)
        for i, p in enumerate(prms):
            tab[i * n:i * n + p.root_of_unity_size] = p.inverse_table_device_order if inv else p.forward_table_device_order
        x = np.concatenate([splitmix64_mod(seed, n, prms[p % mc].modulus.value, offset=p * n) for p in range(batch)])
        d_in = g.to_device(x, dev)
        d_out = torch.empty_like(d_in)
        table = g.to_device(tab, dev)
        mods = g.modulus_array_to_device([p.modulus for p in prms], bits, dev)
        ninv = g.to_device(np.array([p.n_inv for p in prms], dtype=np.uint64), dev)
        c = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE if inv else g.FORWARD, reduction_poly=poly,
                                    mod_inverse=ninv if inv else None)
        xf, xf_in = (g.GPU_INTT, g.GPU_INTT_Inplace) if inv else (g.GPU_NTT, g.GPU_NTT_Inplace)
        src = d_out if inplace else d_in
        mk_plan = lambda: g.NTTPlan(table, [p.modulus for p in prms], logn, poly, g.INVERSE if inv else g.FORWARD,  # noqa: E731
                                    mod_inverse=[p.n_inv for p in prms] if inv else None, batch_hint=batch)
        if api == "plan":
            plan = mk_plan()
            case["first"] = lambda: plan.execute(d_in, d_out, batch)
            case["step"] = lambda: plan.execute(src, d_out, batch)
            case["plan"] = plan
        else:
            case["first"] = lambda: xf(d_in, d_out, table, mods, c, batch, mc)
            case["step"] = (lambda: xf_in(d_out, table, mods, c, batch, mc)) if inplace else \
                           (lambda: xf(d_in, d_out, table, mods, c, batch, mc))
        def other_step():
            if api == "plan":
                return (lambda: xf_in(d_out, table, mods, c, batch, mc)) if inplace else \
                       (lambda: xf(d_in, d_out, table, mods, c, batch, mc))
            p2 = mk_plan()
            case["plan"] = p2
            return lambda: p2.execute(src, d_out, batch)
        case.update(x=x, d_in=d_in, d_out=d_out, table=table, factors=factors, modulus=factors[0][0],
                    other_step=other_step, ninv=ninv,
                    run_shard=lambda a, b: xf(a, b, table, mods, c, batch, mc))
    else:  # 4-step, forward + inverse pair on the transposed-order form the library entry point defines
        p4 = g.NTTParameters4Step(logn, bits)
        distinct = 4  # 4 distinct polynomials repeated: 8 GiB of splitmix on the host would take minutes
        base = np.concatenate([splitmix64_mod(seed + 7 * i, n, p4.modulus.value) for i in range(distinct)])
        d_in = g.to_device(base, dev).repeat(batch // distinct)
        d_mid = torch.empty_like(d_in)
        tf = [g.to_device(t, dev) for t in p4.tables["fwd"]]
        ti = [g.to_device(t, dev) for t in p4.tables["inv"]]
        cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.FORWARD)
        ci = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE, mod_inverse=p4.n_inv)
        d_back = torch.empty_like(d_in)

        def dropin_step():
            g.GPU_4STEP_NTT(d_in, d_mid, *tf, p4.modulus, cf, batch)
            g.GPU_4STEP_NTT(d_mid, d_back, *ti, p4.modulus, ci, batch)

        def plan_step():
            if "plan" not in case:
                case["plan"] = (g.FourStepPlan(*tf, p4.modulus, cf, batch_hint=batch),
                                g.FourStepPlan(*ti, p4.modulus, ci, batch_hint=batch))
            case["plan"][0].execute(d_in, d_mid, batch)
            case["plan"][1].execute(d_mid, d_back, batch)
        step, other_step = (plan_step, dropin_step) if api == "plan" else (dropin_step, plan_step)
        case.update(tables_fwd=tf, tables_inv=ti, cfg_fwd=cf, cfg_inv=ci)
        case.update(step=step, x=base, other_step=lambda: other_step, d_in=d_in, d_out=d_mid, d_back=d_back, distinct=distinct,
                    table=tf[2], modulus=p4.modulus.value, transforms_per_step=2 * batch, p4=p4, run_shard=None)
    return case


def gpu_sample_for_check(g, cfg, case, polys):
    """GPU output of the first `polys` polynomials in the order the CPU path produces.  4-step (c3): polynomial 0 of
    `d_mid`, THE buffer the timed GPU_4STEP_NTT calls write (reference layout n1 x n2), transposed on the device into
    NTT_4STEP_CPU::ntt's order -- what the reference example's closing GPU_Transpose does (test_4step_ntt.cu:170-178)."""
    import torch
    n = case["n"]
    if cfg["kind"] != "4step":
        return g.to_host(case["d_out"])[:polys * n].copy()
    p4 = case["p4"]
    torch.cuda.synchronize()
    return g.to_host(case["d_out"][:n].view(p4.n1, p4.n2).t().contiguous().view(-1))


def fourstep_buffers_check(case):
    """c3, after the timed steps, on the device, over the WHOLE batch: every copy of the `distinct` polynomials in the
    forward output d_mid equals the first one (whose polynomial 0 the CPU leg checks), and the inverse output d_back
    equals the input d_in (forward -> inverse is the identity in the reference layout up to the transposition the two
    calls define: out_inv = in_fwd read as n2 x n1 -> n1 x n2).  Returns the dict for the JSON line."""
    import torch
    p4, n, distinct = case["p4"], case["n"], case["distinct"]
    d_in, d_mid, d_back = case["d_in"], case["d_out"], case["d_back"]
    torch.cuda.synchronize()
    groups = d_mid.view(-1, distinct * n)
    copies_equal = bool((groups == groups[0:1]).all())
    # forward input of polynomial p is x^T (n2 x n1); the inverse call returns x as (n1 x n2)^T-of-the-natural-order,
    # i.e. d_back[p] viewed n1 x n2 and transposed == x == d_in[p] viewed n2 x n1 and transposed back
    ok = True
    for p in range(distinct):
        # natural-order polynomial, flat: the forward call read it transposed (d_in = n2 x n1), the inverse call
        # returns it so that one GPU_Transpose(n1, n2) restores it (test_4step_intt.cu:170-179)
        x_from_in = d_in[p * n:(p + 1) * n].view(p4.n2, p4.n1).t().contiguous().view(-1)
        x_from_back = d_back[p * n:(p + 1) * n].view(p4.n1, p4.n2).t().contiguous().view(-1)
        ok = ok and bool((x_from_in == x_from_back).all())
    back_groups = d_back.view(-1, distinct * n)
    ok = ok and bool((back_groups == back_groups[0:1]).all())
    return {"forward_copies_identical": copies_equal, "inverse_returns_input": ok,
            "polynomials_checked_on_device": int(d_mid.numel() // n)}


def c4_shard_overheads(g, cfg, dev, shard_batch=1024):
    """C4 is the strongly scaled config: on 8 GPUs each rank holds 8192 / 8 = 1024 polynomials of 2^14 x u32 -- about
    40 us of kernel time, the size of the per-call host overhead.  What one such shard costs through the three call
    forms, on ONE GPU (us per call, HIP events over back-to-back calls): the drop-in call (one preparation launch + the
    transform), NTTPlan (the transform only) and NTTPlan replayed from a hipGraph (no per-call host launch path).  The
    8-GPU strong-scaling efficiency of C4 is bounded by these, not by the transform."""
    import torch
    bits, logn = cfg["bits"], cfg["logn"]
    n = 1 << logn
    poly = g.X_N_minus if cfg["poly"] == "minus" else g.X_N_plus
    prm = g.NTTParameters(logn, poly, bits)
    x = splitmix64_mod(0xC4C4, shard_batch * n, prm.modulus.value).astype(g.np_dtype(bits))
    d = g.to_device(x, dev)
    table = g.to_device(prm.forward_table_device_order, dev)
    c = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=poly)
    plan = g.NTTPlan(table, prm.modulus, logn, poly, g.FORWARD, batch_hint=shard_batch)

    def timed(fn, iters=400):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return {"device_us": e0.elapsed_time(e1) * 1e3 / iters, "host_wall_us": (time.perf_counter() - t0) * 1e6 / iters}

    out = {"batch": shard_batch, "log2N": logn, "dtype": "u%d" % bits,
           "dropin": timed(lambda: g.GPU_NTT_Inplace(d, table, prm.modulus, c, shard_batch)),
           "plan": timed(lambda: plan.execute(d, d, shard_batch))}
    try:
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            plan.execute(d, d, shard_batch, stream=torch.cuda.current_stream())
        out["plan_hipgraph"] = timed(graph.replay)
    except Exception as e:  # informational
        out["plan_hipgraph"] = {"error": repr(e)}
    plan.close()
    return out


# ------------------------------------------------------------------ the other BASELINE configs
# What the default run (C2 forward, the headline) times BESIDE its line, so that one driver-owned run witnesses every
# BASELINE.json config: (key, config, direction, steps, warmup).  Same HIP-event bracket on the launch stream, same
# bit-exact check against the reference CPU build, PMC traffic from two child passes over all of them.
OTHER_CONFIGS = (("c2_inv", "c2", "inv", 20, 5), ("c5", "c5", "fwd", 20, 5), ("c5_inv", "c5", "inv", 20, 5),
                 ("c4", "c4", "fwd", 40, 10), ("c4_inv", "c4", "inv", 40, 10), ("c3", "c3", "fwd", 4, 1))
OTHER_CHILD_CALLS = 3


def _other_cfg(name, direction):
    cfg = dict(CONFIGS[name], name=name, direction=direction)
    return cfg, (cfg["kind"] != "4step")


def other_configs_child(g, dev):
    """PMC child pass (`--child --other-configs`): every OTHER_CONFIGS entry, OTHER_CHILD_CALLS calls each, between two
    mark kernels -- the dispatches before the second mark (case set-up, the first out-of-place call) are not counted."""
    import torch
    mark = torch.empty(1, dtype=torch.float64, device=dev)
    for k, (_, name, direction, _, _) in enumerate(OTHER_CONFIGS):
        cfg, inplace = _other_cfg(name, direction)
        case = build_case(g, cfg, 0, 1, dev, "dropin", inplace)
        mark.fill_(float(2 * k))
        case.get("first", case["step"])()
        torch.cuda.synchronize()
        mark.fill_(float(2 * k + 1))
        for _ in range(OTHER_CHILD_CALLS):
            case["step"]()
        torch.cuda.synchronize()
        del case
        torch.cuda.empty_cache()


def other_configs_traffic():
    """HBM bytes per call of every OTHER_CONFIGS entry: two rocprofv3 --pmc child passes (FETCH_SIZE x2 on gfx950,
    WRITE_SIZE -- separate passes, MI355X_MICROARCH.md) of `bench.py --child --other-configs`."""
    import glob
    import shutil
    import sqlite3
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="gpuntt_other_pmc_", dir="/tmp")
    per = [0.0] * len(OTHER_CONFIGS)
    try:
        for counter, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--child", "--other-configs"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                               timeout=600)
            dbs = glob.glob(out + "/**/*.db", recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed: %s" % (counter, (r.stderr or r.stdout)[-300:])
            c = sqlite3.connect(dbs[0])
            names = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
            t = lambda p: [n for n in names if n.startswith(p)][0]  # noqa: E731
            q = (f"select s.kernel_name, d.start, sum(p.value) from {t('rocpd_pmc_event')} p "
                 f"join {t('rocpd_info_pmc')} i on p.pmc_id=i.id "
                 f"join {t('rocpd_kernel_dispatch')} d on p.event_id=d.event_id "
                 f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id=s.id "
                 f"where i.name='{counter}' group by d.id order by d.start")
            idx = -1
            for name, _start, kb in c.execute(q):
                if SWEEP_MARK in name:
                    idx += 1
                elif "gpuntt" in name and idx >= 0 and idx % 2 == 1 and idx // 2 < len(per):
                    per[idx // 2] += kb * 1024.0 * corr / OTHER_CHILD_CALLS
            if idx != 2 * len(OTHER_CONFIGS) - 1:
                return None, "marks in the %s pass: %d, expected %d" % (counter, idx + 1, 2 * len(OTHER_CONFIGS))
        return per, {"method": "rocprofv3 --pmc, two child passes over all the other configs, this run",
                     "fetch_correction": 2.0, "calls_profiled_per_config": OTHER_CHILD_CALLS}
    except Exception as e:
        return None, "traffic measurement failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _time_calls(step, steps, warmup):
    """K calls between two HIP events on the launch stream, after a clock settle and W warm-up calls -> ms per call"""
    import torch
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def run_other_configs(g, dev, args):
    """Every BASELINE config the headline does not time (and the inverse directions), on this GPU, in this run: ms per
    call (HIP events), fraction of the 8 TB/s peak on algorithmic bytes, PMC traffic over algorithmic bytes, and the
    GPU output checked bit for bit against the reference CPU build (the same checker as `cpu_baseline`)."""
    import torch
    t_start = time.perf_counter()
    out = {}
    for key, name, direction, steps, warmup in OTHER_CONFIGS:
        try:
            cfg, inplace = _other_cfg(name, direction)
            case = build_case(g, cfg, 0, 1, dev, "dropin", inplace)
            case.get("first", case["step"])()
            torch.cuda.synchronize()
            n, bits = case["n"], cfg["bits"]
            polys = min(16, case["batch"])
            if cfg["kind"] == "rns":
                polys = max(cfg["mod_count"], polys - polys % cfg["mod_count"])
            y_first = gpu_sample_for_check(g, cfg, case, polys)
            ms = _time_calls(case["step"], steps, warmup)
            per_step = case["transforms_per_step"]
            alg = 2 * n * (bits // 8) * per_step
            ent = {"workload": cfg["workload"], "ms_per_call": ms, "steps": steps, "warmup": warmup,
                   "transforms_per_call": per_step, "value_ntt_per_s": per_step / (ms * 1e-3),
                   "algorithmic_bytes_per_call": alg, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "in_place": inplace, "api": "dropin"}
            if cfg["kind"] != "4step":
                case["x_sample"] = case["x"][:polys * n]
            else:
                p4 = case["p4"]
                case["x_sample"] = np.ascontiguousarray(case["x"][:n].reshape(p4.n2, p4.n1).T).reshape(-1)
                # the two calls of the pair on their own (same buffers)
                fwd_only = lambda: g.GPU_4STEP_NTT(case["d_in"], case["d_out"], *case["tables_fwd"], p4.modulus, case["cfg_fwd"], case["batch"])  # noqa: E731
                inv_only = lambda: g.GPU_4STEP_NTT(case["d_out"], case["d_back"], *case["tables_inv"], p4.modulus, case["cfg_inv"], case["batch"])  # noqa: E731
                ent["forward_call_ms"] = _time_calls(fwd_only, steps, 1)
                ent["inverse_call_ms"] = _time_calls(inv_only, steps, 1)
            if not args.no_cpu_baseline:
                chk = cpu_baseline(cfg, case, y_first, check_only=True)
                ent["bit_exact"] = bool(chk["gpu_output_bit_exact"])
                ent["checked"] = "%d polynomial(s) against the %s CPU transform" % (chk["polynomials"], chk["kind"])
                if cfg["kind"] == "4step":
                    if not np.array_equal(y_first, gpu_sample_for_check(g, cfg, case, 1)):
                        raise RuntimeError("the timed 4-step calls left a different result in their output buffer")
                    ent["batch_check"] = fourstep_buffers_check(case)
                    ent["bit_exact"] = ent["bit_exact"] and ent["batch_check"]["forward_copies_identical"] and \
                        ent["batch_check"]["inverse_returns_input"]
            out[key] = ent
            del case
            torch.cuda.empty_cache()
        except SystemExit as e:  # a failed bit-exact check must be visible, not fatal to the headline
            out[key] = {"error": str(e), "bit_exact": False}
        except Exception as e:
            out[key] = {"error": repr(e)}
    try:
        sh = c4_shard_overheads(g, dict(CONFIGS["c4"]), dev)
        out["c4_shard_of_8"] = {"workload": "Merge-NTT Data32 log2N=14, one GPU's 1024-polynomial shard of the 8-way sharded batch",
                                "dropin_us": sh["dropin"]["device_us"], "plan_us": sh["plan"]["device_us"],
                                "plan_hipgraph_us": sh.get("plan_hipgraph", {}).get("device_us"),
                                "ideal_us_from_whole_batch": (out.get("c4", {}).get("ms_per_call") or 0.0) * 1e3 / 8.0}
    except Exception as e:
        out["c4_shard_of_8"] = {"error": repr(e)}
    if not args.no_traffic:
        per, info = other_configs_traffic()
        if per is not None:
            for (key, *_), b in zip(OTHER_CONFIGS, per):
                if key in out and "algorithmic_bytes_per_call" in out[key]:
                    out[key]["traffic"] = b
                    out[key]["traffic_over_algorithmic"] = b / out[key]["algorithmic_bytes_per_call"]
        out["traffic_info"] = info
    out["seconds"] = time.perf_counter() - t_start
    return out


# ------------------------------------------------------------------------------ self-launch
def spawn_ranks(n):
    """`python bench.py --gpus N` without torchrun: re-execute this command line under torch.distributed.run with N
    ranks on this node (127.0.0.1 rendezvous on a free port) and hand its exit code back."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dry_run(cfg, args, dist_mod):
    """No GPU in this process (CPU container): everything around the kernels still runs -- rendezvous (gloo), the
    shard geometry of every rank, the barrier + MAX-reduced timed region, the JSON line -- with a no-op step.
    `value` is null and the line says "dry_run": nothing here is a measurement."""
    dist, rank, world = dist_mod.init_process_group("gloo", None)
    pkg = sys.modules["gpu_ntt_amd"]
    total = cfg["batch"] * (1 if cfg["scaling"] == "strong" else world)
    lo, hi = pkg.shard_range(total, rank, world, cfg.get("mod_count", 1))
    wall = dist_mod.timed_region(lambda: None, args.steps, dist, None)
    shards = [(lo, hi)]
    if dist is not None:
        shards = [None] * world
        dist.all_gather_object(shards, (lo, hi))
    if rank == 0:
        print(json.dumps({"metric": cfg["metric"], "value": None, "unit": "NTT/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": wall * 1e3 / max(args.steps, 1),
                          "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
                          "dtype": "u%d" % cfg["bits"], "data": "synthetic", "dry_run": True,
                          "config": {"workload": cfg["workload"], "log2N": cfg["logn"],
                                     "parallelism": "batch-shard x%d" % world, "shards": shards},
                          "note": "no GPU visible: launch, rendezvous (gloo), sharding and timing plumbing only"}),
              flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------ sweep
SWEEP_MARK = "FillFunctorIdE"  # the float64 fill kernel that separates two sweep points in a PMC child pass


def sweep_points(args):
    kinds = [k for k in args.sweep_kinds.split(",") if k]
    pts = []
    for kind in kinds:
        for bits in [int(b) for b in args.sweep_bits.split(",")]:
            for logn in range(args.sweep_min, args.sweep_max + 1):
                if kind == "4step" and bits == 32 and logn > 24:
                    continue
                pts.append((kind, bits, logn, max(1, (1 << args.sweep_log_coeffs) >> logn)))
    return pts


def sweep_case(g, kind, bits, logn, batch, dev, rank, inverse=False):
    """(first, step, x0, y0-getter) of one sweep point: forward (or inverse) transform of `batch` polynomials of 2^logn,
    in place for Merge (reference benchmark/bench_merge_ntt.cu:62), in -> out for 4-step (the entry point is out of place)."""
    import torch
    n = 1 << logn
    seed = 0x5EED1000 + logn + 64 * rank
    if kind == "merge":
        prm = g.NTTParameters(logn, g.X_N_minus, bits)
        x = splitmix64_mod(seed, batch * n, prm.modulus.value).astype(g.np_dtype(bits))
        d_in = g.to_device(x, dev)
        d_out = torch.empty_like(d_in)
        if inverse:
            table = g.to_device(prm.inverse_table_device_order, dev)
            c = g.ntt_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=g.X_N_minus, mod_inverse=prm.n_inv)
            first = lambda: g.GPU_INTT(d_in, d_out, table, prm.modulus, c, batch)  # noqa: E731
            step = lambda: g.GPU_INTT_Inplace(d_out, table, prm.modulus, c, batch)  # noqa: E731
        else:
            table = g.to_device(prm.forward_table_device_order, dev)
            c = g.ntt_configuration(n_power=logn, ntt_type=g.FORWARD, reduction_poly=g.X_N_minus)
            first = lambda: g.GPU_NTT(d_in, d_out, table, prm.modulus, c, batch)  # noqa: E731
            step = lambda: g.GPU_NTT_Inplace(d_out, table, prm.modulus, c, batch)  # noqa: E731
        return first, step, x[:n], (lambda: g.to_host(d_out)[:n].copy()), [d_in, d_out, table]
    p4 = g.NTTParameters4Step(logn, bits)
    distinct = min(batch, 4)
    base = np.concatenate([splitmix64_mod(seed + 7 * i, n, p4.modulus.value) for i in range(distinct)]).astype(g.np_dtype(bits))
    d_in = g.to_device(base, dev).repeat(batch // distinct)
    d_out = torch.empty_like(d_in)
    tabs = [g.to_device(t, dev) for t in p4.tables["inv" if inverse else "fwd"]]
    cf = g.ntt4step_configuration(n_power=logn, ntt_type=g.INVERSE if inverse else g.FORWARD,
                                  mod_inverse=p4.n_inv if inverse else 0)
    step = lambda: g.GPU_4STEP_NTT(d_in, d_out, *tabs, p4.modulus, cf, batch)  # noqa: E731

    def out0():
        torch.cuda.synchronize()
        return g.to_host(d_out[:n].view(p4.n1, p4.n2).t().contiguous().view(-1))
    if inverse:
        # the inverse call reads what NTT_4STEP_CPU::intt_first_transpose makes of the spectrum y and one closing
        # GPU_Transpose gives NTT_4STEP_CPU::intt(y) (test_4step_intt.cu:81-179): the CPU leg gets y itself, found by
        # undoing the first transpose of polynomial 0 of the input buffer: flat[i * n2 + j] = y[i + j * n1]
        y_nat = np.ascontiguousarray(base[:n].reshape(p4.n1, p4.n2).T).reshape(-1)
        return step, step, y_nat, out0, [d_in, d_out] + tabs
    # the call reads its input as the n2 x n1 transpose of the natural-order polynomial and writes the spectrum n1 x n2:
    # NTT_4STEP_CPU::ntt(x_nat) is polynomial 0 of THE OUTPUT BUFFER transposed (what the reference example's closing
    # GPU_Transpose does, test_4step_ntt.cu:170-178)
    x_nat = np.ascontiguousarray(base[:n].reshape(p4.n2, p4.n1).T).reshape(-1)
    return step, step, x_nat, out0, [d_in, d_out] + tabs


def sweep_cpu(kind, bits, logn, x0, y0, inverse=False):
    """reference CPU transform of ONE polynomial of this ring on one host core (>= 1 transform, ~0.5 s); also the
    bit-exact check of the GPU's polynomial 0"""
    from oracle import oracle as O
    B = O.Ref(bits) if O.have_ref() else O.Port(bits)
    if kind == "merge":
        prm = B.merge_params(logn, O.X_N_minus)
        run = lambda: B.merge_ntt(x0, prm, inverse)  # noqa: E731
    else:
        prm = B.fourstep_params(logn)
        run = (lambda: B.fourstep_run(x0, prm, 1 if inverse else 0)) if O.have_ref() else \
              (lambda: B.fourstep_ntt(x0, prm, inverse))
    t0 = time.perf_counter()
    y = run()
    k = 1
    while time.perf_counter() - t0 < 0.5:
        run()
        k += 1
    dt = (time.perf_counter() - t0) / k
    return {"value": 1.0 / dt, "unit": "NTT/s", "cores": 1, "kind": "reference" if O.have_ref() else "port",
            "gpu_output_bit_exact": bool(np.array_equal(y, y0)), "sample": "%d transform(s) of one polynomial" % k}


def sweep_traffic(args, points):
    """HBM bytes per call of every sweep point from two rocprofv3 --pmc child passes of `bench.py --sweep --child`
    (FETCH_SIZE x2 on gfx950, WRITE_SIZE): the child separates the points with a float64 fill kernel, the dispatches
    are split at those marks in time order."""
    import glob
    import shutil
    import sqlite3
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    calls = 3
    tmp = tempfile.mkdtemp(prefix="gpuntt_sweep_pmc_", dir="/tmp")
    per_point = [0.0] * len(points)
    try:
        for counter, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "-d", out, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--sweep", "--child", "--steps", str(calls), "--sweep-kinds", args.sweep_kinds, "--sweep-bits",
                   args.sweep_bits, "--sweep-min", str(args.sweep_min), "--sweep-max", str(args.sweep_max),
                   "--sweep-log-coeffs", str(args.sweep_log_coeffs), "--direction", args.direction]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                               timeout=1500)
            dbs = glob.glob(out + "/**/*.db", recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed: %s" % (counter, (r.stderr or r.stdout)[-300:])
            c = sqlite3.connect(dbs[0])
            names = [x[0] for x in c.execute("select name from sqlite_master where type='table'")]
            t = lambda p: [n for n in names if n.startswith(p)][0]  # noqa: E731
            q = (f"select s.kernel_name, d.start, sum(p.value) from {t('rocpd_pmc_event')} p "
                 f"join {t('rocpd_info_pmc')} i on p.pmc_id=i.id "
                 f"join {t('rocpd_kernel_dispatch')} d on p.event_id=d.event_id "
                 f"join {t('rocpd_info_kernel_symbol')} s on d.kernel_id=s.id "
                 f"where i.name='{counter}' group by d.id order by d.start")
            idx = -1
            for name, _start, kb in c.execute(q):
                if SWEEP_MARK in name:
                    idx += 1
                elif "gpuntt" in name and 0 <= idx < len(points):
                    per_point[idx] += kb * 1024.0 * corr / calls
            if idx != len(points) - 1:
                return None, "sweep marks in the %s pass: %d, expected %d" % (counter, idx + 1, len(points))
        return per_point, {"method": "rocprofv3 --pmc, two child passes of the whole sweep, this run",
                           "fetch_correction": 2.0, "calls_profiled_per_point": calls}
    except Exception as e:
        return None, "sweep traffic measurement failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_sweep(g, args, dist_mod, dist, rank, world, dev):
    """north_star's table: log2N x {Merge, 4-Step}, per ring size NTT/s, achieved algorithmic GB/s and fraction of the
    8 TB/s peak, PMC HBM bytes per call and the reference CPU transform beside it -- one JSON line per point, at
    whatever number of ranks the script was started with (weak scaling: every rank transforms its own batch)."""
    import torch
    points = sweep_points(args)
    if args.child:
        mark = torch.empty(1, dtype=torch.float64, device=dev)  # (torch.zeros would itself launch the mark kernel)
        for k, (kind, bits, logn, batch) in enumerate(points):
            first, step, _, _, keep = sweep_case(g, kind, bits, logn, batch, dev, rank, args.direction == "inv")
            torch.cuda.synchronize()
            mark.fill_(float(k))
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            del keep
        return
    traffic, tinfo = (None, "disabled / multi-GPU run")
    if world == 1 and not args.no_traffic:
        traffic, tinfo = sweep_traffic(args, points)
    for k, (kind, bits, logn, batch) in enumerate(points):
        first, step, x0, y0_get, keep = sweep_case(g, kind, bits, logn, batch, dev, rank, args.direction == "inv")
        first()
        torch.cuda.synchronize()
        y0 = y0_get() if rank == 0 else None
        if kind == "merge":
            step()  # the in-place form once before timing
        t0 = time.perf_counter()
        w = 0
        while w < 3 or time.perf_counter() - t0 < 0.08:
            step()
            w += 1
            if w % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        per = max((time.perf_counter() - t0) / w, 1e-6)
        steps = max(5, min(2000, int(0.15 / per)))
        if dist is not None:  # every rank times the SAME number of steps (rank 0's estimate)
            st = torch.tensor([steps], device=REDUCE_DEV or dev, dtype=torch.int64)
            dist.broadcast(st, src=0)
            steps = int(st[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed():
            e0.record()
            for _ in range(steps):
                step()
            e1.record()
        wall = dist_mod.timed_region(timed, 1, dist, dev)
        dev_ms = e0.elapsed_time(e1)
        if dist is not None:
            tt = torch.tensor([dev_ms], device=REDUCE_DEV or dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dev_ms = float(tt[0])
        if rank == 0:
            n = 1 << logn
            alg = 2 * n * (bits // 8) * batch
            call_ms = dev_ms / steps
            achieved = alg / (call_ms * 1e-3) / 1e9
            line = {"sweep": True, "algo": kind, "dtype": "u%d" % bits, "log2N": logn, "batch_per_gpu": batch,
                    "n_gpus": world, "direction": args.direction,
                    "metric": "%s-NTTs/sec + achieved HBM GB/s" % ("inverse" if args.direction == "inv" else "forward"), "unit": "NTT/s",
                    "value": world * batch * steps / wall, "steps": steps, "ms_per_step": wall * 1e3 / steps,
                    "in_place": kind == "merge", "scaling": "weak", "data": "synthetic",
                    "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_call": alg,
                                 "call_ms_hip_events": call_ms,
                                 "traffic": traffic[k] if traffic else None,
                                 "traffic_over_algorithmic": (traffic[k] / alg) if traffic else None}}
            if k == 0:
                line["roofline"]["traffic_info"] = tinfo
            if not args.no_cpu_baseline and world == 1:
                line["cpu_baseline"] = sweep_cpu(kind, bits, logn, x0, y0, args.direction == "inv")
                if not line["cpu_baseline"]["gpu_output_bit_exact"]:
                    raise SystemExit("bench --sweep: GPU result differs from the CPU reference path at %s 2^%d" % (kind, logn))
            print(json.dumps(line), flush=True)
        del keep
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--api", choices=("dropin", "plan"), default="dropin")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-power", action="store_true")
    ap.add_argument("--out-of-place", action="store_true",
                    help="time GPU_NTT(in, out) instead of the in-place call the reference's own benchmark times")
    ap.add_argument("--direction", choices=("fwd", "inv"), default="fwd",
                    help="Merge configs (c2, c4, c5) and --sweep: time GPU_INTT instead of GPU_NTT "
                         "(reference benchmark/bench_merge_ntt.cu:137-141 times both); c3 is a forward + inverse pair")
    ap.add_argument("--cpu-polys", type=int, default=64)
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # PMC child pass: calls only
    ap.add_argument("--other-configs", action="store_true", help=argparse.SUPPRESS)  # with --child: the other configs' pass
    ap.add_argument("--no-shard-overheads", action="store_true",
                    help="c4: do not time the 1024-polynomial shard through the three call forms (kernel-trace runs: keeps the "
                         "per-kernel average about whole-batch launches)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (c2 forward, 1 GPU): do not time the other BASELINE configs beside the headline")
    ap.add_argument("--sweep", action="store_true",
                    help="log2N sweep (Merge and 4-Step, forward), one JSON line per ring size with PMC bytes and the CPU column")
    ap.add_argument("--sweep-kinds", default="merge,4step")
    ap.add_argument("--sweep-bits", default="64")
    ap.add_argument("--sweep-min", type=int, default=12)
    ap.add_argument("--sweep-max", type=int, default=24)
    ap.add_argument("--sweep-log-coeffs", type=int, default=26, help="log2 of the coefficients per GPU and call")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config], name=args.config, direction=args.direction)
    if args.direction == "inv" and cfg["kind"] != "4step":
        cfg["metric"] = cfg["metric"].replace("forward-NTTs/sec", "inverse-NTTs/sec")
        cfg["workload"] = cfg["workload"].replace(" forward", " inverse") if " forward" in cfg["workload"] \
            else cfg["workload"] + " [inverse]"
    if args.config == "c3" and args.steps == 200:
        args.steps, args.warmup = 10, 2  # a step is ~25 ms and 16 GiB of traffic

    # `python bench.py --gpus N` on its own: start the N ranks here (the driver's torchrun form sets WORLD_SIZE)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import importlib
    import torch
    g = _load_pkg()
    g.load_library()  # raises if the HIP extension is missing: there is no fallback
    dist_mod = importlib.import_module("gpu_ntt_amd.dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        return dry_run(cfg, args, dist_mod)
    # GPUNTT_BENCH_BACKEND=gloo (diagnostic): control-plane collectives over gloo and ranks mapped onto the devices that
    # exist (LOCAL_RANK % device count) -- lets a one-GPU box run the N-rank code path end to end; RCCL is the default
    backend = os.environ.get("GPUNTT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist, rank, world = dist_mod.init_process_group(backend, dev)
    global REDUCE_DEV
    REDUCE_DEV = dev if backend == "nccl" else "cpu"
    if args.child and args.other_configs:
        other_configs_child(g, dev)
        return
    if args.sweep:
        run_sweep(g, args, dist_mod, dist, rank, world, dev)
        if dist is not None:
            dist.destroy_process_group()
        return

    inplace = not args.out_of_place and cfg["kind"] != "4step"
    case = build_case(g, cfg, rank, world, dev, args.api, inplace)
    step = case["step"]
    case.get("first", step)()  # d_in -> d_out once: the output the CPU leg checks
    torch.cuda.synchronize()
    bits, logn, n = cfg["bits"], cfg["logn"], case["n"]
    cpu_polys = min(max(args.cpu_polys, os.cpu_count() or 1), case["batch"])  # >= one polynomial per host thread
    if cfg["kind"] == "rns":
        cpu_polys = max(cfg["mod_count"], cpu_polys - cpu_polys % cfg["mod_count"])
    # taken NOW: the in-place timed calls below transform d_out again and again
    y_first = gpu_sample_for_check(g, cfg, case, cpu_polys) if (rank == 0 and not args.child) else None
    if args.child:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return

    # clock settle (untimed, before the W warm-up steps): the part needs ~60 ms of load to leave its
    # idle clocks; without this a short --steps/--warmup run reads 10-15 % slow
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.15:
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)

    def timed_steps():
        # HIP events on the stream the kernels are launched on bracket exactly the K steps
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()

    # barrier + synchronize on both sides, MAX over ranks (gpu-ntt_amd/dist.py)
    wall = dist_mod.timed_region(timed_steps, 1, dist, dev)
    dev_ms = e0.elapsed_time(e1)
    if dist is not None:
        t = torch.tensor([dev_ms], device=REDUCE_DEV or dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dev_ms = float(t[0])

    # the other API form on the same buffers, timed the same way for the record (never `value`)
    other = None
    if "other_step" in case:
        try:
            ostep = case["other_step"]()
            ostep()
            torch.cuda.synchronize()
            for _ in range(max(args.warmup, 10)):
                ostep()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(args.steps):
                ostep()
            a1.record()
            torch.cuda.synchronize()
            other = a0.elapsed_time(a1) / args.steps
        except Exception as e:  # informational only
            other = "failed: %r" % (e,)

    power = None
    if world == 1 and not args.no_power:
        try:
            power = measure_power(step, wall * 1e3 / args.steps)
        except Exception as e:  # informational only
            power = {"error": repr(e)}

    def finish(e2e):
        """rank 0: assemble and print THE line (everything the timed region produced is already known here)"""
        if rank == 0:
            ms_per_step = wall * 1e3 / args.steps
            call_ms = dev_ms / args.steps
            per_step = case["transforms_per_step"]
            alg_bytes = 2 * n * (bits // 8) * per_step  # every coefficient read once + written once per transform
            achieved = alg_bytes / (call_ms * 1e-3) / 1e9
            traffic, traffic_info = None, None
            if world == 1 and not args.no_traffic:
                traffic, traffic_info = measure_traffic(args.config, args.api, args.out_of_place, args.direction)
            quoted = False
            if traffic is None:
                reason = traffic_info
                traffic, src = quoted_traffic() if args.config == "c2" else (None, None)
                quoted = traffic is not None
                traffic_info = {"quoted": quoted, "source": src, "why_not_measured": reason or "disabled / multi-GPU run"}
            line = {
                "metric": cfg["metric"],
                "value": world * per_step * args.steps / wall,
                "unit": "NTT/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": ms_per_step,
                "higher_is_better": True,
                "scaling": cfg["scaling"],
                "vs_baseline": None,
                "dtype": "u%d" % bits,
                "data": "synthetic",
                "config": {"workload": cfg["workload"], "log2N": logn, "batch_per_gpu": case["batch"],
                           "reduction_poly": "X_N_" + cfg["poly"], "modulus": int(case["modulus"]),
                           "in_place": inplace, "api": args.api,
                           "parallelism": "batch-shard x%d" % world},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_info": traffic_info,
                             "algorithmic_bytes_per_call": alg_bytes,
                             "call_ms_hip_events": call_ms,
                             "note": "one call = every launch of one library call; achieved = algorithmic bytes "
                                     "per call / HIP-event time per call; per-kernel averages in profiles/"},
            }
            if isinstance(other, float):
                line["other_api_ms_per_call"] = {"api": "plan" if args.api == "dropin" else "dropin", "ms": other}
            if power is not None and "ms_per_step_while_sampling" in power:
                # the samples describe the timed regime only if the sampled loop ran at the timed rate
                ratio = power["ms_per_step_while_sampling"] / ms_per_step
                power["sampling_vs_timed_ms_ratio"] = ratio
                power["same_regime_as_timed_steps"] = bool(0.9 <= ratio <= 1.1)
            if power is not None:
                if bits == 64 and isinstance(power.get("cap_w"), float) and traffic and power.get("same_regime_as_timed_steps"):
                    # what the measured bytes and the butterflies of one step cost by the per-byte / per-butterfly energies
                    # measured on this part (profiles/r02_power.txt: device copy 0.12 nJ per byte through L2 / fabric / HBM;
                    # profiles/ubench_bfly_r02.txt: register-only 64-bit lazy butterflies 29 nJ per wave of 64; idle 261 W)
                    wave_bfly = per_step * (n // 2) * logn / 64.0
                    dyn_j = float(traffic) * 0.12e-9 + wave_bfly * 29e-9
                    power["energy_model"] = {
                        "dynamic_j_per_step": dyn_j,
                        "ms_per_step_at_cap": dyn_j / (power["cap_w"] - 261.0) * 1e3,
                        "inputs": {"hbm_bytes": float(traffic), "nj_per_byte": 0.12, "wave_butterflies": wave_bfly,
                                   "nj_per_wave_butterfly": 29.0, "idle_w": 261.0},
                        "note": "time the dynamic energy of one step takes at the socket cap: the bound these "
                                "transforms run into (sustained loop: ms_per_step_while_sampling)"}
                line["power"] = power
            if e2e is not None:
                line["end_to_end"] = e2e
            if args.config == "c4" and not args.no_shard_overheads:
                try:
                    line["shard_of_8_gpus_call_overheads"] = c4_shard_overheads(g, cfg, dev)
                except Exception as e:  # informational only
                    line["shard_of_8_gpus_call_overheads"] = {"error": repr(e)}
            if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
                if cfg["kind"] != "4step":
                    case["x_sample"] = case["x"][:cpu_polys * n]
                else:
                    # the timed forward call reads its input as the n2 x n1 transpose of the natural-order polynomial:
                    # that polynomial is what NTT_4STEP_CPU::ntt gets
                    p4 = case["p4"]
                    case["x_sample"] = np.ascontiguousarray(case["x"][:n].reshape(p4.n2, p4.n1).T).reshape(-1)
                line["cpu_baseline"] = cpu_baseline(cfg, case, y_first)
                if cfg["kind"] == "4step":
                    # the same polynomial of the same buffer AFTER the timed steps, and the whole batch on the device
                    if not np.array_equal(y_first, gpu_sample_for_check(g, cfg, case, 1)):
                        raise SystemExit("bench: the timed 4-step calls left a different result in their output buffer")
                    chk = fourstep_buffers_check(case)
                    if not (chk["forward_copies_identical"] and chk["inverse_returns_input"]):
                        raise SystemExit("bench: 4-step batch check failed: %r" % (chk,))
                    line["cpu_baseline"]["checked_buffer"] = "polynomial 0 of the buffer the timed GPU_4STEP_NTT calls wrote"
                    line["batch_check"] = chk
            if (args.config == "c2" and args.direction == "fwd" and args.api == "dropin" and world == 1
                    and not args.out_of_place and not args.no_other_configs):
                # (after everything the headline needs: nothing here can change `value`)
                try:
                    line["other_configs"] = run_other_configs(g, dev, args)
                except Exception as e:
                    line["other_configs"] = {"error": repr(e)}
            print(json.dumps(line), flush=True)

    # The RCCL legs around the transform (table broadcast, scatter, gather) are informational and have never run on
    # more than one physical GPU: a collective that hangs must not cost the run its line.  A watchdog prints the line
    # without them and ends the rank.
    e2e = None
    if dist is not None and not args.no_e2e and case.get("run_shard") is not None:
        import threading
        done = threading.Event()

        def on_timeout():
            if done.is_set():
                return
            try:
                finish({"error": "end-to-end leg did not finish within %d s; skipped" % E2E_TIMEOUT_S})
            finally:
                sys.stdout.flush()
                os._exit(0)

        wd = threading.Timer(E2E_TIMEOUT_S, on_timeout)
        wd.daemon = True
        wd.start()
        try:
            e2e = dist_mod.end_to_end_leg(dist, rank, world, dev, case["table"], case["d_in"], case["d_out"],
                                          case["run_shard"], case["batch"])
        except Exception as e:
            e2e = {"error": repr(e)}
        done.set()
        wd.cancel()
    finish(e2e)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
