#!/usr/bin/env python3
"""Same-session A/B of two builds of the library on bench.py lines (alternating, so that box drift shows):
    python tools/ab_lib.py <base.so> <new.so> [rounds]
Each round runs, per build, `bench.py` for C4 / C2 / C5 inverse and C4 / C2 forward (no CPU leg, no PMC) and prints
ms_per_step; the last lines are the medians."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base, new = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
LINES = (("c4_inv", ["--config", "c4", "--direction", "inv", "--steps", "50", "--warmup", "10"]),
         ("c4_fwd", ["--config", "c4", "--steps", "50", "--warmup", "10"]),
         ("c2_inv", ["--direction", "inv", "--steps", "20", "--warmup", "5"]),
         ("c2_fwd", ["--steps", "20", "--warmup", "5"]),
         ("c5_inv", ["--config", "c5", "--direction", "inv", "--steps", "20", "--warmup", "5"]))
COMMON = ["--no-traffic", "--no-cpu-baseline", "--no-power", "--no-other-configs", "--no-shard-overheads"]
res = {}
for r in range(rounds):
    for tag, so in (("base", base), ("new", new)):
        for name, flags in LINES:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags + COMMON,
                                 env=dict(os.environ, GPUNTT_LIB=os.path.abspath(so)), capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, name, "FAILED", out.stderr[-300:])
                continue
            d = json.loads(line[-1])
            ok = d.get("batch_check", {}).get("bit_exact", d.get("cpu_baseline", {}).get("gpu_output_bit_exact"))
            res.setdefault((name, tag), []).append(d["ms_per_step"])
            print("round %d %-5s %-7s %.4f ms  check=%s" % (r, tag, name, d["ms_per_step"], ok), flush=True)
print("medians (ms per call):")
for name, _ in LINES:
    b, n = statistics.median(res[(name, "base")]), statistics.median(res[(name, "new")])
    print("  %-7s base %.4f  new %.4f  (%+.1f %%)" % (name, b, n, 100 * (n / b - 1)))
