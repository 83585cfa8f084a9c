#!/usr/bin/env python3
"""Compile every HIP translation unit of the library with -Rpass-analysis=kernel-resource-usage (device code only,
nothing is linked) and list the kernels that use scratch memory (register spills or stack objects): there should be none.

    python tools/check_scratch.py [-j 8]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu-ntt_amd", "csrc")


def one(src, out_dir):
    log = os.path.join(out_dir, os.path.basename(src) + ".log")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "--offload-arch=gfx950", "-Wno-unused-result", "-ffp-contract=off", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(out_dir, os.path.basename(src) + ".o")]
    with open(log, "w") as f:
        subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT)
    bad, total = [], 0
    for b in re.split(r"remark: Function Name: ", open(log).read())[1:]:
        total += 1
        name = b.split(" [-Rpass")[0].strip()
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
        v = re.search(r" VGPRs: (\d+)", b)
        if m and int(m.group(1)) > 0:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            bad.append((dem, int(m.group(1)), int(v.group(1)) if v else -1))
    return os.path.basename(src), total, bad


def main():
    jobs = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else 8
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with tempfile.TemporaryDirectory(prefix="gpuntt_scratch_") as tmp, ThreadPoolExecutor(jobs) as ex:
        n_bad = 0
        for name, total, bad in ex.map(lambda s: one(s, tmp), srcs):
            print("%-24s %4d kernels, %d with scratch" % (name, total, len(bad)))
            for dem, sc, vg in bad:
                print("    %d B/lane, %d VGPRs: %s" % (sc, vg, dem[:150]))
            n_bad += len(bad)
    print("kernels with scratch:", n_bad)
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
