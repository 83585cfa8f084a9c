"""The CMake package (reference src/CMakeLists.txt:54-85, cmake/Config.cmake.in: target GPUNTT::ntt, archive libntt-1.0.a,
find_package(GPUNTT)) configured, built, installed under a prefix and consumed by tests/cmake_consumer -- the downstream
project a user of the reference would write.  The CPU part cross-compiles everything (no GPU needed); the -m gpu part runs
the consumer's binary.  Everything lands under tests/cmake_consumer/_pkg (git-ignored; the library's object directory is
also gpurun-ignored, the installed prefix and the consumer travel to the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tests", "cmake_consumer", "_pkg")
BUILD, PREFIX, CONSUMER = os.path.join(PKG, "build"), os.path.join(PKG, "prefix"), os.path.join(PKG, "consumer")
HIP_CXX = "/opt/rocm/lib/llvm/bin/clang++"
EXE = os.path.join(CONSUMER, "consumer_merge_ntt")


def _run(cmd, timeout=1500):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _build_all():
    cmake = shutil.which("cmake")
    if cmake is None or not os.path.exists(HIP_CXX):
        pytest.skip("cmake / ROCm clang++ not available")
    jobs = str(min(8, os.cpu_count() or 1))
    _run([cmake, "-S", ROOT, "-B", BUILD, "-DCMAKE_HIP_COMPILER=" + HIP_CXX, "-DCMAKE_PREFIX_PATH=/opt/rocm",
          "-DCMAKE_BUILD_TYPE=Release"])
    _run([cmake, "--build", BUILD, "-j", jobs])
    _run([cmake, "--install", BUILD, "--prefix", PREFIX])
    _run([cmake, "-S", os.path.join(ROOT, "tests", "cmake_consumer"), "-B", CONSUMER, "-DCMAKE_HIP_COMPILER=" + HIP_CXX,
          "-DCMAKE_PREFIX_PATH=%s;/opt/rocm" % PREFIX, "-DCMAKE_BUILD_TYPE=Release"])
    _run([cmake, "--build", CONSUMER, "-j", jobs])


def test_cmake_package_builds_installs_and_serves_a_consumer():
    _build_all()
    lib = os.path.join(PREFIX, "lib")
    assert os.path.exists(os.path.join(lib, "libntt-1.0.a")), "the reference's archive name (src/CMakeLists.txt:54-60)"
    assert any(f.startswith("libgpuntt.so") for f in os.listdir(lib))
    cfg = os.path.join(lib, "cmake", "GPUNTT-1.0")
    for f in ("GPUNTTConfig.cmake", "GPUNTTConfigVersion.cmake", "GPUNTTTargets.cmake"):
        assert os.path.exists(os.path.join(cfg, f)), f
    inc = os.path.join(PREFIX, "include", "GPUNTT-1.0")
    for h in ("gpuntt/ntt_merge/ntt.cuh", "gpuntt/ntt_4step/ntt_4step.cuh", "gpuntt/common/modular_arith.cuh", "gpuntt_c.h"):
        assert os.path.exists(os.path.join(inc, h)), h
    assert "GPUNTT::ntt" in open(os.path.join(cfg, "GPUNTTTargets.cmake")).read()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cmake_consumer_runs_on_the_gpu():
    if not os.path.exists(EXE):
        _build_all()
    for args in (("12", "2"), ("16", "3"), ("13", "2", "u32")):
        out = _run([EXE, *args], timeout=600)
        assert "All Correct for PerPolynomial NTT." in out and "All Correct for PerPolynomial INTT." in out
