"""Host-side Python mirror of the GPU-NTT operator interface on top of libgpuntt.so.

The product is the C++/HIP library (gpu-ntt_amd/csrc -> gpu-ntt_amd/lib/libgpuntt.so, C ABI
in include/gpuntt_c.h).  This module only binds that C ABI with ctypes so tests, bench.py and
multi-GPU drivers can call the same entry points with the reference's names and argument
meaning (reference src/include/gpuntt/ntt_merge/ntt.cuh:315-421,
src/include/gpuntt/ntt_4step/ntt_4step.cuh:46-49,278-308):

    GPU_NTT / GPU_INTT / GPU_NTT_Inplace / GPU_INTT_Inplace   (single modulus or RNS)
    GPU_4STEP_NTT / GPU_Transpose
    Modulus, ntt_configuration, ntt_rns_configuration, ntt4step_configuration,
    ntt4step_rns_configuration, NTTParameters, NTTParameters4Step

Device buffers are torch tensors (torch is used for device memory and streams only);
64-bit words are carried in int64 tensors, 32-bit words in int32 tensors -- the library
reinterprets the bits as unsigned unless a signed dtype is requested explicitly.

There is NO CPU fallback: importing works anywhere, but every compute entry point raises
if libgpuntt.so is missing or no GPU is present.
"""
import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPUNTT_LIB: an alternative build of the library (A/B experiments under tools/: same sources, one macro changed)
LIB_PATH = os.environ.get("GPUNTT_LIB") or os.path.join(_HERE, "lib", "libgpuntt.so")
CSRC = os.path.join(_HERE, "csrc")

# enum values (reference src/include/gpuntt/common/nttparameters.cuh:19-36)
FORWARD, INVERSE = 0, 1
PerPolynomial, PerCoefficient = 0, 1
X_N_plus, X_N_minus = 0, 1

GPUNTT_OK = 0
_ERR_INVALID, _ERR_HIP = -1, -2


class GpuNttError(RuntimeError):
    """HipException / CudaException of the C++ API (failed launch)."""


class _M32(ctypes.Structure):
    _fields_ = [("value", ctypes.c_uint32), ("bit", ctypes.c_uint32), ("mu", ctypes.c_uint32)]


class _M64(ctypes.Structure):
    _fields_ = [("value", ctypes.c_uint64), ("bit", ctypes.c_uint64), ("mu", ctypes.c_uint64)]


_lib = None


def build_library(jobs=8):
    """Compile every HIP source for gfx950 into gpu-ntt_amd/lib (hipcc cross-compiles
    without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", CSRC, "-j%d" % jobs])


def load_library():
    """dlopen libgpuntt.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s not found: build it with `make -C gpu-ntt_amd/csrc -j8` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    try:  # make sure the HIP runtime torch already loaded is the one we bind to
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the host-only helpers
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.gpuntt_last_error.restype = ctypes.c_char_p
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name)  # AttributeError if the ABI is incomplete
    _lib = lib
    for var, (opt, conv) in ENV_OPTIONS.items():
        if var in os.environ and "gpuntt_set_option" in EXPORTED_SYMBOLS:
            val = os.environ[var]
            set_option(opt, conv(val) if conv else val)
    return lib


# every symbol include/gpuntt_c.h declares
EXPORTED_SYMBOLS = ["gpuntt_last_error", "gpuntt_version"] + [
    "gpuntt_%s_%s" % (f, s)
    for f in ("modulus", "ntt", "intt", "ntt_rns", "intt_rns", "ntt_modulus_ordered",
              "ntt_poly_ordered", "polymul", "polymul_rns", "4step", "4step_rns", "4step_natural", "transpose",
              "merge_params",
              "4step_params", "plan_workspace_bytes", "plan_create", "plan_execute", "plan_fast_path",
              "plan_destroy", "operator_gpu", "4step_plan_workspace_bytes", "4step_plan_create",
              "4step_plan_execute", "4step_plan_fast_path", "4step_plan_destroy",
              "generate_power_table", "generate_4step_w", "butterfly_unit", "debug_recip_norm")
    for s in ("u32", "u64")] + ["gpuntt_release_workspaces", "gpuntt_set_option"]

# GPUNTT_* environment variables of the A/B scripts and tests -> library options.  The C++ library reads no
# environment variable; this harness forwards them once, when it loads the library.
ENV_OPTIONS = {"GPUNTT_PATH": ("path", None), "GPUNTT_U32_E32": ("u32_e32", lambda v: int(v, 0)),
               "GPUNTT_TWO_SWEEP_BIG": ("two_sweep_big", None)}


TEST_HOOKS = {"no_scratch", "rns_force_fallback", "u32_e32", "reset_predictions", "two_sweep_big"}
TEST_PATHS = {"fast-strict", "generic-capped"}


def set_option(name, value):
    """GPU_NTT_SetOption (product options: path = default | generic | fast, check_4step_tables, rns_predict; names in
    include/gpuntt/ntt_merge/ntt.cuh).  The TEST HOOKS of this repository (csrc/test_hooks.h: path = fast-strict |
    generic-capped, no_scratch, rns_force_fallback, u32_e32) are not options of the public interface; this harness
    forwards them to gpuntt_test_set_hook so that the tests and tools/ keep one call."""
    name, value = str(name), str(value)
    if name in TEST_HOOKS or (name == "path" and value in TEST_PATHS):
        return set_test_hook(name, value)
    _check(load_library().gpuntt_set_option(name.encode(), value.encode()))


def set_test_hook(name, value):
    """gpuntt_test_set_hook (csrc/test_hooks.h)."""
    _check(load_library().gpuntt_test_set_hook(str(name).encode(), str(value).encode()))


def scratch_stats():
    """gpuntt_test_scratch_stats (csrc/test_hooks.h): the twiddle scratch of captured calls, which their graph owns."""
    out = (ctypes.c_ulonglong * 6)()
    _check(load_library().gpuntt_test_scratch_stats(out))
    names = ("graph_owned", "died", "pooled", "reused", "chains_erased", "chains")
    return dict(zip(names, (int(v) for v in out)))


class launch_log:
    """with launch_log() as log: ...calls...; log.kernels -> the kernels the library enqueued inside the block, in order
    (gpuntt_test_launch_log_start / _take, csrc/test_hooks.h): ["prep_twiddles", "merge_pass_lazy:31", ...]"""

    def __enter__(self):
        _check(load_library().gpuntt_test_launch_log_start())
        self.kernels = []
        return self

    def __exit__(self, *exc):
        lib = load_library()
        buf = ctypes.create_string_buffer(1 << 16)
        lib.gpuntt_test_launch_log_take(buf, len(buf))
        self.kernels = buf.value.decode().split()
        return False


def _check(rc):
    if rc == GPUNTT_OK:
        return
    msg = load_library().gpuntt_last_error().decode()
    if rc == _ERR_INVALID:
        raise ValueError(msg)  # std::invalid_argument
    raise GpuNttError(msg)


def _bits_of(dtype):
    return {"u32": 32, "s32": 32, "u64": 64, "s64": 64}[dtype]


def _ct(bits):
    return ctypes.c_uint32 if bits == 32 else ctypes.c_uint64


def np_dtype(bits):
    return np.uint32 if bits == 32 else np.uint64


# ------------------------------------------------------------------------------ structs
@dataclass
class Modulus:
    """Modulus<T>{value, bit, mu} (reference modular_arith.cuh:28-57)."""
    value: int
    bit: int = 0
    mu: int = 0
    bits: int = 64

    def __post_init__(self):
        if self.bit == 0:
            lib = load_library()
            m = _M32() if self.bits == 32 else _M64()
            fn = lib.gpuntt_modulus_u32 if self.bits == 32 else lib.gpuntt_modulus_u64
            _check(fn(_ct(self.bits)(self.value), ctypes.byref(m)))
            self.bit, self.mu = int(m.bit), int(m.mu)

    def c(self):
        return (_M32 if self.bits == 32 else _M64)(self.value, self.bit, self.mu)

    def words(self):
        return [self.value, self.bit, self.mu]


@dataclass
class ntt_configuration:
    """reference ntt.cuh:31-40; mod_inverse is a host value."""
    n_power: int
    ntt_type: int = FORWARD
    ntt_layout: int = PerPolynomial
    reduction_poly: int = X_N_minus
    zero_padding: bool = False
    mod_inverse: int = 0
    stream: Optional[object] = None


@dataclass
class ntt_rns_configuration:
    """reference ntt.cuh:42-51; mod_inverse is a device tensor (one word per modulus)."""
    n_power: int
    ntt_type: int = FORWARD
    ntt_layout: int = PerPolynomial
    reduction_poly: int = X_N_minus
    zero_padding: bool = False
    mod_inverse: Optional[object] = None
    stream: Optional[object] = None


@dataclass
class ntt4step_configuration:
    """reference ntt_4step.cuh:19-25"""
    n_power: int
    ntt_type: int = FORWARD
    mod_inverse: int = 0
    stream: Optional[object] = None


@dataclass
class ntt4step_rns_configuration:
    """reference ntt_4step.cuh:27-33"""
    n_power: int
    ntt_type: int = FORWARD
    mod_inverse: Optional[object] = None
    stream: Optional[object] = None


# --------------------------------------------------------------- host-side parameters
class NTTParameters:
    """NTTParameters<T> (reference nttparameters.cuh:56-104) generated by the library's own
    host code; tables are exposed in DEVICE (bit-reversed) order as numpy arrays."""

    def __init__(self, logn, poly_reduction, bits=64, factors=None):
        lib = load_library()
        T = _ct(bits)
        self.bits, self.logn, self.n, self.poly_reduction = bits, logn, 1 << logn, poly_reduction
        size = (1 << (logn - 1)) if poly_reduction == X_N_minus else (1 << logn)
        info = (ctypes.c_uint64 * 8)()
        fwd = np.empty(size, dtype=np_dtype(bits))
        inv = np.empty(size, dtype=np_dtype(bits))
        fac = (T * 3)(*factors) if factors is not None else None
        fn = getattr(lib, "gpuntt_merge_params_u%d" % bits)
        _check(fn(logn, poly_reduction, fac, info, fwd.ctypes.data_as(ctypes.c_void_p),
                  inv.ctypes.data_as(ctypes.c_void_p)))
        self.modulus = Modulus(int(info[0]), int(info[1]), int(info[2]), bits)
        self.omega, self.psi, self.n_inv = int(info[3]), int(info[4]), int(info[5])
        self.root_of_unity_size = int(info[6])
        self.forward_table_device_order = fwd
        self.inverse_table_device_order = inv


class NTTParameters4Step:
    """NTTParameters4Step<T> (reference nttparameters.cuh:106-170)."""

    def __init__(self, logn, bits=64):
        lib = load_library()
        self.bits, self.logn, self.n = bits, logn, 1 << logn
        fn = getattr(lib, "gpuntt_4step_params_u%d" % bits)
        info = (ctypes.c_uint64 * 9)()
        _check(fn(logn, 0, info, None, None, None))
        self.modulus = Modulus(int(info[0]), int(info[1]), int(info[2]), bits)
        self.omega, self.psi, self.n_inv = int(info[3]), int(info[4]), int(info[5])
        self.n1, self.n2 = int(info[6]), int(info[7])
        self.tables = {}
        for inverse, tag in ((0, "fwd"), (1, "inv")):
            t1 = np.empty(self.n1 >> 1, dtype=np_dtype(bits))
            t2 = np.empty(self.n2 >> 1, dtype=np_dtype(bits))
            w = np.empty(self.n, dtype=np_dtype(bits))
            _check(fn(logn, inverse, info, t1.ctypes.data_as(ctypes.c_void_p),
                      t2.ctypes.data_as(ctypes.c_void_p), w.ctypes.data_as(ctypes.c_void_p)))
            self.tables[tag] = (t1, t2, w)


# -------------------------------------------------------------------- device plumbing
def to_device(a, device="cuda:0"):
    """numpy unsigned/signed 32/64-bit array -> torch tensor (int32/int64 storage) on GPU."""
    import torch
    a = np.ascontiguousarray(a)
    view = a.view(np.int32 if a.dtype.itemsize == 4 else np.int64)
    return torch.from_numpy(view.copy()).to(device)


def to_host(t, signed=False):
    a = t.detach().cpu().numpy()
    if signed:
        return a
    return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


def modulus_array_to_device(moduli, bits=64, device="cuda:0"):
    """[Modulus...] -> device array of Modulus<T> (3 words each)."""
    words = np.array([w for m in moduli for w in m.words()], dtype=np_dtype(bits))
    return to_device(words, device)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream(s):
    if s is None:
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if isinstance(s, int):
        return ctypes.c_void_p(s)
    return ctypes.c_void_p(s.cuda_stream)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gpu-ntt_amd has no CPU path: tensors must live on the GPU")


def _width(t, dtype):
    if dtype is not None:
        return _bits_of(dtype)
    return t.element_size() * 8


# --------------------------------------------------------------------- the operator API
def GPU_NTT(device_in, device_out, root_of_unity_table, modulus, cfg, batch_size,
            mod_count=None, dtype=None):
    """Forward Merge NTT, natural order in -> bit-reversed out.  `modulus` is a Modulus
    (single-modulus overload, ntt.cuh:315-321) or a device tensor of Modulus<T> words with
    `mod_count` (RNS overload, ntt.cuh:395-401).  dtype 's32'/'s64' selects the signed-input
    instantiation."""
    lib = load_library()
    _require_gpu(device_in, device_out, root_of_unity_table)
    bits = _width(device_out, dtype)
    signed = int(dtype in ("s32", "s64"))
    if isinstance(modulus, Modulus):
        fn = getattr(lib, "gpuntt_ntt_u%d" % bits)
        _check(fn(_ptr(device_in), _ptr(device_out), _ptr(root_of_unity_table), modulus.c(),
                  cfg.n_power, cfg.ntt_layout, cfg.reduction_poly, signed, _stream(cfg.stream),
                  batch_size))
    else:
        _require_gpu(modulus)
        fn = getattr(lib, "gpuntt_ntt_rns_u%d" % bits)
        _check(fn(_ptr(device_in), _ptr(device_out), _ptr(root_of_unity_table), _ptr(modulus),
                  cfg.n_power, cfg.ntt_layout, cfg.reduction_poly, signed, _stream(cfg.stream),
                  batch_size, int(mod_count)))


def GPU_INTT(device_in, device_out, root_of_unity_table, modulus, cfg, batch_size,
             mod_count=None, dtype=None):
    """Inverse Merge NTT, bit-reversed in -> natural out, scaled by cfg.mod_inverse.
    dtype 's32'/'s64' selects the centred signed-output instantiation."""
    lib = load_library()
    _require_gpu(device_in, device_out, root_of_unity_table)
    bits = _width(device_in, dtype)
    signed = int(dtype in ("s32", "s64"))
    if isinstance(modulus, Modulus):
        fn = getattr(lib, "gpuntt_intt_u%d" % bits)
        _check(fn(_ptr(device_in), _ptr(device_out), _ptr(root_of_unity_table), modulus.c(),
                  cfg.n_power, cfg.ntt_layout, cfg.reduction_poly, _ct(bits)(cfg.mod_inverse),
                  signed, _stream(cfg.stream), batch_size))
    else:
        _require_gpu(modulus, cfg.mod_inverse)
        fn = getattr(lib, "gpuntt_intt_rns_u%d" % bits)
        _check(fn(_ptr(device_in), _ptr(device_out), _ptr(root_of_unity_table), _ptr(modulus),
                  cfg.n_power, cfg.ntt_layout, cfg.reduction_poly, _ptr(cfg.mod_inverse), signed,
                  _stream(cfg.stream), batch_size, int(mod_count)))


def GPU_NTT_Inplace(device_inout, root_of_unity_table, modulus, cfg, batch_size, mod_count=None):
    GPU_NTT(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size, mod_count)


def GPU_INTT_Inplace(device_inout, root_of_unity_table, modulus, cfg, batch_size, mod_count=None):
    GPU_INTT(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size, mod_count)


def _ordered(fname, device_in, device_out, root_of_unity_table, modulus, cfg, batch_size, mod_count, order):
    lib = load_library()
    _require_gpu(device_in, device_out, root_of_unity_table, modulus, order)
    bits = device_in.element_size() * 8
    fn = getattr(lib, "gpuntt_%s_u%d" % (fname, bits))
    _check(fn(_ptr(device_in), _ptr(device_out), _ptr(root_of_unity_table), _ptr(modulus), cfg.n_power,
              cfg.ntt_type, cfg.reduction_poly, _ptr(cfg.mod_inverse), _stream(cfg.stream), batch_size,
              int(mod_count), _ptr(order)))


def GPU_NTT_Modulus_Ordered(device_in, device_out, root_of_unity_table, modulus, cfg, batch_size,
                            mod_count, order):
    """polynomial p uses prime order[p % mod_count] (reference ntt.cuh:515-529); cfg.ntt_type picks
    FORWARD / INVERSE; `order` is an int32 device tensor."""
    _ordered("ntt_modulus_ordered", device_in, device_out, root_of_unity_table, modulus, cfg,
             batch_size, mod_count, order)


def GPU_NTT_Poly_Ordered(device_in, device_out, root_of_unity_table, modulus, cfg, batch_size,
                         mod_count, order):
    """polynomial p is the one in slot order[p] and uses modulus p % mod_count
    (reference ntt.cuh:589-603)."""
    _ordered("ntt_poly_ordered", device_in, device_out, root_of_unity_table, modulus, cfg, batch_size,
             mod_count, order)


def GPU_Transpose(polynomial_in, polynomial_out, row, col, n_power, batch_size):
    """per polynomial (row x col) -> (col x row); default stream (ntt_4step.cuh:46-49)."""
    lib = load_library()
    _require_gpu(polynomial_in, polynomial_out)
    bits = polynomial_in.element_size() * 8
    fn = getattr(lib, "gpuntt_transpose_u%d" % bits)
    _check(fn(_ptr(polynomial_in), _ptr(polynomial_out), row, col, n_power, batch_size))


def GPU_4STEP_NTT(device_in, device_out, n1_root_of_unity_table, n2_root_of_unity_table,
                  W_root_of_unity_table, modulus, cfg, batch_size, mod_count=None):
    """4-step transform of an already transposed (n2 x n1) input into an (n1 x n2) output
    (ntt_4step.cuh:278-308); cfg.ntt_type selects FORWARD / INVERSE."""
    lib = load_library()
    _require_gpu(device_in, device_out, n1_root_of_unity_table, n2_root_of_unity_table,
                 W_root_of_unity_table)
    bits = device_in.element_size() * 8
    if isinstance(modulus, Modulus):
        fn = getattr(lib, "gpuntt_4step_u%d" % bits)
        _check(fn(_ptr(device_in), _ptr(device_out), _ptr(n1_root_of_unity_table),
                  _ptr(n2_root_of_unity_table), _ptr(W_root_of_unity_table), modulus.c(),
                  cfg.n_power, cfg.ntt_type, _ct(bits)(cfg.mod_inverse), _stream(cfg.stream),
                  batch_size))
    else:
        fn = getattr(lib, "gpuntt_4step_rns_u%d" % bits)
        _check(fn(_ptr(device_in), _ptr(device_out), _ptr(n1_root_of_unity_table),
                  _ptr(n2_root_of_unity_table), _ptr(W_root_of_unity_table), _ptr(modulus),
                  cfg.n_power, cfg.ntt_type, _ptr(cfg.mod_inverse), _stream(cfg.stream),
                  batch_size, int(mod_count)))


def GPU_PolyMul(device_a, device_b, device_out, forward_table, inverse_table, modulus, cfg, batch_size,
                mod_count=None):
    """Extension: device_out = INTT(NTT(a) (.) NTT(b)) -- the product in Z_q[X]/(X^N -+ 1) that the
    reference's CPU example builds from NTTCPU::ntt / mult / intt.  a and b are overwritten with their
    transforms; cfg carries n_power, reduction_poly, mod_inverse (N^-1; device array for RNS), stream."""
    lib = load_library()
    _require_gpu(device_a, device_b, device_out, forward_table, inverse_table)
    bits = device_a.element_size() * 8
    if isinstance(modulus, Modulus):
        fn = getattr(lib, "gpuntt_polymul_u%d" % bits)
        _check(fn(_ptr(device_a), _ptr(device_b), _ptr(device_out), _ptr(forward_table), _ptr(inverse_table),
                  modulus.c(), cfg.n_power, cfg.reduction_poly, _ct(bits)(cfg.mod_inverse),
                  _stream(cfg.stream), batch_size))
    else:
        fn = getattr(lib, "gpuntt_polymul_rns_u%d" % bits)
        _check(fn(_ptr(device_a), _ptr(device_b), _ptr(device_out), _ptr(forward_table), _ptr(inverse_table),
                  _ptr(modulus), cfg.n_power, cfg.reduction_poly, _ptr(cfg.mod_inverse), _stream(cfg.stream),
                  batch_size, int(mod_count)))


def GPU_4STEP_NTT_NaturalOrder(device_in, device_out, n1_root_of_unity_table, n2_root_of_unity_table,
                               W_root_of_unity_table, modulus, cfg, batch_size):
    """Extension: natural-order input -> NTT_4STEP_CPU::ntt / ::intt order in one call (what the
    reference's examples do with GPU_Transpose -> GPU_4STEP_NTT -> GPU_Transpose).  device_in is
    overwritten; device_in is not device_out."""
    lib = load_library()
    _require_gpu(device_in, device_out, n1_root_of_unity_table, n2_root_of_unity_table,
                 W_root_of_unity_table)
    bits = device_in.element_size() * 8
    fn = getattr(lib, "gpuntt_4step_natural_u%d" % bits)
    _check(fn(_ptr(device_in), _ptr(device_out), _ptr(n1_root_of_unity_table),
              _ptr(n2_root_of_unity_table), _ptr(W_root_of_unity_table), modulus.c(), cfg.n_power,
              cfg.ntt_type, _ct(bits)(cfg.mod_inverse), _stream(cfg.stream), batch_size))


class NTTPlan:
    """Extension NTTPlan<T> (include/gpuntt/ntt_merge/ntt.cuh): twiddles prepared once into a workspace;
    execute() launches the transform kernels only (no allocation, synchronisation or preparation).
    `moduli` is a Modulus or a list of Modulus (RNS: polynomial p uses modulus p % len); `mod_inverse`
    an int or list of ints (inverse plans); `workspace` an optional uint8 device tensor of
    workspace_bytes() bytes owned by the caller."""

    def __init__(self, table_device, moduli, n_power, reduction_poly=X_N_minus, ntt_type=FORWARD,
                 mod_inverse=None, batch_hint=1024, stream=None, workspace=None):
        lib = load_library()
        _require_gpu(table_device)
        moduli = [moduli] if isinstance(moduli, Modulus) else list(moduli)
        self.bits = moduli[0].bits
        self.n_power, self.ntt_type, self.mod_count = n_power, ntt_type, len(moduli)
        T = _ct(self.bits)
        marr = ((_M32 if self.bits == 32 else _M64) * len(moduli))(*[m.c() for m in moduli])
        ninv = None
        if mod_inverse is not None:
            vals = [mod_inverse] if isinstance(mod_inverse, int) else list(mod_inverse)
            ninv = (T * len(vals))(*vals)
        self._keep = (table_device, workspace)
        self._h = ctypes.c_void_p()
        fn = getattr(lib, "gpuntt_plan_create_u%d" % self.bits)
        _check(fn(ctypes.byref(self._h), _ptr(table_device), marr, len(moduli), n_power, reduction_poly,
                  ntt_type, ninv, int(batch_hint), _ptr(workspace), _stream(stream)))

    @staticmethod
    def workspace_bytes(n_power, mod_count=1, bits=64):
        lib = load_library()
        out = ctypes.c_uint64()
        _check(getattr(lib, "gpuntt_plan_workspace_bytes_u%d" % bits)(n_power, mod_count, ctypes.byref(out)))
        return int(out.value)

    @property
    def fast_path(self):
        return bool(getattr(load_library(), "gpuntt_plan_fast_path_u%d" % self.bits)(self._h))

    def execute(self, device_in, device_out, batch_size, stream=None, io_signed=False):
        _require_gpu(device_in, device_out)
        fn = getattr(load_library(), "gpuntt_plan_execute_u%d" % self.bits)
        _check(fn(self._h, _ptr(device_in), _ptr(device_out), int(batch_size), int(io_signed), _stream(stream)))

    def close(self):
        if self._h:
            getattr(load_library(), "gpuntt_plan_destroy_u%d" % self.bits)(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FourStepPlan:
    """Extension FourStepPlan<T> (include/gpuntt/ntt_4step/ntt_4step.cuh): the Shoup pairs of the n1 / n2 / W
    tables are prepared once; execute() launches the sweeps only.  natural_order False: execute ==
    GPU_4STEP_NTT (n2 x n1 in, n1 x n2 out); True: == GPU_4STEP_NTT_NaturalOrder (device_in is scratch).
    `cfg` is an ntt4step_configuration (n_power, ntt_type, mod_inverse, stream of the preparation);
    `workspace` an optional uint8 device tensor of workspace_bytes() bytes owned by the caller."""

    def __init__(self, n1_root_of_unity_table, n2_root_of_unity_table, W_root_of_unity_table, modulus, cfg,
                 natural_order=False, batch_hint=1, workspace=None):
        lib = load_library()
        _require_gpu(n1_root_of_unity_table, n2_root_of_unity_table, W_root_of_unity_table)
        self.bits = modulus.bits
        self.n_power, self.ntt_type, self.natural_order = cfg.n_power, cfg.ntt_type, bool(natural_order)
        self._keep = (n1_root_of_unity_table, n2_root_of_unity_table, W_root_of_unity_table, workspace)
        self._h = ctypes.c_void_p()
        fn = getattr(lib, "gpuntt_4step_plan_create_u%d" % self.bits)
        _check(fn(ctypes.byref(self._h), _ptr(n1_root_of_unity_table), _ptr(n2_root_of_unity_table),
                  _ptr(W_root_of_unity_table), modulus.c(), cfg.n_power, cfg.ntt_type,
                  _ct(self.bits)(cfg.mod_inverse), int(self.natural_order), int(batch_hint), _ptr(workspace),
                  _stream(cfg.stream)))

    @staticmethod
    def workspace_bytes(n_power, bits=64):
        out = ctypes.c_uint64()
        _check(getattr(load_library(), "gpuntt_4step_plan_workspace_bytes_u%d" % bits)(n_power, ctypes.byref(out)))
        return int(out.value)

    @property
    def fast_path(self):
        return bool(getattr(load_library(), "gpuntt_4step_plan_fast_path_u%d" % self.bits)(self._h))

    def execute(self, device_in, device_out, batch_size, stream=None):
        _require_gpu(device_in, device_out)
        fn = getattr(load_library(), "gpuntt_4step_plan_execute_u%d" % self.bits)
        _check(fn(self._h, _ptr(device_in), _ptr(device_out), int(batch_size), _stream(stream)))

    def close(self):
        if self._h:
            getattr(load_library(), "gpuntt_4step_plan_destroy_u%d" % self.bits)(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def GPU_GeneratePowerTable(device_out, base, modulus, log_count, bit_reversed=True, stream=None):
    """Extension: device_out[k] = base^(bitreverse(k, log_count) if bit_reversed else k), k < 2^log_count -- the
    device-order root tables of GPU_NTT / GPU_INTT / the 4-step n1, n2 slots, built on the device."""
    _require_gpu(device_out)
    fn = getattr(load_library(), "gpuntt_generate_power_table_u%d" % modulus.bits)
    _check(fn(_ptr(device_out), _ct(modulus.bits)(base), modulus.c(), int(log_count), int(bool(bit_reversed)),
              _stream(stream)))


def GPU_Generate4StepW(device_W, root, modulus, n_power, ntt_type=FORWARD, stream=None):
    """Extension: the 4-step W matrix on the device: FORWARD W[i*n2+j] = root^(bitreverse(i)*j) (root_of_unity),
    INVERSE W[i*n2+j] = root^(bitreverse(j)*i) (inverse_root_of_unity)."""
    _require_gpu(device_W)
    fn = getattr(load_library(), "gpuntt_generate_4step_w_u%d" % modulus.bits)
    _check(fn(_ptr(device_W), _ct(modulus.bits)(root), modulus.c(), int(n_power), int(ntt_type), _stream(stream)))


def release_workspaces():
    """GPU_NTT_ReleaseWorkspaces(): frees the library-owned twiddle scratch of the drop-in calls."""
    _check(load_library().gpuntt_release_workspaces())


def operator_gpu(op, a, b, modulus):
    """diagnostic: OPERATOR_GPU<T>::{add, sub, mult, reduce, reduce(signed), centered_reduction}
    (op 0..5) elementwise on device tensors; returns a new tensor"""
    import torch
    lib = load_library()
    _require_gpu(a, b)
    out = torch.empty_like(a)
    fn = getattr(lib, "gpuntt_operator_gpu_u%d" % modulus.bits)
    _check(fn(int(op), _ptr(a), _ptr(b), _ptr(out), modulus.c(), ctypes.c_uint64(a.numel()), _stream(None)))
    return out


def debug_recip_norm(q):
    """diagnostic: the preparation kernels' normalised reciprocal of every word of the device tensor q; returns a new tensor"""
    import torch
    _require_gpu(q)
    out = torch.empty_like(q)
    fn = getattr(load_library(), "gpuntt_debug_recip_norm_u%d" % (q.element_size() * 8))
    _check(fn(_ptr(q), _ptr(out), ctypes.c_uint64(q.numel()), _stream(None)))
    return out


def butterfly_unit(u, v, roots, modulus, gentleman_sande=False):
    """diagnostic: the public device helpers CooleyTukeyUnit / GentlemanSandeUnit (reference ntt.cuh:69-92) applied to
    the pairs (u[i], v[i]) with roots[i], in place on the device tensors"""
    lib = load_library()
    _require_gpu(u, v, roots)
    fn = getattr(lib, "gpuntt_butterfly_unit_u%d" % modulus.bits)
    _check(fn(int(bool(gentleman_sande)), _ptr(u), _ptr(v), _ptr(roots), modulus.c(), ctypes.c_uint64(u.numel()),
              _stream(None)))


# ------------------------------------------------------------------ multi-GPU batch shard
def shard_range(batch_size, rank, world_size, mod_count=1):
    """Rank r of G owns polynomials [lo, hi) of the batch; shards are aligned to mod_count so
    the local p % mod_count equals the global one.  Polynomials are
    independent: no data-path collective exists anywhere in a transform."""
    if batch_size % mod_count:
        raise ValueError("batch_size must be a multiple of mod_count")
    groups = batch_size // mod_count
    lo = (groups * rank) // world_size
    hi = (groups * (rank + 1)) // world_size
    return lo * mod_count, hi * mod_count
