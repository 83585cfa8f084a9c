"""ctypes bindings for the two CPU checkers -- TEST INFRASTRUCTURE ONLY.

* ``Port(bits)``  -> oracle/_build/libntt_oracle.so, this repo's C restatement
  (oracle/ntt_oracle.c) of the reference CPU algorithms.
* ``Ref(bits)``   -> oracle/_ref/libgpuntt_ref.so, the reference's own CPU classes
  compiled from /root/reference by oracle/Makefile (only buildable where the
  reference is mounted; the built file travels to the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package (gpu-ntt_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "_build", "libntt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libgpuntt_ref.so")
REFERENCE_ROOT = "/root/reference"

X_N_plus, X_N_minus = 0, 1  # enum ReductionPolynomial, nttparameters.cuh:32-36


def build(ref=True):
    """(Re)build the checkers; the reference build is attempted only where the
    reference tree is mounted."""
    subprocess.check_call(["make", "-s", "-C", HERE, "port"])
    if ref and os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref():
    return os.path.exists(REF_SO)


def _np_t(bits):
    return np.uint32 if bits == 32 else np.uint64


def _c_t(bits):
    return ctypes.c_uint32 if bits == 32 else ctypes.c_uint64


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Port:
    """The C restatement.  All tables handled here are NATURAL order unless a
    method says otherwise (``bitrev_table`` converts to the GPU order)."""

    def __init__(self, bits):
        assert bits in (32, 64)
        if not os.path.exists(PORT_SO):
            build(ref=False)
        self.bits = bits
        self.T = _np_t(bits)
        self.c = _c_t(bits)
        self.lib = ctypes.CDLL(PORT_SO)
        self.p = "ora%d_" % bits
        self.lib.ora_bitreverse.restype = ctypes.c_int
        for name in ("add", "sub", "mult", "exp", "modinv"):
            getattr(self.lib, self.p + name).restype = self.c

    def f(self, name):
        return getattr(self.lib, self.p + name)

    def modulus(self, q):
        bit, mu = self.c(), self.c()
        self.f("modulus")(self.c(q), ctypes.byref(bit), ctypes.byref(mu))
        return int(q), int(bit.value), int(mu.value)

    def mult(self, a, b, mod):
        q, bit, mu = mod
        return int(self.f("mult")(self.c(a), self.c(b), self.c(q), self.c(bit), self.c(mu)))

    def exp(self, a, e, mod):
        q, bit, mu = mod
        return int(self.f("exp")(self.c(a), self.c(e), self.c(q), self.c(bit), self.c(mu)))

    def modinv(self, a, mod):
        q, bit, mu = mod
        return int(self.f("modinv")(self.c(a), self.c(q), self.c(bit), self.c(mu)))

    def merge_pool(self, logn):
        q, w, p = self.c(), self.c(), self.c()
        self.f("merge_pool")(logn, ctypes.byref(q), ctypes.byref(w), ctypes.byref(p))
        return int(q.value), int(w.value), int(p.value)

    def power_table(self, root, size, mod):
        q, bit, mu = mod
        out = np.empty(size, dtype=self.T)
        self.f("power_table")(self.c(root), ctypes.c_uint64(size), self.c(q), self.c(bit),
                              self.c(mu), _ptr(out))
        return out

    def bitrev_table(self, table):
        table = np.ascontiguousarray(table, dtype=self.T)
        out = np.empty_like(table)
        self.f("bitrev_table")(_ptr(table), ctypes.c_uint64(table.size), _ptr(out))
        return out

    def merge_params(self, logn, poly, factors=None):
        """Restates NTTParameters<T>(LOGN[, factors], poly) (nttparameters.cu:22-82).
        Returns a dict with modulus triple, roots, n_inv and the NATURAL tables."""
        if factors is None:
            q, omega, psi = self.merge_pool(logn)
        else:
            q, omega, psi = factors
        mod = self.modulus(q)
        root = omega if poly == X_N_minus else psi
        inv_root = self.modinv(root, mod)
        size = (1 << (logn - 1)) if poly == X_N_minus else (1 << logn)
        return dict(logn=logn, n=1 << logn, poly=poly, mod=mod, omega=omega, psi=psi,
                    root=root, inv_root=inv_root, root_size=size,
                    n_inv=self.modinv(1 << logn, mod),
                    fwd=self.power_table(root, size, mod),
                    inv=self.power_table(inv_root, size, mod))

    def merge_ntt(self, x, prm, inverse=False):
        """NTTCPU::ntt / ::intt on a (batch, n) or (n,) array; returns a new array."""
        q, bit, mu = prm["mod"]
        x = np.array(x, dtype=self.T, copy=True, order="C")
        rows = x.reshape(-1, prm["n"])
        tab = prm["inv"] if inverse else prm["fwd"]
        fn = self.f("merge_intt" if inverse else "merge_ntt")
        for r in rows:
            fn(_ptr(r), prm["logn"], prm["poly"], _ptr(tab), self.c(q), self.c(bit), self.c(mu))
        return x

    def merge_batch(self, x, logn, poly, inverse, tables, table_stride, moduli, nthreads):
        """In-place batch transform with per-poly modulus p % mod_count (timing leg)."""
        moduli = np.ascontiguousarray(moduli, dtype=self.T)
        self.f("merge_batch")(_ptr(x), x.size >> logn, logn, poly, int(inverse), _ptr(tables),
                              ctypes.c_uint64(table_stride), _ptr(moduli), moduli.size,
                              nthreads)

    def pointwise(self, a, b, mod):
        q, bit, mu = mod
        out = np.empty_like(a)
        self.f("pointwise")(_ptr(a), _ptr(b), _ptr(out), ctypes.c_uint64(a.size), self.c(q),
                            self.c(bit), self.c(mu))
        return out

    def schoolbook(self, a, b, poly, mod):
        q, bit, mu = mod
        out = np.empty_like(a)
        rc = self.f("schoolbook")(_ptr(a), _ptr(b), _ptr(out), a.size, poly, self.c(q),
                                  self.c(bit), self.c(mu))
        assert rc == 0
        return out

    def fourstep_params(self, logn, with_W=True):
        """Restates NTTParameters4Step<T>(LOGN, X_N_minus) (nttparameters.cu:191-227)."""
        q, w, p = self.c(), self.c(), self.c()
        n1, n2 = ctypes.c_int(), ctypes.c_int()
        rc = self.f("fourstep_pool")(logn, ctypes.byref(q), ctypes.byref(w), ctypes.byref(p),
                                     ctypes.byref(n1), ctypes.byref(n2))
        if rc != 0:
            raise ValueError("4-step logn must be in 12..24")
        n1, n2, n = n1.value, n2.value, 1 << logn
        mod = self.modulus(q.value)
        qq, bit, mu = mod
        root = int(w.value)  # X_N_minus only
        inv_root = self.modinv(root, mod)
        prm = dict(logn=logn, n=n, n1=n1, n2=n2, mod=mod, omega=root, psi=int(p.value),
                   root=root, inv_root=inv_root, n_inv=self.modinv(n, mod))
        for inverse, tag in ((0, "fwd"), (1, "inv")):
            t1 = np.empty(n1 >> 1, dtype=self.T)
            t2 = np.empty(n2 >> 1, dtype=self.T)
            self.f("fourstep_small_tables")(self.c(root), ctypes.c_uint64(n), n1, n2, inverse,
                                            self.c(qq), self.c(bit), self.c(mu), _ptr(t1),
                                            _ptr(t2))
            prm["n1_" + tag], prm["n2_" + tag] = t1, t2
            if with_W:
                W = np.empty(n, dtype=self.T)
                self.f("fourstep_W")(self.c(inv_root if inverse else root), n1, n2, inverse,
                                     self.c(qq), self.c(bit), self.c(mu), _ptr(W))
                prm["W_" + tag] = W
        return prm

    def fourstep_ntt(self, x, prm, inverse=False):
        q, bit, mu = prm["mod"]
        x = np.ascontiguousarray(x, dtype=self.T)
        out = np.empty_like(x)
        rin, rout = x.reshape(-1, prm["n"]), out.reshape(-1, prm["n"])
        for a, b in zip(rin, rout):
            if inverse:
                rc = self.f("fourstep_intt")(_ptr(a), _ptr(b), prm["n1"], prm["n2"],
                                             _ptr(prm["n1_inv"]), _ptr(prm["n2_inv"]),
                                             _ptr(prm["W_inv"]), self.c(prm["n_inv"]),
                                             self.c(q), self.c(bit), self.c(mu))
            else:
                rc = self.f("fourstep_ntt")(_ptr(a), _ptr(b), prm["n1"], prm["n2"],
                                            _ptr(prm["n1_fwd"]), _ptr(prm["n2_fwd"]),
                                            _ptr(prm["W_fwd"]), self.c(q), self.c(bit),
                                            self.c(mu))
            assert rc == 0
        return out

    def fourstep_ntt_tables(self, x, prm, n1_table, n2_table, W, inverse=False, q=None, n_inv=None):
        """NTT_4STEP_CPU::ntt / ::intt on CALLER-SUPPLIED natural-order tables (n1/2, n2/2, N words) instead of the
        parameter set's own (ntt_4step_cpu.cu:33-111: the class computes whatever its public tables say); q / n_inv
        replace the modulus / the final scaling."""
        d = dict(prm)
        tag = "inv" if inverse else "fwd"
        d["n1_" + tag] = np.ascontiguousarray(n1_table[:prm["n1"] >> 1], dtype=self.T)
        d["n2_" + tag] = np.ascontiguousarray(n2_table[:prm["n2"] >> 1], dtype=self.T)
        d["W_" + tag] = np.ascontiguousarray(W[:prm["n"]], dtype=self.T)
        if q is not None:
            d["mod"] = self.modulus(q)
        if n_inv is not None:
            d["n_inv"] = int(n_inv)
        return self.fourstep_ntt(x, d, inverse)

    def fourstep_intt_first_transpose(self, x, prm):
        x = np.ascontiguousarray(x, dtype=self.T)
        out = np.empty_like(x)
        for a, b in zip(x.reshape(-1, prm["n"]), out.reshape(-1, prm["n"])):
            self.f("fourstep_intt_first_transpose")(_ptr(a), _ptr(b), prm["n1"], prm["n2"])
        return out

    def splitmix(self, seed, offset, count, q):
        out = np.empty(count, dtype=self.T)
        self.f("splitmix_fill")(ctypes.c_uint64(seed), ctypes.c_uint64(offset),
                                ctypes.c_uint64(count), self.c(q), _ptr(out))
        return out


class Ref:
    """The reference's own NTTParameters / NTTCPU / NTTParameters4Step /
    NTT_4STEP_CPU behind flat handles (oracle/ref_driver.cpp)."""

    def __init__(self, bits):
        assert bits in (32, 64)
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        self.bits = bits
        self.T = _np_t(bits)
        self.c = _c_t(bits)
        # lazy binding: CudaDevice()'s cuda* symbols are never called
        self.lib = ctypes.CDLL(REF_SO, mode=os.RTLD_LAZY)
        self.p = "ref%d_" % bits
        self.f("merge_create").restype = ctypes.c_void_p
        self.f("4step_create").restype = ctypes.c_void_p
        for name in ("mult", "exp", "modinv"):
            self.f(name).restype = self.c

    def f(self, name):
        return getattr(self.lib, self.p + name)

    def modulus(self, q):
        bit, mu = self.c(), self.c()
        self.f("modulus")(self.c(q), ctypes.byref(bit), ctypes.byref(mu))
        return int(q), int(bit.value), int(mu.value)

    def mult(self, a, b, q):
        return int(self.f("mult")(self.c(a), self.c(b), self.c(q)))

    def exp(self, a, e, q):
        return int(self.f("exp")(self.c(a), self.c(e), self.c(q)))

    def modinv(self, a, q):
        return int(self.f("modinv")(self.c(a), self.c(q)))

    def merge_params(self, logn, poly, factors=None):
        q, w, p = factors if factors else (0, 0, 0)
        h = ctypes.c_void_p(self.f("merge_create")(logn, poly, 1 if factors else 0, self.c(q),
                                                   self.c(w), self.c(p)))
        info = (ctypes.c_uint64 * 10)()
        self.f("merge_info")(h, info)
        size = int(info[8])
        prm = dict(handle=h, logn=logn, n=int(info[9]), poly=poly,
                   mod=(int(info[0]), int(info[1]), int(info[2])), omega=int(info[3]),
                   psi=int(info[4]), n_inv=int(info[5]), root=int(info[6]),
                   inv_root=int(info[7]), root_size=size)
        for which, tag in ((0, "fwd"), (1, "inv"), (2, "fwd_gpu"), (3, "inv_gpu")):
            t = np.empty(size, dtype=self.T)
            self.f("merge_table")(h, which, _ptr(t))
            prm[tag] = t
        return prm

    def merge_free(self, prm):
        self.f("merge_destroy")(prm["handle"])

    def merge_ntt(self, x, prm, inverse=False):
        x = np.ascontiguousarray(x, dtype=self.T)
        out = np.empty_like(x)
        self.f("merge_run")(prm["handle"], int(inverse), _ptr(x), _ptr(out),
                            x.size // prm["n"])
        return out

    def merge_ntt_mt(self, x, prm, inverse=False, nthreads=1):
        """the same transforms, the batch loop on `nthreads` OpenMP threads inside the reference build"""
        x = np.ascontiguousarray(x, dtype=self.T)
        out = np.empty_like(x)
        self.f("merge_run_mt")(prm["handle"], int(inverse), _ptr(x), _ptr(out), x.size // prm["n"], int(nthreads))
        return out

    def pointwise(self, a, b, prm):
        out = np.empty_like(a)
        self.f("pointwise")(prm["handle"], _ptr(a), _ptr(b), _ptr(out))
        return out

    def schoolbook(self, a, b, poly, q):
        out = np.empty_like(a)
        self.f("schoolbook")(_ptr(a), _ptr(b), _ptr(out), a.size, self.c(q), poly)
        return out

    def fourstep_params(self, logn):
        h = ctypes.c_void_p(self.f("4step_create")(logn))
        info = (ctypes.c_uint64 * 11)()
        self.f("4step_info")(h, info)
        n1, n2, n = int(info[8]), int(info[9]), int(info[10])
        prm = dict(handle=h, logn=logn, n=n, n1=n1, n2=n2,
                   mod=(int(info[0]), int(info[1]), int(info[2])), omega=int(info[3]),
                   psi=int(info[4]), n_inv=int(info[5]), root=int(info[6]),
                   inv_root=int(info[7]))
        sizes = {0: n1 >> 1, 1: n2 >> 1, 2: n, 3: n1 >> 1, 4: n2 >> 1, 5: n}
        names = {0: "n1_fwd", 1: "n2_fwd", 2: "W_fwd", 3: "n1_inv", 4: "n2_inv", 5: "W_inv"}
        for which in range(6):
            t = np.empty(sizes[which], dtype=self.T)
            self.f("4step_table")(h, which, _ptr(t))
            prm[names[which]] = t
            if which not in (2, 5):
                g = np.empty(sizes[which], dtype=self.T)
                self.f("4step_table")(h, which | 8, _ptr(g))
                prm[names[which] + "_gpu"] = g
        return prm

    def fourstep_free(self, prm):
        self.f("4step_destroy")(prm["handle"])

    def fourstep_run(self, x, prm, mode):
        """mode 0 ntt, 1 intt, 2 intt_first_transpose"""
        x = np.ascontiguousarray(x, dtype=self.T)
        out = np.empty_like(x)
        self.f("4step_run")(prm["handle"], mode, _ptr(x), _ptr(out), x.size // prm["n"])
        return out

    def fourstep_run_tables(self, x, prm, n1_table, n2_table, W, inverse=False, q=None, n_inv=None):
        """the reference's NTT_4STEP_CPU on caller-supplied natural-order tables (ref_driver.cpp: fourstep_run_tables)"""
        x = np.ascontiguousarray(x, dtype=self.T)
        out = np.empty_like(x)
        t1 = np.ascontiguousarray(n1_table[:prm["n1"] >> 1], dtype=self.T)
        t2 = np.ascontiguousarray(n2_table[:prm["n2"] >> 1], dtype=self.T)
        w = np.ascontiguousarray(W[:prm["n"]], dtype=self.T)
        self.f("4step_run_tables")(prm["logn"], int(inverse), _ptr(t1), _ptr(t2), _ptr(w), self.c(q or 0),
                                   self.c(prm["n_inv"] if n_inv is None else n_inv), _ptr(x), _ptr(out),
                                   x.size // prm["n"])
        return out

    def mt19937_uniform(self, seed, q, count):
        out = np.empty(count, dtype=np.uint64)
        self.lib.ref_mt19937_uniform(ctypes.c_uint32(seed), ctypes.c_uint64(q),
                                     ctypes.c_uint64(count), _ptr(out))
        return out
