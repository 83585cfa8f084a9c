"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the oracle)."""
import hashlib

import numpy as np

from oracle import oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class MergeCase:
    """Parameters + device tables for one (bits, logn, poly[, factors]) Merge configuration,
    built by the product's own host generator and cross-checked against the oracle's."""

    def __init__(self, g, bits, logn, poly, factors=None):
        self.g, self.bits, self.logn, self.poly = g, bits, logn, poly
        self.P = O.Port(bits)
        self.oprm = self.P.merge_params(logn, poly, factors)
        self.prm = g.NTTParameters(logn, poly, bits, factors)
        assert (self.prm.modulus.value, self.prm.modulus.bit, self.prm.modulus.mu) == self.oprm["mod"]
        assert np.array_equal(self.prm.forward_table_device_order, self.P.bitrev_table(self.oprm["fwd"]))
        self.n = 1 << logn
        self.q = self.oprm["mod"][0]
        self.fwd_dev = g.to_device(self.prm.forward_table_device_order)
        self.inv_dev = g.to_device(self.prm.inverse_table_device_order)

    def cfg(self, inverse=False, stream=None):
        return self.g.ntt_configuration(n_power=self.logn,
                                        ntt_type=self.g.INVERSE if inverse else self.g.FORWARD,
                                        reduction_poly=self.poly,
                                        mod_inverse=self.prm.n_inv if inverse else 0,
                                        stream=stream)

    def random(self, batch, seed):
        return self.P.splitmix(seed, 0, batch * self.n, self.q)

    def gpu_forward(self, x, inplace=False):
        import torch
        g = self.g
        batch = x.size // self.n
        d_in = g.to_device(x)
        if inplace:
            g.GPU_NTT_Inplace(d_in, self.fwd_dev, self.prm.modulus, self.cfg(), batch)
            d_out = d_in
        else:
            d_out = torch.zeros_like(d_in)
            g.GPU_NTT(d_in, d_out, self.fwd_dev, self.prm.modulus, self.cfg(), batch)
            torch.cuda.synchronize()
            assert np.array_equal(g.to_host(d_in), x), "out-of-place call modified its input"
        torch.cuda.synchronize()
        return g.to_host(d_out)

    def gpu_inverse(self, y, inplace=False):
        import torch
        g = self.g
        batch = y.size // self.n
        d_in = g.to_device(y)
        if inplace:
            g.GPU_INTT_Inplace(d_in, self.inv_dev, self.prm.modulus, self.cfg(True), batch)
            d_out = d_in
        else:
            d_out = torch.zeros_like(d_in)
            g.GPU_INTT(d_in, d_out, self.inv_dev, self.prm.modulus, self.cfg(True), batch)
        torch.cuda.synchronize()
        return g.to_host(d_out)


def oracle_batch(cases, x, inverse=False):
    """NTTCPU::ntt / ::intt of EVERY polynomial of a batch -- polynomial p under cases[p % len(cases)] -- through the
    oracle's OpenMP batch entry (oracle/ntt_oracle_impl.h: merge_batch) on all host threads; returns a new array."""
    import os
    c0 = cases[0]
    n = c0.n
    size = c0.oprm["root_size"]
    tables = np.concatenate([np.ascontiguousarray(c.oprm["inv" if inverse else "fwd"], dtype=c0.P.T) for c in cases])
    y = np.array(x, dtype=c0.P.T, copy=True, order="C")
    c0.P.merge_batch(y, c0.logn, c0.poly, inverse, tables, size, [c.q for c in cases], os.cpu_count() or 1)
    return y


def _is_probable_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 325, 9375, 28178, 450775, 9780504, 1795265022):
        a %= n
        if a == 0:
            continue
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def find_ntt_factors(bits, logn, skip=0, clear_of_top=False):
    """(q, omega, psi) for an NTT prime with exactly `bits` bits and 2^(logn+1) | q-1
    (psi of order 2N, omega = psi^2), found by search -- used to exercise moduli the
    reference's pools do not contain (61/62-bit: no lazy headroom).  clear_of_top: start 2^(bits-40) below
    2^bits -- closer than ~2^(bits-46) the double log2() behind Modulus<T>::bit rounds up to `bits` exactly
    and the modulus gets bit = bits + 1; for tiny rings the first primes found lie that close."""
    step = 1 << (logn + 1)
    top = (1 << bits) - 1 - ((1 << (bits - 40)) if (clear_of_top and bits > 40) else 0)
    q = top // step * step + 1
    found = 0
    while True:
        if q.bit_length() < bits:
            raise ValueError("no %d-bit prime = 1 mod 2^%d left (skip=%d)" % (bits, logn + 1, skip))
        if q.bit_length() == bits and _is_probable_prime(q):
            if found == skip:
                break
            found += 1
        q -= step
    g = 2
    while True:
        psi = pow(g, (q - 1) >> (logn + 1), q)
        if pow(psi, 1 << logn, q) == q - 1:
            return q, psi * psi % q, psi
        g += 1


def distinct_factors(widths, logn):
    seen, out = {}, []
    for b in widths:
        out.append(find_ntt_factors(b, logn, skip=seen.get(b, 0), clear_of_top=True))
        seen[b] = seen.get(b, 0) + 1
    return out


def distinct_factors_scaled(widths, logn):
    """distinct NTT primes of the given widths with (q, omega, psi) for a ring of 2^logn.  Searched for 2^max(logn, 12) and
    brought down by squaring psi: the topmost primes = 1 mod 2^(logn + 1) of a small ring lie within double rounding of
    the power of two, get an over-stated `bit` and (61 -> 62) a mu that does not fit the word (Modulus refuses them)."""
    out, seen = [], set()
    lg = max(logn, 12)
    for w in widths:
        skip = 0
        while True:
            q, _, psi = find_ntt_factors(w, lg, skip)
            if q not in seen:
                break
            skip += 1
        seen.add(q)
        psi = pow(psi, 1 << (lg - logn), q)
        assert pow(psi, 1 << logn, q) == q - 1
        out.append((q, psi * psi % q, psi))
    return out


def rns_stack(g, bits, logn, widths, poly):
    cases = [MergeCase(g, bits, logn, poly, f) for f in distinct_factors(widths, logn)]
    n, mc = 1 << logn, len(cases)
    fwd = np.zeros(mc * n, dtype=cases[0].P.T)
    inv = np.zeros_like(fwd)
    for i, c in enumerate(cases):
        fwd[i * n:i * n + c.prm.root_of_unity_size] = c.prm.forward_table_device_order
        inv[i * n:i * n + c.prm.root_of_unity_size] = c.prm.inverse_table_device_order
    return cases, g.to_device(fwd), g.to_device(inv)


def cpu_class_on_tables(P, oprm, tabs, x_dev_layout, batch, inverse, n_inv):
    """What the reference's CPU class makes of a GPU_4STEP_NTT call with these (device-order) tables: NTT_4STEP_CPU::ntt /
    ::intt on the same tables in natural order (oracle: Port.fourstep_ntt_tables, pinned to the reference build with its
    public table vectors overwritten -- tests/test_oracle_vs_reference.py), between the example programs' transposes
    (test_4step_ntt.cu:90-178, test_4step_intt.cu:81-179), returned in the layout the GPU call writes (n1 x n2)."""
    n, n1, n2 = oprm["n"], oprm["n1"], oprm["n2"]
    t1 = P.bitrev_table(np.ascontiguousarray(tabs[0][:n1 >> 1]))  # the kernels read the first n1/2 (n2/2) words
    t2 = P.bitrev_table(np.ascontiguousarray(tabs[1][:n2 >> 1]))
    out = np.empty_like(x_dev_layout)
    for p in range(batch):
        a = x_dev_layout[p * n:(p + 1) * n]
        # forward: the call reads the n2 x n1 transpose of the natural-order polynomial; inverse: what
        # intt_first_transpose made of the spectrum
        nat = np.ascontiguousarray(a.reshape(n1, n2).T if inverse else a.reshape(n2, n1).T).reshape(-1)
        r = P.fourstep_ntt_tables(nat, oprm, t1, t2, tabs[2], inverse, n_inv=n_inv if inverse else None)
        out[p * n:(p + 1) * n] = np.ascontiguousarray(r.reshape(n2, n1).T).reshape(-1)  # undo the closing GPU_Transpose
    return out
