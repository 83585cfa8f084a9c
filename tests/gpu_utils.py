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
