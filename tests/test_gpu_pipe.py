"""-m gpu parity tests of the persistent, software-pipelined contiguous pass (gpu-ntt_amd/csrc/
merge_pipe_kernels.hpp): the second pass of forward and the first pass of inverse 64-bit transforms of
2^13 .. 2^16 coefficients.  Opt-in (GPUNTT_PIPE=1): measured slower than the default one-tile-per-workgroup
kernels (profiles/r02_pipelined_contig_pass.md), kept bit-exact as the record of the experiment."""
import pytest

from test_gpu_merge import _run_in_subprocess, g  # noqa: F401  (module fixture: library built and loaded)

pytestmark = pytest.mark.gpu


_FORCED = r'''
import numpy as np, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.environ["PYTHONPATH"]), ""))
from conftest import load_pkg
from gpu_utils import MergeCase
from oracle import oracle as O
import test_gpu_merge as tm
g = load_pkg(); g.load_library()
# single modulus: every ring size of the pipelined pass, batches below, at and above the workgroups' stride
for logn, batches in ((13, (1, 7)), (14, (3, 130)), (15, (1, 65)), (16, (1, 5, 33, 67))):
    for poly in (O.X_N_minus, O.X_N_plus):
        c = MergeCase(g, 64, logn, poly)
        for batch in batches:
            x = c.random(batch, 31 * logn + batch)
            want = c.P.merge_ntt(x, c.oprm)
            assert np.array_equal(c.gpu_forward(x, inplace=bool(batch & 1)), want), ("fwd", logn, poly, batch)
            assert np.array_equal(c.gpu_inverse(want, inplace=not (batch & 1)), x), ("inv", logn, poly, batch)
            assert np.array_equal(c.gpu_inverse(x), c.P.merge_ntt(x, c.oprm, inverse=True)), ("inv raw", logn, batch)
# RNS stacks: the workgroups' polynomial stride is rounded to a multiple of mod_count
P = O.Port(64)
for logn, batch, mc in ((14, 9, 3), (16, 70, 3), (16, 21, 7), (15, 40, 8)):
    for poly in (O.X_N_plus, O.X_N_minus):
        cases, fwd, inv, mods, ninv = tm._rns_setup(g, 64, logn, poly, tm._small_prime_factors(P, logn, mc))
        n = 1 << logn
        x = np.concatenate([cases[p % mc].P.splitmix(50 + p, 0, n, cases[p % mc].q) for p in range(batch)])
        want = np.concatenate([cases[p % mc].P.merge_ntt(x[p * n:(p + 1) * n], cases[p % mc].oprm)
                               for p in range(batch)])
        d = g.to_device(x)
        g.GPU_NTT_Inplace(d, fwd, mods, g.ntt_rns_configuration(n_power=logn, reduction_poly=poly), batch, mc)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), want), ("rns fwd", logn, mc, poly)
        icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
        g.GPU_INTT_Inplace(d, inv, mods, icfg, batch, mc)
        torch.cuda.synchronize()
        assert np.array_equal(g.to_host(d), x), ("rns inv", logn, mc, poly)
# modulus-ordered entry point (prime of slot mi is order[mi])
logn, poly = 15, O.X_N_plus
n = 1 << logn
cases, fwd, inv, mods, ninv = tm._prime_stack(g, 64, logn, poly, 6)
order = [1, 3, 4]
d_order = torch.tensor(order, dtype=torch.int32, device="cuda")
batch, mc = 9, 3
prime_of = [order[p % mc] for p in range(batch)]
x = np.concatenate([cases[pr].P.splitmix(300 + p, 0, n, cases[pr].q) for p, pr in enumerate(prime_of)])
want = np.concatenate([cases[pr].P.merge_ntt(x[p * n:(p + 1) * n], cases[pr].oprm) for p, pr in enumerate(prime_of)])
d = g.to_device(x)
o = torch.zeros_like(d)
g.GPU_NTT_Modulus_Ordered(d, o, fwd, mods, g.ntt_rns_configuration(n_power=logn, reduction_poly=poly), batch, mc, d_order)
torch.cuda.synchronize()
assert np.array_equal(g.to_host(o), want), "mod-ordered fwd"
icfg = g.ntt_rns_configuration(n_power=logn, ntt_type=g.INVERSE, reduction_poly=poly, mod_inverse=ninv)
g.GPU_NTT_Modulus_Ordered(o, o, inv, mods, icfg, batch, mc, d_order)
torch.cuda.synchronize()
assert np.array_equal(g.to_host(o), x), "mod-ordered inv"
print("pipe forced OK")
'''


@pytest.mark.parametrize("big_tiles", ["14", "0"])
def test_forced_for_every_batch(g, big_tiles):
    """GPUNTT_PIPE=1: the pipelined pass serves every batch size (workgroups with a single tile, ragged
    rounds), RNS stacks and the modulus-ordered entry point; GPUNTT_U64_BIG_TILES=0 puts 2^13 and 2^14 on
    4096-coefficient tiles so that they take it as well"""
    out = _run_in_subprocess(_FORCED, {"GPUNTT_PIPE": "1", "GPUNTT_U64_BIG_TILES": big_tiles})
    assert "pipe forced OK" in out
