"""The dispatch table of the drop-in entry points as DATA: which kernels a call enqueues, and who serves it, for every
(entry point x overload x modulus class x table state x prediction x path) the host code distinguishes.

Mirrored in DESIGN.md section 3.8 (`python tests/dispatch_rows.py --markdown` prints that table from this list) and walked
row by row by tests/test_gpu_dispatch_table.py, which compares the library's launch log (csrc/test_hooks.h) with `launches`
and the result with the oracle.  Reference counterpart of the decision: the reference's entry points are launches and
nothing else (src/lib/ntt_merge/ntt.cu:2076-2256, src/lib/ntt_4step/ntt_4step.cu:2293-3229) -- one kernel family, one
arithmetic; everything below exists because this library has two arithmetic families and keeps the moduli / tables of a
drop-in call on the device.

Row fields
  entry     merge | merge_rns | merge_ordered | 4step | 4step_rns            (C ABI entry the Python harness calls)
  bits, logn, batch, inverse
  modulus   host-side modulus (merge, 4step): "pool" (the reference's pool prime of the ring: <= 60 bit, 31 q < 2^64;
            32-bit: < 2^29), "b61" / "b62" (searched primes), "b30" (32-bit, 30 bits), "tiny" (q = 2: outside the fast
            kernels' domain)
  stack     device-side moduli (RNS overloads): list of widths, e.g. [60, 60], [60, 61], [62, 60]; "tiny" adds q = 2
  tables    4-step: "ok" | "vetoed" (one W word changed)
  predict   RNS overloads: "unknown" (moduli buffer never seen, no stack of this shape seen either), "right" (third call
            of the same buffer), "wrong" (the buffer held a [60, 60] stack for three calls, then was rewritten in place)
  hooks     options / test hooks set for the call, e.g. {"path": "generic"}
  launches  kernels the call enqueues, in order (launch log; fast kernels carry their lazy range: 0 = the word size's
            default, 31 / 8 / 4)
  serves    which of them transforms the batch: "lazy:<range>" (a forward 4-step of a 31 q modulus gathers on the 16 q
            kernels and finishes on the 31 q ones: "lazy:0+31"), "e32:<range>", "generic", "prep" (the preparation
            kernel's own fall-back), "self" (the fast kernels' own element-by-element path, 4-step)
"""

ROWS = [
    # ------------------------------------------------------------------ Merge, host-side modulus (no preparation decisions)
    dict(id="merge u64 2^16 pool fwd", entry="merge", bits=64, logn=16, batch=4, modulus="pool",
         launches=["prep_twiddles", "merge_pass_lazy:31", "merge_pass_lazy:31"], serves="lazy:31"),
    dict(id="merge u64 2^16 pool inv", entry="merge", bits=64, logn=16, batch=4, modulus="pool", inverse=True,
         launches=["prep_twiddles", "merge_pass_lazy:0", "merge_pass_lazy:0"], serves="lazy:0"),
    dict(id="merge u64 2^16 61-bit fwd", entry="merge", bits=64, logn=16, batch=4, modulus="b61",
         launches=["prep_twiddles", "merge_pass_lazy:8", "merge_pass_lazy:8"], serves="lazy:8"),
    dict(id="merge u64 2^16 62-bit fwd", entry="merge", bits=64, logn=16, batch=4, modulus="b62",
         launches=["prep_twiddles", "merge_pass_lazy:4", "merge_pass_lazy:4"], serves="lazy:4"),
    dict(id="merge u64 2^16 q=2 (out of domain)", entry="merge", bits=64, logn=16, batch=4, modulus="tiny",
         launches=["merge_pass", "merge_pass"], serves="generic"),
    dict(id="merge u64 2^16 pool, path=generic", entry="merge", bits=64, logn=16, batch=4, modulus="pool",
         hooks={"path": "generic"}, launches=["merge_pass", "merge_pass"], serves="generic"),
    dict(id="merge u64 2^12 pool fwd (one tile)", entry="merge", bits=64, logn=12, batch=4, modulus="pool",
         launches=["prep_twiddles", "merge_pass_lazy:31"], serves="lazy:31"),
    dict(id="merge u64 2^3 pool (tiny job: size heuristic)", entry="merge", bits=64, logn=3, batch=2, modulus="pool",
         launches=["merge_pass"], serves="generic"),
    dict(id="merge u32 2^14 pool fwd (32 coefficients per lane)", entry="merge", bits=32, logn=14, batch=4, modulus="pool",
         launches=["prep_twiddles", "merge_ring_e32:8"], serves="e32:8"),
    dict(id="merge u32 2^14 30-bit fwd", entry="merge", bits=32, logn=14, batch=4, modulus="b30",
         launches=["prep_twiddles", "merge_ring_e32:0"], serves="e32:0"),
    dict(id="merge u32 2^14 pool, u32_e32=0 (16 coefficients per lane)", entry="merge", bits=32, logn=14, batch=4, modulus="pool",
         hooks={"u32_e32": "0"}, launches=["prep_twiddles", "merge_pass_lazy:8"], serves="lazy:8"),
    dict(id="merge u32 2^15 pool inv (one sweep)", entry="merge", bits=32, logn=15, batch=4, modulus="pool", inverse=True,
         launches=["prep_twiddles", "merge_ring_e32:8"], serves="e32:8"),
    dict(id="merge u32 2^16 pool fwd", entry="merge", bits=32, logn=16, batch=4, modulus="pool",
         launches=["prep_twiddles", "merge_pass_lazy:8", "merge_pass_lazy:8"], serves="lazy:8"),
    dict(id="merge u32 2^20 pool fwd (contiguous pass on 32 coefficients per lane)", entry="merge", bits=32, logn=20, batch=2,
         modulus="pool", launches=["prep_twiddles", "merge_pass_lazy:8", "merge_ring_e32:8"], serves="lazy:8+e32:8"),
    dict(id="merge u32 2^23 pool inv (two sweeps)", entry="merge", bits=32, logn=23, batch=2, modulus="pool", inverse=True,
         launches=["prep_twiddles", "merge_ring_e32:8", "merge_pass_lazy:8"], serves="e32:8+lazy:8"),
    dict(id="merge u64 2^16 pool, no_scratch", entry="merge", bits=64, logn=16, batch=4, modulus="pool",
         hooks={"no_scratch": "1"}, launches=["merge_pass", "merge_pass"], serves="generic"),
    # ------------------------------------------------------------------ Merge, RNS overload (moduli on the device)
    dict(id="rns u64 2^13 [60,60] unknown", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, 60], predict="unknown",
         launches=["prep_twiddles", "merge_pass_lazy:0"], serves="lazy:0",
         note="a stack nothing is known about starts on the default 16 q family, which serves every stack of <= 60-bit primes"),
    dict(id="rns u64 2^13 [60,60] right", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, 60], predict="right",
         launches=["prep_twiddles", "merge_pass_lazy:0"], serves="lazy:0",
         note="the stack would also fit the 31 q family; the prediction narrows to it after 16 calls in a row that needed less"),
    dict(id="rns u64 2^13 [60,60] right inv", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, 60], predict="right",
         inverse=True, launches=["prep_twiddles", "merge_pass_lazy:0"], serves="lazy:0"),
    dict(id="rns u64 2^13 [60,61] right", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, 61], predict="right",
         launches=["prep_twiddles", "merge_pass_lazy:8", "merge_pass_lazy:8"], serves="lazy:8"),
    dict(id="rns u64 2^13 [62,60] right", entry="merge_rns", bits=64, logn=13, batch=4, stack=[62, 60], predict="right",
         launches=["prep_twiddles", "merge_pass_lazy:4", "merge_pass_lazy:4"], serves="lazy:4"),
    dict(id="rns u64 2^13 [60,61] wrong (rewritten in place)", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, 61],
         predict="wrong", launches=["prep_twiddles", "merge_pass_lazy:0"], serves="prep"),
    dict(id="rns u64 2^13 [60,tiny] right (out of domain)", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, "tiny"],
         predict="right", launches=["prep_twiddles", "merge_pass_lazy:0"], serves="prep",
         note="an out-of-domain state is remembered as 'unsure', never as a family: small rings keep the in-preparation "
              "fall-back"),
    dict(id="rns u64 2^17 [60,61] unknown (wide net)", entry="merge_rns", bits=64, logn=17, batch=4, stack=[60, 61],
         predict="unknown",
         launches=["prep_twiddles", "merge_pass_lazy:0", "merge_pass_lazy:0", "merge_pass_lazy:31", "merge_pass_lazy:31", "merge_pass_lazy:8", "merge_pass_lazy:8", "merge_pass_lazy:4", "merge_pass_lazy:4", "merge_pass", "merge_pass"],
         serves="lazy:8"),
    dict(id="rns u64 2^17 [60,61] right", entry="merge_rns", bits=64, logn=17, batch=4, stack=[60, 61], predict="right",
         launches=["prep_twiddles", "merge_pass_lazy:8", "merge_pass_lazy:8"], serves="lazy:8"),
    dict(id="rns u64 2^17 [60,tiny] right (out of domain, large ring)", entry="merge_rns", bits=64, logn=17, batch=4,
         stack=[60, "tiny"], predict="right",
         launches=["prep_twiddles", "merge_pass_lazy:0", "merge_pass_lazy:0", "merge_pass_lazy:31", "merge_pass_lazy:31", "merge_pass_lazy:8", "merge_pass_lazy:8", "merge_pass_lazy:4", "merge_pass_lazy:4", "merge_pass", "merge_pass"],
         serves="generic"),
    dict(id="rns u64 2^13 [60,60], rns_predict=0 (every family)", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, 60],
         predict="right", hooks={"rns_predict": "0"},
         launches=["prep_twiddles", "merge_pass_lazy:0", "merge_pass_lazy:31", "merge_pass_lazy:8", "merge_pass_lazy:8", "merge_pass_lazy:4", "merge_pass_lazy:4"],
         serves="lazy:31"),
    dict(id="rns u64 2^13 [60,60], path=generic", entry="merge_rns", bits=64, logn=13, batch=4, stack=[60, 60], predict="right",
         hooks={"path": "generic"}, launches=["merge_pass", "merge_pass"], serves="generic"),
    dict(id="rns u64 2^8 [60,61] right (per-lane moduli)", entry="merge_rns", bits=64, logn=8, batch=64, stack=[60, 61],
         predict="right", launches=["prep_twiddles", "merge_pass_lazy_vqc:8"], serves="lazy:8"),
    dict(id="rns u32 2^14 [29,30] right", entry="merge_rns", bits=32, logn=14, batch=4, stack=[29, 30], predict="right",
         launches=["prep_twiddles", "merge_ring_e32:0"], serves="e32:0"),
    dict(id="ordered u64 2^13 [60,61] right", entry="merge_ordered", bits=64, logn=13, batch=4, stack=[60, 61], predict="right",
         launches=["prep_twiddles", "merge_pass_lazy:8", "merge_pass_lazy:8"], serves="lazy:8"),
    # ------------------------------------------------------------------ 4-step, host-side modulus
    dict(id="4step u64 2^12 ok", entry="4step", bits=64, logn=12, batch=2, modulus="pool", tables="ok",
         launches=["prep_merge_from_fourstep", "fourstep_small_lazy:0"], serves="lazy:0"),
    dict(id="4step u64 2^12 vetoed", entry="4step", bits=64, logn=12, batch=2, modulus="pool", tables="vetoed",
         launches=["prep_merge_from_fourstep", "fourstep_small_lazy:0"], serves="self"),
    dict(id="4step u64 2^16 ok fwd", entry="4step", bits=64, logn=16, batch=2, modulus="pool", tables="ok",
         launches=["prep_merge_from_fourstep", "fourstep_first_lazy:0", "merge_pass_lazy:31"], serves="lazy:0+31"),
    dict(id="4step u64 2^16 vetoed fwd", entry="4step", bits=64, logn=16, batch=2, modulus="pool", tables="vetoed",
         launches=["prep_merge_from_fourstep", "fourstep_first_lazy:0", "merge_pass_lazy:31"], serves="self"),
    dict(id="4step u64 2^16 vetoed inv", entry="4step", bits=64, logn=16, batch=2, modulus="pool", tables="vetoed", inverse=True,
         launches=["prep_merge_from_fourstep", "fourstep_inv_first_lazy:0", "merge_pass_lazy:0"], serves="self"),
    dict(id="4step u64 2^20 ok fwd", entry="4step", bits=64, logn=20, batch=2, modulus="pool", tables="ok",
         launches=["prep_merge_from_fourstep", "fourstep_first_lazy:0", "merge_pass_lazy:31", "merge_pass", "merge_pass"],
         serves="lazy:0+31"),
    dict(id="4step u64 2^20 vetoed fwd", entry="4step", bits=64, logn=20, batch=2, modulus="pool", tables="vetoed",
         launches=["prep_merge_from_fourstep", "fourstep_first_lazy:0", "merge_pass_lazy:31", "merge_pass", "merge_pass"],
         serves="generic"),
    dict(id="4step u64 2^20 ok fwd, check_4step_tables=0", entry="4step", bits=64, logn=20, batch=2, modulus="pool", tables="ok",
         hooks={"check_4step_tables": "0"},
         launches=["prep_merge_from_fourstep", "fourstep_first_lazy:0", "merge_pass_lazy:31"], serves="lazy:0+31"),
    dict(id="4step u32 2^20 ok fwd (last pass on 32 coefficients per lane)", entry="4step", bits=32, logn=20, batch=2,
         modulus="pool", tables="ok",
         launches=["prep_merge_from_fourstep", "fourstep_first_lazy:0", "merge_ring_e32:8", "merge_pass", "merge_pass"],
         serves="lazy:0+e32:8"),
    dict(id="4step u64 2^16 61-bit ok fwd", entry="4step", bits=64, logn=16, batch=2, modulus="b61", tables="ok",
         launches=["prep_merge_from_fourstep", "fourstep_first_lazy:8", "merge_pass_lazy:8"], serves="lazy:8"),
    dict(id="4step u64 2^16 ok, path=generic", entry="4step", bits=64, logn=16, batch=2, modulus="pool", tables="ok",
         hooks={"path": "generic"}, launches=["merge_pass", "merge_pass"], serves="generic"),
    # ------------------------------------------------------------------ 4-step, RNS overload
    dict(id="4step_rns u64 2^12 [60] right ok", entry="4step_rns", bits=64, logn=12, batch=2, stack=[60], predict="right",
         tables="ok", launches=["prep_merge_from_fourstep", "fourstep_small_lazy:0"], serves="lazy:0"),
    dict(id="4step_rns u64 2^12 [60] right vetoed", entry="4step_rns", bits=64, logn=12, batch=2, stack=[60], predict="right",
         tables="vetoed", launches=["prep_merge_from_fourstep", "fourstep_small_lazy:0"], serves="self"),
    dict(id="4step_rns u64 2^12 [60,60] (mod_count 2: shared tables)", entry="4step_rns", bits=64, logn=12, batch=2,
         stack=[60, 60], predict="right", tables="ok", launches=["merge_pass", "merge_pass"], serves="generic"),
]


SHORT = {"prep_twiddles": "prep", "prep_merge_from_fourstep": "prep4", "merge_pass_lazy": "lazy", "merge_ring_e32": "e32",
         "merge_pass_lazy_vqc": "vqc", "merge_pass": "generic", "fourstep_small_lazy": "fs_small", "fourstep_first_lazy": "fs_first",
         "fourstep_inv_first_lazy": "fs_inv_first"}


def short(kernels):
    """launch list in the table's shorthand: kernel template names abbreviated (SHORT), repeats as x2"""
    out = []
    for k in kernels:
        name, _, fam = k.partition(":")
        k = SHORT.get(name, name) + (":" + fam if fam else "")
        if out and out[-1][0] == k:
            out[-1][1] += 1
        else:
            out.append([k, 1])
    return " ".join("`%s`%s" % (k, "" if c == 1 else "x%d" % c) for k, c in out)


def markdown():
    out = ["| row | options / hooks | launches enqueued, in order | serves the call |", "|---|---|---|---|"]
    for r in ROWS:
        hooks = ", ".join("%s=%s" % kv for kv in r.get("hooks", {}).items()) or "-"
        out.append("| %s | %s | %s | %s |" % (r["id"], hooks, short(r["launches"]), r["serves"]))
    return "\n".join(out)


if __name__ == "__main__":
    import sys
    if "--markdown" in sys.argv:
        print(markdown())
