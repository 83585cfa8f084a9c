#!/usr/bin/env python3
"""Static instruction budget of one kernel from hipcc --save-temps assembly.

    python tools/isa_budget.py <file.s> <demangled-name substring> [--top N]

Prints the register / LDS footprint and the instruction histogram of the first kernel whose demangled
name contains the substring, with VALU / SALU / LDS / VMEM / s_nop totals.  The count is STATIC (every
path of the kernel); the executed count per wave is the PMC figure SQ_INSTS_VALU / SQ_WAVES
(tools/pmc_sq.sh).  profiles/*_isa_budget.txt are outputs of this script."""
import collections
import subprocess
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    lines = open(path).read().split("\n")
    labels = [(i, l.split(":")[0]) for i, l in enumerate(lines)
              if l.startswith("_Z") and ":" in l and not l.startswith("_Z") is False and "@" in l]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in labels), capture_output=True,
                           text=True).stdout.split("\n")
    start = None
    for (i, m), d in zip(labels, names):
        if key in d:
            start, mangled, dem = i, m, d
            break
    if start is None:
        raise SystemExit("kernel not found: " + key)
    end = next(j for j in range(start, len(lines)) if lines[j].startswith(".Lfunc_end"))
    body = lines[start:end]
    ops = collections.Counter(l.split()[0] for l in body if l.startswith("\t") and not l.startswith("\t.")
                              and not l.startswith("\t;"))
    meta = {}
    for l in lines[end:end + 400]:
        for k in ("; NumVgprs:", "; NumSgprs:", "; LDSByteSize:", "; ScratchSize:", "; Occupancy:"):
            if l.startswith(k) and k not in meta:
                meta[k] = l[len(k):].split()[0]
        if l.startswith("; Occupancy:"):
            break
    cls = collections.Counter()
    for op, n in ops.items():
        c = ("VALU" if op.startswith("v_") else "s_nop" if op == "s_nop" else "s_waitcnt" if op == "s_waitcnt" else
             "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else
             "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        cls[c] += n
    print("kernel:", dem)
    print("vgpr %s  sgpr %s  lds %s B  scratch %s B  occupancy %s waves/SIMD" % tuple(
        meta.get(k) for k in ("; NumVgprs:", "; NumSgprs:", "; LDSByteSize:", "; ScratchSize:", "; Occupancy:")))
    print("static instructions: %d  " % sum(ops.values()) + "  ".join("%s %d" % kv for kv in sorted(cls.items())))
    for op, n in ops.most_common(top):
        print("%6d  %s" % (n, op))


if __name__ == "__main__":
    main()
