#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE build (oracle/_ref/libgpuntt_ref.so).

Run in the build container (where /root/reference is mounted):
    make -C oracle ref && python tools/make_golden.py

Fixtures are data only (parameters, inputs' seeds, expected outputs, digests); the
inputs are regenerated at test time from the portable splitmix64 stream
    x[k] = splitmix64(seed ^ (offset + k)) mod q          (oracle/ntt_oracle.c)
so only seeds are stored.  Outputs come from the reference's own
NTTCPU<T>::ntt/intt and NTT_4STEP_CPU<T>::ntt/intt.

Files written:
  merge_u32.npz / merge_u64.npz   full forward+inverse outputs, logn in FULL_LOGN, both polys
  fourstep_u32.npz / _u64.npz     full 4-step forward+inverse outputs for logn 12, 13
  digests.json                    SHA-256 of tables/outputs for the large sizes + the
                                  mt19937(0) known answers of example/ntt_merge/test_merge_ntt.cu
  rns_c5.json                     the 8-prime RNS set of BASELINE config 5 with per-prime digests
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FULL_LOGN = (1, 2, 3, 5, 9, 10, 11, 12)
DIGEST_LOGN = (14, 16, 17, 20)
FOURSTEP_FULL = (12, 13)
FOURSTEP_DIGEST = (14, 15, 16, 17, 19, 20, 21, 22, 23, 24)
SEED = 0x5EED0000

# BASELINE config 5: 8 distinct ~60-bit primes from the reference's own pools
# (nttparameters.cu:84-142 merge pool; :244-299 4-step pools), each with its pool psi and
# the pool's logN, from which psi_16 = psi_pool^(2^(logN_pool-16)) (order 2^17).
C5_PRIMES = [
    # (q, psi_pool, logN_pool)
    (576460756061519873, 4517306222, 28),
    (576460752308273153, 3760097055997, 16),
    (576460752315482113, 328867687796, 18),
    (576460752340123649, 2298846063117, 19),
    (576460752364240897, 731868219707, 20),
    (576460752475389953, 409596963254, 21),
    (576460752597024769, 189266227206, 22),
    (576460753024843777, 31864818375, 23),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fourstep_record(R, P, bits, logn):
    """digest record of one 4-step shape from the reference build (+ the full outputs)"""
    prm = R.fourstep_params(logn)
    q = prm["mod"][0]
    seed = SEED + 1000 + logn * 2 + (bits == 32)
    x = P.splitmix(seed, 0, prm["n"], q)
    fwd = R.fourstep_run(x, prm, 0)
    inv = R.fourstep_run(x, prm, 1)
    ft = R.fourstep_run(x, prm, 2)
    rec = dict(bits=bits, logn=logn, seed=seed, q=q, bit=prm["mod"][1],
               mu=prm["mod"][2], omega=prm["omega"], psi=prm["psi"],
               n_inv=prm["n_inv"], n1=prm["n1"], n2=prm["n2"], sha_in=sha(x),
               sha_W_fwd=sha(prm["W_fwd"]), sha_W_inv=sha(prm["W_inv"]),
               sha_n1_fwd_gpu=sha(prm["n1_fwd_gpu"]),
               sha_n2_fwd_gpu=sha(prm["n2_fwd_gpu"]),
               sha_n1_inv_gpu=sha(prm["n1_inv_gpu"]),
               sha_n2_inv_gpu=sha(prm["n2_inv_gpu"]), sha_fwd=sha(fwd),
               sha_inv=sha(inv), sha_first_transpose=sha(ft))
    R.fourstep_free(prm)
    return rec, fwd, inv


def add_fourstep(shapes):
    """`make_golden.py --add-fourstep 64:19 64:21 ...`: append digest records of further 4-step shapes to
    digests.json without regenerating the rest (2^24 parameter generation alone takes minutes)"""
    path = os.path.join(OUT, "digests.json")
    digests = json.load(open(path))
    have = {(r["bits"], r["logn"]) for r in digests["fourstep"]}
    for item in shapes:
        bits, logn = (int(v) for v in item.split(":"))
        if (bits, logn) in have:
            continue
        rec, _, _ = fourstep_record(O.Ref(bits), O.Port(bits), bits, logn)
        digests["fourstep"].append(rec)
        print("4step", bits, logn, "added", flush=True)
    digests["fourstep"].sort(key=lambda r: (r["bits"], r["logn"]))
    with open(path, "w") as f:
        json.dump(digests, f, indent=1)


def add_merge(shapes):
    """`make_golden.py --add-merge 64:24 32:24`: append Merge digest records (both reduction polynomials) of further
    ring sizes to digests.json without regenerating the rest (SURVEY.md 8c asks for 2^24)"""
    path = os.path.join(OUT, "digests.json")
    digests = json.load(open(path))
    have = {(r["bits"], r["logn"], r["poly"]) for r in digests["merge"]}
    for item in shapes:
        bits, logn = (int(v) for v in item.split(":"))
        R, P = O.Ref(bits), O.Port(bits)
        for poly in (O.X_N_plus, O.X_N_minus):
            if (bits, logn, poly) in have:
                continue
            prm = R.merge_params(logn, poly)
            q = prm["mod"][0]
            seed = SEED + logn * 4 + poly * 2 + (bits == 32)
            x = P.splitmix(seed, 0, prm["n"], q)
            fwd = R.merge_ntt(x, prm, False)
            inv = R.merge_ntt(x, prm, True)
            digests["merge"].append(dict(bits=bits, poly=poly, logn=logn, batch=1, seed=seed, q=q, bit=prm["mod"][1],
                                         mu=prm["mod"][2], omega=prm["omega"], psi=prm["psi"], n_inv=prm["n_inv"],
                                         root=prm["root"], inv_root=prm["inv_root"], sha_in=sha(x),
                                         sha_fwd_gpu_table=sha(prm["fwd_gpu"]), sha_inv_gpu_table=sha(prm["inv_gpu"]),
                                         sha_fwd=sha(fwd), sha_inv=sha(inv)))
            R.merge_free(prm)
            print("merge", bits, logn, poly, "added", flush=True)
    with open(path, "w") as f:
        json.dump(digests, f, indent=1)


def add_c3(count=8):
    """`make_golden.py --add-c3`: BASELINE config 3's own call (u64, 2^24) on `count` distinct polynomials -- digests of
    NTT_4STEP_CPU::ntt / ::intt from the reference build for seeds seed0 + 7 i (polynomial 0 is the existing 2^24
    record), written to tests/golden/c3_polys.json"""
    R, P = O.Ref(64), O.Port(64)
    prm = R.fourstep_params(24)
    q = prm["mod"][0]
    seed0 = SEED + 1000 + 24 * 2
    out = dict(bits=64, logn=24, q=q, n1=prm["n1"], n2=prm["n2"], seed0=seed0, stride=7, polys=[])
    for i in range(count):
        x = P.splitmix(seed0 + 7 * i, 0, prm["n"], q)
        out["polys"].append(dict(seed=seed0 + 7 * i, sha_in=sha(x), sha_fwd=sha(R.fourstep_run(x, prm, 0)),
                                 sha_inv=sha(R.fourstep_run(x, prm, 1))))
        print("c3 poly", i, "done", flush=True)
    R.fourstep_free(prm)
    with open(os.path.join(OUT, "c3_polys.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--add-fourstep":
        return add_fourstep(sys.argv[2:])
    if len(sys.argv) > 2 and sys.argv[1] == "--add-merge":
        return add_merge(sys.argv[2:])
    if len(sys.argv) > 1 and sys.argv[1] == "--add-c3":
        return add_c3()
    os.makedirs(OUT, exist_ok=True)
    digests = {"seed": SEED, "merge": [], "fourstep": [], "mt19937": []}
    for bits in (32, 64):
        R, P = O.Ref(bits), O.Port(bits)
        store = {}
        for poly in (O.X_N_plus, O.X_N_minus):
            for logn in FULL_LOGN + DIGEST_LOGN:
                prm = R.merge_params(logn, poly)
                q = prm["mod"][0]
                batch = 2 if logn <= 12 else 1
                seed = SEED + logn * 4 + poly * 2 + (bits == 32)
                x = P.splitmix(seed, 0, batch * prm["n"], q)
                fwd = R.merge_ntt(x, prm, False)
                inv = R.merge_ntt(x, prm, True)  # INTT applied to the same raw input
                rec = dict(bits=bits, poly=poly, logn=logn, batch=batch, seed=seed,
                           q=q, bit=prm["mod"][1], mu=prm["mod"][2], omega=prm["omega"],
                           psi=prm["psi"], n_inv=prm["n_inv"], root=prm["root"],
                           inv_root=prm["inv_root"], sha_in=sha(x),
                           sha_fwd_gpu_table=sha(prm["fwd_gpu"]),
                           sha_inv_gpu_table=sha(prm["inv_gpu"]), sha_fwd=sha(fwd),
                           sha_inv=sha(inv))
                digests["merge"].append(rec)
                if logn in FULL_LOGN:
                    key = "p%d_l%d" % (poly, logn)
                    store[key + "_fwd"] = fwd
                    store[key + "_inv"] = inv
                    store[key + "_tabf"] = prm["fwd_gpu"][:64]
                    store[key + "_tabi"] = prm["inv_gpu"][:64]
                R.merge_free(prm)
        np.savez(os.path.join(OUT, "merge_u%d.npz" % bits), **store)

        store = {}
        for logn in FOURSTEP_FULL + FOURSTEP_DIGEST:
            if bits == 32 and logn == 24:
                continue  # one 2^24 parameter generation (u64) is enough wall-clock
            rec, fwd, inv = fourstep_record(R, P, bits, logn)
            digests["fourstep"].append(rec)
            if logn in FOURSTEP_FULL:
                store["l%d_fwd" % logn] = fwd
                store["l%d_inv" % logn] = inv
            print("4step", bits, logn, "done", flush=True)
        np.savez(os.path.join(OUT, "fourstep_u%d.npz" % bits), **store)

    # known answers on the reference examples' own deterministic stream
    # (test_merge_ntt.cu:70-96: mt19937(0), uniform_int_distribution<uint64_t>(0,q-1), X^N-1)
    R = O.Ref(64)
    for logn in (5, 12, 16):
        prm = R.merge_params(logn, O.X_N_minus)
        x = R.mt19937_uniform(0, prm["mod"][0], prm["n"])
        y = R.merge_ntt(x, prm)
        digests["mt19937"].append(dict(logn=logn, q=prm["mod"][0],
                                       first_in=[int(v) for v in x[:4]],
                                       first_out=[int(v) for v in y[:4]], sha_in=sha(x),
                                       sha_out=sha(y)))
        R.merge_free(prm)
    with open(os.path.join(OUT, "digests.json"), "w") as f:
        json.dump(digests, f, indent=1)

    # config 5 RNS set, negacyclic and cyclic, one polynomial per prime
    rns = {"logn": 16, "primes": []}
    P = O.Port(64)
    for i, (q, psi_pool, lg_pool) in enumerate(C5_PRIMES):
        psi = pow(psi_pool, 1 << (lg_pool - 16), q)
        omega = psi * psi % q
        assert pow(psi, 1 << 16, q) == q - 1 and pow(omega, 1 << 15, q) == q - 1
        ent = dict(q=q, psi=psi, omega=omega)
        for poly in (O.X_N_plus, O.X_N_minus):
            prm = R.merge_params(16, poly, (q, omega, psi))
            seed = SEED + 5000 + i * 2 + poly
            x = P.splitmix(seed, 0, prm["n"], q)
            tag = "plus" if poly == O.X_N_plus else "minus"
            ent["seed_" + tag] = seed
            ent["n_inv"] = prm["n_inv"]
            ent["sha_fwd_" + tag] = sha(R.merge_ntt(x, prm, False))
            ent["sha_inv_" + tag] = sha(R.merge_ntt(x, prm, True))
            ent["sha_tab_" + tag] = sha(prm["fwd_gpu"])
            R.merge_free(prm)
        rns["primes"].append(ent)
    with open(os.path.join(OUT, "rns_c5.json"), "w") as f:
        json.dump(rns, f, indent=1)
    print("golden written to", OUT)


if __name__ == "__main__":
    main()
