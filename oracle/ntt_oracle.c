/*
 * ntt_oracle.c -- CPU restatement of the reference's NTT algorithms.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is the parity checker for
 * the HIP library: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.  The product path
 * (gpu-ntt_amd/csrc) never links it and has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks every
 * function below bit-for-bit against oracle/_ref/libgpuntt_ref.so -- the
 * reference's own CPU classes (NTTParameters, NTTCPU, NTT_4STEP_CPU,
 * NTTParameters4Step) compiled from the sources where they lie under
 * /root/reference by oracle/Makefile -- and against the committed fixtures in
 * tests/golden/ that were generated from that same build
 * (tools/make_golden.py).
 *
 * Each function cites the reference file:line it restates (see
 * ntt_oracle_impl.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* gpuntt::bitreverse  src/lib/common/nttparameters.cu:10-20 */
int ora_bitreverse(int index, int n_power)
{
    int res_1 = 0;
    for (int i = 0; i < n_power; i++)
    {
        res_1 <<= 1;
        res_1 = (index & 1) | res_1;
        index >>= 1;
    }
    return res_1;
}

/* portable PRNG for synthetic inputs (not part of the reference) */
uint64_t ora_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

#define ORA_BITS 32
#define ORA_T uint32_t
#define ORA_T2 uint64_t
#define ORA_TMAX UINT32_MAX
#define ORA_(x) ora32_##x
#include "ntt_oracle_impl.h"
#undef ORA_BITS
#undef ORA_T
#undef ORA_T2
#undef ORA_TMAX
#undef ORA_

#define ORA_BITS 64
#define ORA_T uint64_t
#define ORA_T2 unsigned __int128
#define ORA_TMAX UINT64_MAX
#define ORA_(x) ora64_##x
#include "ntt_oracle_impl.h"
#undef ORA_BITS
#undef ORA_T
#undef ORA_T2
#undef ORA_TMAX
#undef ORA_
