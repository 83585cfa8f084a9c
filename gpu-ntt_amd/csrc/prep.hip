// prep.hip -- twiddle preparation for the fast (lazy-residue) kernels, 32- and 64-bit, + its workspace.
//
// The caller's tables hold plain residues in the reference's bit-reversed order
// (reference test_merge_ntt.cu:115-122).  Each call re-derives from them, on the call's
// stream, the table the fast kernels read: Shoup pairs {w, floor(w * 2^W / q)} laid out by
// stage (stage with m = 2^S groups occupies slots [2^S, 2^(S+1)) for both reduction
// polynomials), with the distance-1/2/4 stages permuted to [tile][k][thread] so the last
// contiguous round loads them fully coalesced.  The quotients come from one multiplication by
// the modulus' normalised reciprocal plus a two-step correction (~40 instructions per entry; the
// reciprocal itself is a host division for a single modulus, one restoring division per block for
// an RNS stack).  Nothing is cached between calls, so a caller that rewrites its table in place
// is always honoured.
#include <atomic>
#include <cstdlib>
#include <initializer_list>
#include <string>
#include <map>
#include <tuple>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "lazy_launch.hpp"

namespace gpuntt
{
    namespace kern
    {
        // floor(w * 2^W / q) for w < q by restoring division (q < 2^(W-2))
        template <typename T> __device__ __forceinline__ T shoup_quotient(T w, T q)
        {
            T rem = w, quo = 0;
#pragma unroll 8
            for (int i = 0; i < static_cast<int>(8 * sizeof(T)); i++)
            {
                rem <<= 1;
                const bool ge = rem >= q;
                rem = ge ? rem - q : rem;
                quo = (quo << 1) | (ge ? 1u : 0u);
            }
            return quo;
        }

        // Normalised reciprocal R = floor(2^(W-1+b) / q), b = bit length of q: in [2^(W-1), 2^W) for
        // every q that is not a power of two.  0 = "none" (power of two / q < 3): callers then
        // fall back to the restoring division.
        // One thread of every preparation block derives it while the others wait at the barrier, so its LATENCY is on
        // the critical path of every drop-in RNS call: a double-precision estimate (2^-52 relative: within 2^13 of R) and
        // two exact 128-bit remainder corrections instead of a 64-step restoring division (~2500 dependent cycles).
        __device__ __forceinline__ uint64_t recip_norm64(uint64_t q)
        {
            const int b = 64 - __clzll(static_cast<long long>(q));
            const double dq = static_cast<double>(q);
            const double est = ldexp(1.0, 63 + b) / dq; // (2^63, 2^64]
            uint64_t R = est >= 18446744073709549568.0 ? 0xfffffffffffff800ull : static_cast<uint64_t>(est);
            // rem = 2^(63+b) - R * q as a signed 128-bit number (hi : lo); |rem| < 2^14 * q
            auto remainder = [&](uint64_t r, long long& hi, uint64_t& lo) {
                const uint64_t plo = r * q, phi = __umul64hi(r, q);
                lo = 0ull - plo;
                hi = static_cast<long long>((1ull << (b - 1)) - phi - (plo != 0ull ? 1ull : 0ull));
            };
            long long hi;
            uint64_t lo;
            remainder(R, hi, lo);
            // first correction from the estimate of rem / q (|.| < 2^14: exact in a double up to +-1)
            const double drem = static_cast<double>(hi) * 18446744073709551616.0 + static_cast<double>(lo);
            const long long adj = static_cast<long long>(floor(drem / dq));
            R += static_cast<uint64_t>(adj);
            remainder(R, hi, lo);
            // exact finish: 0 <= rem < q
            for (int i = 0; i < 4 && hi < 0; i++)
            {
                R--;
                const uint64_t nl = lo + q;
                hi += (nl < lo) ? 1 : 0;
                lo = nl;
            }
            for (int i = 0; i < 4 && (hi > 0 || lo >= q); i++)
            {
                R++;
                hi -= (lo < q) ? 1 : 0;
                lo -= q;
            }
            return R;
        }
        template <typename T> __device__ __forceinline__ T recip_norm(T q)
        {
            if (q < 3 || (q & (q - 1)) == 0)
                return 0;
            if constexpr (sizeof(T) == 8)
                return recip_norm64(q);
            else
            {
                const int b = 32 - __clz(static_cast<int>(q));
                return static_cast<T>((1ull << (31 + b)) / q);
            }
        }
        // diagnostic (C ABI gpuntt_debug_recip_norm_*): the reciprocal of every q[i], checked against integers in the tests
        template <typename T>
        __global__ __launch_bounds__(256) void debug_recip_norm(const T* __restrict__ q, T* __restrict__ out, unsigned long long count)
        {
            const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
            if (i < count)
                out[i] = recip_norm<T>(q[i]);
        }

        // floor(w * 2^W / q) for w < q < 2^(W-2) from R = recip_norm(q):
        //   est = (w * R) >> (b-1) lies in {Q-2, Q-1, Q}  (R > 2^(W-1+b)/q - 1 and w < 2^b),
        //   and the remainder w*2^W - est*q < 3q < 2^W is exact in W bits, so two conditional
        //   subtractions finish it.
        template <typename T> __device__ __forceinline__ T shoup_quotient_r(T w, T q, T rinv)
        {
            if (rinv == 0)
                return shoup_quotient<T>(w, q);
            constexpr int W = static_cast<int>(8 * sizeof(T));
            const int sh = W - 1 - ((W == 64) ? __clzll(static_cast<long long>(q)) : __clz(static_cast<int>(q)));
            const T hi = dev::mulhi(w, rinv), lo = w * rinv;
            T quo = (lo >> sh) | (hi << (W - sh)); // 1 <= sh <= W-3
            T rem = static_cast<T>(0) - quo * q;
#pragma unroll
            for (int i = 0; i < 2; i++)
            {
                const bool ge = rem >= q;
                rem = ge ? rem - q : rem;
                quo += ge ? 1u : 0u;
            }
            return quo;
        }

        // a * b mod q for a, b < q < 2^(W-1), through the Shoup pair of b
        template <typename T> __device__ __forceinline__ T mulmod_r(T a, T b, T q, T rinv)
        {
            const T bp = shoup_quotient_r<T>(b, q, rinv);
            const T r = a * b - dev::mulhi(a, bp) * q; // [0, 2q)
            return (r >= q) ? (r - q) : r;
        }

        // The fall-back of a drop-in RNS Merge call (kern::SlowArgs): polynomial p of the batch -- modulus p % mod_count,
        // prime index through mod_order, memory slot through poly_order -- transformed by ONE block, a stage at a time
        // through global memory, with the caller's plain table and the public Barrett arithmetic (what the reference's
        // kernels compute, src/lib/ntt_merge/ntt.cu:596-761, 1086-1318; same stage / twiddle indexing as
        // merge_kernels.hpp: column_ntt_small).  Correctness path: runs when the host's family prediction was wrong.
        template <typename T>
        __device__ void slow_rns_transform(const SlowArgs<T>& sa, const T* __restrict__ roots, const Modulus<T>* __restrict__ mods,
                                           const T* __restrict__ ninv_arr, const int* __restrict__ mod_order, int mod_count,
                                           int n, int negacyclic)
        {
            using S = typename std::make_signed<T>::type;
            const unsigned long long N = 1ull << n;
            for (unsigned long long p = blockIdx.x; p < sa.polys; p += gridDim.x)
            {
                const int mi = static_cast<int>(p % static_cast<unsigned>(mod_count));
                const int prime = (mod_order != nullptr) ? mod_order[mi] : mi;
                const Modulus<T> md = mods[prime];
                const dev::ModCtx<T> m{md.value, md.bit, md.mu};
                const unsigned long long slot = (sa.poly_order != nullptr) ? static_cast<unsigned>(sa.poly_order[p]) : p;
                // coefficient e of the polynomial lies at base + (e << es): rows of a PerPolynomial batch, or a column of the
                // PerCoefficient matrix (reference ForwardCoreTranspose / InverseCoreTranspose, ntt.cu:1693-1835, 1957-2074)
                const int es = sa.col_log >= 0 ? sa.col_log : 0;
                const unsigned long long base = sa.col_log >= 0 ? slot : (slot << n);
                const T* src = static_cast<const T*>(sa.in) + base;
                T* dst = sa.out + base;
                const T* tab = roots + (static_cast<unsigned long long>(prime) << n);
                for (unsigned long long e = threadIdx.x; e < N; e += 256)
                {
                    T x = src[e << es];
                    if ((sa.flags & F_SIGNED_IN) && static_cast<S>(x) < 0)
                        x = static_cast<T>(m.q + x);
                    dst[e << es] = x;
                }
                __threadfence();
                __syncthreads();
                for (int st = 0; st < n; st++)
                {
                    const int P = sa.inverse ? st : (n - 1 - st); // butterfly distance 2^P
                    for (unsigned long long b = threadIdx.x; b < (N >> 1); b += 256)
                    {
                        const unsigned long long lo = b & ((1ull << P) - 1ull), grp = b >> P;
                        const unsigned long long i0 = (grp << (P + 1)) | lo, i1 = i0 + (1ull << P);
                        const T w = tab[grp + (negacyclic ? (1ull << (n - 1 - P)) : 0ull)];
                        T U = dst[i0 << es], V = dst[i1 << es];
                        if (sa.inverse)
                            dev::gs_butterfly(U, V, w, m);
                        else
                            dev::ct_butterfly(U, V, w, m);
                        dst[i0 << es] = U;
                        dst[i1 << es] = V;
                    }
                    __threadfence();
                    __syncthreads();
                }
                const bool scale = (sa.flags & F_SCALE) != 0u && ninv_arr != nullptr;
                if (scale || sa.mul_in != nullptr)
                {
                    const T ninv = scale ? ninv_arr[prime] : static_cast<T>(0);
                    for (unsigned long long e = threadIdx.x; e < N; e += 256)
                    {
                        T x = dst[e << es];
                        if (sa.mul_in != nullptr)
                            x = m.mul(x, sa.mul_in[base + (e << es)]);
                        if (scale)
                        {
                            x = m.mul(x, ninv);
                            if (sa.flags & F_CENTERED)
                                x = (x > (m.q >> 1)) ? static_cast<T>(x - m.q) : x;
                        }
                        dst[e << es] = x;
                    }
                }
                __syncthreads();
            }
        }

        // fold_ninv (inverse transforms): the single twiddle of the final Gentleman-Sande stage
        // (slot 1) is stored pre-multiplied by n^-1, so that stage scales both outputs itself --
        // U' = (U + V) * n^-1, V' = (U - V) * (w * n^-1) -- instead of a separate n^-1 product on
        // every coefficient afterwards.
        template <typename T>
        __global__ __launch_bounds__(256) void prep_twiddles(const T* __restrict__ roots,
                                                             lazy::Tw<T>* __restrict__ ws,
                                                             const Modulus<T>* __restrict__ mods, T q_single,
                                                             T rinv_single, int mod_count, int n, int negacyclic,
                                                             int perm_tile_log,
                                                             const T* __restrict__ ninv_arr,
                                                             lazy::Tw<T>* __restrict__ ws_ninv,
                                                             unsigned* __restrict__ go_flag,
                                                             lazy::NormConst* __restrict__ norm_arr,
                                                             const int* __restrict__ mod_order,
                                                             T ninv_single, int fold_ninv,
                                                             unsigned* __restrict__ host_state, int allow_31q,
                                                             unsigned family, SlowArgs<T> slow)
        {
            const unsigned long long gid = blockIdx.x * 256ull + threadIdx.x;
            // RNS stacks (moduli in device memory): classify the stack -- kern::GO_GENERIC (a modulus outside the fast
            // kernels' domain), GO_LAZY (every modulus leaves the default lazy range of the word size its headroom: bit <= 60 /
            // 30), GO_LAZY_8Q / GO_LAZY_4Q (64-bit words, widest modulus 61 / 62 bits: the 8 q / 4 q kernels, 4096-coefficient
            // tiles only).  Block 0 publishes the state; every block needs it when the default family would run on a bigger
            // tile, because the per-tile permutation of the last three stages must match the family that will really run.
            // family != 0: the ONE lazy family the host enqueued behind this launch (its prediction for the stack); the
            // flag then names that family when the stack fits it, else GO_GENERIC -- and this kernel transforms the batch
            // itself (slow_rns_transform).  family == 0: every family is enqueued, the flag names the exact state.
            // EVERY block classifies (its fall-back role needs the verdict; so does the per-tile permutation): wave 0 reads the
            // moduli and votes, thread 0 derives the reciprocal of the block's modulus meanwhile (RNS: a block never straddles
            // two moduli when n >= 8; below that every thread derives its own), ONE barrier publishes both
            __shared__ unsigned s_state;
            __shared__ T s_rinv;
            const unsigned long long per_mod = 1ull << n;
            const bool block_recip = (mods != nullptr) && n >= 8;
            // the thread's table word is requested FIRST: its latency then runs beside the classification and the
            // reciprocal instead of behind the barrier (the kernel is a chain of latencies, not of work)
            const bool in_table = gid < per_mod * mod_count;
            const int mi = static_cast<int>(gid >> n);
            // prepared tables are indexed by the compact slot mi; the caller's moduli, tables and
            // n^-1 values by the prime index (identical unless *_Modulus_Ordered remaps it)
            const int prime = (in_table && mod_order != nullptr) ? mod_order[mi] : mi;
            const unsigned slot = static_cast<unsigned>(gid & (per_mod - 1));
            const int S = slot ? (31 - __clz(slot)) : 0; // stage: m = 2^S groups
            const unsigned i = slot - (1u << S);         // group index = position in the caller's table
            T w = 0;
            if (in_table && slot != 0u)
                w = roots[(static_cast<unsigned long long>(prime) << n) + (negacyclic ? ((1u << S) + i) : i)];
            unsigned state = GO_LAZY;
            if (mods != nullptr)
            {
                if (threadIdx.x < 64)
                {
                    bool bad = false, w61 = false, w62 = false, over31 = false;
                    for (int i = static_cast<int>(threadIdx.x); i < mod_count; i += 64)
                    {
                        const Modulus<T> md = mods[mod_order != nullptr ? mod_order[i] : i];
                        if (md.value < 3 || md.bit > static_cast<T>(sizeof(T) == 8 ? 62 : lazy::Mod<T>::MAX_BIT))
                            bad = true;
                        else if (sizeof(T) == 8 && md.bit == static_cast<T>(62))
                            w62 = true;
                        else if (sizeof(T) == 8 && md.bit == static_cast<T>(61))
                            w61 = true;
                        if (sizeof(T) == 8 && static_cast<unsigned long long>(md.value) > 0xffffffffffffffffull / 31)
                            over31 = true;
                        // normalisation constants of modulus i (one 64-bit division each): one LANE per modulus
                        if (blockIdx.x == 0 && norm_arr != nullptr && go_flag != nullptr)
                            norm_arr[i] = lazy::norm_const_of(md.value, md.bit);
                    }
                    const bool any_bad = __ballot(bad) != 0ull, any62 = __ballot(w62) != 0ull, any61 = __ballot(w61) != 0ull,
                               any_over31 = __ballot(over31) != 0ull;
                    // forward calls (no n^-1 folded) of 64-bit words whose every modulus has 31 q < 2^64: the 31 q kernels
                    const bool wide_range = sizeof(T) == 8 && allow_31q != 0 && fold_ninv == 0 && !any_over31;
                    if (threadIdx.x == 0)
                        s_state = any_bad ? GO_GENERIC
                                          : (any62 ? GO_LAZY_4Q : (any61 ? GO_LAZY_8Q : (wide_range ? GO_LAZY_31Q : GO_LAZY)));
                }
                if (block_recip && threadIdx.x == 64)
                {
                    const int bm = static_cast<int>((blockIdx.x * 256ull) >> n);
                    // (blocks beyond the table exist when the grid was enlarged for the fall-back: nothing to prepare)
                    s_rinv = (bm < mod_count) ? recip_norm<T>(mods[mod_order != nullptr ? mod_order[bm] : bm].value) : static_cast<T>(0);
                }
                __syncthreads();
                state = s_state;
            }
            // host-mapped word (or nullptr): what the host predicts the NEXT call of this stack from (RnsGuess)
            if (gid == 0 && host_state != nullptr)
                *host_state = state;
            if (mods != nullptr && go_flag != nullptr)
            {
                const bool fits = (family == 0u) ? (state != GO_GENERIC)
                                                 : (state != GO_GENERIC && family_rank(state) <= family_rank(family));
                if (!fits || slow.force != 0)
                {
                    if (gid == 0)
                        *go_flag = GO_GENERIC; // every fast kernel behind this launch returns
                    if (slow.enabled != 0)
                        slow_rns_transform<T>(slow, roots, mods, ninv_arr, mod_order, mod_count, n, negacyclic);
                    return; // (no table: nobody reads it)
                }
                if (family != 0u)
                    state = family; // the stack runs on the (equal or wider) family that was enqueued
            }
            if ((state == GO_LAZY_8Q || state == GO_LAZY_4Q) && perm_tile_log > 12)
                perm_tile_log = 12;
            if (gid == 0 && go_flag != nullptr)
                *go_flag = state;
            if (!in_table)
                return;
            const T q = (mods != nullptr) ? mods[prime].value : q_single;
            const T rinv = (mods == nullptr) ? rinv_single : (block_recip ? s_rinv : recip_norm<T>(q));
            if (slot == 0)
            {
                if (ninv_arr != nullptr && ws_ninv != nullptr)
                {
                    const T v = ninv_arr[prime];
                    ws_ninv[mi] = lazy::Tw<T>{v, shoup_quotient_r<T>(v, q, rinv)};
                }
                ws[gid] = lazy::Tw<T>{0, 0};
                return;
            }
            const int P = n - 1 - S;              // butterfly distance 2^P
            unsigned long long dst = gid;
            if (perm_tile_log > 0 && P <= 2)
            {
                // [tile][k][thread] layout of the stages whose twiddles differ per thread: entry t * rp + k of a tile's
                // rp * nt entries goes to position k * nt + t.  The THREAD enumerates the caller's table (fully coalesced
                // reads) and scatters its pair in runs of 64 / rp lanes = 128 .. 512 B; enumerating the destination
                // instead read the table with a stride of rp words -- 7/8 of all entries, 5 x the bytes (PMC: the
                // prepared table of 2^24 x 4 cost 0.65 GiB of fetch for a 128 MiB source)
                // (powers of two throughout: shifts and masks, no integer division)
                const int nt_log = perm_tile_log - 4; // threads per tile (16 coefficients each)
                const int rp_log = 3 - P;             // twiddles per thread = 16 >> (P + 1)
                const unsigned tile = i >> (rp_log + nt_log), rem = i & ((1u << (rp_log + nt_log)) - 1u);
                const unsigned kk = rem & ((1u << rp_log) - 1u), t = rem >> rp_log;
                dst = (gid - i) + (tile << (rp_log + nt_log)) + (kk << nt_log) + t;
            }
            if (fold_ninv && slot == 1)
                w = mulmod_r<T>(w, (ninv_arr != nullptr) ? ninv_arr[prime] : ninv_single, q, rinv);
            ws[dst] = lazy::Tw<T>{w, shoup_quotient_r<T>(w, q, rinv)};
        }

        // a^e (mod q) by square and multiply (table check, plan construction)
        template <typename T> __device__ __forceinline__ T powmod_r(T base, unsigned long long e, T q, T rinv)
        {
            T r = 1, b = base;
            while (e != 0ull)
            {
                if (e & 1ull)
                    r = mulmod_r<T>(r, b, q, rinv);
                b = mulmod_r<T>(b, b, q, rinv);
                e >>= 1;
            }
            return r;
        }

        // ---- table check of the 4-step entry points (on by default, no host involvement) ---------------------------------
        // The fast 4-step path derives every twiddle from n1_table and ONE row of W (prep_merge_from_fourstep below) where
        // the reference multiplies by W[address] element by element and runs the rows through n2_table
        // (src/lib/ntt_4step/ntt_4step.cu:1049-1058, 776-779).  The two agree iff the three tables are the ones
        // NTTParameters4Step generates for ONE root g of order N (src/lib/common/nttparameters.cu:356-444):
        //   forward  W[r * n2 + j] = g^(brev(r, log n1) * j)      inverse  W[r * n2 + c] = g^(r * brev(c, log n2))
        //   n1_table[i] = (g^n2)^brev(i, log n1 - 1)              n2_table[i] = (g^n1)^brev(i, log n2 - 1)
        // Thread gid of the preparation kernel verifies entry gid of W (and of the small tables) with ONE modular product
        // against its neighbours, which pins ALL N + n1/2 + n2/2 words:
        //   a table t of bit-reversed powers (t[i] = b^brev(i, L)):  t[0] = 1,  t[2^k + i'] = t[i'] * t[2^k],
        //       t[2^k] = t[2^(k+1)]^2,  t[2^(L-1)] = b                                          (brev_powers_ok)
        //   forward W: column 1 is such a table (b = g: the entry that DEFINES g), row r is geometric with ratio W[r*n2 + 1];
        //   inverse W: row 1 is such a table (L = log n2), column c is geometric with ratio W[n2 + c];
        //   g^(N/2) = -1: the entry holding g^(N/4) squares to q - 1;   the bases g^n2, g^n1 of the small tables are W entries.
        // A thread that finds a mismatch publishes kern::GO_GENERIC through the call's veto word (publish_state): the
        // fast kernels of the call return, the element-by-element Barrett kernels enqueued behind them run -- whatever
        // the tables say, like the reference, only slower.
        __device__ __forceinline__ void publish_state(unsigned long long* word, unsigned epoch, unsigned state)
        {
            // smallest high half = latest call of the chain; within a call the smallest state wins (GO_GENERIC = 0)
            atomicMin(word, (static_cast<unsigned long long>(~epoch) << 32) | state);
        }
        // t[i * stride], i < 2^L, are b^brev(i, L); `top` = b, or nullptr when t[2^(L-1)] itself defines b
        template <typename T>
        __device__ __forceinline__ bool brev_powers_ok(const T* __restrict__ t, unsigned long long stride, unsigned i, int L,
                                                       const T* top, T q, T rinv)
        {
            const T x = t[i * stride];
            if (x >= q)
                return false;
            if (i == 0u)
                return x == static_cast<T>(1);
            const int k = 31 - __clz(i);
            const unsigned rest = i - (1u << k);
            if (rest != 0u)
            {
                const T u = t[rest * stride], v = t[(static_cast<unsigned long long>(1u) << k) * stride];
                return u < q && v < q && x == mulmod_r<T>(u, v, q, rinv);
            }
            if (k == L - 1)
                return top == nullptr || x == *top;
            const T y = t[(static_cast<unsigned long long>(2u) << k) * stride];
            return y < q && x == mulmod_r<T>(y, y, q, rinv);
        }
        // Bulk of the check for rings from 2^17 (n2 >= 4096): a thread verifies FOUR_E entries that share their multiplier --
        // forward: the entries e = chunk * 256 * FOUR_E + k * 256 + t of one row (ratio W[r*n2 + 1]); inverse: column c of
        // FOUR_E consecutive rows (ratio W[n2 + c]) -- so the Shoup quotient of the multiplier is derived once per thread
        // and an entry costs one multiply-subtract and a compare (~12 instructions instead of ~50).
        constexpr int FOUR_E = 8;
        template <typename T>
        __device__ __forceinline__ bool fourstep_bulk_ok(const T* __restrict__ w, unsigned t8, int l1, int l2, int inverse, T q, T rinv)
        {
            const unsigned long long n2 = 1ull << l2;
            bool ok = true;
            if (!inverse)
            {
                const unsigned long long e0 = (static_cast<unsigned long long>(t8 >> 8) << 11) + (t8 & 255u);
                const unsigned long long row = e0 >> l2;
                const T v = w[(row << l2) + 1ull];
                const T vp = shoup_quotient_r<T>(v < q ? v : static_cast<T>(0), q, rinv);
#pragma unroll
                for (int k = 0; k < FOUR_E; k++)
                {
                    const unsigned long long e = e0 + (static_cast<unsigned long long>(k) << 8);
                    const unsigned long long j = e & (n2 - 1ull);
                    const T x = w[e];
                    ok = ok && x < q;
                    if (j >= 2ull)
                    {
                        const T u = w[e - 1ull];
                        T r = u * v - dev::mulhi(u, vp) * q; // [0, 2q) for any u
                        r = (r >= q) ? (r - q) : r;
                        ok = ok && x == r;
                    }
                }
            }
            else
            {
                const unsigned long long c = t8 & (n2 - 1ull);
                const unsigned long long r0 = static_cast<unsigned long long>(t8 >> l2) * FOUR_E;
                const T v = w[n2 + c];
                const T vp = shoup_quotient_r<T>(v < q ? v : static_cast<T>(0), q, rinv);
                T u = (r0 > 0ull) ? w[((r0 - 1ull) << l2) + c] : static_cast<T>(0);
#pragma unroll
                for (int k = 0; k < FOUR_E; k++)
                {
                    const unsigned long long r = r0 + static_cast<unsigned long long>(k);
                    const T x = w[(r << l2) + c];
                    ok = ok && x < q;
                    if (r >= 2ull)
                    {
                        T m = u * v - dev::mulhi(u, vp) * q;
                        m = (m >= q) ? (m - q) : m;
                        ok = ok && x == m;
                    }
                    u = x;
                }
            }
            (void) l1;
            return ok;
        }
        template <typename T>
        __device__ __forceinline__ bool fourstep_tables_ok(const T* __restrict__ n1_table, const T* __restrict__ n2_table,
                                                           const T* __restrict__ w, unsigned gid, int l1, int l2, int inverse,
                                                           T q, T rinv)
        {
            const unsigned long long n2 = 1ull << l2;
            const unsigned r = gid >> l2, c = gid & static_cast<unsigned>(n2 - 1ull);
            const bool bulk = l2 >= 11; // the geometric relations of rows / columns are checked FOUR_E at a time
            bool ok = true;
            if (bulk && gid < (1u << (l1 + l2 - 3)))
                ok = fourstep_bulk_ok<T>(w, gid, l1, l2, inverse, q, rinv);
            if (!inverse)
            {
                if (c == 0u)
                    ok = ok && w[gid] == static_cast<T>(1);
                else if (c == 1u)
                    ok = ok && brev_powers_ok<T>(w + 1, n2, r, l1, nullptr, q, rinv);
                else if (!bulk)
                {
                    const T x = w[gid], u = w[gid - 1u], v = w[(static_cast<unsigned long long>(r) << l2) + 1u];
                    ok = ok && x < q && u < q && v < q && x == mulmod_r<T>(u, v, q, rinv);
                }
            }
            else
            {
                if (r == 0u)
                    ok = ok && w[gid] == static_cast<T>(1);
                else if (r == 1u)
                    ok = ok && brev_powers_ok<T>(w + n2, 1ull, c, l2, nullptr, q, rinv);
                else if (!bulk)
                {
                    const T x = w[gid], u = w[gid - n2], v = w[n2 + c];
                    ok = ok && x < q && u < q && v < q && x == mulmod_r<T>(u, v, q, rinv);
                }
            }
            if (gid == 0u)
            {
                // g^(N/4): forward row 1 has ratio g^(n1/2), entry n2/2; inverse row n1/2, column 1 (brev = n2/2)
                const T h = inverse ? w[(n2 << (l1 - 1)) + 1ull] : w[n2 + (n2 >> 1)];
                ok = ok && h < q && mulmod_r<T>(h, h, q, rinv) == q - 1;
            }
            // small tables: bases g^n2 and g^n1 are entries of W (forward: row 1 = powers of g^(n1/2); inverse: W[r*n2+c] = g^(r*brev(c)))
            if (gid < (1u << (l1 - 1)))
            {
                const T* top = inverse ? (w + 2ull * n2 + 1ull) : (w + n2 + (2ull << (l2 - l1)));
                ok = ok && *top < q && brev_powers_ok<T>(n1_table, 1ull, gid, l1 - 1, top, q, rinv);
            }
            if (gid < (1u << (l2 - 1)))
            {
                const T* top = inverse ? (w + n2 + (1ull << (l2 - 1 - l1))) : (w + n2 + 2ull);
                ok = ok && *top < q && brev_powers_ok<T>(n2_table, 1ull, gid, l2 - 1, top, q, rinv);
            }
            return ok;
        }

        // The 4-step transform IS the Merge transform of the same ring with one transposition on the natural-order
        // side (forward: GPU_4STEP_NTT(in) = MergeNTT(in read as the n2 x n1 transpose of x); inverse: the output is
        // stored transposed), so the Merge kernels can run it from a MERGE table of the ring -- bit-reversed powers of
        // the 4-step root w, rebuilt here from the caller's 4-step tables and written straight into the kernels' stage
        // layout (same slots / permutation as prep_twiddles, cyclic):
        //   w^k = Wrow(k mod n2) * n1_table[brev(k >> log n2, log n1 - 1)],      k = brev(i, n - 1)
        //   forward  Wrow(j) = W[(n1 / 2) * n2 + j]          (W[r * n2 + j] = w^(brev(r, log n1) * j))
        //   inverse  Wrow(j) = W[n2 + brev(j, log n2)]       (W[r * n2 + c] = w^-(r * brev(c, log n2)))
        // (reference table generators: src/lib/common/nttparameters.cu:356-444).  fold: n^-1 into the single twiddle of
        // the final inverse stage (slot 1).  mods != nullptr: one device-side modulus, classified like prep_twiddles does for an RNS stack.
        // veto != nullptr (drop-in calls, plan construction): the state goes out through the veto word, and with
        // n2_table != nullptr every thread also checks its share of the caller's tables (fourstep_tables_ok).
        template <typename T>
        __global__ __launch_bounds__(256) void prep_merge_from_fourstep(
            const T* __restrict__ n1_table, const T* __restrict__ w_table, lazy::Tw<T>* __restrict__ ws, int log_n1,
            int log_n2, int perm_tile_log, int inverse, int fold, T q_single, T rinv_single, T ninv_single,
            const Modulus<T>* __restrict__ mods, const T* __restrict__ ninv_dev, lazy::Tw<T>* __restrict__ ws_ninv,
            unsigned* __restrict__ go_flag, lazy::NormConst* __restrict__ norm_arr, unsigned* __restrict__ host_state,
            const T* __restrict__ n2_table, unsigned long long* __restrict__ veto, unsigned epoch)
        {
            const unsigned long long gid = blockIdx.x * 256ull + threadIdx.x;
            __shared__ T s_rinv;
            T q = q_single, rinv = rinv_single;
            unsigned state = GO_LAZY; // host-side modulus: the host picked the kernels, the word only carries the veto
            if (mods != nullptr)
            {
                const Modulus<T> md = mods[0];
                q = md.value;
                if (threadIdx.x == 0)
                    s_rinv = recip_norm<T>(q);
                __syncthreads();
                rinv = s_rinv;
                // four-state go-flag, like prep_twiddles; a 61- / 62-bit modulus runs the 8 q / 4 q family on 4096-coefficient
                // tiles, so the table takes that tile's permutation
                state = (md.value < 3 || md.bit > static_cast<T>(sizeof(T) == 8 ? 62 : lazy::Mod<T>::MAX_BIT))
                            ? GO_GENERIC
                            : ((sizeof(T) == 8 && md.bit == static_cast<T>(62))
                                   ? GO_LAZY_4Q
                                   : ((sizeof(T) == 8 && md.bit == static_cast<T>(61)) ? GO_LAZY_8Q : GO_LAZY));
                if ((state == GO_LAZY_8Q || state == GO_LAZY_4Q) && perm_tile_log > 12)
                    perm_tile_log = 12;
                if (gid == 0)
                {
                    if (go_flag != nullptr)
                        *go_flag = state;
                    if (host_state != nullptr)
                        *host_state = state;
                    if (norm_arr != nullptr)
                        norm_arr[0] = lazy::norm_const_of(md.value, md.bit);
                }
            }
            if (gid == 0 && veto != nullptr)
                publish_state(veto, epoch, state);
            const T ninv = (ninv_dev != nullptr) ? ninv_dev[0] : ninv_single;
            if (gid == 0 && ninv_dev != nullptr && ws_ninv != nullptr)
                ws_ninv[0] = lazy::Tw<T>{ninv, shoup_quotient_r<T>(ninv, q, rinv)};
            const int n = log_n1 + log_n2;
            if (gid >= (1ull << n))
                return;
            if (veto != nullptr && n2_table != nullptr && state != GO_GENERIC &&
                !fourstep_tables_ok<T>(n1_table, n2_table, w_table, static_cast<unsigned>(gid), log_n1, log_n2, inverse, q, rinv))
                publish_state(veto, epoch, GO_GENERIC);
            const unsigned slot = static_cast<unsigned>(gid);
            if (slot == 0)
            {
                ws[0] = lazy::Tw<T>{0, 0};
                return;
            }
            const int S = 31 - __clz(slot);
            unsigned i = slot - (1u << S);
            const int P = n - 1 - S;
            if (perm_tile_log > 0 && P <= 2)
            {
                const int nt_log = perm_tile_log - 4;
                const int rp_log = 3 - P;
                const unsigned tile = i >> (rp_log + nt_log), rem = i & ((1u << (rp_log + nt_log)) - 1u);
                const unsigned kk = rem >> nt_log, t = rem & ((1u << nt_log) - 1u);
                i = (tile << (rp_log + nt_log)) + (t << rp_log) + kk;
            }
            const unsigned k = (n > 1) ? (__brev(i) >> (33 - n)) : 0u; // brev(i, n - 1)
            const unsigned n2 = 1u << log_n2;
            const unsigned j = k & (n2 - 1u), m = k >> log_n2;
            const unsigned long long widx =
                inverse ? (static_cast<unsigned long long>(n2) + (__brev(j) >> (32 - log_n2)))
                        : ((static_cast<unsigned long long>(n2) << (log_n1 - 1)) + j);
            T w = w_table[widx];
            if (m != 0u)
                w = mulmod_r<T>(w, n1_table[__brev(m) >> (33 - log_n1)], q, rinv);
            if (fold && slot == 1)
                w = mulmod_r<T>(w, ninv, q, rinv);
            ws[slot] = lazy::Tw<T>{w, shoup_quotient_r<T>(w, q, rinv)};
        }

    } // namespace kern

    namespace host
    {
        // ---- options and test hooks -------------------------------------------------------------------
        // The library reads NO environment variable.  PRODUCT options (GPU_NTT_SetOption / gpuntt_set_option, listed in
        // include/gpuntt/ntt_merge/ntt.cuh): path = default | generic | fast, check_4step_tables, rns_predict.  TEST HOOKS
        // (set_test_hook / gpuntt_test_set_hook -- a symbol the public headers do not declare; csrc/test_hooks.h): path =
        // fast-strict | generic-capped, no_scratch, rns_force_fallback, u32_e32.  Every public entry point opens a
        // WorkspaceScope, and the outermost scope of a thread takes ONE snapshot of all of them: a call sees the same values
        // from its first decision to its last launch, whatever another thread sets meanwhile.
        namespace
        {
            struct OptionValues
            {
                int path = 0;               // 0 size heuristic, 1 generic, 2 fast, 3 fast-strict, 4 generic-capped
                int no_scratch = 0;         // test hook: behave as if the twiddle scratch could not be allocated
                int check_4step = 1;        // 4-step entry points: verify the caller's three tables on the device
                int rns_predict = 1;        // drop-in RNS calls: enqueue only the lazy family the stack needed last time
                int rns_force_fallback = 0; // test hook: the preparation kernel's own fall-back serves every drop-in RNS Merge call
                int u32_e32 = 0x1f000;      // test hook: 32-bit Merge rings on the 32-coefficients-per-lane kernels (bit n = ring 2^n; bit 16: the
                                            // full-tile contiguous pass of larger rings)
                int two_sweep_big = 0;      // test hook (experiment): 64-bit rings 2^23 / 2^24 forward in two sweeps on 16384-coefficient tiles
            };
            std::mutex g_opt_mutex;
            std::atomic<int> g_reset_predictions{0};   // test hook reset_predictions: consumed by the next rns_guess
            OptionValues g_opt;                        // guarded by g_opt_mutex
            std::atomic<unsigned> g_opt_generation{1}; // bumped by every successful set
            thread_local OptionValues t_opt;           // this thread's snapshot ...
            thread_local unsigned t_opt_generation = 0; // ... of this generation
            thread_local int t_scope_depth = 0;         // WorkspaceScopes open on this thread

            void refresh_snapshot()
            {
                const unsigned gen = g_opt_generation.load(std::memory_order_acquire);
                if (gen != t_opt_generation)
                {
                    std::lock_guard<std::mutex> lock(g_opt_mutex);
                    t_opt = g_opt;
                    t_opt_generation = g_opt_generation.load(std::memory_order_relaxed);
                }
            }
            // inside an API call: the snapshot its outermost WorkspaceScope took; outside (plan construction helpers, tests
            // poking single functions): the current values
            const OptionValues& options()
            {
                if (t_scope_depth == 0)
                    refresh_snapshot();
                return t_opt;
            }

            bool set_option_impl(const char* name, const char* value, bool test_hooks)
            {
                if (name == nullptr || value == nullptr)
                    return false;
                const std::string k(name), v(value);
                // numeric values: the whole string must be a number of the option's documented set -- "abc" or "7" are
                // refused (false), not silently turned into 0 / the default
                char* end = nullptr;
                const long lv = std::strtol(value, &end, 0);
                const bool is_num = end != value && *end == '\0';
                const int iv = static_cast<int>(lv);
                const bool bit = is_num && (lv == 0 || lv == 1);
                std::lock_guard<std::mutex> lock(g_opt_mutex);
                if (k == "path")
                {
                    const int m = v == "generic" ? 1 : v == "fast" ? 2 : (v == "default" || v.empty()) ? 0
                                  : (test_hooks && v == "fast-strict") ? 3 : (test_hooks && v == "generic-capped") ? 4 : -1;
                    if (m < 0)
                        return false;
                    g_opt.path = m;
                }
                else if (k == "check_4step_tables" && bit)
                    g_opt.check_4step = iv;
                else if (k == "rns_predict" && bit)
                    g_opt.rns_predict = iv;
                else if (test_hooks && k == "no_scratch" && bit)
                    g_opt.no_scratch = iv;
                else if (test_hooks && k == "rns_force_fallback" && bit)
                    g_opt.rns_force_fallback = iv;
                else if (test_hooks && k == "u32_e32" && is_num && lv >= 0 && (lv & ~0x1f000L) == 0)
                    g_opt.u32_e32 = iv; // a mask over the rings 2^12 .. 2^15; bit 16: the contiguous pass of larger rings
                else if (test_hooks && k == "two_sweep_big" && bit)
                    g_opt.two_sweep_big = iv;
                else if (test_hooks && k == "reset_predictions" && bit)
                    g_reset_predictions.store(1, std::memory_order_relaxed); // rns_guess forgets every stack it has seen
                else
                    return false;
                g_opt_generation.fetch_add(1, std::memory_order_release);
                return true;
            }
        } // namespace

        bool set_option(const char* name, const char* value) { return set_option_impl(name, value, false); }
        bool set_test_hook(const char* name, const char* value) { return set_option_impl(name, value, true); }
        unsigned lazy_e32_mask() { return static_cast<unsigned>(options().u32_e32); }
        int forced_path() { return options().path; }
        bool lazy_two_sweep_big() { return options().two_sweep_big != 0; }
        static bool rns_predict_enabled() { return options().rns_predict != 0; }
        static bool rns_force_fallback() { return options().rns_force_fallback != 0; }
        bool check_4step_tables() { return options().check_4step != 0; }

        // ---- launch log (test hook): which kernels did a call enqueue? -----------------------------------------------------
        // Every kernel launch of the library goes through GPUNTT_LAUNCH (launch.hpp), which reports the kernel expression of
        // its call site here.  Off (one relaxed load per launch) unless a test switched it on through
        // gpuntt_test_launch_log_start(); the dispatch-table test (tests/test_gpu_dispatch_table.py) compares what a call
        // enqueued with the row of DESIGN.md's table.
        namespace
        {
            std::atomic<int> g_log_on{0};
            std::mutex g_log_mutex;
            std::vector<std::string> g_log;
        } // namespace
        void note_launch(const char* kernel_expr, int family)
        {
            if (g_log_on.load(std::memory_order_relaxed) == 0)
                return;
            std::string k(kernel_expr);
            // "(kern::merge_pass_lazy<T, TLOG, ...>)" -> "merge_pass_lazy"
            const size_t a = k.find("kern::");
            if (a != std::string::npos)
                k = k.substr(a + 6);
            const size_t b = k.find_first_of("<)( ,");
            if (b != std::string::npos)
                k = k.substr(0, b);
            if (family >= 0)
                k += ":" + std::to_string(family);
            std::lock_guard<std::mutex> lock(g_log_mutex);
            g_log.push_back(k);
        }
        void launch_log_start()
        {
            std::lock_guard<std::mutex> lock(g_log_mutex);
            g_log.clear();
            g_log_on.store(1, std::memory_order_relaxed);
        }
        std::string launch_log_take()
        {
            std::lock_guard<std::mutex> lock(g_log_mutex);
            g_log_on.store(0, std::memory_order_relaxed);
            std::string out;
            for (const std::string& e : g_log)
                out += (out.empty() ? "" : " ") + e;
            g_log.clear();
            return out;
        }

        namespace
        {
            // One scratch chain per (device, stream) for eager calls and one per (device, stream, capture) for calls made
            // while the stream is being captured into a hipGraph.  A buffer of an EAGER chain that has ever been handed out
            // is never freed, and never handed to another chain, before GPU_NTT_ReleaseWorkspaces(): kernels already
            // enqueued keep reading it.  A buffer of a CAPTURE chain lives as long as the graph it was captured into and the
            // executables made from it (below).  Growth allocates a new buffer and RETIRES the old one (steps of at least
            // 1.5 x, so the retired buffers of a chain add up to less than twice the live one).  A captured call never
            // shares its buffer with eager calls: replaying the graph on another stream while eager calls run on the
            // capture stream touches two different buffers.
            using SlotKey = std::tuple<int, hipStream_t, unsigned long long>;
            struct Slot
            {
                SlotKey key{};
                void* ptr = nullptr; // start of the buffer = its header (WS_HEADER bytes); the user area lies behind it
                size_t bytes = 0;    // size of the user area
                unsigned long long seq = 0; // calls that took a veto epoch from this chain (lazy_workspace_veto)
                std::vector<void*> retired; // outgrown buffers, freed by release_workspaces() only
                std::recursive_mutex mu;    // held by a host thread for the duration of one API call
            };
            constexpr size_t WS_HEADER = 256;
            std::mutex g_ws_mutex; // guards the map itself (and g_capture_pool, g_ws_stats)
            std::map<SlotKey, Slot> g_ws;
            thread_local std::vector<std::recursive_mutex*> t_held;

            // ---- capture chains belong to their GRAPH (round 6) ------------------------------------------------------
            // A buffer allocated for a call that is being captured is retained by the graph through a hipUserObject; when the
            // graph AND every executable instantiated from it are gone (ROCm keeps the object alive for the executables like
            // CUDA does: tools/probe_userobject.hip), the object's destructor reports the buffer dead.  Dead buffers are not
            // freed (hipFree synchronises the device) but POOLED: the next capture chain that needs a buffer takes one that is
            // large enough.  A program that re-captures periodically therefore stops growing after its first captures, and the
            // map node of a dead chain is erased.  The destructor runs on a runtime thread and must not call HIP: it only
            // appends to g_dead; lazy_workspace() and release_workspaces() drain that list.
            struct DeadBuffer
            {
                SlotKey key;
                void* ptr;
                size_t total; // bytes of the allocation (header + user area)
            };
            // (both on the heap and never destroyed: the runtime may destroy a leaked graph, and call its user objects'
            // destructors, after this library's static objects are gone)
            std::mutex& g_dead_mutex = *new std::mutex;
            std::vector<DeadBuffer>& g_dead = *new std::vector<DeadBuffer>; // guarded by g_dead_mutex
            std::vector<std::pair<void*, size_t>> g_capture_pool;   // (pointer, bytes of the allocation); guarded by g_ws_mutex
            struct WsStats
            {
                unsigned long long graph_owned = 0, died = 0, pooled_now = 0, reused = 0, chains_erased = 0;
            } g_ws_stats; // guarded by g_ws_mutex
            void capture_buffer_dead(void* p)
            {
                DeadBuffer* d = static_cast<DeadBuffer*>(p);
                if (d->ptr != nullptr) // (null: the graph never took the object, lazy_workspace() keeps the buffer)
                {
                    std::lock_guard<std::mutex> lock(g_dead_mutex);
                    g_dead.push_back(*d);
                }
                delete d;
            }
            // hands the buffers of dead graphs to the pool and erases chains that own nothing any more.  Never blocks on a
            // chain's lock (a chain that is busy keeps its entry for the next drain).
            void drain_dead_captures()
            {
                std::vector<DeadBuffer> dead;
                {
                    std::lock_guard<std::mutex> lock(g_dead_mutex);
                    if (g_dead.empty())
                        return;
                    dead.swap(g_dead);
                }
                std::vector<DeadBuffer> again;
                {
                    std::lock_guard<std::mutex> lock(g_ws_mutex);
                    for (const DeadBuffer& d : dead)
                    {
                        auto it = g_ws.find(d.key);
                        if (it == g_ws.end())
                            continue; // (the chain is gone: release_workspaces() freed the buffer meanwhile)
                        Slot& s = it->second;
                        std::unique_lock<std::recursive_mutex> sl(s.mu, std::try_to_lock);
                        if (!sl.owns_lock())
                        {
                            again.push_back(d);
                            continue;
                        }
                        bool found = false;
                        if (s.ptr == d.ptr)
                        {
                            s.ptr = nullptr;
                            s.bytes = 0;
                            found = true;
                        }
                        else
                            for (auto r = s.retired.begin(); r != s.retired.end(); ++r)
                                if (*r == d.ptr)
                                {
                                    s.retired.erase(r);
                                    found = true;
                                    break;
                                }
                        if (found) // (not found: release_workspaces() freed it and the address may belong to someone else now)
                        {
                            g_capture_pool.emplace_back(d.ptr, d.total);
                            g_ws_stats.died++;
                        }
                        if (s.ptr == nullptr && s.retired.empty() && std::get<2>(d.key) != 0ull)
                        {
                            sl.unlock();
                            sl.release();
                            g_ws.erase(it); // a capture's key is never used again once the capture has ended
                            g_ws_stats.chains_erased++;
                        }
                    }
                    g_ws_stats.pooled_now = g_capture_pool.size();
                }
                if (!again.empty())
                {
                    std::lock_guard<std::mutex> lock(g_dead_mutex);
                    g_dead.insert(g_dead.end(), again.begin(), again.end());
                }
            }

            // 0: the stream is not being captured; else a key unique to the capture.  (The legacy default stream cannot be
            // captured, and asking about it while another stream captures in global mode is itself an error.)
            unsigned long long capture_key(hipStream_t stream)
            {
                if (stream == nullptr)
                    return 0ull;
                hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
                unsigned long long id = 0;
                // A query that FAILS leaves "is this call being captured?" unknown, and guessing "no" would put a captured call
                // on the eager chain: its epoch baked into the graph, no memset node for the veto word, its kernels replayed
                // over the eager calls' scratch.  The call cannot be served safely: it throws.
                GPUNTT_HIP_CHECK(hipStreamGetCaptureInfo(stream, &st, &id));
                if (st == hipStreamCaptureStatusInvalidated)
                    throw std::invalid_argument("the stream's capture has been invalidated");
                return st == hipStreamCaptureStatusActive ? (id | (1ull << 63)) : 0ull;
            }

            // the chain of (current device, stream, capture state).  Inside a WorkspaceScope -- one API call -- neither can
            // change, so the answer of the first lookup is kept (a drop-in call asks three or four times: three HIP runtime
            // calls and a map lookup each)
            struct SlotCache
            {
                hipStream_t stream = nullptr;
                Slot* slot = nullptr;
                bool capturing = false;
            };
            thread_local SlotCache t_slot_cache;

            Slot& slot_of(hipStream_t stream, bool* capturing = nullptr)
            {
                if (t_scope_depth > 0 && t_slot_cache.slot != nullptr && t_slot_cache.stream == stream)
                {
                    if (capturing != nullptr)
                        *capturing = t_slot_cache.capturing;
                    return *t_slot_cache.slot;
                }
                int dev = 0;
                GPUNTT_HIP_CHECK(hipGetDevice(&dev));
                const unsigned long long cap = capture_key(stream);
                if (capturing != nullptr)
                    *capturing = cap != 0ull;
                Slot* sp;
                {
                    std::lock_guard<std::mutex> lock(g_ws_mutex);
                    // map nodes never move; the node of a CAPTURE chain is erased once its graph is gone and it owns nothing
                    // (drain_dead_captures) -- no call can be using it then, a capture's key dies with the capture
                    const SlotKey key = std::make_tuple(dev, stream, cap);
                    sp = &g_ws[key];
                    sp->key = key;
                }
                if (t_scope_depth > 0)
                    t_slot_cache = SlotCache{stream, sp, cap != 0ull};
                return *sp;
            }

            // takes the chain's lock: for the rest of the call inside a WorkspaceScope, else until the guard dies
            struct SlotLock
            {
                std::recursive_mutex* m;
                explicit SlotLock(Slot& s) : m(&s.mu)
                {
                    m->lock();
                    if (t_scope_depth > 0)
                    {
                        t_held.push_back(m); // released by the outermost WorkspaceScope of this thread
                        m = nullptr;
                    }
                }
                ~SlotLock()
                {
                    if (m != nullptr)
                        m->unlock();
                }
            };
        } // namespace

        WorkspaceScope::WorkspaceScope()
        {
            if (t_scope_depth++ == 0)
                refresh_snapshot(); // the options this call runs under
        }
        WorkspaceScope::~WorkspaceScope()
        {
            if (--t_scope_depth == 0)
            {
                for (auto it = t_held.rbegin(); it != t_held.rend(); ++it)
                    (*it)->unlock();
                t_held.clear();
                t_slot_cache = SlotCache{};
            }
        }

        void release_workspaces()
        {
            // lock order: never BLOCK on a slot's lock while holding the map's mutex -- a thread inside an API call holds
            // its slot and re-enters lazy_workspace() (map mutex) for the next buffer of the same call.  So: try each
            // chain under the map's mutex, take what can be taken, and come back for the chains that were busy.
            drain_dead_captures();
            std::vector<void*> doomed;
            for (;;)
            {
                bool busy = false;
                {
                    std::lock_guard<std::mutex> lock(g_ws_mutex);
                    for (auto it = g_ws.begin(); it != g_ws.end();)
                    {
                        Slot& s = it->second;
                        std::unique_lock<std::recursive_mutex> sl(s.mu, std::try_to_lock);
                        if (!sl.owns_lock())
                        {
                            busy = true;
                            ++it;
                            continue;
                        }
                        if (s.ptr != nullptr)
                            doomed.push_back(s.ptr);
                        s.ptr = nullptr;
                        s.bytes = 0;
                        doomed.insert(doomed.end(), s.retired.begin(), s.retired.end());
                        s.retired.clear();
                        // (the node stays: a thread may be between slot_of() and its lock; the user objects of graphs that
                        // are still alive will report buffers this chain no longer lists, and drain ignores those)
                        ++it;
                    }
                    for (auto& pb : g_capture_pool)
                        doomed.push_back(pb.first);
                    g_capture_pool.clear();
                    g_ws_stats.pooled_now = 0;
                }
                if (!busy)
                    break;
                std::this_thread::yield();
            }
            for (void* p : doomed)
                (void) hipFree(p); // synchronises the device
        }

        void scratch_stats(unsigned long long out[6])
        {
            drain_dead_captures();
            std::lock_guard<std::mutex> lock(g_ws_mutex);
            out[0] = g_ws_stats.graph_owned;   // buffers handed to a graph (hipGraphRetainUserObject)
            out[1] = g_ws_stats.died;          // of those, reported dead by their graph and moved to the pool
            out[2] = g_capture_pool.size();    // buffers waiting in the pool
            out[3] = g_ws_stats.reused;        // capture-chain buffers taken from the pool instead of hipMalloc
            out[4] = g_ws_stats.chains_erased; // map nodes of dead capture chains erased
            out[5] = g_ws.size();              // chains alive
        }

        // ---- which lazy family will a drop-in RNS call need?  (RnsGuess, lazy_launch.hpp) ------------------------------
        namespace
        {
            struct GuessSlot
            {
                int word = -1; // index of the slot's word in its device's pool of host-mapped words
                unsigned predicted = kern::GO_LAZY;
                bool have_prediction = false;
                int streak = 0;                          // calls in a row that needed a narrower family than the predicted one
                unsigned streak_state = kern::GO_LAZY;   // the widest of them
                unsigned long long last_use = 0;
                bool generic_seen = false; // the last finished call found a modulus outside the lazy families' domain
            };
            // one pinned, device-mapped allocation per device: GUESS_MAX_KEYS words, one cache line apart.  Never freed
            // (captured graphs keep writing their state to the word they were captured with; a word that has been handed to
            // another stack meanwhile only costs that stack a wrong prediction, which the generic kernels behind it absorb)
            struct GuessPool
            {
                unsigned* host = nullptr;
                unsigned* dev = nullptr;
            };
            constexpr size_t GUESS_MAX_KEYS = 256;
            constexpr size_t GUESS_STRIDE = 16; // words
            constexpr unsigned STATE_UNKNOWN = 0xffffffffu;
            std::mutex g_guess_mutex;
            std::map<int, GuessPool> g_guess_pool;
            // (device, moduli, mod_order, mod_count, word size | 4-step entry | direction): forward and inverse calls of one
            // stack may need different families (31 q serves forward transforms only), a *_Modulus_Ordered call uses the
            // subset its order array names; the Merge entry points otherwise share one slot per stack
            std::map<std::tuple<int, const void*, const void*, int, int>, GuessSlot> g_guess;
            // what stacks of the same SHAPE (device, mod_count, word size | entry | direction, ordered or not) needed last,
            // whatever buffer they lived in: the first prediction for a moduli buffer never seen before -- a caller that
            // uploads its stack to a fresh buffer per call keeps its family
            std::map<std::tuple<int, int, int, bool>, unsigned> g_shape_hint;
            unsigned long long g_guess_clock = 0;
        } // namespace

        static bool rns_predict_enabled(); // option rns_predict (defined with the options below)
        static bool rns_force_fallback();  // option rns_force_fallback (test hook)
        RnsGuess rns_guess(const void* moduli_device, int mod_count, int word_bytes, bool inverse, const void* order,
                           bool exact)
        {
            RnsGuess gss{kern::GO_LAZY, true, nullptr};
            if (!rns_predict_enabled() || forced_path() == 3)
                return gss; // every family (path = fast-strict: the tests want the lazy families to own the call)
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess)
                return gss;
            std::lock_guard<std::mutex> lock(g_guess_mutex);
            if (g_reset_predictions.exchange(0, std::memory_order_relaxed) != 0)
            {
                g_guess.clear(); // (the host-mapped words stay: graphs captured earlier keep writing to theirs)
                g_shape_hint.clear();
            }
            GuessPool& pool = g_guess_pool[dev];
            if (pool.host == nullptr)
            {
                void* hp = nullptr;
                void* dp = nullptr;
                if (hipHostMalloc(&hp, sizeof(unsigned) * GUESS_STRIDE * GUESS_MAX_KEYS, hipHostMallocMapped) != hipSuccess ||
                    hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess)
                {
                    (void) hipGetLastError(); // (not during a stream capture, or no pinned memory left): every family
                    if (hp != nullptr)
                        (void) hipHostFree(hp);
                    return gss;
                }
                pool.host = static_cast<unsigned*>(hp);
                pool.dev = static_cast<unsigned*>(dp);
                for (size_t i = 0; i < GUESS_MAX_KEYS; i++)
                    reinterpret_cast<volatile unsigned*>(pool.host)[i * GUESS_STRIDE] = STATE_UNKNOWN;
            }
            const auto key = std::make_tuple(dev, moduli_device, order, mod_count, word_bytes | (inverse ? 0x100 : 0));
            auto it = g_guess.find(key);
            if (it == g_guess.end())
            {
                GuessSlot slot;
                size_t on_dev = 0;
                auto oldest = g_guess.end();
                for (auto jt = g_guess.begin(); jt != g_guess.end(); ++jt)
                    if (std::get<0>(jt->first) == dev)
                    {
                        on_dev++;
                        if (oldest == g_guess.end() || jt->second.last_use < oldest->second.last_use)
                            oldest = jt;
                    }
                if (on_dev < GUESS_MAX_KEYS)
                    slot.word = static_cast<int>(on_dev); // words are handed out in order and only recycled below
                else
                {
                    slot.word = oldest->second.word; // table full: the least recently used stack gives up its word
                    g_guess.erase(oldest);
                }
                reinterpret_cast<volatile unsigned*>(pool.host)[slot.word * GUESS_STRIDE] = STATE_UNKNOWN;
                const auto hint = g_shape_hint.find(std::make_tuple(dev, mod_count, std::get<4>(key), order != nullptr));
                if (hint != g_shape_hint.end())
                    slot.predicted = hint->second;
                it = g_guess.emplace(key, slot).first;
            }
            GuessSlot& s = it->second;
            s.last_use = ++g_guess_clock;
            const unsigned seen = reinterpret_cast<volatile unsigned*>(pool.host)[s.word * GUESS_STRIDE];
            if (seen != STATE_UNKNOWN && seen <= kern::GO_LAZY_31Q && (exact || seen != kern::GO_GENERIC))
            {
                // the state some earlier call of this stack found (the last one that has finished).  The enqueued family
                // serves every narrower stack too, so: widen at once, narrow after 16 calls in a row that needed less
                const int r_seen = kern::family_rank(seen), r_now = kern::family_rank(s.predicted);
                if (!s.have_prediction || r_seen > r_now || exact)
                {
                    s.predicted = seen;
                    s.streak = 0;
                }
                else if (r_seen == r_now)
                    s.streak = 0;
                else
                {
                    if (s.streak == 0 || r_seen > kern::family_rank(s.streak_state))
                        s.streak_state = seen;
                    if (++s.streak >= 16)
                    {
                        s.predicted = s.streak_state;
                        s.streak = 0;
                    }
                }
                s.have_prediction = true;
                s.generic_seen = false;
                g_shape_hint[std::make_tuple(dev, mod_count, std::get<4>(key), order != nullptr)] = s.predicted;
            }
            else if (seen == kern::GO_GENERIC)
                s.generic_seen = true;
            gss.unsure = !s.have_prediction || s.generic_seen;
            gss.state_out = pool.dev + s.word * GUESS_STRIDE;
            gss.all_families = false;
            gss.state = s.predicted;
            return gss;
        }

        // host twin of kern::recip_norm
        template <typename T> static T recip_norm_host(T q)
        {
            if (q < 3 || (q & (q - 1)) == 0)
                return 0;
            constexpr int W = static_cast<int>(8 * sizeof(T));
            int b = 0;
            while (b < W && (static_cast<unsigned long long>(q) >> b) != 0)
                b++;
            return static_cast<T>((static_cast<unsigned __int128>(1) << (W - 1 + b)) / q);
        }

        int lazy_contig_k(int n)
        {
            // The strided pass is HBM-bound with idle VALU slots while the contiguous pass is
            // VALU-bound, so up to 6 stages (the most a strided tile keeps wave-uniform twiddles
            // for) are moved in front: 2^16 = 6 + 10 measured 2-3 % faster than 4 + 12 (stage splits 8 + 8 ... 4 + 12 lie
            // within 3.5 %, profiles/r03_c2_stage_split.txt).
            if (n > 12 && n <= 18)
                return (n - 6 > 10) ? (n - 6) : 10;
            return 12;
        }

        namespace
        {
            // the graph being captured on `stream` retains the buffer: capture_buffer_dead() runs when the graph and all
            // executables made from it are destroyed.  Failure at any step leaves the buffer with the chain for good, which
            // is what every buffer did before round 6.
            void give_to_graph(hipStream_t stream, const SlotKey& key, void* ptr, size_t total)
            {
                hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
                unsigned long long id = 0;
                hipGraph_t graph = nullptr;
                const hipGraphNode_t* deps = nullptr;
                size_t ndeps = 0;
                if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &ndeps) != hipSuccess ||
                    st != hipStreamCaptureStatusActive || graph == nullptr)
                {
                    (void) hipGetLastError();
                    return;
                }
                DeadBuffer* d = new DeadBuffer{key, ptr, total};
                hipUserObject_t obj = nullptr;
                if (hipUserObjectCreate(&obj, d, capture_buffer_dead, 1, hipUserObjectNoDestructorSync) != hipSuccess)
                {
                    (void) hipGetLastError();
                    delete d;
                    return;
                }
                if (hipGraphRetainUserObject(graph, obj, 1, hipGraphUserObjectMove) != hipSuccess)
                {
                    (void) hipGetLastError();
                    d->ptr = nullptr; // the destructor reports nothing
                    (void) hipUserObjectRelease(obj, 1);
                    return;
                }
                std::lock_guard<std::mutex> lock(g_ws_mutex);
                g_ws_stats.graph_owned++;
            }
        } // namespace

        void* lazy_workspace(hipStream_t stream, size_t bytes, bool or_null)
        {
            if (or_null && options().no_scratch != 0)
                return nullptr; // test hook: the out-of-memory fall-back of the drop-in entry points
            drain_dead_captures();
            bool capturing = false;
            Slot& s = slot_of(stream, &capturing);
            SlotLock lock(s);
            if (s.bytes < bytes)
            {
                // grow: a NEW buffer; the old one is retired, not freed -- earlier calls on this stream may still be
                // reading it, and a graph captured from them may be replayed at any time (no synchronisation either)
                size_t want = bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes;
                if (want < s.bytes + s.bytes / 2)
                    want = s.bytes + s.bytes / 2;
                want = (want + 255u) & ~size_t(255);
                // a capture in global mode refuses hipMalloc: allocate under the relaxed mode, as the capture rules allow
                // for calls that do not touch the capturing stream
                hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
                if (capturing)
                    (void) hipThreadExchangeStreamCaptureMode(&mode);
                void* fresh = nullptr;
                hipError_t err = hipSuccess;
                if (capturing)
                {
                    // a buffer whose graph is gone, if one is large enough (best fit): re-capturing programs stop growing
                    std::lock_guard<std::mutex> lock(g_ws_mutex);
                    auto best = g_capture_pool.end();
                    for (auto it = g_capture_pool.begin(); it != g_capture_pool.end(); ++it)
                        if (it->second >= WS_HEADER + want && (best == g_capture_pool.end() || it->second < best->second))
                            best = it;
                    if (best != g_capture_pool.end())
                    {
                        fresh = best->first;
                        want = best->second - WS_HEADER;
                        g_capture_pool.erase(best);
                        g_ws_stats.reused++;
                        g_ws_stats.pooled_now = g_capture_pool.size();
                    }
                }
                if (fresh == nullptr)
                    err = hipMalloc(&fresh, WS_HEADER + want);
                if (capturing)
                    (void) hipThreadExchangeStreamCaptureMode(&mode);
                if (err == hipSuccess && capturing)
                    give_to_graph(stream, s.key, fresh, WS_HEADER + want);
                // header: the veto word of the 4-step table check (lazy_workspace_veto) starts at "no call yet" = all ones
                if (err == hipSuccess)
                {
                    err = hipMemsetAsync(fresh, 0xff, WS_HEADER, stream);
                    if (err != hipSuccess)
                        (void) hipFree(fresh);
                }
                if (err != hipSuccess)
                {
                    // out of device memory: the caller falls back to the kernels that need no scratch
                    (void) hipGetLastError();
                    if (or_null)
                        return nullptr;
                    GPUNTT_HIP_CHECK(err);
                }
                if (s.ptr != nullptr)
                    s.retired.push_back(s.ptr);
                s.ptr = fresh;
                s.bytes = want;
                s.seq = 0;
            }
            return static_cast<unsigned char*>(s.ptr) + WS_HEADER;
        }

        // The 64-bit veto word in the header of the chain's buffer (the one lazy_workspace() returned last) and a fresh
        // epoch for it: kernels publish  ((~epoch) << 32) | state  with atomicMin, so the word always holds the state of
        // the LATEST call on the chain -- smallest high half -- and, within that call, the smallest state any thread
        // published (kern::GO_GENERIC = 0 = "the table check failed" beats every other).  No per-call reset: the word is
        // set to all ones when the buffer is allocated and whenever the 32-bit epoch wraps.  Call after lazy_workspace().
        void lazy_workspace_veto(hipStream_t stream, unsigned long long** word, unsigned* epoch)
        {
            Slot& s = slot_of(stream);
            SlotLock lock(s);
            if (s.ptr == nullptr)
                throw std::logic_error("internal: veto word requested before the scratch buffer");
            const unsigned e = static_cast<unsigned>(s.seq & 0xffffffffull);
            if (s.seq != 0 && e == 0u)
                GPUNTT_HIP_CHECK(hipMemsetAsync(s.ptr, 0xff, WS_HEADER, stream)); // epoch wrapped: start over
            s.seq++;
            *word = static_cast<unsigned long long*>(s.ptr);
            *epoch = e;
        }

        template <typename T>
        void launch_prep(const T* roots, lazy::Tw<T>* ws, const Modulus<T>* mods, T q, int mod_count, int n,
                         bool negacyclic, int perm_tile_log, const T* ninv_arr, lazy::Tw<T>* ws_ninv,
                         unsigned* go_flag, lazy::NormConst* norm_arr, hipStream_t stream, const int* mod_order,
                         const T* fold_ninv_single, bool fold_ninv_rns, unsigned* host_state, bool allow_31q,
                         unsigned family, const kern::SlowArgs<T>* slow)
        {
            const unsigned long long entries = static_cast<unsigned long long>(mod_count) << n;
            unsigned grid = static_cast<unsigned>((entries + 255) / 256);
            kern::SlowArgs<T> sa{};
            if (slow != nullptr)
            {
                sa = *slow;
                sa.force = (rns_force_fallback() && sa.enabled) ? 1 : 0;
                // the fall-back transforms one polynomial per block: a tiny table must not leave it a one-block grid
                const unsigned long long want = sa.polys < 1024ull ? sa.polys : 1024ull;
                if (sa.enabled && grid < want)
                    grid = static_cast<unsigned>(want);
            }
            GPUNTT_LAUNCH((kern::prep_twiddles<T>), dim3(grid), dim3(256), 0, stream, roots, ws, mods, q,
                               (mods == nullptr) ? recip_norm_host<T>(q) : static_cast<T>(0), mod_count, n, negacyclic ? 1 : 0, perm_tile_log, ninv_arr, ws_ninv, go_flag, norm_arr, mod_order,
                               fold_ninv_single ? *fold_ninv_single : static_cast<T>(0),
                               (fold_ninv_single != nullptr || (fold_ninv_rns && ninv_arr != nullptr)) ? 1 : 0, host_state,
                               allow_31q ? 1 : 0, family, sa);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
        template <typename T>
        void launch_prep_merge_from_fourstep(const T* n1_table, const T* w_table, lazy::Tw<T>* ws, int log_n1, int log_n2,
                                             int perm_tile_log, bool inverse, bool fold, T q, T ninv,
                                             const Modulus<T>* mods, const T* ninv_dev, lazy::Tw<T>* ws_ninv,
                                             unsigned* go_flag, lazy::NormConst* norm_arr, hipStream_t stream,
                                             unsigned* host_state, const FourStepVeto& veto, const T* n2_table)
        {
            const unsigned long long count = 1ull << (log_n1 + log_n2);
            const unsigned grid = static_cast<unsigned>((count + 255) / 256);
            GPUNTT_LAUNCH((kern::prep_merge_from_fourstep<T>), dim3(grid), dim3(256), 0, stream, n1_table, w_table, ws,
                               log_n1, log_n2, perm_tile_log, inverse ? 1 : 0, fold ? 1 : 0, q,
                               mods ? static_cast<T>(0) : recip_norm_host<T>(q), ninv, mods, ninv_dev, ws_ninv,
                               veto.word != nullptr ? nullptr : go_flag, norm_arr, host_state,
                               veto.check ? n2_table : nullptr, veto.word, veto.epoch);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
        template void launch_prep_merge_from_fourstep<uint64_t>(const uint64_t*, const uint64_t*, lazy::Tw64*, int, int, int,
                                                                bool, bool, uint64_t, uint64_t, const Modulus<uint64_t>*,
                                                                const uint64_t*, lazy::Tw64*, unsigned*, lazy::NormConst*,
                                                                hipStream_t, unsigned*, const FourStepVeto&, const uint64_t*);
        template void launch_prep_merge_from_fourstep<uint32_t>(const uint32_t*, const uint32_t*, lazy::Tw32*, int, int, int,
                                                                bool, bool, uint32_t, uint32_t, const Modulus<uint32_t>*,
                                                                const uint32_t*, lazy::Tw32*, unsigned*, lazy::NormConst*,
                                                                hipStream_t, unsigned*, const FourStepVeto&, const uint32_t*);


        template <typename T> void debug_recip_norm(const T* q, T* out, unsigned long long count, hipStream_t stream)
        {
            if (count == 0)
                return;
            GPUNTT_LAUNCH((kern::debug_recip_norm<T>), dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, stream, q,
                               out, count);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
        template void debug_recip_norm<uint64_t>(const uint64_t*, uint64_t*, unsigned long long, hipStream_t);
        template void debug_recip_norm<uint32_t>(const uint32_t*, uint32_t*, unsigned long long, hipStream_t);

        template void launch_prep<uint64_t>(const uint64_t*, lazy::Tw64*, const Modulus<uint64_t>*, uint64_t, int,
                                            int, bool, int, const uint64_t*, lazy::Tw64*, unsigned*, lazy::NormConst*, hipStream_t, const int*,
                                            const uint64_t*, bool, unsigned*, bool, unsigned, const kern::SlowArgs<uint64_t>*);
        template void launch_prep<uint32_t>(const uint32_t*, lazy::Tw32*, const Modulus<uint32_t>*, uint32_t, int,
                                            int, bool, int, const uint32_t*, lazy::Tw32*, unsigned*, lazy::NormConst*, hipStream_t, const int*,
                                            const uint32_t*, bool, unsigned*, bool, unsigned, const kern::SlowArgs<uint32_t>*);
    } // namespace host
} // namespace gpuntt
