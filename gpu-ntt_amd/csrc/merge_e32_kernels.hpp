// merge_e32_kernels.hpp -- the second register geometry of the fast 32-bit kernels: 32 coefficients per lane (gfx950).
//
// A ring of 2^TLOG 32-bit coefficients (TLOG = 12 .. 15) that fits ONE tile is transformed in one HBM sweep by
// 2^(TLOG - 5) threads holding 32 coefficients each: 5 radix-2 stages per register round instead of 4, so the 14
// stages of the ring 2^14 are 4 + 5 + 5 (two LDS exchanges, one of them wave-local) where the 16-coefficient
// geometry of merge_lazy_kernels.hpp needs 2 + 4 + 4 + 4 (three), and the ring 2^15 fits a 32768-coefficient tile
// (1024 threads, 144 KiB of LDS) -- one sweep instead of two.  4 waves per SIMD, a 128-VGPR budget.
//
//   forward (Cooley-Tukey, natural in -> bit-reversed out)
//     round A  stages TLOG-1 .. 10   window bits [TLOG-5, TLOG)   coalesced dword loads, twiddles in SGPRs (block-uniform)
//     exchange through LDS (block barrier)
//     round B  stages 9 .. 5         window bits [5, 10)          per-lane twiddles, 16-byte loads of consecutive pairs
//     exchange through LDS inside the wave's own 2048-coefficient sub-block (no barrier)
//     round C  stages 4 .. 0         window bits [0, 5)           32 contiguous coefficients per lane, ds_read_b128;
//                                                                 per-lane twiddles, 16-byte loads from the prepared
//                                                                 [k][16-coefficient group] layout of prep.hip
//     normalisation, wave-local transposition, 1 KiB runs per store instruction
//   inverse (Gentleman-Sande): the mirror image, n^-1 folded into the last stage.
//
// LDS index of tile element e: e + 4 * (e >> 5) -- the 32 lanes of a half-wave that hold consecutive e hit 32
// different banks (ds_read_b32 / ds_write_b32), and the 16-byte accesses of lane t at 36 t + 4 k fall into 16 different
// 16-byte slots for each of the instruction's 16-lane groups (9 t mod 16 is a permutation of those groups).
//
// The prepared twiddle table is the one the 16-coefficient kernels read (same slots, same permutation of the
// distance-1/2/4 stages: two neighbouring 16-coefficient groups are one lane here, so their two pairs are one
// 16-byte load), so the two geometries are interchangeable per launch.
// Replaces reference ForwardCore / InverseCore for these rings (src/lib/ntt_merge/ntt.cu:596-761, 1086-1318; its plan
// for 2^14 / 2^15 is two kernels, src/include/gpuntt/ntt_merge/ntt.cuh:628-633).
#pragma once

#include "merge_lazy_kernels.hpp"

namespace gpuntt
{
    namespace kern
    {
        constexpr int R5 = 5;
        constexpr int E32 = 1 << R5;

        template <int TLOG> struct ETile
        {
            static_assert(TLOG >= 12 && TLOG <= 15, "one-tile rings of 2^12 .. 2^15 coefficients");
            static constexpr int TL = TLOG;
            static constexpr int NT = 1 << (TLOG - R5);
            static constexpr int TILE = 1 << TLOG;
            static constexpr int LDS_ELEMS = TILE + (TILE >> 3);
            static constexpr int TWB_WORDS = (NT / 64) * 128; // round-B twiddles: 64 pairs (512 B) per wave, behind the tile
            static constexpr int NA = TLOG - 10;   // stages of round A (2 .. 5)
            static constexpr int WLA = TLOG - R5;  // its register window starts here
        };
        __device__ __forceinline__ int epad(int e) { return e + ((e >> 5) << 2); }
        // register j of the window [WL, WL + 5) as a compile-time LDS offset from the lane's base (WL >= 5)
        template <int WL> constexpr int ejoff(int j) { return (j << WL) + ((j << (WL - 5)) << 2); }

        // ---- compile-time schedules of the range corrections ----------------------------------------------------------
        // forward: every register carries the same bound (both outputs of a Cooley-Tukey butterfly have bound U + TB)
        template <int TLOG, int LIMIT, int TB, int IN_BOUND = 1> struct EFwdSched
        {
            struct Data
            {
                int ku[TLOG];
                int final_bound;
            };
            static constexpr Data make()
            {
                Data d{};
                int b = IN_BOUND;
                for (int s = 0; s < TLOG; s++)
                {
                    // ku > 0: conditional subtraction of ku * q; ku = -1: quotient estimate (Mod32::reduce_2q: any word ->
                    // [0, 2q), two instructions like the conditional subtraction) where that buys a stage more than
                    // halving the bound does -- 8 q range: a correction every THIRD stage instead of every second
                    if (b + TB > LIMIT)
                    {
                        const int k = lazy::csub_k(b);
                        d.ku[s] = (k > 2) ? -1 : k;
                        b = (k > 2) ? 2 : k;
                    }
                    b += TB;
                }
                d.final_bound = b;
                return d;
            }
            static constexpr Data d = make();
            static_assert(d.final_bound <= LIMIT, "lazy bound exceeds the headroom");
        };
        // inverse: bounds per register, conservative (max) across an exchange -- kern::PassSched's rule with 32 registers
        template <int TLOG, int LIMIT, int TB> struct EInvSched
        {
            static constexpr int NA = ETile<TLOG>::NA;
            struct Data
            {
                int ku[3][R5][E32 / 2];
                int kv[3][R5][E32 / 2];
                int c[3][R5][E32 / 2];
                int ko[3][R5][E32 / 2];
            };
            // round 0: stages 0 .. 4 (window [0, 5)), round 1: stages 5 .. 9 (window [5, 10)), round 2: stages 10 .. TLOG-1
            // (window [TLOG-5, TLOG): register bit of stage p is p - (TLOG - 5))
            static constexpr int stages_of(int r) { return r == 2 ? NA : R5; }
            static constexpr int jb_of(int r, int s) { return r == 2 ? (10 + s - (TLOG - R5)) : s; }
            static constexpr Data make()
            {
                Data d{};
                int bound_in = 1;
                for (int r = 0; r < 3; r++)
                {
                    int b[E32] = {};
                    for (int j = 0; j < E32; j++)
                        b[j] = bound_in;
                    for (int s = 0; s < stages_of(r); s++)
                    {
                        const int jb = jb_of(r, s);
                        for (int h = 0; h < E32 / 2; h++)
                        {
                            const int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                            const int j1 = j0 | (1 << jb);
                            lazy::GsPlan pl = lazy::gs_plan(b[j0], b[j1], LIMIT);
                            // ko = -1: the sum is corrected by the quotient estimate (Mod32::reduce_2q: -> [0, 2q)) where a
                            // conditional subtraction would only halve its bound -- 8 q range: a sum of two 4 q values every
                            // SECOND stage of a run of sums instead of every stage
                            if (pl.ko > 2)
                            {
                                pl.ko = -1;
                                pl.out_u = 2;
                            }
                            d.ku[r][s][h] = pl.ku;
                            d.kv[r][s][h] = pl.kv;
                            d.c[r][s][h] = pl.c;
                            d.ko[r][s][h] = pl.ko;
                            b[j0] = pl.out_u;
                            b[j1] = TB;
                        }
                    }
                    int mx = 0;
                    for (int j = 0; j < E32; j++)
                        mx = b[j] > mx ? b[j] : mx;
                    bound_in = mx;
                }
                return d;
            }
            static constexpr Data d = make();
        };

        // Every global access of these kernels is a BUFFER instruction: one resource per tile / per modulus table in scalar
        // registers, ONE 32-bit lane offset in a vector register shared by all accesses of a kind, the block-uniform part of
        // the address in the instruction's scalar offset.  (With flat global addressing the compiler folds the uniform
        // part into 64-bit VALU adds of the lane's address: 64 instructions for the 32 loads of a tile.)
        constexpr int BUF_RSRC_WORD3 = 0x00020000; // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
        constexpr int BUF_NT = 2;                  // cache policy: non-temporal (streaming input / output)
        __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes)
        {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), BUF_RSRC_WORD3);
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        // 16-byte load of two consecutive prepared pairs
        struct alignas(16) Tw32x2
        {
            lazy::Tw32 a, b;
        };

        // PART = false: one tile = one polynomial (ring 2^TLOG), grid = polynomials of the call.
        // PART = true: the tile is one of the 2^(n - TLOG) tiles of a LARGER ring and the kernel is the contiguous pass of its
        // plan -- forward: the last pass (TLOG stages on lazy input below LIMIT q from the strided passes, canonical output);
        // inverse: the first pass (canonical input, lazy output below LIMIT / 2 for the strided passes behind it, no n^-1).
        // Blocks in merge_pass_lazy's order (poly-minor from 2^20, consecutive passes in opposite directions).
        template <int TLOG, bool INV, int LIM, bool PART = false>
        __global__ __launch_bounds__(ETile<TLOG>::NT, 4) void merge_ring_e32(LazyArgsT<uint32_t> a)
        {
            using T = uint32_t;
            using M = lazy::Mod<T, LIM>;
            using TW = lazy::Tw32;
            using G = ETile<TLOG>;
            constexpr int NT = G::NT, NA = G::NA, WLA = G::WLA;
            constexpr int NT16 = 1 << (TLOG - 4); // "threads" of the prepared layout's permutation (16-coefficient groups)
            constexpr bool LAST = !(PART && INV);       // canonical output
            constexpr bool CANON_IN = !(PART && !INV);  // canonical input
            __shared__ __attribute__((aligned(16))) T lds[G::LDS_ELEMS + G::TWB_WORDS];

            if (not_my_call<T, LIM>(a.go_flag, a.flags))
                return;
            const int t = threadIdx.x;
            const unsigned tu = threadIdx.x; // lane part of every global address: one 32-bit offset beside a uniform base
            const int n = PART ? a.n : TLOG;
            unsigned poly = blockIdx.x, tip = 0u; // polynomial and tile inside it
            if constexpr (PART)
            {
                const unsigned bx = (a.flags & F_REVERSE) ? (gridDim.x - 1u - blockIdx.x) : blockIdx.x;
                const int tiles_log = n - TLOG;
                if (a.batch > 1)
                    poly_minor_order(bx, static_cast<unsigned>(a.batch), tiles_log, poly, tip, a.flags);
                else
                {
                    poly = bx >> tiles_log;
                    tip = bx & ((1u << tiles_log) - 1u);
                }
                poly = uniform32(poly);
                tip = uniform32(tip);
            }
            T qv = a.q, qbit = a.q_bit, qmu = a.q_mu;
            int mi = 0;
            if (a.mods != nullptr)
            {
                mi = static_cast<int>(uniform32(poly % static_cast<unsigned>(a.mod_count)));
                const Modulus<T> md = a.mods[a.mod_order != nullptr ? a.mod_order[mi] : mi];
                qv = md.value;
                qbit = md.bit;
                qmu = md.mu;
            }
            M m;
            m.set(qv, (a.norm_arr != nullptr) ? a.norm_arr[mi] : a.norm);
            const TW* __restrict__ tw = a.tw + (static_cast<unsigned long long>(mi) << n);
            const __amdgpu_buffer_rsrc_t rtw = make_rsrc(tw, static_cast<unsigned>(sizeof(TW)) << n);
            // first slot of the tile's twiddles of the stage at tile bit p: stage slots [2^(n-1-p), 2^(n-p)), 2^(TLOG-1-p) per tile
            auto sbase = [&](int p) -> unsigned { return (1u << (n - 1 - p)) + (tip << (TLOG - 1 - p)); };
            const unsigned slot_poly = (a.poly_order != nullptr) ? static_cast<unsigned>(a.poly_order[poly]) : poly;
            const unsigned long long base =
                (static_cast<unsigned long long>(uniform32(slot_poly)) << n) + (static_cast<unsigned long long>(tip) << TLOG);
            // (src may alias dst: the whole tile is read before any store)
            const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(static_cast<const T*>(a.in) + base, sizeof(T) << TLOG);
            const __amdgpu_buffer_rsrc_t rdst = make_rsrc(a.out + base, sizeof(T) << TLOG);
            constexpr int POL_IN = CANON_IN ? BUF_NT : 0; // the hand-off between two passes should stay in the caches
            constexpr int POL_OUT = LAST ? BUF_NT : 0;
            auto tw_pair = [&](unsigned voff, unsigned slot) -> TW {
                const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rtw, static_cast<int>(voff), static_cast<int>(slot * 8u), 0);
                return TW{x.x, x.y};
            };
            auto tw_two = [&](unsigned voff, unsigned slot, TW& p0, TW& p1) {
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rtw, static_cast<int>(voff), static_cast<int>(slot * 8u), 0);
                p0 = TW{x.x, x.y};
                p1 = TW{x.z, x.w};
            };

            // ---- round B twiddles (stages 9 .. 5): staged in LDS per wave ----------------------------------------------------
            // The 32 lanes of a half-wave (one value of g = t >> 5) use the SAME 31 pairs -- 2^s consecutive pairs from slot
            // sbase(9 - s) + (g << s) for s = 0 .. 4 -- so a wave needs 62 pairs in all.  Lane l < 62 fetches one of them at the
            // start of the kernel and drops it into the wave's own 512-byte area; the butterflies read them back with 16-byte
            // broadcast reads, stage by stage.  (Per-lane loads of those pairs moved 127 KiB per tile through the vector
            // memory path for 4 KiB of distinct data: knocking them out made C4 5.7 % faster, tools/ko_e32.py; this gets
            // most of that back, frees 62 VGPRs across rounds A and B, and needs no barrier: the area is wave-private.)
            //   area layout (pairs): [h * 32 + 2^s + k] for half h, stage s, k < 2^s; index h * 32 is unused
            TW* const twb_area = reinterpret_cast<TW*>(lds + G::LDS_ELEMS) + (static_cast<unsigned>(t) >> 6) * 64u;
            auto stage_twb = [&]() {
                const unsigned l = tu & 63u, h = l >> 5, r = l & 31u; // r = 2^s + k, r = 0 unused
                if (r != 0u)
                {
                    const unsigned s_ = 31u - static_cast<unsigned>(__clz(r)), k = r - (1u << s_);
                    const unsigned g = ((tu >> 6) << 1) + h;
                    // slot of stage p = 9 - s_: sbase(p) + (g << s_) + k, with sbase(p) = 2^(n - 10 + s_) + (tip << (TLOG - 10 + s_))
                    const unsigned slot = (1u << (n - 10 + s_)) + (tip << (TLOG - 10 + s_)) + (g << s_) + k;
                    TW pr = tw_pair(slot << 3, 0u);
                    // inverse: the odd twiddles of every stage but the round's last feed butterflies whose outputs leave
                    // complemented (gs below), which multiply by the NEGATED twiddle -- negated here, once per tile
                    if (INV && s_ >= 1u && (k & 1u) != 0u)
                        pr.w = 0u - pr.w;
                    twb_area[l] = pr;
                }
            };
            // the 2^s pairs of stage s for this lane's half-wave
            auto read_twb = [&](auto s_, TW(&w)[16]) {
                constexpr int s = decltype(s_)::value;
                const TW* src = twb_area + ((tu >> 5) & 1u) * 32u + (1u << s);
                if constexpr (s == 0)
                    w[0] = src[0];
                else
                    static_for<(1 << s) / 2>([&](auto k_) {
                        constexpr int k = decltype(k_)::value;
                        const u32x4 x = *reinterpret_cast<const u32x4*>(src + 2 * k);
                        w[2 * k] = TW{x.x, x.y};
                        w[2 * k + 1] = TW{x.z, x.w};
                    });
            };
            // round C, stage p = 4 .. 0 (2^(4-p) pairs each): p = 4, 3 natural slots; p <= 2 the [k][16-coefficient group]
            // permutation of prep.hip -- entry (group 2t + h, k) at k * NT16 + 2t + h, i.e. ONE 16-byte load per k
            auto load_tw_c = [&](TW(&w)[E32 - 1]) {
                w[0] = tw_pair(tu << 3, sbase(4));
                const unsigned voff = tu << 4;
                tw_two(voff, sbase(3), w[1], w[2]);
                static_for<3>([&](auto s_) {
                    constexpr int p = 2 - decltype(s_)::value; // 2, 1, 0
                    constexpr int RP = 16 >> (p + 1);          // entries per 16-coefficient group: 2, 4, 8
                    constexpr int O = (1 << (4 - p)) - 1;      // first entry of the stage in w[]: 3, 7, 15
                    static_for<RP>([&](auto k_) {
                        constexpr int k = decltype(k_)::value;
                        // group 2t: twiddle index m = k; group 2t + 1: m = RP + k
                        tw_two(voff, sbase(p) + static_cast<unsigned>(k) * NT16, w[O + k], w[O + RP + k]);
                    });
                });
            };

            T v[E32];
            TW twv[E32 - 1];

            if constexpr (!INV)
            {
                using SCH = EFwdSched<TLOG, M::LIMIT, M::TB, CANON_IN ? 1 : M::LIMIT>;
                // ---- round A: coalesced loads, block-uniform twiddles ---------------------------------------------------
#pragma unroll
                for (int j = 0; j < E32; j++)
                    v[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, static_cast<int>(tu << 2), j << (WLA + 2), POL_IN);
                stage_twb();
                if (CANON_IN && (a.flags & F_SIGNED_IN) != 0u)
                {
#pragma unroll
                    for (int j = 0; j < E32; j++)
                        v[j] = (static_cast<int32_t>(v[j]) < 0) ? static_cast<T>(v[j] + m.q) : v[j];
                }
                auto ct = [&](auto s_, auto uni_, T& u, T& x, const TW& w, bool unit) {
                    constexpr int ku = SCH::d.ku[decltype(s_)::value];
                    constexpr bool UNI = decltype(uni_)::value;
                    T U = u;
                    if constexpr (ku > 0)
                        U = m.template csub<ku>(U);
                    if constexpr (ku < 0)
                        U = m.reduce_2q(U);
                    const T nu = unit ? static_cast<T>(U + x) : m.template mul_acc<UNI>(x, w, U);
                    u = nu;
                    x = static_cast<T>(m.shl1_add(U, m.kq(M::TB)) - nu);
                };
                static_for<NA>([&](auto s_) {
                    constexpr int s = decltype(s_)::value; // stage p = TLOG - 1 - s, register bit jb = 4 - s
                    constexpr int jb = R5 - 1 - s;
                    static_for<E32 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                        constexpr int j1 = j0 | (1 << jb);
                        const TW w = tw[sbase(TLOG - 1 - s) + (j0 >> (jb + 1))]; // scalar load (uniform address)
                        // first stage of a cyclic transform: table[0] = omega^0 = 1 (uniform test; V is canonical there)
                        const bool unit = (s == 0 && !PART) ? (w.w == 1u) : false;
                        ct(std::integral_constant<int, s>{}, std::true_type{}, v[j0], v[j1], w, unit);
                    });
                });
                // ---- exchange A -> B (block-wide) ----------------------------------------------------------------------
                {
                    T* lw = lds + epad(t);
#pragma unroll
                    for (int j = 0; j < E32; j++)
                        lw[ejoff<WLA>(j)] = v[j];
                }
                __syncthreads();
                const T* lb = lds + epad((t & 31) | ((t >> 5) << 10));
#pragma unroll
                for (int j = 0; j < E32; j++)
                    v[j] = lb[ejoff<5>(j)];
                // ---- round B: stages 9 .. 5 ---------------------------------------------------------------------------
                static_for<R5>([&](auto s_) {
                    constexpr int s = decltype(s_)::value; // stage p = 9 - s, register bit jb = 4 - s
                    constexpr int jb = R5 - 1 - s;
                    TW wb[16];
                    read_twb(s_, wb);
                    static_for<E32 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                        constexpr int j1 = j0 | (1 << jb);
                        ct(std::integral_constant<int, NA + s>{}, std::false_type{}, v[j0], v[j1], wb[j0 >> (jb + 1)], false);
                    });
                });
                load_tw_c(twv);
                // ---- exchange B -> C (inside the wave's sub-block) -----------------------------------------------------
                {
                    T* lw = lds + epad((t & 31) | ((t >> 5) << 10));
#pragma unroll
                    for (int j = 0; j < E32; j++)
                        lw[ejoff<5>(j)] = v[j];
                }
                wave_sync();
                {
                    const u32x4* lc = reinterpret_cast<const u32x4*>(lds + 36 * t);
#pragma unroll
                    for (int k = 0; k < E32 / 4; k++)
                    {
                        const u32x4 x = lc[k];
                        v[4 * k] = x.x;
                        v[4 * k + 1] = x.y;
                        v[4 * k + 2] = x.z;
                        v[4 * k + 3] = x.w;
                    }
                }
                // ---- round C: stages 4 .. 0 ---------------------------------------------------------------------------
                static_for<R5>([&](auto s_) {
                    constexpr int s = decltype(s_)::value; // stage p = 4 - s = register bit
                    constexpr int jb = R5 - 1 - s;
                    static_for<E32 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                        constexpr int j1 = j0 | (1 << jb);
                        ct(std::integral_constant<int, NA + R5 + s>{}, std::false_type{}, v[j0], v[j1],
                           twv[(1 << s) - 1 + (j0 >> (jb + 1))], false);
                    });
                });
                static_for<E32>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    v[j] = lazy::normalize<SCH::d.final_bound>(m, v[j]);
                });
                // ---- 32 contiguous coefficients per lane -> 1 KiB runs per store instruction --------------------------
                {
                    u32x4* lc = reinterpret_cast<u32x4*>(lds + 36 * t);
#pragma unroll
                    for (int k = 0; k < E32 / 4; k++)
                        lc[k] = u32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
                }
                wave_sync();
                {
                    const unsigned e0 = (static_cast<unsigned>(t >> 6) << 11) + 4u * (static_cast<unsigned>(t) & 63u);
                    const T* lo = lds + epad(static_cast<int>(e0));
                    const T* mul_in = a.mul_in; // (uniform; the operand lies at the memory slot of the tile, like the output)
                    const dev::ModCtx<T> em{qv, qbit, qmu};
                    const __amdgpu_buffer_rsrc_t rmul = make_rsrc(mul_in != nullptr ? mul_in + base : static_cast<const T*>(a.in) + base,
                                                                  sizeof(T) << TLOG);
#pragma unroll
                    for (int i = 0; i < E32 / 4; i++)
                    {
                        u32x4 x = *reinterpret_cast<const u32x4*>(lo + 288 * i); // epad(256 i) = 288 i
                        if (mul_in != nullptr)
                        {
                            // GPU_PolyMul: the pointwise product with the other operand's transform rides on the final store
                            const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rmul, static_cast<int>(e0 << 2), 1024 * i, 0);
                            x = u32x4{em.mul(x.x, o.x), em.mul(x.y, o.y), em.mul(x.z, o.z), em.mul(x.w, o.w)};
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(x, rdst, static_cast<int>(e0 << 2), 1024 * i, POL_OUT);
                    }
                }
            }
            else
            {
                using SCH = EInvSched<TLOG, M::LIMIT, M::TB>;
                T ninv_w = a.ninv.w, ninv_wp = a.ninv.wp;
                if (a.ninv_arr != nullptr)
                {
                    ninv_w = a.ninv_arr[mi].w;
                    ninv_wp = a.ninv_arr[mi].wp;
                }
                const TW ninv{ninv_w, ninv_wp};
                // ---- 1 KiB runs per load instruction -> 32 contiguous coefficients per lane ----------------------------
                {
                    const unsigned e0 = (static_cast<unsigned>(t >> 6) << 11) + 4u * (static_cast<unsigned>(t) & 63u);
                    u32x4 x[E32 / 4];
#pragma unroll
                    for (int i = 0; i < E32 / 4; i++)
                        x[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(e0 << 2), 1024 * i, POL_IN);
                    load_tw_c(twv);
                    stage_twb();
                    T* lo = lds + epad(static_cast<int>(e0));
#pragma unroll
                    for (int i = 0; i < E32 / 4; i++)
                        *reinterpret_cast<u32x4*>(lo + 288 * i) = x[i];
                }
                wave_sync();
                {
                    const u32x4* lc = reinterpret_cast<const u32x4*>(lds + 36 * t);
#pragma unroll
                    for (int k = 0; k < E32 / 4; k++)
                    {
                        const u32x4 x = lc[k];
                        v[4 * k] = x.x;
                        v[4 * k + 1] = x.y;
                        v[4 * k + 2] = x.z;
                        v[4 * k + 3] = x.w;
                    }
                }
                // Gentleman-Sande butterfly  (U, V) -> (U + V, (U + c q - V) w)  with the SECOND operand of every stage but a
                // round's first arriving COMPLEMENTED (W = ~V): then the sum is one v_xad_u32 ((W ^ -1) + U) or, where the
                // next stage wants it complemented, one v_sub (W - U = ~(U + V)); the difference is one v_add3 (U + W +
                // (c q + 1)); and the product comes out complemented for the same three instructions (Mod32::mulc: the
                // twiddle negated, the accumulator started from -1).  Which form a register holds is a compile-time fact:
                // a stage's outputs are complemented iff the next stage of the round uses them as second operands (VC / OC
                // below); a round starts and ends with plain values (the exchanges and the stores see nothing of this).
                // 5 instructions per butterfly instead of 6, like the forward one.
                auto gs = [&](auto r_, auto s_, auto h_, auto uni_, auto last_, auto vc_, auto oc_, T& u, T& x, const TW& w) {
                    constexpr int r = decltype(r_)::value, s = decltype(s_)::value, h = decltype(h_)::value;
                    constexpr bool NEGATED = (r == 1); // round B: stage_twb() stored the negated twiddle where OC holds
                    constexpr bool UNI = decltype(uni_)::value, LASTST = decltype(last_)::value;
                    constexpr bool VC = decltype(vc_)::value, OC = decltype(oc_)::value;
                    constexpr int ku = SCH::d.ku[r][s][h], kv = SCH::d.kv[r][s][h], c = SCH::d.c[r][s][h];
                    T U = u, V = x;
                    if constexpr (ku != 0)
                        U = m.template csub<ku>(U);
                    if constexpr (kv != 0)
                    {
                        if constexpr (VC)
                            V = m.template csub_c<kv>(V);
                        else
                            V = m.template csub<kv>(V);
                    }
                    if constexpr (LASTST)
                    {
                        const T S = VC ? lazy::xad_not(V, U) : static_cast<T>(U + V);
                        u = m.template mul<true>(S, ninv); // the last twiddle was prepared as w * n^-1
                    }
                    else
                    {
                        constexpr int ko = SCH::d.ko[r][s][h];
                        if constexpr (!OC || ko < 0)
                        {
                            T S = VC ? lazy::xad_not(V, U) : static_cast<T>(U + V);
                            if constexpr (ko > 0)
                                S = m.template csub<ko>(S);
                            if constexpr (ko < 0)
                                S = m.reduce_2q(S);
                            u = OC ? static_cast<T>(~S) : S;
                        }
                        else
                        {
                            T Sc = VC ? static_cast<T>(V - U) : static_cast<T>(~(U + V));
                            if constexpr (ko > 0)
                                Sc = m.template csub_c<ko>(Sc);
                            u = Sc;
                        }
                    }
                    const T D = VC ? static_cast<T>(U + V + (m.kq(c) + 1u)) : static_cast<T>(U + m.kq(c) - V);
                    if constexpr (OC)
                        x = m.template mulc<UNI>(D, NEGATED ? w.w : static_cast<T>(0u - w.w), w.wp);
                    else
                        x = m.template mul<UNI>(D, w);
                };
                // ---- round C: stages 0 .. 4 ---------------------------------------------------------------------------
                static_for<R5>([&](auto s_) {
                    constexpr int s = decltype(s_)::value; // stage p = s = register bit
                    constexpr int jb = s;
                    static_for<E32 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                        constexpr int j1 = j0 | (1 << jb);
                        constexpr int kk = j0 >> (jb + 1);
                        gs(std::integral_constant<int, 0>{}, s_, h_, std::false_type{}, std::false_type{},
                           std::integral_constant<bool, (s > 0)>{}, std::integral_constant<bool, (s < R5 - 1) && (kk & 1)>{},
                           v[j0], v[j1], twv[(1 << (4 - s)) - 1 + kk]);
                    });
                });
                {
                    u32x4* lc = reinterpret_cast<u32x4*>(lds + 36 * t);
#pragma unroll
                    for (int k = 0; k < E32 / 4; k++)
                        lc[k] = u32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
                }
                wave_sync();
                {
                    const T* lb = lds + epad((t & 31) | ((t >> 5) << 10));
#pragma unroll
                    for (int j = 0; j < E32; j++)
                        v[j] = lb[ejoff<5>(j)];
                }
                // ---- round B: stages 5 .. 9 ---------------------------------------------------------------------------
                static_for<R5>([&](auto s_) {
                    constexpr int s = decltype(s_)::value; // stage p = 5 + s, register bit jb = s
                    constexpr int jb = s;
                    TW wb[16];
                    read_twb(std::integral_constant<int, 4 - s>{}, wb); // stage p = 9 - (4 - s)
                    static_for<E32 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                        constexpr int j1 = j0 | (1 << jb);
                        constexpr int kk = j0 >> (jb + 1);
                        gs(std::integral_constant<int, 1>{}, s_, h_, std::false_type{}, std::false_type{},
                           std::integral_constant<bool, (s > 0)>{}, std::integral_constant<bool, (s < R5 - 1) && (kk & 1)>{},
                           v[j0], v[j1], wb[kk]);
                    });
                });
                {
                    T* lw = lds + epad((t & 31) | ((t >> 5) << 10));
#pragma unroll
                    for (int j = 0; j < E32; j++)
                        lw[ejoff<5>(j)] = v[j];
                }
                __syncthreads();
                {
                    const T* la = lds + epad(t);
#pragma unroll
                    for (int j = 0; j < E32; j++)
                        v[j] = la[ejoff<WLA>(j)];
                }
                // ---- round A: stages 10 .. TLOG-1, block-uniform twiddles, n^-1 in the last one ------------------------
                static_for<NA>([&](auto s_) {
                    constexpr int s = decltype(s_)::value; // stage p = 10 + s, register bit jb = p - WLA
                    constexpr int p = 10 + s;
                    constexpr int jb = p - WLA;
                    static_for<E32 / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                        constexpr int j1 = j0 | (1 << jb);
                        constexpr int kk = j0 >> (jb + 1);
                        const TW w = tw[sbase(p) + kk];
                        gs(std::integral_constant<int, 2>{}, s_, h_, std::true_type{},
                           std::integral_constant<bool, LAST && s == NA - 1>{}, std::integral_constant<bool, (s > 0)>{},
                           std::integral_constant<bool, (s < NA - 1) && (kk & 1)>{}, v[j0], v[j1], w);
                    });
                });
                // (signed callers, GPU_INTT<Data32s>: a BRANCH on the block-uniform flag -- as a select the centring costs
                // every caller four instructions per coefficient beside the two of the normalisation)
                auto store_all = [&](auto centred_) {
                    static_for<E32>([&](auto j_) {
                        constexpr int j = decltype(j_)::value;
                        T x = v[j]; // PART: lazy hand-over, below LIMIT / 2 (sums corrected to it, products below 2 q)
                        if constexpr (LAST)
                        {
                            x = lazy::normalize<M::TB>(m, x);
                            if constexpr (decltype(centred_)::value)
                                x = (x > (m.q >> 1)) ? (x - m.q) : x;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(x, rdst, static_cast<int>(tu << 2), j << (WLA + 2), POL_OUT);
                    });
                };
                if (LAST && __builtin_amdgcn_readfirstlane(a.flags & F_CENTERED) != 0u)
                    store_all(std::true_type{});
                else
                    store_all(std::false_type{});
            }
        }
    } // namespace kern
} // namespace gpuntt
