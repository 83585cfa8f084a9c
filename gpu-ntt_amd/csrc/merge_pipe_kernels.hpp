// merge_pipe_kernels.hpp -- persistent, software-pipelined form of the 10-stage contiguous pass of the fast
// 64-bit Merge kernels (gfx950): the second pass of a forward transform of 2^13 .. 2^16 coefficients and the
// first pass of the inverse one (reference ForwardCore / InverseCore, src/lib/ntt_merge/ntt.cu:596-761,
// 1086-1318, second / first launch of the rows ntt.cuh:634-636).
//
// merge_pass_lazy runs one 4096-coefficient tile per workgroup: every wave first waits for its 16 coefficient
// loads, then for two rounds of per-lane twiddle loads (30 x 16 B per thread, as many bytes as the data),
// and only the three other waves of its SIMD cover those waits.  Measured on C2 (profiles/r01_contig_pass_
// experiments.txt): 267 us against 177 us for the loads / exchanges / stores alone and ~150-190 us of pure
// butterfly issue -- the two halves overlap badly.  Here a workgroup is resident for the whole launch and owns
// ONE tile position of the ring: it walks that position through the polynomials of the batch, so
//   * the twiddles of all three rounds depend on the tile position only: they are loaded once -- wave-uniform
//     round in 12 SGPRs, 16-contiguous round in 60 VGPRs, the round between them (16 lanes share each pair) in
//     4 KiB of LDS -- at two waves per SIMD, which the register-only butterfly microbenchmark prices at 74
//     instead of 70 cycles per butterfly (profiles/ubench_bfly_r02.txt);
//   * the coefficients of the NEXT polynomial are requested before the butterflies of the current one start
//     (32 more VGPRs), so the arithmetic never waits for HBM after the first tile;
//   * every exchange stays inside the wave (windows with WL <= 6, see merge_lazy_kernels.hpp), so the four
//     waves of a workgroup never synchronise.
// Measured (MI355X, C2: u64 2^16 x 1024): 279 us against 267 us for merge_pass_lazy, C2 0.438 against 0.427 ms --
// the part is at its 1400 W power cap in both (profiles/r02_power.txt), the pipelining leaves the energy per
// call unchanged, and two waves per SIMD cover the exchanges less well than four.  Kept opt-in (GPUNTT_PIPE=1,
// tests/test_gpu_pipe.py) as the record of the experiment; profiles/r02_pipelined_contig_pass.md.
// RNS: polynomial p uses modulus p % mod_count; the host sizes the grid so that the polynomial stride of a
// workgroup is a multiple of mod_count and its modulus (twiddles, constants) never changes.
#pragma once

#include "merge_lazy_kernels.hpp"

#ifndef GPUNTT_PIPE_SCHED_GROUP
#define GPUNTT_PIPE_SCHED_GROUP 4
#endif

namespace gpuntt
{
    namespace kern
    {
        // PREFETCH: request the next polynomial's coefficients before the current butterflies (32 more VGPRs: two
        // waves per SIMD); without it the kernel fits three waves per SIMD and only keeps the twiddles resident
        template <typename T, bool INV, int K, int IN_BOUND, bool LAST, bool PREFETCH = true>
        __global__ __launch_bounds__(256, (PREFETCH ? 2 : 3)) void merge_contig_pipe(LazyArgsT<T> a)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            constexpr int TLOG = 12;
            using M = lazy::Mod<T, 0>;
            using SCH = PassSched<TLOG, INV, true, K, IN_BOUND, M::LIMIT, M::TB>;
            using TW = lazy::Tw<T>;
            constexpr int NR = SCH::NR;
            constexpr int NT = LTile<TLOG>::NT;
            constexpr int SCHED_GROUP = GPUNTT_PIPE_SCHED_GROUP;
            constexpr int IOW = 6; // global loads / stores use the 64-contiguous window: 512-byte runs per wave
            static_assert(sizeof(T) == 8 && K >= 9 && K <= 10, "pipelined pass: 64-bit words, 9 or 10 stages");
            static_assert(SCH::wl_of(0) <= 6 && SCH::wl_of(NR - 1) <= 6 && SCH::wl_of(1) <= 6, "wave-local windows only");
            static_assert(!(INV && LAST) && !(!INV && !LAST), "forward: last pass; inverse: first pass");

            __shared__ T lds[LTile<TLOG>::LDS_ELEMS];
            // twiddles of the round on tile bits 4..7: they depend on t >> 4 only (16 lanes share each), so the
            // workgroup keeps its 16 x 15 pairs in LDS and a butterfly reads its pair when it needs it -- in
            // registers they (60 VGPRs) pushed the kernel past its 256-register budget (scratch spills whose
            // reloads wait on vmcnt(0), i.e. on the prefetch as well)
            __shared__ TW lds_tw[(NT >> 4) * TW_PER_ROUND];
            if (a.go_flag != nullptr && *a.go_flag == 0u)
                return;

            const int t = threadIdx.x;
            const int tpl = a.n - TLOG;
            const unsigned tp = blockIdx.x & ((1u << tpl) - 1u); // tile position inside the ring
            const unsigned lane_p = blockIdx.x >> tpl;             // first polynomial of this workgroup
            const unsigned stride_p = gridDim.x >> tpl;            // polynomials between two of its tiles
            const unsigned polys = static_cast<unsigned>(a.total >> a.n);
            if (lane_p >= polys)
                return;
            const unsigned count = (polys - lane_p + stride_p - 1u) / stride_p;
            // F_REVERSE: the batch is walked from its end (run_transform_lazy: the part of the hand-off that the
            // previous pass wrote last is still in the Infinity Cache)
            const bool rev = (a.flags & F_REVERSE) != 0u;
            const unsigned p_first = rev ? (polys - 1u - lane_p) : lane_p;

            T qv = a.q;
            int mi = 0;
            if (a.mods != nullptr)
            {
                mi = static_cast<int>(p_first % static_cast<unsigned>(a.mod_count));
                qv = a.mods[a.mod_order != nullptr ? a.mod_order[mi] : mi].value;
            }
            M m;
            m.set(qv, (a.norm_arr != nullptr) ? a.norm_arr[mi] : a.norm);
            const TW* __restrict__ tw_mod = a.tw + (static_cast<unsigned long long>(mi) << a.n);
            const unsigned tb = tp << TLOG; // offset of the tile inside its polynomial

            // ---- twiddles of every round, once ------------------------------------------------------------
            TW tws[NR][TW_PER_ROUND];
            static_for<NR>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                constexpr int STAGES = SCH::stages_of(r);
                constexpr int FIRST_POS = SCH::first_pos(r);
                constexpr int WL = SCH::wl_of(r);
                constexpr bool UNIFORM = (WL >= 6); // lanes of a wave differ below the window only
                const int t_uni = __builtin_amdgcn_readfirstlane(t);
                int off = 0;
                if constexpr (WL == 4)
                {
                    // entry e = t & 15 of group g = t >> 4 (the group's first thread is t = g << 4)
                    const int e = t & 15;
                    static_for<STAGES>([&](auto s_) {
                        constexpr int s = decltype(s_)::value;
                        constexpr int p = INV ? (FIRST_POS + s) : (FIRST_POS - s);
                        constexpr int CNT = 1 << (R - 1 - (p - WL));
                        static_assert(p > 2, "distance-1/2/4 stages sit in the 16-contiguous round");
                        if (e >= off && e < off + CNT)
                            lds_tw[(t >> 4) * TW_PER_ROUND + e] =
                                tw_mod[(1u << (a.n - 1 - p)) +
                                       ((tb + static_cast<unsigned>(elem_of<WL>(t & ~15, 0))) >> (p + 1)) + (e - off)];
                        off += CNT;
                    });
                }
                else
                static_for<STAGES>([&](auto s_) {
                    constexpr int s = decltype(s_)::value;
                    constexpr int p = INV ? (FIRST_POS + s) : (FIRST_POS - s);
                    constexpr int jb = p - WL;
                    constexpr int CNT = 1 << (R - 1 - jb);
                    constexpr bool PERM = (WL == 0) && (p <= 2); // prepared layout [tile][k][thread]
                    static_assert(PERM || p > 2, "distance-1/2/4 stages sit in the 16-contiguous round");
                    const unsigned stage_base = 1u << (a.n - 1 - p);
                    const TW* ps;
                    if constexpr (PERM)
                        ps = tw_mod + stage_base + tp * (CNT * NT) + t;
                    else
                        ps = tw_mod + stage_base + ((tb + static_cast<unsigned>(elem_of<WL>(UNIFORM ? t_uni : t, 0))) >> (p + 1));
                    static_for<CNT>([&](auto k_) {
                        constexpr int kk = decltype(k_)::value;
                        tws[r][off + kk] = ps[PERM ? kk * NT : kk];
                    });
                    off += CNT;
                });
            });

            __syncthreads(); // lds_tw is complete (the only workgroup-wide synchronisation of the kernel)

            // ---- the polynomials of this tile position ---------------------------------------------------------
            const T* src = static_cast<const T*>(a.in); // may alias a.out (in-place pass): a tile is read before it is written
            const unsigned io_lane = static_cast<unsigned>(elem_of<IOW>(t, 0));
            auto tile_base = [&](unsigned i) -> unsigned long long {
                const unsigned p = rev ? (p_first - i * stride_p) : (p_first + i * stride_p);
                return (static_cast<unsigned long long>(p) << a.n) + tb + io_lane;
            };
            T nxt[EPT];
            if constexpr (PREFETCH)
            {
                const T* g = src + tile_base(0);
#pragma unroll
                for (int j = 0; j < EPT; j++)
                    nxt[j] = ld_stream<(IN_BOUND == 1), false>(g + (j << IOW));
            }
            for (unsigned i = 0; i < count; i++)
            {
                T v[EPT];
                const unsigned long long out_base = tile_base(i);
                if constexpr (PREFETCH)
                {
#pragma unroll
                    for (int j = 0; j < EPT; j++)
                        v[j] = nxt[j];
                    if (i + 1u < count)
                    {
                        const T* g = src + tile_base(i + 1u);
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            nxt[j] = ld_stream<(IN_BOUND == 1), false>(g + (j << IOW));
                    }
                }
                else
                {
                    const T* g = src + out_base;
#pragma unroll
                    for (int j = 0; j < EPT; j++)
                        v[j] = ld_stream<(IN_BOUND == 1), false>(g + (j << IOW));
                }
                static_for<NR>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    constexpr int STAGES = SCH::stages_of(r);
                    constexpr int FIRST_POS = SCH::first_pos(r);
                    constexpr int WL = SCH::wl_of(r);
                    constexpr bool UNIFORM_R = (WL >= 6);
                    // ---- gather: the loaded window (round 0) or the previous round's window -> this round's
                    if constexpr (r > 0 || WL != IOW)
                    {
                        constexpr int PW = (r == 0) ? IOW : SCH::wl_of(r == 0 ? 0 : r - 1);
                        T* lw = lds + lds_pad(elem_of<PW>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            lw[lds_joff<PW>(j)] = v[j];
                        wave_sync();
                        const T* lr = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            v[j] = lr[lds_joff<WL>(j)];
                    }
                    // ---- butterflies (same arithmetic and range schedule as pass_body)
                    int off = 0;
                    static_for<STAGES>([&](auto s_) {
                        constexpr int s = decltype(s_)::value;
                        constexpr int p = INV ? (FIRST_POS + s) : (FIRST_POS - s);
                        constexpr int jb = p - WL;
                        static_for<EPT / 2>([&](auto h_) {
                            constexpr int h = decltype(h_)::value;
                            constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                            constexpr int j1 = j0 | (1 << jb);
                            constexpr int kk = j0 >> (jb + 1);
                            TW tw;
                            if constexpr (WL == 4)
                                tw = lds_tw[(t >> 4) * TW_PER_ROUND + off + kk];
                            else
                                tw = tws[r][off + kk];
                            constexpr int ku = SCH::d.ku[r][s][h];
                            if constexpr (!INV)
                            {
                                T U = v[j0];
                                if constexpr (ku != 0)
                                    U = m.template csub<ku>(U);
                                const T nu = m.template mul_acc<UNIFORM_R>(v[j1], tw, U);
                                v[j0] = nu;
                                v[j1] = static_cast<T>((U << 1) + m.kq(M::TB) - nu);
                            }
                            else
                            {
                                constexpr int kv = SCH::d.kv[r][s][h];
                                constexpr int c = SCH::d.c[r][s][h];
                                T U = v[j0], V = v[j1];
                                if constexpr (ku != 0)
                                    U = m.template csub<ku>(U);
                                if constexpr (kv != 0)
                                    V = m.template csub<kv>(V);
                                constexpr int ko = SCH::d.ko[r][s][h];
                                T S = U + V;
                                if constexpr (ko != 0)
                                    S = m.template csub<ko>(S);
                                v[j0] = S;
                                v[j1] = m.template mul<UNIFORM_R>(U + m.kq(c) - V, tw);
                            }
                            // the eight butterflies of a stage are independent; scheduled all at once their
                            // temporaries push the kernel past its 256 registers (spills)
                            if constexpr ((h % SCHED_GROUP) == SCHED_GROUP - 1)
                                __builtin_amdgcn_sched_barrier(0);
                        });
                        off += 1 << (R - 1 - jb);
                    });
                    // ---- scatter of the last round: canonical (forward, last pass) or lazy (inverse, first pass)
                    if constexpr (r == NR - 1)
                    {
                        if constexpr (LAST)
                        {
                            static_for<EPT>([&](auto j_) {
                                constexpr int j = decltype(j_)::value;
                                v[j] = lazy::normalize<SCH::d.bout[r][j]>(m, v[j]);
                            });
                        }
                        if constexpr (WL != IOW)
                        {
                            T* lw = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                lw[lds_joff<WL>(j)] = v[j];
                            wave_sync();
                            const T* lo = lds + lds_pad(elem_of<IOW>(t, 0));
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                v[j] = lo[lds_joff<IOW>(j)];
                        }
                        T* g = a.out + out_base;
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            st_stream<LAST>(g + (j << IOW), v[j]);
                    }
                });
            }
#else
            (void) a;
#endif
        }
    } // namespace kern
} // namespace gpuntt
