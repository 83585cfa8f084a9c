// merge_lazy_kernels.hpp -- the fast (lazy-residue) Merge-NTT tile-pass kernels, 32- and 64-bit (gfx950).
//
// Same tile/round geometry as merge_kernels.hpp (4096-coefficient tiles, 16 coefficients per
// thread, <= 4 radix-2 stages per register round, padded-LDS exchanges), but
//   * twiddles are read from the library's prepared table of Shoup pairs {w, floor(w*2^64/q)}
//     (prep.hip): block-uniform rounds fetch them through the scalar cache (s_load), the
//     last contiguous round reads them fully coalesced from a per-thread permuted layout;
//   * butterflies work on lazy residues in [0, B*q) with B tracked at compile time
//     (lazy.hpp); passes hand over lazy values, only the final pass normalises.
// Replaces the hot loops of reference ForwardCore/InverseCore
// (src/lib/ntt_merge/ntt.cu:596-761, 1086-1318).
#pragma once

#include "lazy.hpp"
#include "merge_kernels.hpp"

namespace gpuntt
{
    namespace kern
    {
        template <typename T> struct LazyArgsT
        {
            const void* in;
            T* out;
            const lazy::Tw<T>* tw;           // prepared twiddles: modulus slot mi at (mi << n)
            const Modulus<T>* mods;          // device array (RNS) or nullptr
            T q;                             // single modulus {value, bit, mu}
            T q_bit;
            T q_mu;
            const lazy::Tw<T>* ninv_arr;     // prepared n^-1 pairs per modulus (RNS) or nullptr
            lazy::Tw<T> ninv;                // single modulus n^-1 pair
            lazy::NormConst norm;            // single modulus: final-normalisation constants
            const lazy::NormConst* norm_arr; // RNS: per modulus (written by the prep kernel) or nullptr
            const unsigned* go_flag;         // RNS: device word, 1 = every modulus has lazy headroom
            const int* mod_order;            // *_Modulus_Ordered: prime of slot mi is mod_order[mi]
            const int* poly_order;           // *_Poly_Ordered: polynomial p lives in slot poly_order[p]
            const T* mul_in;                 // GPU_PolyMul: canonical operand multiplied into the final forward store, or nullptr
            const T* fs_n1;                  // one-tile 4-step calls with F_SELF_FALLBACK: the caller's own three tables
            const T* fs_n2;
            const T* fs_w;
            int lim;                         // 64-bit words: 0, or 8 / 4 = a 61- / 62-bit modulus in the call -> the LIMIT = 8 / 4 kernels (host-side switch)
            int host_allow_31q;              // host side only: the preparation kernel of this drop-in RNS call may name the 31 q family
            int n2_log;                      // 4-step transposing passes: log2 of the row stride of the transposed side
            int row_log;                     // natural-order 4-step row passes (Fst::nat_rows): log2 of the row stride of the row-major side (n2); a.n stays the ring (twiddle indices)
            int batch;                       // > 1: polynomials of the call, blocks are ordered poly-minor (big-ring contiguous passes)
            int col_log;                     // per-lane moduli (VQ, PerCoefficient RNS): log2 of the matrix row = number of columns
            int mod_shift;                   // per-lane moduli (VQ): log2 of the table stride per modulus (the ring size n_power)
            unsigned long long total;
            int n;
            int poly_shift;
            int mod_count;
            int p_lo;
            unsigned flags;
        };
        using LazyArgs = LazyArgsT<uint64_t>;

        // lds_pad(elem_of<WL>(t, j)) == lds_pad(elem_of<WL>(t, 0)) + lds_joff<WL>(j): the register part
        // of every LDS index is a compile-time constant (ds_read/ds_write immediate offset), so a
        // round's 16 accesses share one per-thread base address
        template <int WL> constexpr int lds_joff(int j)
        {
            return (j << WL) + ((WL >= 4) ? (j << (WL >= 4 ? WL - 4 : 0)) : (j >> (WL < 4 ? 4 - WL : 0)));
        }

        // ---- tile geometry of the fast kernels: 2^TLOG coefficients, 2^(TLOG-R) threads --------
        template <int TLOG> struct LTile
        {
            static constexpr int TL = TLOG;
            static constexpr int NT = 1 << (TLOG - R);
            static constexpr int TILE = 1 << TLOG;
            static constexpr int LDS_ELEMS = TILE + (TILE >> 4);
            static constexpr int LDS_ELEMS_FST = TILE + (TILE >> 4) + (TILE >> 5);
        };
        template <int TLOG, bool CONTIG, int K> struct LGeo
        {
            static constexpr int L = CONTIG ? 0 : (TLOG - K);
            static constexpr int NR = (K + R - 1) / R;
        };
        // SEG (CONTIG only): the tile is 2^(TLOG-K) runs of 2^K contiguous coefficients, run r at
        // base + (r << row_shift) -- the same column range of consecutive rows of a row-major matrix
        template <int TLOG, bool CONTIG, int K, bool SEG = false> struct LTileMap
        {
            unsigned long long base;
            int p_lo; // STRIDED: first stage bit of the pass; SEG: log2 of the row stride
            // CONTIG tile at an explicit flat base (the transposing 4-step passes order their blocks themselves)
            __device__ __forceinline__ explicit LTileMap(unsigned long long flat_base, int row_shift = 0)
                : base(flat_base), p_lo(row_shift)
            {
            }
            __device__ __forceinline__ LTileMap(int n, int pass_p_lo, unsigned long long blk)
            {
                constexpr int L = LGeo<TLOG, CONTIG, K>::L;
                if constexpr (CONTIG)
                {
                    base = blk << TLOG;
                    p_lo = 0;
                }
                else
                {
                    p_lo = pass_p_lo;
                    const unsigned long long poly = blk >> (n - TLOG);
                    const unsigned long long b = blk & ((1ull << (n - TLOG)) - 1);
                    const unsigned long long xb = b & ((1ull << (p_lo - L)) - 1);
                    const unsigned long long hi = b >> (p_lo - L);
                    base = (poly << n) | (hi << (p_lo + K)) | (xb << L);
                }
            }
            __device__ __forceinline__ LTileMap(int n, int pass_p_lo) : LTileMap(n, pass_p_lo, blockIdx.x) {}
            __device__ __forceinline__ unsigned long long flat(int e) const
            {
                return base + part(static_cast<unsigned>(e));
            }
            __device__ __forceinline__ int gpos(int p) const
            {
                constexpr int L = LGeo<TLOG, CONTIG, K>::L;
                if constexpr (CONTIG)
                    return p;
                else
                    return p_lo + (p - L);
            }
            // *_Poly_Ordered: move the tile to the memory slot of its polynomial (tiles never straddle
            // polynomials on this path: the host requires n >= TLOG)
            __device__ __forceinline__ void remap_poly(const int* order, int n)
            {
                const unsigned long long poly = base >> n;
                base = (static_cast<unsigned long long>(static_cast<unsigned>(order[poly])) << n) |
                       (base & ((1ull << n) - 1));
            }
            // flat(e) = base + part(e), and part() is additive over disjoint tile bits: the part of
            // the register bits is block-uniform (goes into the scalar base address), the part of
            // the thread bits is one 32-bit lane offset shared by all 16 accesses of a thread
            __device__ __forceinline__ unsigned part(unsigned e) const
            {
                constexpr int L = LGeo<TLOG, CONTIG, K>::L;
                if constexpr (CONTIG && SEG)
                    return ((e >> K) << p_lo) | (e & ((1u << K) - 1u));
                else if constexpr (CONTIG)
                    return e;
                else
                    return ((e >> L) << p_lo) | (e & ((1u << L) - 1u));
            }
        };

        // ---- compile-time schedule of range corrections for one pass --------------------
        // SKIP (inverse contiguous passes only): the pass runs the stages on tile bits [SKIP, K) -- the low SKIP stages of
        // the K-bit rows were done by the pass before it (inverse 4-step of the rings 2^15 / 2^16 in Merge form)
        template <int TLOG, bool INV, bool CONTIG, int K, int IN_BOUND, int LIMIT, int TB, int SKIP = 0> struct PassSched
        {
            using G = LGeo<TLOG, CONTIG, K>;
            static_assert(SKIP == 0 || (INV && CONTIG && SKIP < K), "partial passes: inverse, contiguous");
            static constexpr int TL = TLOG;
            static constexpr int NR = (K - SKIP + R - 1) / R;
            struct Data
            {
                int ku[NR][R][EPT / 2];
                int kv[NR][R][EPT / 2];
                int c[NR][R][EPT / 2];
                int ko[NR][R][EPT / 2];
                int bout[NR][EPT];
                int bin[NR];
                int final_bound;
            };
            // Round shapes.  The short round (K mod 4 stages) is the one FARTHEST from bit 0: first
            // for the forward transform (stages run top-down), last for the inverse (bottom-up), so
            // the remaining rounds are aligned to bit groups [..][7..4][3..0] of the tile: the
            // 16-contiguous-coefficient round always owns distances 1,2,4,8 and windows of STRIDED
            // passes never dip below the contiguous-run bits (lanes of a wave then differ only in
            // those bits and every twiddle of the pass is wave-uniform).
            static constexpr int SHORT = (K - SKIP) - R * (NR - 1);
            static constexpr int stages_of(int r)
            {
                return INV ? ((r == NR - 1) ? SHORT : R) : ((r == 0) ? SHORT : R);
            }
            static constexpr int first_pos(int r)
            {
                if (INV)
                    return G::L + SKIP + r * R;
                return (r == 0) ? (G::L + K - 1) : (G::L + K - 1 - SHORT - (r - 1) * R);
            }
            static constexpr int wl_of(int r)
            {
                // the register window is aligned to the TOP stage of the round in both directions (a short
                // inverse round 8,9 of a 10-stage contiguous pass owns tile bits 6..9, like the forward one)
                int w = INV ? (first_pos(r) + stages_of(r) - R) : (first_pos(r) - R + 1);
                const int lo = CONTIG ? 0 : (G::L > TL - R ? TL - R : G::L);
                return w < lo ? lo : (w > TL - R ? TL - R : w);
            }
            static constexpr Data make()
            {
                Data d{};
                int bound_in = IN_BOUND;
                for (int r = 0; r < NR; r++)
                {
                    int b[EPT] = {};
                    for (int j = 0; j < EPT; j++)
                        b[j] = bound_in;
                    d.bin[r] = bound_in;
                    for (int s = 0; s < stages_of(r); s++)
                    {
                        const int p = INV ? (first_pos(r) + s) : (first_pos(r) - s);
                        const int jb = p - wl_of(r);
                        for (int h = 0; h < EPT / 2; h++)
                        {
                            const int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                            const int j1 = j0 | (1 << jb);
                            if (!INV)
                            {
                                const lazy::CtPlan pl = lazy::ct_plan(b[j0], LIMIT, TB);
                                d.ku[r][s][h] = pl.ku;
                                b[j0] = b[j1] = pl.out;
                            }
                            else
                            {
                                const lazy::GsPlan pl = lazy::gs_plan(b[j0], b[j1], LIMIT);
                                d.ku[r][s][h] = pl.ku;
                                d.kv[r][s][h] = pl.kv;
                                d.c[r][s][h] = pl.c;
                                d.ko[r][s][h] = pl.ko;
                                b[j0] = pl.out_u;
                                b[j1] = TB;
                            }
                        }
                    }
                    int mx = 0;
                    for (int j = 0; j < EPT; j++)
                    {
                        d.bout[r][j] = b[j];
                        mx = b[j] > mx ? b[j] : mx;
                    }
                    bound_in = mx;
                }
                d.final_bound = bound_in;
                return d;
            }
            static constexpr Data d = make();
            static_assert(d.final_bound <= LIMIT, "lazy bound exceeds the headroom");
            // inverse passes are instantiated with IN_BOUND = LIMIT / 2 behind another inverse pass (lazy_launch_impl.hpp)
            static_assert(!INV || d.final_bound <= LIMIT / 2, "an inverse pass must hand over values below LIMIT / 2");
        };

        // waves per SIMD requested from the register allocator: four 256-thread tiles per CU
        // (128 VGPRs) or two 1024-thread tiles per CU (64 VGPRs; the 32-bit kernels fit once the
        // twiddles of a round are loaded at the start of that round instead of a round ahead)
        template <int TLOG, typename T = uint32_t> struct LOcc
        {
            // 64-bit kernels need the 128-VGPR budget at every tile size
            static constexpr int WAVES = (TLOG >= 14 && sizeof(T) == 4) ? 8 : 4;
        };

        // twiddles of one register round, in stage order: stage s (register bit jb) contributes
        // 2^(R-1-jb) entries -> at most 1 + 2 + 4 + 8 = 15 per thread
        constexpr int TW_PER_ROUND = EPT - 1;

        // Lazy residues throughout; moduli outside the lazy families' domain are served by the generic kernels
        // (merge_kernels.hpp) -- fusing both arithmetic policies behind a run-time branch cost the lazy path its register
        // allocation (round 1).
        // FST: the transposing passes of the 4-step entry points (reference FourStepForwardCoreT1..4 /
        // FourStepInverseCoreT1..4 + FourStepPartial*Core, src/lib/ntt_4step/ntt_4step.cu:68-743, 1049-1058, 1177-1872).
        // Every 4-step transform is the ring's own Merge plan with a transposition on the natural-order side (DESIGN.md
        // 3.5): the forward one gathers the transposed input in its first strided pass (Xp::first_gather below /
        // fourstep_first_lazy), the inverse one stores transposed from its first contiguous pass (Fst::inv_first /
        // fourstep_inv_first_lazy); no W product, no W stream.  (The W-multiplying phase 1 of rounds 1-3 is gone.)
        // Fst::nat_rows, natural-order 4-step, last forward pass: CONTIG stages on the same 2^K-column range
        // of 2^(TL-K) consecutive rows (lazy input from the strided row passes, or canonical input
        // when the rows fit one pass), canonical output stored transposed (no W product).
        // streaming forms (global_load / global_store ... nt): the first pass reads input nobody reads again, the
        // last pass writes output nobody reads again -- what should stay in the caches is the hand-off between the
        // passes (C2 0.437 -> 0.422 ms, C5 0.235 -> 0.227 ms, C4 0.310 -> 0.303 ms; nt on the hand-off loads as well
        // changes nothing)
        template <bool NT_, typename T> __device__ __forceinline__ T ld_stream(const T* p)
        {
            if constexpr (NT_)
                return __builtin_nontemporal_load(p);
            else
                return *p;
        }
        template <bool NT_, typename T> __device__ __forceinline__ void st_stream(T* p, T v)
        {
            if constexpr (NT_)
                __builtin_nontemporal_store(v, p);
            else
                *p = v;
        }

        // exchange point of a pass whose LDS traffic never leaves the wave: LDS operations of one wave
        // execute in order, so only the compiler has to be kept from moving the reads above the writes
        __device__ __forceinline__ void wave_sync()
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // XP: the 4-step entry points on rings that fit ONE tile.  GPU_4STEP_NTT is the Merge transform of the ring with
        // a transposition on the natural-order side (prep.hip: prep_merge_from_fourstep), n1 = 32 for every such ring
        // (reference launch table src/lib/ntt_4step/ntt_4step.cu:2306-2330: 2^12 .. 2^14 are 32 x n2):
        //   Xp::small_fwd (forward):  the tile is read as it lies (n2 x 32, coalesced) and dropped into LDS transposed, i.e. at
        //                      its natural position  e = (i << log n2) | c  for  f = (c << 5) | i;
        //   Xp::small_inv (inverse):  the natural-order result leaves through LDS transposed: o = (b << log n2) | a for
        //                      e = (a << 5) | b, and is stored as it lies (32 x n2, coalesced).
        // One HBM sweep instead of the reference's two kernels (FourStepForwardCoreT1 + FourStepPartialForwardCore).
        // LDS layout of the transposition: one pad element per n2-row, so the 32 lanes that write one column and the
        // lanes that read along a row both hit distinct banks.
        //   Xp::small_nat_fwd (natural-order forward): natural-order input, the spectrum leaves transposed -- out[(a << 5) | b] =
        //                      y[(b << log n2) | a] (NTT_4STEP_CPU::ntt order): wave-local turn into the 64-contiguous
        //                      window, then through LDS at (o + (o >> 5)) and out as it lies;
        //   Xp::small_nat_inv (natural-order inverse): the mirror image on the way in.
        constexpr int XP_L1 = 5;
        template <int K> __device__ __forceinline__ unsigned xp_swap_fwd(unsigned f) // f = (c << 5) | i  ->  (i << l2) | c
        {
            constexpr unsigned M = (1u << K) - 1u;
            const unsigned w = f & M;
            return (f & ~M) | ((w & 31u) << (K - XP_L1)) | (w >> XP_L1);
        }
        template <int K> __device__ __forceinline__ unsigned xp_swap_inv(unsigned e) // e = (a << 5) | b  ->  (b << l2) | a
        {
            return xp_swap_fwd<K>(e); // the same bit rotation: low five bits to the top of the ring index
        }
        template <int K> __device__ __forceinline__ unsigned xp_lds(unsigned e) { return e + (e >> (K - XP_L1)); }
        // natural-order side of Xp::small_nat_fwd / small_nat_inv: spectrum position e = (b << l2) | a  <->  o = (a << 5) | b; one pad element per
        // 32-element row of o, so the 64 lanes that hold consecutive a (stride 32 in o) hit distinct banks
        template <int K> __device__ __forceinline__ unsigned xp_nat_pos(unsigned e)
        {
            constexpr unsigned M = (1u << K) - 1u;
            const unsigned w = e & M;
            return (e & ~M) | ((w & ((1u << (K - XP_L1)) - 1u)) << XP_L1) | (w >> (K - XP_L1));
        }
        __device__ __forceinline__ unsigned xp_nat_lds(unsigned o) { return o + (o >> 5); }

        // Block-uniform values that went through a division or a select end up in vector registers, and so does
        // every address derived from them (64-bit VALU adds + v_readfirstlane per access).  Reading them back
        // through v_readfirstlane tells the compiler they are scalar: tile bases, twiddle bases and the modulus
        // index stay in SGPRs and the address arithmetic moves to the scalar unit.
        __device__ __forceinline__ unsigned uniform32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
        __device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
        {
            return (static_cast<unsigned long long>(uniform32(static_cast<unsigned>(v >> 32))) << 32) |
                   uniform32(static_cast<unsigned>(v));
        }

        // LDS reads that must have COMPLETED before the barrier that follows (other waves overwrite the buffer behind it):
        // the compiler is free to sink a load whose value is first used after the barrier below that barrier -- a release
        // fence only keeps earlier loads ahead of later STORES of the same thread -- and it does (found in round 3: the
        // one-launch 4-step kernel on the 16384-coefficient tile returned a wrong polynomial once in ~1000).  A volatile asm
        // that "uses" every value forces the loads and their s_waitcnt in front of it.
        template <typename T, int N> __device__ __forceinline__ void pin_loaded(T (&v)[N])
        {
#pragma unroll
            for (int j = 0; j < N; j++)
            {
                if constexpr (sizeof(T) == 8)
                    asm volatile("" : "+v"(v[j])::"memory");
                else
                    asm volatile("" : "+v"(v[j])::"memory");
            }
        }

        // ---- pass modes: which gather (round 0) and which store (last round) a pass_body instantiation uses ----------------
        //   tag                  gather (first round)                                  store (last round)                       kernel
        //   Fst::none, Xp::none  the tile as the plan maps it (direct / coalescing     the same on the way out                  merge_pass_lazy
        //                        pass / wave-local 64-contiguous window)
        //   Xp::small_fwd        n2 x 32 tile read as it lies, dropped into LDS        plain                                    fourstep_small_lazy (forward)
        //                        transposed = natural order
        //   Xp::small_inv        plain                                                 natural-order result through LDS at its    fourstep_small_lazy (inverse)
        //                                                                              transposed position, stored as it lies
        //   Xp::small_nat_fwd    plain (natural-order input)                           wave-local turn, then LDS at o + (o >> 5),  fourstep_small_lazy<NAT> (forward)
        //                                                                              spectrum stored transposed
        //   Xp::small_nat_inv    tile read as it lies, staged in LDS, picked up at     plain                                    fourstep_small_lazy<NAT> (inverse)
        //                        the spectrum positions of the 64-contiguous window
        //   Xp::first_gather     STRIDED first pass of a forward 4-step: 2^(K - l1)    plain (lazy hand-over)                   fourstep_first_lazy
        //                        coalesced runs of the transposed input through LDS
        //   Fst::nat_rows        (inverse) 2^RB-row runs of the column-major side      (forward) rows of 2^K through LDS, stored  fourstep_nat_last_lazy /
        //                        transposed through LDS                                transposed                               fourstep_nat_first_inv_lazy
        //   Fst::inv_first       plain (the spectrum as it lies)                       rows of 2^ROWLEN = n1 through LDS, stored  fourstep_inv_first_lazy
        //                                                                              transposed (lazy hand-over)
        // Every mode that changes the LDS layout between two uses of the buffer goes through relayout_barrier() below.
        enum class Fst : int
        {
            none = 0,
            nat_rows = 2,
            inv_first = 3
        };
        enum class Xp : int
        {
            none = 0,
            small_fwd = 1,
            small_inv = 2,
            small_nat_fwd = 3,
            small_nat_inv = 4,
            first_gather = 5
        };

        // "The LDS layout changes here": everything a thread has read from the buffer in the old layout must have ARRIVED
        // (pin_loaded: the compiler may otherwise sink the reads below the barrier) before any wave writes the new layout.
        // The ONE way to separate two layouts of the same LDS buffer -- a bare __syncthreads() at such a point is the
        // round-3 race.
        template <typename T, int N> __device__ __forceinline__ void relayout_barrier(T (&loaded)[N])
        {
            pin_loaded(loaded);
            __syncthreads();
        }

        // VQ: per-lane moduli -- the PerCoefficient layout with an RNS stack (reference ForwardCoreTranspose /
        // InverseCoreTranspose, src/lib/ntt_merge/ntt.cu:1554-2074: column c uses modulus c % mod_count with its own table
        // slot and n^-1).  Strided passes only; the 16 coefficients of a thread lie in ONE column in every round (register
        // windows of strided passes never dip below the contiguous-run bits), so a thread keeps one modulus for the pass
        // and only the operand classes change: q, -q, twiddles and n^-1 in vector registers (lazy::Mod<T, LIM, true>).
        template <typename T, int TLOG, bool INV, bool CONTIG, int K, int IN_BOUND, bool LAST,
                  Fst FST = Fst::none, int LIM = 0, Xp XP = Xp::none, int SKIP = 0, bool VQ = false, int ROWLEN = 0>
        __device__ __forceinline__ void pass_body(const LazyArgsT<T>& a, T* lds, T q_value, T q_bit, T q_mu,
                                                  int mi, unsigned long long fst_poly = 0,
                                                  unsigned fst_tile = 0, long long blk_override = -1,
                                                  unsigned fst_seg = 0)
        {
            using G = LGeo<TLOG, CONTIG, K>;
            using M = lazy::Mod<T, LIM, VQ>;
            using SCH = PassSched<TLOG, INV, CONTIG, K, IN_BOUND, M::LIMIT, M::TB, SKIP>;
            using TW = lazy::Tw<T>;
            static_assert(!VQ || (FST == Fst::none && XP == Xp::none && SKIP == 0), "per-lane moduli: plain passes");
            static_assert(!(VQ && CONTIG) || (K >= R && K < TLOG && IN_BOUND == 1 && LAST),
                          "per-lane moduli, contiguous: single-pass transforms of rings of 16 .. tile/2 coefficients");
            constexpr bool HAS_FST = (FST != Fst::none);
            constexpr int TL = TLOG;
            constexpr int NT = LTile<TLOG>::NT;
            constexpr int NR_ = SCH::NR;

            // single-pass transforms of rings smaller than a tile: the tile holds several polynomials
            constexpr bool MULTI_POLY = CONTIG && (K < TL) && (IN_BOUND == 1) && (LAST || HAS_FST);
            // Wave-local exchanges.  elem_of<WL> sends thread bit b >= WL to tile bit b + 4, so in every register
            // window with WL <= 6 the wave index (thread bits >= 6) IS the index of the 1024-coefficient
            // sub-block (tile bits >= 10) the wave's 16 x 64 coefficients lie in.  An exchange between two such
            // windows never leaves the wave's own LDS region: the block barrier becomes a wave-level ordering
            // point (wave_sync) and the waves of a workgroup run unsynchronised -- all of a contiguous pass of
            // <= 10 stages, the last exchange of every longer one (u64 K = 11, 12; u32 big tiles).  Full-tile
            // contiguous passes also enter / leave through the 64-contiguous window (512-byte runs per wave
            // instruction, WIO) with a wave-local transposition instead of the block-wide coalescing pass.
            constexpr bool WIO_OK = CONTIG && (!HAS_FST || FST == Fst::inv_first) && !MULTI_POLY && (TL >= 10);
            constexpr int WIO = 6;
            const int t = threadIdx.x;
            constexpr bool SEG = (FST == Fst::nat_rows);
            using Map = LTileMap<TLOG, CONTIG, K, SEG>;
            Map map = SEG   ? Map((fst_poly << a.poly_shift) +
                                      ((static_cast<unsigned long long>(fst_tile) << (TL - K)) << a.row_log) +
                                      (static_cast<unsigned long long>(fst_seg) << K),
                                  a.row_log)
                      : HAS_FST ? Map((fst_poly << a.poly_shift) + (static_cast<unsigned long long>(fst_tile) << TL))
                            : Map(a.n, a.p_lo,
                                  blk_override >= 0 ? static_cast<unsigned long long>(blk_override)
                                                    : static_cast<unsigned long long>(blockIdx.x));
            // whole tile inside the batch (always true for N >= 4096); taken before *_Poly_Ordered moves
            // the tile to its memory slot, which may lie beyond batch * N
            const bool tile_in_range = (CONTIG && !HAS_FST) ? ((map.base + LTile<TLOG>::TILE) <= a.total) : true;
            if (a.poly_order != nullptr)
                map.remap_poly(a.poly_order, a.n); // twiddle indices use flat & (N-1): unaffected
            if constexpr (VQ)
            {
                unsigned owner;
                if constexpr (CONTIG)
                    // RNS stack of rings below one tile: the tile holds 2^(TL - K) polynomials.  Every register window is
                    // aligned to the top stage of its round and K >= 4, so no window reaches tile bit K: thread bit b >= WL
                    // is tile bit b + 4 in every round, and the thread's 16 coefficients lie in polynomial t >> (K - 4) of
                    // the tile in EVERY round (reference ForwardCoreLowRing / InverseCoreLowRing RNS forms,
                    // src/lib/ntt_merge/ntt.cu:116-219, 326-433: mod_index = polynomial % mod_count)
                    owner = static_cast<unsigned>(map.base >> K) + (static_cast<unsigned>(t) >> (K - R));
                else
                    // the thread's column (the same in every round, see above) picks its modulus
                    owner = static_cast<unsigned>(map.flat(elem_of<SCH::wl_of(0)>(t, 0))) & ((1u << a.col_log) - 1u);
                mi = static_cast<int>(owner % static_cast<unsigned>(a.mod_count));
                const Modulus<T> md = a.mods[a.mod_order != nullptr ? a.mod_order[mi] : mi];
                q_value = md.value;
                q_bit = md.bit;
                q_mu = md.mu;
            }
            // lazy::Mod64::mul_acc_raw names the fixed register pair v[126:127]: every 64-bit kernel must be compiled for a
            // budget of AT LEAST 128 VGPRs, i.e. at most 4 waves per SIMD in its __launch_bounds__
            static_assert(sizeof(T) != 8 || LOcc<TLOG, T>::WAVES <= 4,
                          "64-bit lazy kernels need the 128-VGPR budget (v126 / v127 are named in lazy.hpp)");
            M m;
            m.set(q_value, (a.norm_arr != nullptr) ? a.norm_arr[mi] : a.norm);
            const dev::ModCtx<T> em{q_value, q_bit, q_mu};
            const unsigned long long root_base = static_cast<unsigned long long>(mi) << ((VQ && !CONTIG) ? a.mod_shift : a.n);
            // (word by word: selecting between the two 8-byte structs as a whole left a dead 16-byte stack slot --
            // and with it a scratch allocation -- in every 32-bit inverse kernel)
            T ninv_w = a.ninv.w, ninv_wp = a.ninv.wp;
            if (LAST && INV && a.ninv_arr != nullptr)
            {
                ninv_w = a.ninv_arr[mi].w;
                ninv_wp = a.ninv_arr[mi].wp;
            }
            const TW ninv{ninv_w, ninv_wp};
            const unsigned nmask = (1u << a.n) - 1u;
            const TW* __restrict__ tw_mod = a.tw + root_base;

            // Issues every twiddle load of round r (15 x 16 B per thread, or scalar loads when the
            // round is block-uniform).  Called one round ahead, in front of the LDS barrier, so the
            // L2 latency is covered by the previous round's butterflies and the exchange.
            auto load_twiddles = [&](auto r_, TW(&tws)[TW_PER_ROUND]) {
                constexpr int r = decltype(r_)::value;
                constexpr int STAGES = SCH::stages_of(r);
                constexpr int FIRST_POS = SCH::first_pos(r);
                constexpr int WL = SCH::wl_of(r);
                // block-uniform: no thread bits above the register window; wave-uniform: the 64 lanes
                // of a wave differ only in tile bits below the window (WL >= 6) -> scalar loads with
                // the wave's first thread id
                constexpr bool UNIFORM = (WL + R == TL) || (WL >= 6);
                const int t_uni = (WL + R == TL) ? 0 : __builtin_amdgcn_readfirstlane(t);
                int off = 0;
                static_for<STAGES>([&](auto s_) {
                    constexpr int s = decltype(s_)::value;
                    constexpr int p = INV ? (FIRST_POS + s) : (FIRST_POS - s);
                    constexpr int jb = p - WL;
                    constexpr int CNT = 1 << (R - 1 - jb);
                    // prepared layout of the distance-1/2/4 stages of full tiles: [tile][k][thread]
                    constexpr bool PERM = CONTIG && !MULTI_POLY && !SEG && (WL == 0) && (p <= 2);
                    const int P = map.gpos(p);
                    const unsigned stage_base = 1u << (a.n - 1 - P); // slots [2^S, 2^(S+1)), S = n-1-P
                    const TW* ps;
                    if constexpr (PERM)
                    {
                        const unsigned tile_in_poly = (static_cast<unsigned>(map.flat(0)) & nmask) >> TL;
                        ps = tw_mod + stage_base + tile_in_poly * (CNT * NT) + t;
                    }
                    else
                    {
                        // UNIFORM: blockIdx and compile-time bits only -> scalar loads
                        const unsigned idx0 =
                            static_cast<unsigned>(map.flat(elem_of<WL>(UNIFORM ? t_uni : t, 0))) & nmask;
                        ps = tw_mod + stage_base + (idx0 >> (P + 1));
                    }
                    static_for<CNT>([&](auto k_) {
                        constexpr int kk = decltype(k_)::value;
                        if constexpr (CONTIG && !MULTI_POLY && !SEG && (WL != 0) && (p <= 2))
                        {
                            // distance-1/2/4 stage that is not in the 16-contiguous-coefficient round
                            // (contiguous passes of 9 or 10 stages): address the permuted layout
                            // [tile][k][16-coefficient group] entry by entry
                            const unsigned e0 = elem_of<WL>(t, kk << (jb + 1));
                            const unsigned tile_in_poly = (static_cast<unsigned>(map.flat(0)) & nmask) >> TL;
                            constexpr int RP = EPT >> (p + 1);
                            tws[off + kk] = tw_mod[stage_base + tile_in_poly * (RP * NT) +
                                                   ((e0 & (EPT - 1)) >> (p + 1)) * NT + (e0 >> R)];
                        }
                        else if constexpr (MULTI_POLY)
                        {
                            // a tile holds several rings of length 2^K (n == K here): window bits
                            // at or above the ring size select the ring, not the twiddle, so the
                            // offset of entry kk is a compile-time constant
                            constexpr int KOFF = ((kk << (jb + 1 + WL)) & ((1 << K) - 1)) >> (p + 1);
                            tws[off + kk] = ps[KOFF];
                        }
                        else
                        {
                            tws[off + kk] = ps[PERM ? kk * NT : kk];
                        }
                    });
                    off += CNT;
                });
            };

            const bool full_tile = tile_in_range;
            const bool plain_io = full_tile; // signed input is converted after the unguarded loads
            // (only the first pass of a forward transform can see signed words)
            const bool signed_in = (!INV && IN_BOUND == 1 && !HAS_FST) ? ((a.flags & F_SIGNED_IN) != 0u) : false;
            auto to_residue = [&](T x) -> T {
                using S = typename std::make_signed<T>::type;
                return (static_cast<S>(x) < 0) ? static_cast<T>(x + m.q) : x;
            };

            // 64-bit: request a round's twiddles one round ahead (in front of the exchange barrier);
            // 32-bit big tiles: at the start of the round (halves the live twiddle registers; the
            // 8 waves per SIMD cover the latency)
            constexpr bool TW_AHEAD = (LOcc<TLOG, T>::WAVES <= 4);

            T v[EPT];
            TW tw_next[TW_PER_ROUND];
            if constexpr (TW_AHEAD)
                load_twiddles(std::integral_constant<int, 0>{}, tw_next);

            static_for<NR_>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                constexpr int STAGES = SCH::stages_of(r);
                constexpr int FIRST_POS = SCH::first_pos(r);
                constexpr int WL = SCH::wl_of(r);
                constexpr bool DIRECT_IO = (WL >= 4);
                constexpr bool UNIFORM_R = (WL + R == TL) || (WL >= 6); // scalar twiddles this round

                // ---- gather -----------------------------------------------------------
                if constexpr (r == 0)
                {
                    // fast path: whole tile in range and unsigned input -> 16 independent loads
                    // in flight; the guarded path only serves ragged last tiles / signed input
                    // ragged last tile: the load itself is unconditional (a clamped, always valid address), only the
                    // value is selected -- a branch per element serialised the 16 loads of a thread (a single 2^13
                    // polynomial in a 16384-coefficient tile took longer than a 2^14 one)
                    auto load_guarded = [&](unsigned long long f) -> T {
                        const bool inside = f < a.total;
                        const T raw = static_cast<const T*>(a.in)[inside ? f : 0ull];
                        T val = raw;
                        if (a.flags & F_SIGNED_IN)
                        {
                            using S = typename std::make_signed<T>::type;
                            val = (static_cast<S>(raw) < 0) ? static_cast<T>(m.q + raw) : raw;
                        }
                        return inside ? val : static_cast<T>(0);
                    };
                    const T* src = static_cast<const T*>(a.in); // may alias a.out (in-place calls)
                    if constexpr (XP == Xp::small_fwd)
                    {
                        // coalesced read of the n2 x 32 tile, transposed into natural order through LDS
                        // (groups of XG loads in flight: the 64-VGPR kernels of the 32-bit big tile spill with all 16)
                        constexpr int XG = 8;
#pragma unroll
                        for (int part = 0; part < EPT / XG; part++)
                        {
                            T tmp[XG];
                            if (plain_io)
                            {
#pragma unroll
                                for (int jj = 0; jj < XG; jj++)
                                    tmp[jj] = ld_stream<true>((src + (map.base + static_cast<unsigned>(NT * (part * XG + jj)))) + t);
                            }
                            else
                            {
#pragma unroll
                                for (int jj = 0; jj < XG; jj++)
                                    tmp[jj] = load_guarded(map.base + static_cast<unsigned>(t + NT * (part * XG + jj)));
                            }
#pragma unroll
                            for (int jj = 0; jj < XG; jj++)
                                lds[xp_lds<K>(xp_swap_fwd<K>(static_cast<unsigned>(t + NT * (part * XG + jj))))] = tmp[jj];
                        }
                        __syncthreads();
                        if constexpr (WL >= K - XP_L1)
                        {
                            // the register bits lie above the row shift: one base address + compile-time offsets
                            const T* lx = lds + xp_lds<K>(static_cast<unsigned>(elem_of<WL>(t, 0)));
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                v[j] = lx[(j << WL) + ((j << WL) >> (K - XP_L1))];
                        }
                        else
                        {
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                v[j] = lds[xp_lds<K>(static_cast<unsigned>(elem_of<WL>(t, j)))];
                        }
                        relayout_barrier(v); // the exchanges below reuse the buffer in the e + (e >> 4) layout
                    }
                    else if constexpr (SEG && INV)
                    {
                        // natural-order inverse 4-step, first pass: the tile's 2^RB rows x 2^K columns
                        // come from the column-major side -- in[((seg << K) + c) * n1 + row0 + r] -- as
                        // runs of 2^RB rows, and are transposed through LDS into the row-major tile
                        constexpr int RB = TL - K;
                        const unsigned lane = (static_cast<unsigned>(t >> RB) << a.n2_log) + (t & ((1 << RB) - 1));
                        const unsigned long long in_base =
                            (fst_poly << a.poly_shift) + ((static_cast<unsigned long long>(fst_seg) << K) << a.n2_log) +
                            (fst_tile << RB);
                        T tmp[EPT];
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            tmp[j] = (src + (in_base + (static_cast<unsigned long long>((NT >> RB) * j) << a.n2_log)))[lane];
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const int o = t + NT * j;
                            lds[lds_pad_t<K>(((o & ((1 << RB) - 1)) << K) | (o >> RB))] = tmp[j];
                        }
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            v[j] = lds[lds_pad_t<K>(elem_of<WL>(t, j))];
                        relayout_barrier(v); // the exchanges below reuse the buffer in the e + (e >> 4) layout
                    }
                    else if constexpr (XP == Xp::first_gather)
                    {
                        // forward 4-step, first pass in Merge form: an ordinary strided first pass of the ring's Merge
                        // plan -- the tile is 2^K rows x 2^L columns of the natural layout, rows = the top K index
                        // bits (r, c_top) with r the n1-index -- whose INPUT lies transposed: in[(c << l1) | r].
                        // For a fixed c_top the tile's 2^L columns x n1 rows are ONE run of 2^(L + l1) words there,
                        // so the tile is read as 2^(K - l1) such runs (coalesced), dropped into LDS at its natural
                        // in-tile position (one pad word per r: the lanes of a run walk r first) and picked up in
                        // the register window of the first round.  a.n2_log = log2 n1.
                        static_assert(!CONTIG && !INV && DIRECT_IO && TL == 12 && K >= 5 && K <= 8, "strided forward first pass");
                        constexpr int L = TL - K;
                        const int l1 = a.n2_log;
                        const int ch = TL - (K - l1); // log2 of a run: L + l1, 9 .. 12
                        const unsigned long long in_base =
                            ((map.base >> a.n) << a.n) + (static_cast<unsigned long long>((static_cast<unsigned>(map.base) & nmask) >> L) << ch);
                        const unsigned rr = static_cast<unsigned>(t) & ((1u << l1) - 1u);
                        const unsigned lds_t = (rr << (TL - l1)) + rr;
                        T tmp[EPT];
#pragma unroll
                        for (int jj = 0; jj < EPT; jj++)
                        {
                            const unsigned ctop = static_cast<unsigned>(jj) >> (ch - 8);
                            const unsigned within = (static_cast<unsigned>(jj) & ((1u << (ch - 8)) - 1u)) << 8;
                            tmp[jj] = ld_stream<true>((src + (in_base + (static_cast<unsigned long long>(ctop) << (a.p_lo + l1)) + within)) + t);
                        }
#pragma unroll
                        for (int jj = 0; jj < EPT; jj++)
                        {
                            const unsigned f = static_cast<unsigned>(t + NT * jj);
                            const unsigned ctop = static_cast<unsigned>(jj) >> (ch - 8);
                            lds[lds_t + (ctop << L) + ((f >> l1) & ((1u << L) - 1u))] = tmp[jj];
                        }
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const unsigned e = static_cast<unsigned>(elem_of<WL>(t, j));
                            v[j] = lds[e + (e >> (TL - l1))];
                        }
                        relayout_barrier(v); // the exchanges below reuse the buffer in the e + (e >> 4) layout
                    }
                    else if constexpr (DIRECT_IO)
                    {
                        if (plain_io)
                        {
                            const unsigned lane = map.part(elem_of<WL>(t, 0));
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                v[j] = ld_stream<(IN_BOUND == 1)>((src + (map.base + map.part(static_cast<unsigned>(j) << WL))) + lane);
                            if (signed_in)
                            {
#pragma unroll
                                for (int j = 0; j < EPT; j++)
                                    v[j] = to_residue(v[j]);
                            }
                        }
                        else
                        {
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                v[j] = load_guarded(map.flat(elem_of<WL>(t, j)));
                        }
                    }
                    else if constexpr (WIO_OK && WL <= 6)
                    {
                        // (inverse first pass) 512-byte runs per load instruction into the wave's own sub-block,
                        // transposed to the register window through the wave's own LDS region
                        constexpr int IWL = WIO;
                        const unsigned lane = map.part(elem_of<IWL>(t, 0));
                        T tmp[EPT];
                        if constexpr (XP == Xp::small_nat_inv)
                        {
                            // natural-order inverse 4-step: the tile is read as it lies (n2 x 32), staged in LDS and
                            // picked up at the spectrum positions of the 64-contiguous window
#pragma unroll
                            for (int half = 0; half < 2; half++)
                            {
                                T ld[EPT / 2];
#pragma unroll
                                for (int jj = 0; jj < EPT / 2; jj++)
                                    ld[jj] = ld_stream<true>((src + (map.base + static_cast<unsigned>(NT * (half * (EPT / 2) + jj)))) + t);
#pragma unroll
                                for (int jj = 0; jj < EPT / 2; jj++)
                                    lds[xp_nat_lds(static_cast<unsigned>(t + NT * (half * (EPT / 2) + jj)))] = ld[jj];
                            }
                            __syncthreads();
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                tmp[j] = lds[xp_nat_lds(xp_nat_pos<K>(static_cast<unsigned>(elem_of<IWL>(t, j))))];
                            relayout_barrier(tmp); // the wave-local turn below rewrites the buffer in the e + (e >> 4) layout
                        }
                        else
                        {
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                tmp[j] = ld_stream<(IN_BOUND == 1)>((src + (map.base + map.part(static_cast<unsigned>(j) << IWL))) + lane);
                        }
                        T* li = lds + lds_pad(elem_of<IWL>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            li[lds_joff<IWL>(j)] = tmp[j];
                        wave_sync();
                        const T* lw = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            v[j] = lw[lds_joff<WL>(j)];
                    }
                    else
                    {
                        if (plain_io)
                        {
                            T tmp[EPT];
                            const unsigned lane = map.part(static_cast<unsigned>(t));
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                tmp[j] = ld_stream<(IN_BOUND == 1)>((src + (map.base + map.part(static_cast<unsigned>(NT * j)))) + lane);
                            if (signed_in)
                            {
#pragma unroll
                                for (int j = 0; j < EPT; j++)
                                    tmp[j] = to_residue(tmp[j]);
                            }
                            T* lc = lds + lds_pad(t);
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                lc[NT * j + ((NT * j) >> 4)] = tmp[j];
                        }
                        else
                        {
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                lds[lds_pad(t + NT * j)] = load_guarded(map.flat(t + NT * j));
                        }
                        __syncthreads();
                        const T* lw = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            v[j] = lw[lds_joff<WL>(j)];
                    }
                }
                else
                {
                    const T* lw = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                    for (int j = 0; j < EPT; j++)
                        v[j] = lw[lds_joff<WL>(j)];
                }

                TW tw_cur[TW_PER_ROUND];
                if constexpr (TW_AHEAD)
                {
#pragma unroll
                    for (int i = 0; i < TW_PER_ROUND; i++)
                        tw_cur[i] = tw_next[i];
                }
                else
                {
                    load_twiddles(r_, tw_cur);
                }

                // ---- butterflies ------------------------------------------------------
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
                        const TW tw = tw_cur[off + kk];
                        constexpr bool UNI_TW = UNIFORM_R && !VQ; // twiddle in scalar registers
                        if constexpr (!INV)
                        {
                            constexpr int ku = SCH::d.ku[r][s][h];
                            T U = v[j0];
                            if constexpr (ku != 0)
                                U = m.template csub<ku>(U);
                            // first stage of a cyclic transform: table[0] = omega^0 = 1 for every
                            // butterfly (block-uniform scalar test, so tables that differ still work);
                            // V is a canonical input there, so V itself is the product
                            bool unit = false;
                            if constexpr (UNI_TW && r == 0 && s == 0 && IN_BOUND <= M::TB && !HAS_FST)
                                unit = (tw.w == 1);
                            // U' = U + T comes out of the product's own multiply-add chain;
                            // V' = U - T + TB q = 2 U + TB q - U'  (mod 2^W; the true value is below LIMIT q)
                            {
                                const T nu = unit ? static_cast<T>(U + v[j1]) : m.template mul_acc<UNI_TW>(v[j1], tw, U);
                                v[j0] = nu;
                                // 2 U + TB q as ONE v_lshl_add_u64 / _u32, then one subtract: kept opaque so that
                                // the sum is not re-associated into shift, subtract, add (a fourth instruction)
                                v[j1] = static_cast<T>(m.shl1_add(U, m.kq(M::TB)) - nu);
                            }
                        }
                        else
                        {
                            constexpr int ku = SCH::d.ku[r][s][h];
                            constexpr int kv = SCH::d.kv[r][s][h];
                            constexpr int c = SCH::d.c[r][s][h];
                            T U = v[j0], V = v[j1];
                            if constexpr (ku != 0)
                                U = m.template csub<ku>(U);
                            if constexpr (kv != 0)
                                V = m.template csub<kv>(V);
                            // final stage of an inverse transform: its twiddle was prepared as
                            // w * n^-1, so scaling the sum as well finishes the n^-1 product (the product
                            // takes any 64-bit value: no range correction of the sum there)
                            if constexpr (LAST && r == NR_ - 1 && s == STAGES - 1)
                                v[j0] = m.template mul<!VQ>(U + V, ninv);
                            else
                            {
                                constexpr int ko = SCH::d.ko[r][s][h];
                                T S = U + V;
                                if constexpr (ko != 0)
                                    S = m.template csub<ko>(S);
                                v[j0] = S;
                            }
                            v[j1] = m.template mul<UNI_TW>(U + m.kq(c) - V, tw);
                        }
                    });
                    off += 1 << (R - 1 - jb);
                });

                // ---- scatter ----------------------------------------------------------
                if constexpr (r == NR_ - 1)
                {
                    constexpr bool PMUL_OK = LAST && !INV && !HAS_FST;
                    const T* mul_in = PMUL_OK ? a.mul_in : nullptr;
                    (void) mul_in;
                    if constexpr (LAST && !INV && sizeof(T) == 8)
                    {
                        // moduli of >= 48 bits: quotient estimate from the high word (one shift less per coefficient);
                        // the test is wave-uniform, both forms are straight-line code
                        if (!VQ && m.hi_norm())
                            static_for<EPT>([&](auto j_) {
                                constexpr int j = decltype(j_)::value;
                                v[j] = lazy::normalize<SCH::d.bout[r][j], true>(m, v[j]);
                            });
                        else
                            static_for<EPT>([&](auto j_) {
                                constexpr int j = decltype(j_)::value;
                                v[j] = lazy::normalize<SCH::d.bout[r][j], false>(m, v[j]);
                            });
                    }
                    else if constexpr (LAST && INV)
                    {
                        // (signed callers, GPU_INTT<Data64s>: a BRANCH on the block-uniform flag -- as a select the centring
                        // costs every caller a compare, a select and a subtraction per coefficient)
                        if (__builtin_amdgcn_readfirstlane(a.flags & F_CENTERED) != 0u)
                            static_for<EPT>([&](auto j_) {
                                constexpr int j = decltype(j_)::value;
                                const T x = lazy::normalize<M::TB>(m, v[j]); // n^-1 went in with the last stage
                                v[j] = (x > (m.q >> 1)) ? (x - m.q) : x;
                            });
                        else
                            static_for<EPT>([&](auto j_) {
                                constexpr int j = decltype(j_)::value;
                                v[j] = lazy::normalize<M::TB>(m, v[j]);
                            });
                    }
                    else if constexpr (LAST)
                    {
                        static_for<EPT>([&](auto j_) {
                            constexpr int j = decltype(j_)::value;
                            v[j] = lazy::normalize<SCH::d.bout[r][j]>(m, v[j]);
                        });
                    }
                    if constexpr (XP == Xp::small_inv)
                    {
                        // natural-order result -> LDS at its transposed position -> coalesced stores of the 32 x n2 tile
                        relayout_barrier(v); // every wave is done with the e + (e >> 4) layout
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            lds[xp_lds<K>(xp_swap_inv<K>(static_cast<unsigned>(elem_of<WL>(t, j))))] = v[j];
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const unsigned o = static_cast<unsigned>(t + NT * j);
                            const T x = lds[xp_lds<K>(o)];
                            if (full_tile)
                                st_stream<true>(a.out + (map.base + o), x);
                            else if (map.base + o < a.total)
                                a.out[map.base + o] = x;
                        }
                    }
                    else if constexpr (HAS_FST && !(SEG && INV))
                    {
                        // Fst::inv_first: the tile is 2^(TL - TK) rows of 2^TK = n1 coefficients (TK = ROWLEN), all TL stages of the
                        // ring's inverse Merge plan done on it; Fst::nat_rows: rows of 2^K, K stages
                        constexpr int TK = (FST == Fst::inv_first) ? ROWLEN : K;
                        static_assert(FST == Fst::nat_rows || FST == Fst::inv_first, "transposing store: natural-order last pass / inverse first pass");
                        static_assert(CONTIG && TK >= 4 && TK <= 9, "4-step row runs are 16..512 long");
                        static_assert(FST != Fst::inv_first || (INV && K == TL && !LAST), "Merge-form inverse first pass");
                        constexpr int RB = TL - TK; // log2 rows per tile
                        relayout_barrier(v); // all gathers from the e + (e >> 4) layout are done
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            lds[lds_pad_t<TK>(elem_of<WL>(t, j))] = v[j];
                        __syncthreads();
                        const unsigned row0 = fst_tile << RB;
                        // SEG: output row of tile column i is (seg << K) + i
                        const unsigned long long seg_base =
                            SEG ? ((static_cast<unsigned long long>(fst_seg) << TK) << a.n2_log) : 0ull;
#pragma unroll
                        for (int half = 0; half < 2; half++)
                        {
                            T x[EPT / 2];
#pragma unroll
                            for (int jj = 0; jj < EPT / 2; jj++)
                            {
                                // o = t + NT * j: column i = o >> RB splits into a uniform and a lane part
                                const int jr = half * (EPT / 2) + jj;
                                const int o = t + NT * jr;
                                const int jl = o & ((1 << RB) - 1);
                                const int i = o >> RB;
                                const unsigned lane = (static_cast<unsigned>(t >> RB) << a.n2_log) + (t & ((1 << RB) - 1));
                                const unsigned long long ubase =
                                    (static_cast<unsigned long long>((NT >> RB) * jr) << a.n2_log) + row0;
                                x[jj] = lds[lds_pad_t<TK>((jl << TK) | i)];
                            }
#pragma unroll
                            for (int jj = 0; jj < EPT / 2; jj++)
                            {
                                const int jr = half * (EPT / 2) + jj;
                                const unsigned lane = (static_cast<unsigned>(t >> RB) << a.n2_log) + (t & ((1 << RB) - 1));
                                const unsigned long long ubase =
                                    (static_cast<unsigned long long>((NT >> RB) * jr) << a.n2_log) + row0;
                                // (Fst::nat_rows: canonical; Fst::inv_first: lazy hand-over to the row passes)
                                (a.out + ((fst_poly << a.poly_shift) + seg_base + ubase))[lane] = x[jj];
                            }
                        }
                    }
                    else if constexpr (WIO_OK && !DIRECT_IO && WL <= 6)
                    {
                        // registers (16 contiguous coefficients per thread) -> LDS -> the wave's own
                        // 64-contiguous window -> 512-byte runs per store instruction
                        constexpr int OWL = WIO;
                        T* lw = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            lw[lds_joff<WL>(j)] = v[j];
                        wave_sync();
                        const T* lo = lds + lds_pad(elem_of<OWL>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            v[j] = lo[lds_joff<OWL>(j)];
                        if constexpr (XP == Xp::small_nat_fwd)
                        {
                            // natural-order forward 4-step: from the 64-contiguous window (lanes = consecutive a) into LDS
                            // at the transposed position, then out as it lies (n2 x 32, coalesced)
                            relayout_barrier(v); // every wave is done with the e + (e >> 4) layout
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                lds[xp_nat_lds(xp_nat_pos<K>(static_cast<unsigned>(elem_of<OWL>(t, j))))] = v[j];
                            __syncthreads();
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                            {
                                const unsigned o = static_cast<unsigned>(t + NT * j);
                                st_stream<true>(a.out + (map.base + o), lds[xp_nat_lds(o)]);
                            }
                            return;
                        }
                        const unsigned lane = map.part(elem_of<OWL>(t, 0));
                        if (PMUL_OK && mul_in != nullptr)
                        {
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                v[j] = em.mul(v[j], (mul_in + (map.base + map.part(static_cast<unsigned>(j) << OWL)))[lane]);
                        }
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            st_stream<LAST>((a.out + (map.base + map.part(static_cast<unsigned>(j) << OWL))) + lane, v[j]);
                    }
                    else if constexpr (DIRECT_IO)
                    {
                        if (full_tile)
                        {
                            const unsigned lane = map.part(elem_of<WL>(t, 0));
                            if (PMUL_OK && mul_in != nullptr)
                            {
#pragma unroll
                                for (int j = 0; j < EPT; j++)
                                    v[j] = em.mul(v[j], (mul_in + (map.base + map.part(static_cast<unsigned>(j) << WL)))[lane]);
                            }
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                                st_stream<LAST>((a.out + (map.base + map.part(static_cast<unsigned>(j) << WL))) + lane, v[j]);
                        }
                        else
                        {
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                            {
                                const unsigned long long f = map.flat(elem_of<WL>(t, j));
                                if (f < a.total)
                                    a.out[f] = (PMUL_OK && mul_in != nullptr) ? em.mul(v[j], mul_in[f]) : v[j];
                            }
                        }
                    }
                    else
                    {
                        T* lw = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            lw[lds_joff<WL>(j)] = v[j];
                        __syncthreads();
                        if (full_tile)
                        {
                            const T* lc = lds + lds_pad(t);
                            const unsigned lane = map.part(static_cast<unsigned>(t));
                            if (PMUL_OK && mul_in != nullptr)
                            {
                                // GPU_PolyMul: the pointwise product with the other operand's transform rides
                                // on the final store (halves of 8 bound the live operands)
#pragma unroll
                                for (int half = 0; half < 2; half++)
                                {
                                    T o[EPT / 2];
#pragma unroll
                                    for (int jj = 0; jj < EPT / 2; jj++)
                                        o[jj] = (mul_in + (map.base + map.part(static_cast<unsigned>(NT * (half * (EPT / 2) + jj)))))[lane];
#pragma unroll
                                    for (int jj = 0; jj < EPT / 2; jj++)
                                    {
                                        const int j = half * (EPT / 2) + jj;
                                        (a.out + (map.base + map.part(static_cast<unsigned>(NT * j))))[lane] =
                                            em.mul(lc[NT * j + ((NT * j) >> 4)], o[jj]);
                                    }
                                }
                            }
                            else
                            {
#pragma unroll
                                for (int j = 0; j < EPT; j++)
                                    (a.out + (map.base + map.part(static_cast<unsigned>(NT * j))))[lane] =
                                        lc[NT * j + ((NT * j) >> 4)];
                            }
                        }
                        else
                        {
#pragma unroll
                            for (int j = 0; j < EPT; j++)
                            {
                                const unsigned long long f = map.flat(t + NT * j);
                                if (f < a.total)
                                {
                                    const T x = lds[lds_pad(t + NT * j)];
                                    a.out[f] = (PMUL_OK && mul_in != nullptr) ? em.mul(x, mul_in[f]) : x;
                                }
                            }
                        }
                    }
                }
                else
                {
                    // next round's twiddles are requested before the exchange barrier
                    if constexpr (TW_AHEAD)
                        load_twiddles(std::integral_constant<int, r + 1>{}, tw_next);
                    T* lw = lds + lds_pad(elem_of<WL>(t, 0));
#pragma unroll
                    for (int j = 0; j < EPT; j++)
                        lw[lds_joff<WL>(j)] = v[j];
                    if constexpr (WL <= 6 && SCH::wl_of(r + 1 < NR_ ? r + 1 : r) <= 6)
                        wave_sync(); // both windows inside the wave's sub-block
                    else
                        __syncthreads();
                }
            });
        }

        // Drop-in RNS calls keep their moduli in device memory, so the host cannot pick the kernel family: the preparation
        // kernel classifies the stack and publishes a go-flag (prep.hip: 0 = generic Barrett kernels, 1 = the default lazy
        // range of the word size, 2 / 3 = 64-bit words whose widest modulus has 61 / 62 bits: the 8 q / 4 q range, 4 below);
        // the kernels of a family the flag does not name return at once.
        // GO_LAZY_31Q (forward Merge calls of 64-bit words only): every modulus of the stack has 31 q < 2^64 -- the 31 q
        // kernels (a range correction every fourth stage), what NTTPlan picks from host moduli for the same stack
        constexpr unsigned GO_GENERIC = 0u, GO_LAZY = 1u, GO_LAZY_8Q = 2u, GO_LAZY_4Q = 3u, GO_LAZY_31Q = 4u;
        // generality of a lazy family: a family serves every stack a narrower one serves (31 q needs 31 q < 2^64, 16 q
        // bit <= 60, 8 q bit <= 61, 4 q bit <= 62).  GO_GENERIC: -1
        __host__ __device__ constexpr int family_rank(unsigned state)
        {
            return state == GO_LAZY_31Q ? 0 : (state == GO_LAZY ? 1 : (state == GO_LAZY_8Q ? 2 : (state == GO_LAZY_4Q ? 3 : -1)));
        }
        // Drop-in RNS Merge calls carry their own fall-back INSIDE the preparation kernel (prep.hip: prep_twiddles): when
        // the stack the caller's buffer holds does not fit the one lazy family the host enqueued -- first call of a stack with
        // a 61- / 62-bit prime, moduli rewritten in place with wider ones, moduli outside the documented domain -- the
        // preparation kernel itself transforms the batch, one polynomial per block, stage by stage through global memory
        // with the public Barrett arithmetic (OPERATOR_GPU<T>), and publishes GO_GENERIC so that every fast kernel behind
        // it returns.  Slow (milliseconds), rare, never wrong -- and no generic shadow launch behind any call (rounds 3-4
        // paid two skipped launches, ~6 us each, on EVERY drop-in RNS call).
        template <typename T> struct SlowArgs
        {
            const void* in;
            T* out;
            const T* mul_in;       // GPU_PolyMul: multiplied into the forward result
            const int* poly_order; // *_Poly_Ordered: polynomial p lives in slot poly_order[p]
            unsigned long long polys;
            int col_log;           // >= 0: PerCoefficient layout -- polynomial p is COLUMN p of a 2^n x 2^col_log row-major
                                   // matrix (coefficient i at i * 2^col_log + p); -1: polynomial p at p << n (PerPolynomial)
            unsigned flags;        // F_SIGNED_IN | F_SCALE | F_CENTERED
            int inverse;
            int enabled;           // 0: publish GO_GENERIC only (path = fast-strict: the tests want the lazy families to own the call)
            int force;             // test hook (option rns_force_fallback): treat every stack as not fitting
        };

        template <typename T, int LIM> __device__ __forceinline__ bool not_my_call(const unsigned* go_flag, unsigned flags = 0u)
        {
            constexpr unsigned mine = sizeof(T) != 8 ? GO_LAZY
                                      : (LIM == 4 ? GO_LAZY_4Q : (LIM == 8 ? GO_LAZY_8Q : (LIM == 31 ? GO_LAZY_31Q : GO_LAZY)));
            if (go_flag == nullptr)
                return false;
            const unsigned st = *go_flag;
            // F_VETO_ONLY: the host chose this kernel (host-side modulus); only the table check of a 4-step call can
            // still take the call away (GO_GENERIC)
            return (flags & F_VETO_ONLY) ? (st == GO_GENERIC) : (st != mine);
        }

        // One tile per block.  (Rounds 3-4 ran the 8 q / 4 q families of 64-bit words on a capped grid whose blocks walked
        // the tiles, because they were enqueued behind EVERY drop-in RNS call as shadows and a skipped launch costs ~0.4 ns
        // per block.  Since the family prediction -- host::RnsGuess -- a family is enqueued when it is expected to OWN the
        // call, and the tile loop cost those kernels 150 .. 340 bytes of scratch per lane: removed in round 5.)
        template <typename T, int LIM> struct WalksTiles
        {
            static constexpr bool value = false;
        };
        template <bool WALK, typename F> __device__ __forceinline__ void for_each_block(unsigned, F&& f)
        {
            static_assert(!WALK, "tile-walking grids are gone");
            f(blockIdx.x, gridDim.x);
        }

        // Poly-minor block order (the polynomials of a batch that share a slice of the twiddle / W table run back
        // to back), XCD-aware: the dispatcher sends workgroup b to XCD b % 8 and every XCD has its own L2, so a
        // slice shared by consecutive block indices is fetched from the fabric once per XCD -- up to 8 times
        // (PMC, round 2, the W-streaming phase 1 of C3: 2.15 GB fetched for a 256 MiB table).  With 8 | tiles the tile index takes its
        // TOP three bits from b % 8, so all polynomials of a tile position run on ONE XCD and its slice crosses the
        // fabric once, and the eight tiles in flight at a time lie an eighth of the ring apart (in the low bits
        // they would be neighbouring 512-byte runs of the same rows in a strided pass -- one HBM channel for all
        // eight XCDs: the natural-order forward 4-step ran 9 % slower that way).  Falls back to the plain order
        // for rings of fewer than 8 tiles.
        __device__ __forceinline__ void poly_minor_order(unsigned bx, unsigned batch, int tiles_log, unsigned& poly,
                                                         unsigned& tile, unsigned flags = 0u)
        {
            if (tiles_log >= 3 && (flags & F_PLAIN_ORDER) == 0u)
            {
                const unsigned x = bx & 7u, k = bx >> 3;
                poly = k % batch;
                tile = (x << (tiles_log - 3)) | (k / batch);
            }
            else
            {
                poly = bx % batch;
                tile = bx / batch;
            }
        }

        // ---- the element-by-element 4-step algorithm of the generic kernels, restated for the fast kernels' own blocks ---------
        // (merge_kernels.hpp: merge_pass<FST> -- n1-point transforms of the rows of the n2 x n1 input with n1_table, product
        // with W[address], transposed store -- and the n2-point row transforms with n2_table, n^-1 at the end of an inverse;
        // reference src/lib/ntt_4step/ntt_4step.cu:68-743, 1049-1058, 776-779), with the Barrett operators the generic
        // kernels use: bit for bit their results for ANY three tables.  What a checked 4-step call computes when the table
        // check took the call away from the fast kernels (GO_GENERIC) and the host asked them to be their own fall-back
        // (F_SELF_FALLBACK): the first kernel of every plan does phase 1, one-tile kernels and the inverse row pass of the
        // rings 2^14 .. 2^16 phase 2 as well, so fewer (or no) generic launches sit behind the call.  Speed is beside the
        // point: one thread per butterfly and stage, block barriers.
        // 2^lg-point transforms of the `nel` consecutive words at buf (LDS, or the block's own words of global memory)
        template <typename T, bool INV, int NT>
        __device__ __forceinline__ void fs_stages(T* buf, int nel, const T* table, int lg, const dev::ModCtx<T>& m)
        {
            const int t = threadIdx.x;
            for (int s = 0; s < lg; s++)
            {
                const int P = INV ? s : (lg - 1 - s); // distance 2^P: Cooley-Tukey from the top, Gentleman-Sande from the bottom
                for (int b = t; b < nel / 2; b += NT)
                {
                    const int e0 = ((b >> P) << (P + 1)) | (b & ((1 << P) - 1)), e1 = e0 | (1 << P);
                    const unsigned idx = static_cast<unsigned>(e0) & ((1u << lg) - 1u);
                    const T w = table[idx >> (P + 1)];
                    T U = buf[e0], V = buf[e1];
                    if constexpr (INV)
                        dev::gs_butterfly(U, V, w, m);
                    else
                        dev::ct_butterfly(U, V, w, m);
                    buf[e0] = U;
                    buf[e1] = V;
                }
                __syncthreads();
            }
        }
        // phase 1 on the 4096 words [tile << 12, ...) of polynomial poly: 2^(12 - l1) rows of n1
        template <typename T, bool INV, int NT>
        __device__ __forceinline__ void fs_phase1_tile(const LazyArgsT<T>& a, T* lds, unsigned long long poly, unsigned tile, int l1,
                                                       int l2, const dev::ModCtx<T>& m)
        {
            const int t = threadIdx.x;
            const T* in = static_cast<const T*>(a.in) + (poly << (l1 + l2)) + (static_cast<unsigned long long>(tile) << 12);
            T* out = a.out + (poly << (l1 + l2));
            for (int e = t; e < 4096; e += NT)
                lds[e] = in[e];
            __syncthreads();
            fs_stages<T, INV, NT>(lds, 4096, a.fs_n1, l1, m);
            const unsigned row0 = tile << (12 - l1);
            for (int e = t; e < 4096; e += NT)
            {
                const unsigned r = row0 + (static_cast<unsigned>(e) >> l1), i = static_cast<unsigned>(e) & ((1u << l1) - 1u);
                const unsigned long long widx = (static_cast<unsigned long long>(i) << l2) + r;
                out[widx] = m.mul(lds[e], a.fs_w[widx]);
            }
            __syncthreads();
        }
        // phase 2 on the 4096 words [chunk << 12, ...) of `out`, in place (rows of n2 <= 4096)
        template <typename T, bool INV, int NT>
        __device__ __forceinline__ void fs_phase2_chunk(const LazyArgsT<T>& a, unsigned long long chunk, int l2, const dev::ModCtx<T>& m,
                                                        T ninv)
        {
            T* buf = a.out + (chunk << 12);
            fs_stages<T, INV, NT>(buf, 4096, a.fs_n2, l2, m);
            if constexpr (INV)
                for (int e = threadIdx.x; e < 4096; e += NT)
                    buf[e] = m.mul(buf[e], ninv);
        }
        // "this call is not mine and the host enqueued nothing else for it": the table check vetoed it (host-side modulus), or the
        // device-side modulus turned out to need another family than the one the host predicted -- or none
        template <typename T, int LIM> __device__ __forceinline__ bool self_fallback_call(const LazyArgsT<T>& a)
        {
            return (a.flags & F_SELF_FALLBACK) != 0u && not_my_call<T, LIM>(a.go_flag, a.flags);
        }
        // modulus and n^-1 of a 4-step call (one modulus: host-side, or slot 0 of the device array) for the Barrett operators
        template <typename T> __device__ __forceinline__ dev::ModCtx<T> fs_modulus(const LazyArgsT<T>& a)
        {
            if (a.mods != nullptr)
            {
                const Modulus<T> md = a.mods[0];
                return dev::ModCtx<T>{md.value, md.bit, md.mu};
            }
            return dev::ModCtx<T>{a.q, a.q_bit, a.q_mu};
        }
        template <typename T> __device__ __forceinline__ T fs_ninv(const LazyArgsT<T>& a)
        {
            return (a.ninv_arr != nullptr) ? a.ninv_arr[0].w : a.ninv.w;
        }
        // both phases on ONE polynomial that fills the tile (one-tile rings, n1 = 32), in LDS
        template <typename T, int TLOG, bool INV>
        __device__ __forceinline__ void fourstep_tile_generic(const LazyArgsT<T>& a, T* lds, unsigned long long poly,
                                                              const dev::ModCtx<T>& m, T ninv)
        {
            constexpr int NT = LTile<TLOG>::NT, N = 1 << TLOG, L1 = XP_L1, L2 = TLOG - XP_L1;
            static_assert(N == NT * EPT, "16 coefficients per thread");
            const int t = threadIdx.x;
            const T* in = static_cast<const T*>(a.in) + (poly << TLOG);
            T* out = a.out + (poly << TLOG);
            for (int e = t; e < N; e += NT)
                lds[e] = in[e];
            __syncthreads();
            fs_stages<T, INV, NT>(lds, N, a.fs_n1, L1, m);
            // out[(i << l2) + r] = row r, element i, times W[(i << l2) + r]
            T tmp[EPT];
#pragma unroll
            for (int k = 0; k < EPT; k++)
            {
                const unsigned o = static_cast<unsigned>(t + NT * k);
                const unsigned i = o >> L2, r = o & ((1u << L2) - 1u);
                tmp[k] = m.mul(lds[(r << L1) | i], a.fs_w[o]);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EPT; k++)
                lds[t + NT * k] = tmp[k];
            __syncthreads();
            fs_stages<T, INV, NT>(lds, N, a.fs_n2, L2, m);
            for (int e = t; e < N; e += NT)
                out[e] = INV ? m.mul(lds[e], ninv) : lds[e];
        }

        template <typename T, int TLOG, bool INV, bool CONTIG, int K, int IN_BOUND, bool LAST, int LIM = 0, int SKIP = 0>
        __global__ __launch_bounds__(LTile<TLOG>::NT, (LOcc<TLOG, T>::WAVES)) void merge_pass_lazy(LazyArgsT<T> a)
        {
            using M = lazy::Mod<T, LIM>;
            // forward kernels: SKIP != 0 is only a MARKER -- the last pass of a forward 4-step whose n2-long rows fit one tile
            // (host::launch_fourstep_fwd_last_lazy), instantiations of their own that can carry the fall-back below
            constexpr int SK = INV ? SKIP : 0;
            using SCH = PassSched<TLOG, INV, CONTIG, K, IN_BOUND, M::LIMIT, M::TB, SK>;
            // single-round passes with coalesced register windows never touch LDS
            constexpr bool NEEDS_LDS = (SCH::NR > 1) || (SCH::wl_of(0) < 4);
            __shared__ T lds[NEEDS_LDS ? LTile<TLOG>::LDS_ELEMS : 1];

            // (the inverse row pass of the 4-step rings 2^14 .. 2^16 and the forward last pass of 2^14 .. 2^17 as their own
            // fall-back: phase 2 of the generic algorithm on the block's 4096 words; inverse: a.n = log2 n2, forward: a.n =
            // log2 N and a.n2_log = log2 n1)
            if constexpr (SKIP != 0 && TLOG == 12 && CONTIG)
                if (self_fallback_call<T, LIM>(a))
                {
                    if constexpr (INV)
                        fs_phase2_chunk<T, true, LTile<12>::NT>(a, blockIdx.x, a.n, fs_modulus(a), fs_ninv(a));
                    else
                        fs_phase2_chunk<T, false, LTile<12>::NT>(a, blockIdx.x, a.n - a.n2_log, fs_modulus(a), T(0));
                    return;
                }
            // RNS calls: the twiddle-prep kernel publishes which kernel family the stack of moduli needs (not_my_call);
            // the other families, launched alongside, return here
            if (not_my_call<T, LIM>(a.go_flag, a.flags))
                return;
            for_each_block<WalksTiles<T, LIM>::value>(
                static_cast<unsigned>((a.total + LTile<TLOG>::TILE - 1) >> TLOG), [&](unsigned bidx, unsigned nblk) {
            // the block's modulus: polynomial index of the tile % mod_count (tiles never straddle
            // polynomials with different moduli here: RNS calls with N < tile and mod_count > 1
            // are routed to the generic kernels by the host)
            T qv = a.q, qb = a.q_bit, qm = a.q_mu;
            int mi = 0;
            // a.batch > 1: poly-minor block order (block b -> tile b / batch of polynomial b % batch), asked
            // for by the host for big rings so that the polynomials of a batch read each slice of the
            // twiddle table back to back (L2 hits instead of one HBM read per polynomial)
            // F_REVERSE: consecutive passes of one transform walk the batch in opposite directions, so a pass
            // starts on the data its predecessor wrote last -- the part still in the 256 MiB Infinity Cache
            const unsigned bx = (a.flags & F_REVERSE) ? (nblk - 1u - bidx) : bidx;
            long long blk = static_cast<long long>(bx);
            if (a.batch > 1)
            {
                unsigned poly, tile;
                poly_minor_order(bx, static_cast<unsigned>(a.batch), a.n - TLOG, poly, tile, a.flags);
                blk = static_cast<long long>(
                    uniform64((static_cast<unsigned long long>(poly) << (a.n - TLOG)) | tile));
            }
            if (a.mods != nullptr)
            {
                const LTileMap<TLOG, CONTIG, K> map(a.n, a.p_lo, static_cast<unsigned long long>(blk));
                unsigned long long poly = map.flat(0) >> a.poly_shift;
                // RNS stack of rings below one tile, from 1024 coefficients: a wave's 16 x 64 coefficients lie in ONE
                // polynomial (thread bits >= K - 4 >= 6 select it), so the modulus is wave-uniform and stays in scalar
                // registers; smaller rings take the per-lane-modulus kernels (merge_pass_lazy_vqc)
                if constexpr (CONTIG && K < TLOG && K - R >= 6 && IN_BOUND == 1 && LAST)
                    poly += threadIdx.x >> (K - R);
                mi = static_cast<int>(uniform32(static_cast<unsigned>(poly % static_cast<unsigned>(a.mod_count))));
                const Modulus<T> md = a.mods[a.mod_order != nullptr ? a.mod_order[mi] : mi];
                qv = md.value;
                qb = md.bit;
                qm = md.mu;
            }
            pass_body<T, TLOG, INV, CONTIG, K, IN_BOUND, LAST, Fst::none, LIM, Xp::none, SK>(a, lds, qv, qb, qm, mi, 0, 0, blk);
                });
        }

        // PerCoefficient layout with an RNS stack: strided pass with per-lane moduli (pass_body VQ).  One lazy family serves
        // every modulus of the documented domain -- 64-bit words the 4 q range (<= 62 bit), 32-bit words the default one
        // -- so the go-flag only separates it from the generic kernels.  a.n = log2(N * batch) (the virtual ring the
        // columns form), a.col_log = log2 batch, a.mod_shift = log2 N, a.mods / a.ninv_arr / a.norm_arr per modulus.
        template <typename T> struct VqLim
        {
            static constexpr int LIM = (sizeof(T) == 8) ? 4 : 0;
        };
        template <typename T, bool INV, int K, int IN_BOUND, bool LAST>
        __global__ __launch_bounds__(LTile<12>::NT, (LOcc<12, T>::WAVES)) void merge_pass_lazy_vq(LazyArgsT<T> a)
        {
            constexpr int LIM = VqLim<T>::LIM;
            using M = lazy::Mod<T, LIM, true>;
            using SCH = PassSched<12, INV, false, K, IN_BOUND, M::LIMIT, M::TB, 0>;
            constexpr bool NEEDS_LDS = (SCH::NR > 1) || (SCH::wl_of(0) < 4);
            __shared__ T lds[NEEDS_LDS ? LTile<12>::LDS_ELEMS : 1];
            if (a.go_flag != nullptr && *a.go_flag == GO_GENERIC)
                return;
            pass_body<T, 12, INV, false, K, IN_BOUND, LAST, Fst::none, LIM, Xp::none, 0, true>(a, lds, 0, 0, 0, 0, 0, 0,
                                                                                       static_cast<long long>(blockIdx.x));
        }

        // RNS stacks of rings of 16 .. 512 coefficients (reference ForwardCoreLowRing / InverseCoreLowRing, RNS forms,
        // src/lib/ntt_merge/ntt.cu:116-219, 326-433): one contiguous pass over a 4096-coefficient tile that holds 2^(12 - K)
        // polynomials of DIFFERENT moduli, several of them per wave -- pass_body with per-lane moduli (VQ).  LIM: the lazy
        // range the stack needs (0 the default range, 8 / 4: 64-bit words with a 61- / 62-bit prime); a.lim = 31 on a
        // LIM = 0 launch: the launch stands in for the 31 q family (it accepts the go-flag state GO_LAZY_31Q instead of GO_LAZY)
        template <typename T, bool INV, int K, int LIM>
        __global__ __launch_bounds__(LTile<12>::NT, (LOcc<12, T>::WAVES)) void merge_pass_lazy_vqc(LazyArgsT<T> a)
        {
            __shared__ T lds[LTile<12>::LDS_ELEMS];
            if (a.go_flag != nullptr)
            {
                const unsigned mine = sizeof(T) != 8 ? GO_LAZY
                                      : (LIM == 4 ? GO_LAZY_4Q : (LIM == 8 ? GO_LAZY_8Q : (a.lim == 31 ? GO_LAZY_31Q : GO_LAZY)));
                if (*a.go_flag != mine)
                    return;
            }
            for_each_block<WalksTiles<T, LIM>::value>(
                static_cast<unsigned>((a.total + LTile<12>::TILE - 1) >> 12), [&](unsigned bidx, unsigned) {
                    pass_body<T, 12, INV, true, K, 1, true, Fst::none, LIM, Xp::none, 0, true>(
                        a, lds, 0, 0, 0, 0, 0, 0, static_cast<long long>(bidx));
                });
        }

        // block -> (polynomial, tile of the polynomial) for the transposing row passes of the natural-order 4-step: tile =
        // (column run << log2 row blocks) | row block.  a.batch > 1 (the host asks for it from 2^20): poly-minor order, so
        // the polynomials of a batch read each slice of the ring's Merge table back to back (an L2 hit instead of one HBM
        // read per polynomial: the table is as large as the polynomial).  Plain poly-minor order, not the XCD-aware one of
        // the Merge passes: with the grouping 2^20 x 64 measured 0.734 against 0.693 ms (2^24 x 64: 11.78 / 11.75 ms)
        template <typename T>
        __device__ __forceinline__ void nat_block(const LazyArgsT<T>& a, int tlog, unsigned& poly, unsigned& tile)
        {
            const int tiles_log = a.n - tlog;
            if (a.batch > 1)
                poly_minor_order(blockIdx.x, static_cast<unsigned>(a.batch), tiles_log, poly, tile, a.flags | F_PLAIN_ORDER);
            else
            {
                poly = blockIdx.x >> tiles_log;
                tile = blockIdx.x & ((1u << tiles_log) - 1u);
            }
            poly = uniform32(poly);
            tile = uniform32(tile);
        }

        // natural-order 4-step, last forward pass: block b -> nat_block;
        // a.row_log = log2 n2 (row length and stride of the row-major side), a.n2_log = log2 n1 (row stride of the
        // column-major side), a.n = log2 N: the stages are the low K stages of the Merge transform of the whole ring, so
        // the twiddles come from the ring's Merge table and depend on the row as well (4-step in Merge form)
        template <typename T, int TLOG, int K, int IN_BOUND, int LIM = 0>
        __global__ __launch_bounds__(LTile<TLOG>::NT, (LOcc<TLOG, T>::WAVES)) void fourstep_nat_last_lazy(LazyArgsT<T> a)
        {
            __shared__ T lds[LTile<TLOG>::LDS_ELEMS_FST];
            if (not_my_call<T, LIM>(a.go_flag, a.flags))
                return;
            constexpr int RB = TLOG - K;
            const int rb_log = a.n2_log - RB;
            unsigned poly, tile;
            nat_block(a, TLOG, poly, tile);
            const unsigned rb = tile & ((1u << rb_log) - 1u), seg = tile >> rb_log;
            pass_body<T, TLOG, false, true, K, IN_BOUND, true, Fst::nat_rows, LIM>(a, lds, a.q, a.q_bit, a.q_mu, 0, poly, rb, -1, seg);
        }

        // natural-order 4-step, inverse direction (the forward passes run backwards):
        //   first pass  transposed load of the column-major input + the 2^K low row stages
        //               (Gentleman-Sande), stored row-major, lazy; block order as fourstep_nat_last_lazy
        template <typename T, int TLOG, int K, int LIM = 0>
        __global__ __launch_bounds__(LTile<TLOG>::NT, (LOcc<TLOG, T>::WAVES)) void fourstep_nat_first_inv_lazy(LazyArgsT<T> a)
        {
            __shared__ T lds[LTile<TLOG>::LDS_ELEMS_FST];
            if (not_my_call<T, LIM>(a.go_flag, a.flags))
                return;
            constexpr int RB = TLOG - K;
            const int rb_log = a.n2_log - RB;
            unsigned poly, tile;
            nat_block(a, TLOG, poly, tile);
            const unsigned rb = tile & ((1u << rb_log) - 1u), seg = tile >> rb_log;
            pass_body<T, TLOG, true, true, K, 1, false, Fst::nat_rows, LIM>(a, lds, a.q, a.q_bit, a.q_mu, 0, poly, rb, -1, seg);
        }

        // forward 4-step, first pass in Merge form with the transposed gather (Xp::first_gather): the first strided pass of the ring's
        // Merge plan, K >= log2 n1 stages, reading the n2 x n1 input; everything behind it is the Merge plan itself.
        // a.n = log2 N, a.p_lo = log2 N - K, a.n2_log = log2 n1; grid = batch * N / 4096 blocks in tile order
        template <typename T, int K, int LIM = 0>
        __global__ __launch_bounds__(LTile<12>::NT, (LOcc<12, T>::WAVES)) void fourstep_first_lazy(LazyArgsT<T> a)
        {
            __shared__ T lds[LTile<12>::LDS_ELEMS];
            if (self_fallback_call<T, LIM>(a)) // phase 1 of the generic algorithm on 4096-word tile blockIdx.x (a.n2_log = log2 n1 here)
            {
                const dev::ModCtx<T> em = fs_modulus(a);
                const unsigned tiles_log = static_cast<unsigned>(a.n - 12);
                fs_phase1_tile<T, false, LTile<12>::NT>(a, lds, blockIdx.x >> tiles_log, blockIdx.x & ((1u << tiles_log) - 1u),
                                                         a.n2_log, a.n - a.n2_log, em);
                return;
            }
            if (not_my_call<T, LIM>(a.go_flag, a.flags))
                return;
            T qv = a.q, qb = a.q_bit, qm = a.q_mu;
            if (a.mods != nullptr)
            {
                const Modulus<T> md = a.mods[0];
                qv = md.value;
                qb = md.bit;
                qm = md.mu;
            }
            for_each_block<WalksTiles<T, LIM>::value>(static_cast<unsigned>(a.total >> 12), [&](unsigned bidx, unsigned nblk) {
                const unsigned bx = (a.flags & F_REVERSE) ? (nblk - 1u - bidx) : bidx;
                pass_body<T, 12, false, false, K, 1, false, Fst::none, LIM, Xp::first_gather>(a, lds, qv, qb, qm, 0, 0, 0,
                                                                              static_cast<long long>(bx));
            });
        }

        // inverse 4-step, first pass in Merge form (Fst::inv_first).  With e = (a << l1) | b the natural index of the result x =
        // MergeINTT_w(in), GPU_4STEP_NTT stores out[(b << l2) | a] = x[e]: the LOW l1 index bits go to the top.  The first
        // pass of the ring's inverse Merge plan -- 12 contiguous Gentleman-Sande stages on a tile that lies as it stands in
        // the spectrum -- holds all of b and the low 12 - l1 bits of a, so it can store the tile transposed: 2^l1 rows
        // (b) of 2^(12 - l1) consecutive words (128 B .. 1 KiB runs).  Behind it every remaining stage works on index bits
        // of a, i.e. inside the n2-long rows of `out`: ordinary strided inverse passes of an n2-point ring (host side:
        // fourstep_run_lazy), whose twiddle slots are a prefix of the ring's own table.  No W stream, no W product.
        // a.n = log2 N, a.n2_log = log2 n2, a.poly_shift = log2 N; blocks in merge_pass_lazy's order.
        // TLOG = 13 / 14 (host::fourstep_inv_tile: 64-bit rings 2^21 / 2^22, 32-bit rings 2^20 .. 2^22): the big tile of the
        // ring's inverse Merge plan does 13 / 14 stages, which leaves ONE strided pass of at most 8 -- two sweeps, like
        // the Merge plan of that ring
        template <typename T, int L1, int LIM = 0, int TLOG = 12>
        __global__ __launch_bounds__(LTile<TLOG>::NT, (LOcc<TLOG, T>::WAVES)) void fourstep_inv_first_lazy(LazyArgsT<T> a)
        {
            __shared__ T lds[LTile<TLOG>::LDS_ELEMS_FST];
            if (self_fallback_call<T, LIM>(a)) // phase 1 of the generic algorithm on the 2^(TLOG - 12) 4096-word tiles of block blockIdx.x
            {
                const dev::ModCtx<T> em = fs_modulus(a);
                const unsigned tiles_log = static_cast<unsigned>(a.n - TLOG);
                const unsigned long long poly = blockIdx.x >> tiles_log;
                const unsigned big = blockIdx.x & ((1u << tiles_log) - 1u);
                for (unsigned sub = 0; sub < (1u << (TLOG - 12)); sub++)
                    fs_phase1_tile<T, true, LTile<TLOG>::NT>(a, lds, poly, (big << (TLOG - 12)) + sub, L1, a.n - L1, em);
                return;
            }
            if (not_my_call<T, LIM>(a.go_flag, a.flags))
                return;
            T qv = a.q, qb = a.q_bit, qm = a.q_mu;
            if (a.mods != nullptr)
            {
                const Modulus<T> md = a.mods[0];
                qv = md.value;
                qb = md.bit;
                qm = md.mu;
            }
            for_each_block<WalksTiles<T, LIM>::value>(static_cast<unsigned>(a.total >> TLOG), [&](unsigned bidx, unsigned nblk) {
                const unsigned bx = (a.flags & F_REVERSE) ? (nblk - 1u - bidx) : bidx;
                const int tiles_log = a.n - TLOG;
                unsigned poly, tile;
                if (a.batch > 1)
                    poly_minor_order(bx, static_cast<unsigned>(a.batch), tiles_log, poly, tile, a.flags);
                else
                {
                    poly = bx >> tiles_log;
                    tile = bx & ((1u << tiles_log) - 1u);
                }
                pass_body<T, TLOG, true, true, TLOG, 1, false, Fst::inv_first, LIM, Xp::none, 0, false, L1>(a, lds, qv, qb, qm, 0, uniform32(poly),
                                                                                      uniform32(tile));
            });
        }

        // 4-step transform of a ring that fits one tile (2^12 .. 2^14): ONE contiguous Merge pass over the whole ring
        // with the transposition of the natural-order side done in LDS (XP above).  a.tw = Merge table of the ring.
        // NAT: the natural-order extension (NTT_4STEP_CPU order on the spectrum side, Xp::small_nat_fwd / small_nat_inv) instead of the
        // reference layout (Xp::small_fwd / small_inv)
        template <typename T, int TLOG, bool INV, int K, int LIM = 0, bool NAT = false>
        __global__ __launch_bounds__(LTile<TLOG>::NT, (LOcc<TLOG, T>::WAVES)) void fourstep_small_lazy(LazyArgsT<T> a)
        {
            static_assert(K >= 12 && K == TLOG, "one-tile 4-step rings fill their tile: 32 x n2 with n2 >= 128");
            __shared__ T lds[LTile<TLOG>::LDS_ELEMS];
            T qv = a.q, qb = a.q_bit, qm = a.q_mu;
            if (a.mods != nullptr)
            {
                const Modulus<T> md = a.mods[0];
                qv = md.value;
                qb = md.bit;
                qm = md.mu;
            }
            if constexpr (!NAT)
                if (self_fallback_call<T, LIM>(a))
                {
                    fourstep_tile_generic<T, TLOG, INV>(a, lds, blockIdx.x, fs_modulus(a), fs_ninv(a));
                    return;
                }
            if (not_my_call<T, LIM>(a.go_flag, a.flags))
                return;
            for_each_block<WalksTiles<T, LIM>::value>(
                static_cast<unsigned>((a.total + LTile<TLOG>::TILE - 1) >> TLOG), [&](unsigned bidx, unsigned) {
                    pass_body<T, TLOG, INV, true, K, 1, true, Fst::none, LIM,
                              NAT ? (INV ? Xp::small_nat_inv : Xp::small_nat_fwd) : (INV ? Xp::small_inv : Xp::small_fwd)>(
                        a, lds, qv, qb, qm, 0, 0, 0, static_cast<long long>(bidx));
                });
        }

    } // namespace kern
} // namespace gpuntt
