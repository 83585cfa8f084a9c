// merge_kernels.hpp -- tile-pass kernels for the Merge NTT on gfx950 (wave64, 160 KiB LDS).
//
// Replaces the reference's ForwardCore / InverseCore / *LowRing kernel family
// (reference src/lib/ntt_merge/ntt.cu:11-1552) with ONE templated kernel:
//
//   a "pass" performs K consecutive radix-2 stages on tiles of 2^TL = 4096 coefficients;
//   each of the 256 threads keeps 2^R = 16 coefficients in VGPRs and runs up to R stages
//   per "round" entirely in registers; rounds exchange through a padded LDS tile
//   (one ds_write + one ds_read per coefficient per round, conflict-free by the
//   e + (e >> 4) padding for both the 4-byte and 8-byte element sizes).
//
//   CONTIG pass : tile = 4096 contiguous coefficients (stage distances 2^(K-1) .. 1)
//   STRIDED pass: tile = 2^K rows of 2^(TL-K) contiguous coefficients, row stride 2^p_lo
//                 (stage distances 2^(p_lo+K-1) .. 2^p_lo), coalesced in runs of >= 128 B
//
// A transform of size 2^n is a host-planned list of passes (merge_ntt.hip): n <= 12 is a
// single CONTIG pass (several polynomials per tile when n < 12), larger n add STRIDED
// passes in front (forward) or behind (inverse).
//
// Twiddle indexing follows the reference's bit-reversed tables (ntt.cu:507-514, 670-679):
// for the stage whose butterfly distance is 2^P, element index idx uses
//   table[(mod_index << n) + (idx >> (P+1))]                       (X^N - 1)
//   table[(mod_index << n) + (1 << (n-1-P)) + (idx >> (P+1))]      (X^N + 1)
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "modarith.hpp"

namespace gpuntt
{
    namespace kern
    {
        constexpr int R = 4;   // log2 coefficients per thread
        constexpr int NT = 256; // threads per workgroup (4 waves)
        constexpr int TL = 12; // log2 tile size
        constexpr int TILE = 1 << TL;
        constexpr int EPT = 1 << R;
        constexpr int LDS_ELEMS = TILE + (TILE >> 4) + (TILE >> 5);

        enum : unsigned
        {
            F_NEGACYCLIC = 1u, // X^N + 1 tables
            F_SIGNED_IN = 2u,  // forward first pass: input is signed, reduce to [0, q)
            F_SCALE = 4u,      // inverse last pass: multiply by n^-1
            F_CENTERED = 8u,   // inverse last pass: emit centred signed residues
            F_FOURSTEP_T = 16u, // 4-step phase 1: transposed store with W multiply
            F_MULTI = 32u, // a tile may span polynomials with different moduli (RNS, N < tile)
            F_REVERSE = 128u, // fast kernels: walk the tiles from the last to the first (see run_transform_lazy)
            F_COLMOD = 64u, // PerCoefficient RNS: the modulus follows the COLUMN (flat & (2^n2_log - 1)) % mod_count
            F_PLAIN_ORDER = 256u, // fast kernels: poly-minor block order without the XCD grouping (GPUNTT_XCD_ORDER=0, A/B timing)
            F_VETO_ONLY = 512u, // fast kernels: the go-flag only carries a veto (4-step calls with a host-side modulus: the
                                // host picked the kernel families, the table check may still hand the call to the generic kernels)
            F_SELF_FALLBACK = 1024u // one-tile 4-step kernels: when the veto fires, run the element-by-element algorithm on
                                    // the tile yourself (kern::fourstep_tile_generic) -- nothing is enqueued behind the call
        };

        template <typename T> struct PassArgs
        {
            const void* in;
            T* out;
            const T* roots;
            const Modulus<T>* mods; // device array (RNS) or nullptr
            Modulus<T> mod;         // single modulus, used when mods == nullptr
            const T* ninv_arr;      // device array (RNS) or nullptr
            T ninv;
            const T* w_table;       // 4-step W matrix (F_FOURSTEP_T) or nullptr
            const unsigned* skip_flag; // device word or nullptr: the go-flag of a drop-in RNS call (prep.hip)
            unsigned skip_value;       // 0: return when *skip_flag != 0 (a lazy family owns the call); s > 0: return when
                                       // *skip_flag == s (only the lazy family of state s was enqueued in front of this
                                       // launch -- every other state is this kernel's job, lazy_launch.hpp: RnsGuess)
            const int* mod_order;  // *_Modulus_Ordered: polynomial p uses prime mod_order[p % mod_count]
            const int* poly_order; // *_Poly_Ordered: polynomial p lives in slot poly_order[p] of in/out
            unsigned long long total; // batch * N coefficients
            int n;          // log2 of the transform length (twiddle indexing)
            int poly_shift; // log2 of the polynomial length (modulus selection: flat >> poly_shift)
            int root_shift; // RNS table stride log2 (table of modulus i at i << root_shift), -1: shared
            int mod_count;
            int p_lo;   // STRIDED: global bit position of the lowest stage of this pass
            int n2_log; // F_FOURSTEP_T: log2 n2 (rows of the phase-1 input)
            unsigned flags;
        };

        __device__ __forceinline__ int lds_pad(int e) { return e + (e >> 4); }

        // padding of the final exchange of the 4-step phase-1 kernel: writes have lane stride 16,
        // the transposed reads lane stride 2^K (one row of the n2 x n1 input per lane)
        template <int K> __device__ __forceinline__ int lds_pad_t(int e) { return e + (e >> 4) + (e >> K); }

        template <int WL> __device__ __forceinline__ int elem_of(int t, int j)
        {
            return (t & ((1 << WL) - 1)) | (j << WL) | ((t >> WL) << (WL + R));
        }

        template <typename F, int... Is>
        __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>)
        {
            (f(std::integral_constant<int, Is>{}), ...);
        }
        template <int N, typename F> __device__ __forceinline__ void static_for(F&& f)
        {
            static_for_impl(f, std::make_integer_sequence<int, N>{});
        }

        // ---- tile geometry -------------------------------------------------------------
        template <bool CONTIG, int K> struct Geo
        {
            static constexpr int L = CONTIG ? 0 : (TL - K); // contiguous run bits of a STRIDED tile
            static constexpr int NR = (K + R - 1) / R;      // rounds
        };

        template <typename T, bool CONTIG, int K> struct TileMap
        {
            unsigned long long base; // CONTIG: flat base; STRIDED: flat index bits supplied by the block
            int p_lo;
            __device__ __forceinline__ TileMap(const PassArgs<T>& a, unsigned tile) : TileMap(a.n, a.p_lo, tile) {}
            __device__ __forceinline__ TileMap(int n, int pass_p_lo, unsigned tile)
            {
                constexpr int L = Geo<CONTIG, K>::L;
                if constexpr (CONTIG)
                {
                    base = static_cast<unsigned long long>(tile) << TL;
                    p_lo = 0;
                }
                else
                {
                    p_lo = pass_p_lo;
                    const unsigned long long blk = tile;
                    const unsigned long long poly = blk >> (n - TL);
                    const unsigned long long b = blk & ((1ull << (n - TL)) - 1);
                    const unsigned long long xb = b & ((1ull << (p_lo - L)) - 1);
                    const unsigned long long hi = b >> (p_lo - L);
                    base = (poly << n) | (hi << (p_lo + K)) | (xb << L);
                }
            }
            // flat coefficient index of tile element e
            __device__ __forceinline__ unsigned long long flat(int e) const
            {
                constexpr int L = Geo<CONTIG, K>::L;
                if constexpr (CONTIG)
                    return base + static_cast<unsigned>(e);
                else
                    return base | (static_cast<unsigned long long>(e >> L) << p_lo) |
                           static_cast<unsigned>(e & ((1 << L) - 1));
            }
            // global bit position of the stage that sits at tile bit position p
            __device__ __forceinline__ int gpos(int p) const
            {
                constexpr int L = Geo<CONTIG, K>::L;
                if constexpr (CONTIG)
                    return p;
                else
                    return p_lo + (p - L);
            }
        };

        template <typename T> struct Ctx
        {
            dev::ModCtx<T> m;
            T ninv;
            unsigned long long root_base;
        };

        template <typename T>
        __device__ __forceinline__ Ctx<T> make_ctx(const PassArgs<T>& a, unsigned long long poly)
        {
            Ctx<T> c;
            if (a.mods != nullptr)
            {
                int mi = static_cast<int>(poly % static_cast<unsigned>(a.mod_count));
                if (a.mod_order != nullptr)
                    mi = a.mod_order[mi]; // reference ForwardCoreModulusOrdered, ntt.cu:3117-3118
                const Modulus<T> md = a.mods[mi];
                c.m = dev::ModCtx<T>{md.value, md.bit, md.mu};
                c.ninv = (a.ninv_arr != nullptr) ? a.ninv_arr[mi] : a.ninv;
                c.root_base = (a.root_shift >= 0) ? (static_cast<unsigned long long>(mi) << a.root_shift) : 0ull;
            }
            else
            {
                c.m = dev::ModCtx<T>{a.mod.value, a.mod.bit, a.mod.mu};
                c.ninv = a.ninv;
                c.root_base = 0;
            }
            return c;
        }

        // index of the polynomial a flat coefficient index belongs to: the row-major batch slot, or --
        // PerCoefficient layout -- the matrix column (reference ForwardCoreTranspose, ntt.cu:1737-1741:
        // batch_index % mod_count picks the modulus, tables at mod_index << log_row)
        template <typename T>
        __device__ __forceinline__ unsigned long long poly_of(const PassArgs<T>& a, unsigned long long flat)
        {
            if (a.flags & F_COLMOD)
                return flat & ((1ull << a.n2_log) - 1ull);
            return flat >> a.poly_shift;
        }

        // logical flat index (polynomial p, coefficient i) -> memory index; only *_Poly_Ordered
        // calls remap the polynomial slot (reference ForwardCorePolyOrdered, ntt.cu:3797-3798)
        template <typename T>
        __device__ __forceinline__ unsigned long long phys(const PassArgs<T>& a, unsigned long long flat)
        {
            if (a.poly_order == nullptr)
                return flat;
            const unsigned long long slot = static_cast<unsigned>(a.poly_order[flat >> a.poly_shift]);
            return (slot << a.poly_shift) | (flat & ((1ull << a.poly_shift) - 1));
        }

        template <typename T>
        __device__ __forceinline__ T load_in(const PassArgs<T>& a, unsigned long long flat, T q)
        {
            using S = typename std::make_signed<T>::type;
            if (flat >= a.total)
                return 0;
            flat = phys(a, flat);
            if (a.flags & F_SIGNED_IN)
            {
                S v = static_cast<const S*>(a.in)[flat];
                return (v < 0) ? static_cast<T>(q + static_cast<T>(v)) : static_cast<T>(v);
            }
            return static_cast<const T*>(a.in)[flat];
        }

        // one register round: r stages at tile positions [P0 .. ] in pass order
        template <typename T, bool INV, bool CONTIG, int K, int ROUND>
        struct Round
        {
            using G = Geo<CONTIG, K>;
            static constexpr int L = G::L;
            static constexpr int STAGES = (ROUND == G::NR - 1) ? (K - R * (G::NR - 1)) : R;
            // forward (CT): stages run from the highest tile position downwards
            // inverse (GS): from the lowest upwards
            static constexpr int FIRST_POS = INV ? (L + ROUND * R) : (L + K - 1 - ROUND * R);
            static constexpr int WL_RAW = INV ? FIRST_POS : (FIRST_POS - R + 1);
            static constexpr int WL = WL_RAW < 0 ? 0 : (WL_RAW > TL - R ? TL - R : WL_RAW);
            static constexpr bool DIRECT_IO = (WL >= 4); // >= 16 contiguous coefficients per row

            template <typename MAP>
            static __device__ __forceinline__ void butterflies(T (&v)[EPT], const PassArgs<T>& a,
                                                               const MAP& map, const Ctx<T>& blk_ctx,
                                                               int t)
            {
                static_for<STAGES>([&](auto s_) {
                    constexpr int s = decltype(s_)::value;
                    constexpr int p = INV ? (FIRST_POS + s) : (FIRST_POS - s);
                    constexpr int jb = p - WL;
                    static_assert(jb >= 0 && jb < R, "stage bit outside the register window");
                    const int P = map.gpos(p);
                    static_for<EPT / 2>([&](auto h_) {
                        constexpr int h = decltype(h_)::value;
                        // insert a 0 at bit jb of h
                        constexpr int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                        constexpr int j1 = j0 | (1 << jb);
                        const unsigned long long flat = map.flat(elem_of<WL>(t, j0));
                        const unsigned idx = static_cast<unsigned>(flat) & ((1u << a.n) - 1u);
                        Ctx<T> c = blk_ctx;
                        if (a.flags & F_MULTI)
                            c = make_ctx(a, poly_of(a, flat));
                        unsigned ti = idx >> (P + 1);
                        if (a.flags & F_NEGACYCLIC)
                            ti += 1u << (a.n - 1 - P);
                        const T w = a.roots[c.root_base + ti];
                        if constexpr (INV)
                            dev::gs_butterfly(v[j0], v[j1], w, c.m);
                        else
                            dev::ct_butterfly(v[j0], v[j1], w, c.m);
                    });
                });
            }
        };

        // FST = 4-step phase 1: CONTIG pass over the rows (length n1 = 2^K) of the n2 x n1 input,
        // stored transposed into the n1 x n2 output with the W twiddle multiply fused
        // (reference FourStepForwardCoreT1..4 + the W product of FourStepPartialForwardCore,
        //  src/lib/ntt_4step/ntt_4step.cu:68-743, :1049-1058)
        template <typename T, bool INV, bool CONTIG, int K, bool FST = false>
        __global__ __launch_bounds__(NT) void merge_pass(PassArgs<T> a)
        {
            using G = Geo<CONTIG, K>;
            using S = typename std::make_signed<T>::type;
            __shared__ T lds[LDS_ELEMS];

            if (a.skip_flag != nullptr)
            {
                const unsigned st = *a.skip_flag;
                if (a.skip_value == 0u ? (st != 0u) : (st == a.skip_value))
                    return;
            }
            const int t = threadIdx.x;
            // one tile per block, except for the shadow launches of RNS calls (skip_flag set), which
            // use a capped grid that walks the tiles so that a skipped launch costs ~1 us, not ~7
            const unsigned ntiles = static_cast<unsigned>((a.total + TILE - 1) >> TL);
            for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
            {
            if (tile != blockIdx.x)
                __syncthreads(); // LDS of the previous tile is free
            const TileMap<T, CONTIG, K> map(a, tile);
            // block-uniform context (exact when the tile lies inside one polynomial; with
            // F_MULTI it is refreshed per butterfly)
            const Ctx<T> ctx = make_ctx(a, poly_of(a, map.flat(0)));

            T v[EPT];

            static_for<G::NR>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                using RD = Round<T, INV, CONTIG, K, r>;
                constexpr int WL = RD::WL;

                // ---- gather this round's 16 coefficients -----------------------------
                if constexpr (r == 0)
                {
                    if constexpr (RD::DIRECT_IO)
                    {
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const unsigned long long f = map.flat(elem_of<WL>(t, j));
                            T q = ctx.m.q;
                            if ((a.flags & (F_MULTI | F_SIGNED_IN)) == (F_MULTI | F_SIGNED_IN))
                                q = make_ctx(a, poly_of(a, f)).m.q;
                            v[j] = load_in(a, f, q);
                        }
                    }
                    else
                    {
                        // coalesced global -> LDS, then gather the register window
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const int e = t + NT * j;
                            const unsigned long long f = map.flat(e);
                            T q = ctx.m.q;
                            if ((a.flags & (F_MULTI | F_SIGNED_IN)) == (F_MULTI | F_SIGNED_IN))
                                q = make_ctx(a, poly_of(a, f)).m.q;
                            lds[lds_pad(e)] = load_in(a, f, q);
                        }
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            v[j] = lds[lds_pad(elem_of<WL>(t, j))];
                    }
                }
                else
                {
#pragma unroll
                    for (int j = 0; j < EPT; j++)
                        v[j] = lds[lds_pad(elem_of<WL>(t, j))];
                }

                RD::butterflies(v, a, map, ctx, t);

                // ---- scatter ---------------------------------------------------------
                if constexpr (r == G::NR - 1)
                {
                    // final round of the pass: optional n^-1 scaling / centring, store
                    auto finish = [&](T x, unsigned long long f) -> T {
                        if (a.flags & F_SCALE)
                        {
                            Ctx<T> c = ctx;
                            if (a.flags & F_MULTI)
                                c = make_ctx(a, poly_of(a, f));
                            x = c.m.mul(x, c.ninv);
                            if (a.flags & F_CENTERED)
                            {
                                S sx = (x > (c.m.q >> 1)) ? static_cast<S>(x - c.m.q) : static_cast<S>(x);
                                x = static_cast<T>(sx);
                            }
                        }
                        return x;
                    };
                    if constexpr (FST)
                    {
                        static_assert(CONTIG && K >= 4 && K <= 8, "phase-1 rows are 32..256 long");
                        constexpr int RB = TL - K; // log2 rows per tile
                        __syncthreads();           // all gathers from the e + (e >> 4) layout are done
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            lds[lds_pad_t<K>(elem_of<WL>(t, j))] = v[j];
                        __syncthreads();
                        const unsigned tiles_log = a.poly_shift - TL;
                        const unsigned long long poly = tile >> tiles_log;
                        const unsigned row0 = (tile & ((1u << tiles_log) - 1u)) << RB;
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const int o = t + NT * j;
                            const int jl = o & ((1 << RB) - 1);
                            const int i = o >> RB;
                            const unsigned long long widx =
                                (static_cast<unsigned long long>(i) << a.n2_log) + row0 + jl;
                            const T x = lds[lds_pad_t<K>((jl << K) | i)];
                            a.out[(poly << a.poly_shift) + widx] = ctx.m.mul(x, a.w_table[widx]);
                        }
                    }
                    else if constexpr (RD::DIRECT_IO)
                    {
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const unsigned long long f = map.flat(elem_of<WL>(t, j));
                            if (f < a.total)
                                a.out[phys(a, f)] = finish(v[j], f);
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                            lds[lds_pad(elem_of<WL>(t, j))] = v[j];
                        __syncthreads();
#pragma unroll
                        for (int j = 0; j < EPT; j++)
                        {
                            const int e = t + NT * j;
                            const unsigned long long f = map.flat(e);
                            if (f < a.total)
                                a.out[phys(a, f)] = finish(lds[lds_pad(e)], f);
                        }
                    }
                }
                else
                {
                    // exchange through LDS for the next round (a thread scatters to exactly the
                    // slots it gathered from, so only the scatter -> next gather edge needs a barrier)
#pragma unroll
                    for (int j = 0; j < EPT; j++)
                        lds[lds_pad(elem_of<WL>(t, j))] = v[j];
                    __syncthreads();
                }
            });
            } // tile loop
        }

        // Column-wise (PerCoefficient) transform for matrices too small for a 4096-coefficient
        // strided tile: one thread per butterfly, stages separated by block barriers, one block per
        // group of `cols_per_block` columns (all N rows).  Correctness path, not a fast path.
        template <typename T, bool INV>
        __global__ __launch_bounds__(256) void column_ntt_small(PassArgs<T> a, int n, int log_w, int cols_log)
        {
            using S = typename std::make_signed<T>::type;
            // column c uses modulus c % mod_count, its table at (c % mod_count) << n and its n^-1
            auto ctx_of = [&](unsigned c) -> Ctx<T> {
                Ctx<T> r;
                if (a.mods != nullptr)
                {
                    const int mi = static_cast<int>(c % static_cast<unsigned>(a.mod_count));
                    const Modulus<T> md = a.mods[mi];
                    r.m = dev::ModCtx<T>{md.value, md.bit, md.mu};
                    r.ninv = (a.ninv_arr != nullptr) ? a.ninv_arr[mi] : a.ninv;
                    r.root_base = (a.mod_count > 1) ? (static_cast<unsigned long long>(mi) << n) : 0ull;
                }
                else
                {
                    r.m = dev::ModCtx<T>{a.mod.value, a.mod.bit, a.mod.mu};
                    r.ninv = a.ninv;
                    r.root_base = 0;
                }
                return r;
            };
            const unsigned w = 1u << log_w, nrow = 1u << n, cols = 1u << cols_log;
            const unsigned col0 = blockIdx.x << cols_log;
            const unsigned half = (nrow >> 1) << cols_log; // butterflies per stage in this block
            // copy in -> out (with signed conversion) so the stages can run in place on `out`
            for (unsigned e = threadIdx.x; e < (nrow << cols_log); e += 256)
            {
                const unsigned r = e >> cols_log, c = col0 + (e & (cols - 1));
                const unsigned long long f = static_cast<unsigned long long>(r) * w + c;
                T x;
                if (a.flags & F_SIGNED_IN)
                {
                    const S sv = static_cast<const S*>(a.in)[f];
                    x = (sv < 0) ? static_cast<T>(ctx_of(c).m.q + static_cast<T>(sv)) : static_cast<T>(sv);
                }
                else
                    x = static_cast<const T*>(a.in)[f];
                a.out[f] = x;
            }
            __syncthreads();
            for (int st = 0; st < n; st++)
            {
                const int P = INV ? st : (n - 1 - st); // butterfly distance 2^P rows
                for (unsigned b = threadIdx.x; b < half; b += 256)
                {
                    const unsigned c = col0 + (b & (cols - 1));
                    const unsigned k = b >> cols_log;                     // butterfly index in the column
                    const unsigned lo = k & ((1u << P) - 1), grp = k >> P;
                    const unsigned r0 = (grp << (P + 1)) | lo;
                    unsigned ti = grp;
                    if (a.flags & F_NEGACYCLIC)
                        ti += 1u << (n - 1 - P);
                    const Ctx<T> cx = ctx_of(c);
                    const dev::ModCtx<T>& m = cx.m;
                    const T tw = a.roots[cx.root_base + ti];
                    const unsigned long long f0 = static_cast<unsigned long long>(r0) * w + c;
                    const unsigned long long f1 = f0 + (static_cast<unsigned long long>(1u << P) * w);
                    T U = a.out[f0], V = a.out[f1];
                    if (INV)
                        dev::gs_butterfly(U, V, tw, m);
                    else
                        dev::ct_butterfly(U, V, tw, m);
                    a.out[f0] = U;
                    a.out[f1] = V;
                }
                __syncthreads();
            }
            if (a.flags & F_SCALE)
                for (unsigned e = threadIdx.x; e < (nrow << cols_log); e += 256)
                {
                    const unsigned r = e >> cols_log, c = col0 + (e & (cols - 1));
                    const unsigned long long f = static_cast<unsigned long long>(r) * w + c;
                    const Ctx<T> cx = ctx_of(c);
                    T x = cx.m.mul(a.out[f], cx.ninv);
                    if (a.flags & F_CENTERED)
                        x = (x > (cx.m.q >> 1)) ? static_cast<T>(x - cx.m.q) : x;
                    a.out[f] = x;
                }
        }

    } // namespace kern
} // namespace gpuntt
