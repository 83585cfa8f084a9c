// prep.hip -- twiddle preparation for the fast 64-bit path + its workspace.
//
// The caller's tables hold plain residues in the reference's bit-reversed order
// (reference test_merge_ntt.cu:115-122).  Each call re-derives from them, on the call's
// stream, the table the fast kernels read: Shoup pairs {w, floor(w * 2^64 / q)} laid out by
// stage (stage with m = 2^S groups occupies slots [2^S, 2^(S+1)) for both reduction
// polynomials), with the distance-1/2/4 stages permuted to [tile][k][thread] so the last
// contiguous round loads them fully coalesced.  Cost: N-1 64-step divisions per modulus,
// ~1 us of chip time at N = 2^16 -- nothing is cached between calls, so a caller that
// rewrites its table in place is always honoured.
#include <cstdlib>
#include <map>
#include <mutex>

#include "lazy_launch.hpp"

namespace gpuntt
{
    namespace kern
    {
        __device__ __forceinline__ uint64_t shoup_quotient(uint64_t w, uint64_t q)
        {
            // floor(w * 2^64 / q) for w < q < 2^62 by restoring division
            uint64_t rem = w, quo = 0;
#pragma unroll 8
            for (int i = 0; i < 64; i++)
            {
                rem <<= 1;
                const bool ge = rem >= q;
                rem = ge ? rem - q : rem;
                quo = (quo << 1) | (ge ? 1u : 0u);
            }
            return quo;
        }

        __global__ __launch_bounds__(256) void prep_twiddles(const uint64_t* __restrict__ roots,
                                                             lazy::Tw64* __restrict__ ws,
                                                             const Modulus<uint64_t>* __restrict__ mods,
                                                             uint64_t q_single, int mod_count, int n,
                                                             int negacyclic, int perm_low,
                                                             const uint64_t* __restrict__ ninv_arr,
                                                             lazy::Tw64* __restrict__ ws_ninv)
        {
            const unsigned long long gid = blockIdx.x * 256ull + threadIdx.x;
            const unsigned long long per_mod = 1ull << n;
            if (gid >= per_mod * mod_count)
                return;
            const int mi = static_cast<int>(gid >> n);
            const unsigned slot = static_cast<unsigned>(gid & (per_mod - 1));
            const uint64_t q = (mods != nullptr) ? mods[mi].value : q_single;
            if (slot == 0)
            {
                if (ninv_arr != nullptr && ws_ninv != nullptr)
                {
                    const uint64_t v = ninv_arr[mi];
                    ws_ninv[mi] = lazy::Tw64{v, shoup_quotient(v, q)};
                }
                ws[gid] = lazy::Tw64{0, 0};
                return;
            }
            const int S = 31 - __clz(slot);       // stage: m = 2^S groups
            unsigned i = slot - (1u << S);        // permuted group index
            const int P = n - 1 - S;              // butterfly distance 2^P
            if (perm_low && P <= 2)
            {
                const unsigned rp = 16u >> (P + 1);            // twiddles per thread
                const unsigned tile = i / (rp * 256u), rem = i % (rp * 256u);
                const unsigned kk = rem / 256u, t = rem % 256u;
                i = tile * (rp * 256u) + t * rp + kk;
            }
            const unsigned src = negacyclic ? ((1u << S) + i) : i;
            const uint64_t w = roots[(static_cast<unsigned long long>(mi) << n) + src];
            ws[gid] = lazy::Tw64{w, shoup_quotient(w, q)};
        }
    } // namespace kern

    namespace host
    {
        namespace
        {
            struct Slot
            {
                void* ptr = nullptr;
                size_t bytes = 0;
            };
            std::mutex g_ws_mutex;
            std::map<std::pair<int, hipStream_t>, Slot> g_ws;
        } // namespace

        int lazy_contig_k(int n)
        {
            static const int forced = [] {
                const char* e = std::getenv("GPUNTT_CONTIG_K");
                return e ? std::atoi(e) : 0;
            }();
            if (forced >= 8 && forced <= 12)
                return forced;
            (void) n;
            return kern::TL;
        }

        void* lazy_workspace(hipStream_t stream, size_t bytes)
        {
            int dev = 0;
            GPUNTT_HIP_CHECK(hipGetDevice(&dev));
            std::lock_guard<std::mutex> lock(g_ws_mutex);
            Slot& s = g_ws[std::make_pair(dev, stream)];
            if (s.bytes < bytes)
            {
                if (s.ptr != nullptr)
                {
                    GPUNTT_HIP_CHECK(hipStreamSynchronize(stream)); // earlier calls may still read it
                    GPUNTT_HIP_CHECK(hipFree(s.ptr));
                    s.ptr = nullptr;
                    s.bytes = 0;
                }
                size_t want = bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes;
                GPUNTT_HIP_CHECK(hipMalloc(&s.ptr, want));
                s.bytes = want;
            }
            return s.ptr;
        }

        void launch_prep(const uint64_t* roots, lazy::Tw64* ws, const Modulus<uint64_t>* mods, uint64_t q,
                         int mod_count, int n, bool negacyclic, bool perm_low, const uint64_t* ninv_arr,
                         lazy::Tw64* ws_ninv, hipStream_t stream)
        {
            const unsigned long long entries = static_cast<unsigned long long>(mod_count) << n;
            const unsigned grid = static_cast<unsigned>((entries + 255) / 256);
            hipLaunchKernelGGL(kern::prep_twiddles, dim3(grid), dim3(256), 0, stream, roots, ws, mods, q,
                               mod_count, n, negacyclic ? 1 : 0, perm_low ? 1 : 0, ninv_arr, ws_ninv);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
    } // namespace host
} // namespace gpuntt
