// Stable LSD radix sort of (u64 key, u32 value) pairs (or of keys alone: null value arrays), 8 bits
// per pass, for the paths that order
// points by a sparse integer key (voxel ids, cluster labels).  One wave owns a contiguous chunk of
// 1024 pairs in both kernels of a pass, so ranks inside a chunk need no block barrier:
//   histogram : per-chunk digit counts (LDS atomics) into a digit-major table [256][chunks]
//   scan      : exclusive scan of the table = first output slot of every (digit, chunk)
//   scatter   : the chunk is re-read 64 pairs at a time; lanes holding the same digit find each
//               other with eight ballots, the rank inside the group is a popcount, and the lowest
//               lane of each group advances the chunk's cursor for that digit.
// Equal keys therefore keep their input order in every pass, which makes the whole sort stable.
#include "knn_common.h"

namespace pcrs {

constexpr int RS_CHUNK = 1024;

__global__ __launch_bounds__(64) void radix_hist_kernel(const uint64_t *keys, uint32_t n, int shift,
                                                        uint32_t nchunks, uint32_t *table) {
    __shared__ uint32_t hist[256];
    const int lane = threadIdx.x;
    for (int t = lane; t < 256; t += 64) hist[t] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t begin = blockIdx.x * RS_CHUNK;
    const uint32_t end = min(n, begin + RS_CHUNK);
    for (uint32_t i = begin + lane; i < end; i += 64)
        atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __syncthreads();
    for (int t = lane; t < 256; t += 64) table[(size_t)t * nchunks + blockIdx.x] = hist[t];
}

__global__ __launch_bounds__(64) void radix_scatter_kernel(const uint64_t *keys,
                                                           const uint32_t *vals, uint32_t n,
                                                           int shift, uint32_t nchunks,
                                                           const uint32_t *table,
                                                           uint64_t *keys_out, uint32_t *vals_out) {
    __shared__ uint32_t cursor[256];
    const int lane = threadIdx.x;
    for (int t = lane; t < 256; t += 64) cursor[t] = table[(size_t)t * nchunks + blockIdx.x];
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t begin = blockIdx.x * RS_CHUNK;
    const uint32_t end = min(n, begin + RS_CHUNK);
    for (uint32_t base = begin; base < end; base += 64) {
        const uint32_t i = base + lane;
        const bool live = i < end;
        uint64_t key = 0;
        uint32_t val = 0;
        if (live) {
            key = keys[i];
            if (vals) val = vals[i];  // keys only: no values to carry
        }
        const uint32_t digit = (uint32_t)(key >> shift) & 255u;
        // lanes with the same digit (dead lanes match nobody)
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long set = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? set : ~set;
        }
        uint32_t first_slot = 0;
        if (live) {
            first_slot = cursor[digit];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __syncthreads();  // every lane has read the cursors before any group leader moves one
        if (live) {
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            const uint32_t at = first_slot + rank;
            keys_out[at] = key;
            if (vals) vals_out[at] = val;
            if (rank == 0) cursor[digit] = first_slot + (uint32_t)__popcll(peers);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __syncthreads();
    }
}

// The whole sort of a short array in ONE launch: a single workgroup of sixteen waves runs every
// pass -- each wave owns a contiguous sixteenth of the array, counts its digits into its own row of
// an LDS table, the table is scanned digit-major (so the passes stay stable), and the wave scatters
// its part with the ballot ranking of radix_scatter_kernel.  Three launches a pass cost a frame of
// a LiDAR (four sorts of 2-3 passes over 2-57 K keys) more than the sorting itself.
constexpr uint32_t RS_SMALL_MAX = 65536;   // what the kernel can take
constexpr uint32_t RS_SMALL_DEFAULT = 16384; // up to where it beats three launches a pass
constexpr int RS_SMALL_WAVES = 16;

__global__ __launch_bounds__(RS_SMALL_WAVES * 64) void radix_sort_small_kernel(
    uint64_t *k0, uint32_t *v0, uint64_t *k1, uint32_t *v1, uint32_t n, int key_bits) {
    __shared__ uint32_t cur[RS_SMALL_WAVES][256];
    __shared__ uint32_t wave_total[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t per = (((n + RS_SMALL_WAVES - 1) / RS_SMALL_WAVES) + 63u) & ~63u;
    const uint32_t begin = min(n, (uint32_t)w * per), end = min(n, begin + per);
    uint64_t *kin = k0, *kout = k1;
    uint32_t *vin = v0, *vout = v1;
    for (int shift = 0; shift < key_bits; shift += 8) {
        for (int t = lane; t < 256; t += 64) cur[w][t] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t i = begin + lane; i < end; i += 64)
            atomicAdd(&cur[w][(uint32_t)(kin[i] >> shift) & 255u], 1u);
        __syncthreads();
        // first slot of every (digit, wave): digits in order, the waves of a digit in order
        uint32_t total = 0, incl = 0;
        if (threadIdx.x < 256) {
            for (int j = 0; j < RS_SMALL_WAVES; ++j) total += cur[j][threadIdx.x];
            incl = total;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (lane == 63) wave_total[w] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t run = incl - total;
            for (int j = 0; j < w; ++j) run += wave_total[j];
            for (int j = 0; j < RS_SMALL_WAVES; ++j) {
                const uint32_t c = cur[j][threadIdx.x];
                cur[j][threadIdx.x] = run;
                run += c;
            }
        }
        __syncthreads();
        for (uint32_t base = begin; base < end; base += 64) {
            const uint32_t i = base + lane;
            const bool live = i < end;
            uint64_t key = 0;
            uint32_t val = 0;
            if (live) {
                key = kin[i];
                if (vin) val = vin[i];
            }
            const uint32_t digit = (uint32_t)(key >> shift) & 255u;
            unsigned long long peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long set = __ballot((digit >> b) & 1u);
                peers &= ((digit >> b) & 1u) ? set : ~set;
            }
            const uint32_t first_slot = live ? cur[w][digit] : 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();  // every lane has read the cursors before a leader moves one
            if (live) {
                const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
                const uint32_t at = first_slot + rank;
                kout[at] = key;
                if (vin) vout[at] = val;
                if (rank == 0) cur[w][digit] = first_slot + (uint32_t)__popcll(peers);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // the next pass reads what the other waves of this workgroup have just written
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint64_t *tk = kin;
        kin = kout;
        kout = tk;
        uint32_t *tv = vin;
        vin = vout;
        vout = tv;
    }
}

}  // namespace pcrs

using namespace pcrs;

int pcrs_sort_pairs_u64(pcrs_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt,
                        uint32_t *vals_alt, size_t n, int key_bits, bool *in_alt) {
    *in_alt = false;
    if (n <= 1 || key_bits <= 0) return PCRS_OK;
    if (n >= 0xFFFFFFFFull) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "sort too long");
    if (key_bits > 64) key_bits = 64;
    static const size_t small_max = [] {  // experiments only: PCRS_SORT_SMALL=<keys>, 0 = never
        const char *e = getenv("PCRS_SORT_SMALL");
        if (!e) return (size_t)RS_SMALL_DEFAULT;
        const long v = atol(e);
        return (size_t)(v < 0 ? 0 : (v > (long)RS_SMALL_MAX ? (long)RS_SMALL_MAX : v));
    }();
    if (n <= small_max) {
        hipLaunchKernelGGL(radix_sort_small_kernel, dim3(1), dim3(RS_SMALL_WAVES * 64), 0, ctx->stream, keys,
                           vals, keys_alt, vals ? vals_alt : nullptr, (uint32_t)n, key_bits);
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        *in_alt = (((key_bits + 7) / 8) & 1) != 0;
        return PCRS_OK;
    }
    const uint32_t nchunks = (uint32_t)((n + RS_CHUNK - 1) / RS_CHUNK);
    const size_t entries = (size_t)256 * nchunks;
    DevBuf table, sums;
    PCRS_HIP_CHECK(ctx, table.alloc(ctx, (entries + 8) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, (entries / 1024 + 4) * sizeof(uint32_t)));
    uint64_t *kin = keys, *kout = keys_alt;
    uint32_t *vin = vals, *vout = vals_alt;
    for (int shift = 0; shift < key_bits; shift += 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nchunks), dim3(64), 0, ctx->stream, kin,
                           (uint32_t)n, shift, nchunks, table.as<uint32_t>());
        PCRS_TRY(pcrs_launch_scan_u32(ctx, table.as<uint32_t>(), entries, sums.as<uint32_t>(),
                                      table.as<uint32_t>() + entries));
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nchunks), dim3(64), 0, ctx->stream, kin, vin,
                           (uint32_t)n, shift, nchunks, table.as<uint32_t>(), kout, vout);
        std::swap(kin, kout);
        std::swap(vin, vout);
        *in_alt = !*in_alt;
    }
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}
