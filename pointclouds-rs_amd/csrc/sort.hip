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

}  // namespace pcrs

using namespace pcrs;

int pcrs_sort_pairs_u64(pcrs_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt,
                        uint32_t *vals_alt, size_t n, int key_bits, bool *in_alt) {
    *in_alt = false;
    if (n <= 1 || key_bits <= 0) return PCRS_OK;
    if (n >= 0xFFFFFFFFull) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "sort too long");
    if (key_bits > 64) key_bits = 64;
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
