// Streaming kernels around the neighbour search: inlier mask, rigid transform, radius count and
// the ICP consumer reduction.  All are HBM-bound single passes over SoA columns.
#include "knn_common.h"
#include "solve_math.h"

#pragma clang fp contract(off)

namespace pcrs {

// ---------------------------------------------------------------------------------------------
// keep[i] = mean[i] <= thr  (crates/filters/src/statistical_outlier.rs:64-66); NaN/inf means drop.
__global__ __launch_bounds__(256) void mask_kernel(const float *__restrict__ mean, size_t n,
                                                   float thr, uint8_t *__restrict__ keep,
                                                   unsigned long long *count) {
    unsigned local = 0;
    // four means per thread: one 16-byte load, one 4-byte store (hipMalloc'd columns are aligned)
    const size_t n4 = n / 4;
    const float4 *mean4 = reinterpret_cast<const float4 *>(mean);
    uint32_t *keep4 = reinterpret_cast<uint32_t *>(keep);
    const bool vec = ((reinterpret_cast<uintptr_t>(mean) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(keep) & 3u) == 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if (vec) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const float4 m = mean4[i];
            const uint32_t k0 = m.x <= thr, k1 = m.y <= thr, k2 = m.z <= thr, k3 = m.w <= thr;
            keep4[i] = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24);
            local += k0 + k1 + k2 + k3;
        }
        done = n4 * 4;
    }
    for (size_t i = done + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool k = mean[i] <= thr;
        keep[i] = k ? 1 : 0;
        local += k ? 1u : 0u;
    }
    if (!count) return;
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    __shared__ unsigned wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (s) atomicAdd(count, s);
    }
}

// ---------------------------------------------------------------------------------------------
// p' = R p + t, crates/registration/src/icp.rs:39-47: ((r0*x + r1*y) + r2*z) + t, no fma.
struct Rigid {
    float r[9];
    float t[3];
};

__global__ __launch_bounds__(256) void transform_kernel(const float *x, const float *y,
                                                        const float *z, size_t n, Rigid T,
                                                        float *ox, float *oy, float *oz) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const float px = x[i], py = y[i], pz = z[i];
        const float a = T.r[0] * px + T.r[1] * py + T.r[2] * pz + T.t[0];
        const float b = T.r[3] * px + T.r[4] * py + T.r[5] * pz + T.t[1];
        const float c = T.r[6] * px + T.r[7] * py + T.r[8] * pz + T.t[2];
        ox[i] = a;
        oy[i] = b;
        oz[i] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// Count of reference points with d2 <= r2 per query (kdtree.rs:114-129), LDS-tiled scan.
// FILL: the hits are written to indices[offsets[query] ...] instead of being counted; the scan
// runs in index order, so every list comes out ascending (kdtree.rs:131-132).
constexpr int RC_TILE = 1024;
template <bool FILL>
__global__ __launch_bounds__(256) void radius_count_kernel(const float *rx, const float *ry,
                                                           const float *rz, uint32_t m,
                                                           const float *qx, const float *qy,
                                                           const float *qz, uint32_t n, float r2,
                                                           uint32_t *count, const uint32_t *offsets,
                                                           uint32_t *indices) {
    __shared__ __attribute__((aligned(16))) float tx[RC_TILE];
    __shared__ __attribute__((aligned(16))) float ty[RC_TILE];
    __shared__ __attribute__((aligned(16))) float tz[RC_TILE];
    const uint32_t tid = threadIdx.x;
    const uint32_t qi = blockIdx.x * 256 + tid;
    float x = 0, y = 0, z = 0;
    if (qi < n) {
        x = qx[qi];
        y = qy[qi];
        z = qz[qi];
    }
    const bool qvalid = qi < n && finite3(x, y, z);
    uint32_t c = 0;
    uint32_t at = (FILL && qi < n) ? offsets[qi] : 0u;
    for (uint32_t base = 0; base < m; base += RC_TILE) {
        __syncthreads();
        for (int t = tid; t < RC_TILE; t += 256) {
            const uint32_t g = base + t;
            float px = __builtin_nanf(""), py = 0, pz = 0;
            if (g < m) {
                px = rx[g];
                py = ry[g];
                pz = rz[g];
                if (!finite3(px, py, pz)) px = __builtin_nanf("");
            }
            tx[t] = px;
            ty[t] = py;
            tz[t] = pz;
        }
        __syncthreads();
        const uint32_t lim = min((uint32_t)RC_TILE, m - base);
        for (uint32_t j = 0; j < lim; ++j) {
            const float d2 = sqdist(x, y, z, tx[j], ty[j], tz[j]);
            if (FILL) {
                if (qvalid && d2 <= r2) indices[at++] = base + j;
            } else {
                c += (d2 <= r2) ? 1u : 0u;
            }
        }
    }
    if (!FILL && qi < n) count[qi] = qvalid ? c : 0u;
}

// ---------------------------------------------------------------------------------------------
// ICP consumer reduction.  Stage 1: every block folds its grid-stride share of the source into
// PCRS_ICP_NSUMS doubles (fixed lane -> wave -> block order); stage 2: one block folds the block
// partials in block order.  No floating-point atomics, so the result is run-to-run identical.
constexpr int ICP_THREADS = 256;

__device__ __forceinline__ void icp_tail_step(pcrs_math::Work &w, int mode, const double *sums,
                                              float n_source, float tolerance, bool metrics_only,
                                              const float *exact_sum, IcpLoopState *st);

template <int MODE>
__global__ __launch_bounds__(ICP_THREADS) void icp_partial_kernel(
    const float *cx, const float *cy, const float *cz, uint32_t n, const float *tx,
    const float *ty, const float *tz, const float *tnx, const float *tny, const float *tnz,
    const uint32_t *nn_idx, const float *nn_d2, float max_d, uint32_t *tgt_idx_out,
    float *dist_out, double *partials, float *d2seq, const uint32_t *perm,
    const IcpLoopState *st) {
    constexpr int NS = (MODE == PCRS_ICP_POINT_TO_PLANE) ? 29 : 17;
    if (st && st->done) return;  // queued behind the end of the loop
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;

    for (uint32_t i = blockIdx.x * ICP_THREADS + threadIdx.x; i < n;
         i += gridDim.x * ICP_THREADS) {
        const uint32_t ti = nn_idx[i];
        bool valid = ti != PCRS_NO_INDEX;
        float dist = __builtin_nanf("");
        if (valid) {
            dist = sqrtf(nn_d2[i]);  // kdtree.rs:76
            valid = dist <= max_d;   // correspondence.rs:28
        }
        if (tgt_idx_out) tgt_idx_out[i] = valid ? ti : PCRS_NO_INDEX;
        if (dist_out) dist_out[i] = valid ? dist : __builtin_nanf("");
        const float dd = dist * dist;  // icp.rs:279, product rounded to f32
        if (d2seq) d2seq[perm ? perm[i] : i] = valid ? dd : 0.0f;
        if (!valid) continue;
        acc[0] += 1.0;
        acc[1] += (double)dd;
        const double sx = cx[i], sy = cy[i], sz = cz[i];
        const double px = tx[ti], py = ty[ti], pz = tz[ti];
        if (MODE == PCRS_ICP_POINT_TO_PLANE) {
            // icp_plane.rs:152-179
            const double nx = tnx[ti], ny = tny[ti], nz = tnz[ti];
            const double a[6] = {sy * nz - sz * ny, sz * nx - sx * nz, sx * ny - sy * nx,
                                 nx,                ny,                nz};
            const double b = (px - sx) * nx + (py - sy) * ny + (pz - sz) * nz;
            int s = 2;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) acc[s++] += a[r] * a[c];
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[23 + r] += a[r] * b;
        } else {
            const double s3[3] = {sx, sy, sz}, t3[3] = {px, py, pz};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                acc[2 + r] += s3[r];
                acc[5 + r] += t3[r];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[8 + r * 3 + c] += s3[r] * t3[c];
            }
        }
    }

    // Lane totals -> block totals through LDS: the lanes of a wave lay their NS sums down as one
    // row each, then two lanes per sum add 32 rows apiece, top to bottom (174 f64 shuffles per
    // wave cost 6 us of this kernel's 14).  The order is fixed: even lanes, odd lanes, their sum,
    // then the waves in order.
    constexpr int ROW = NS | 1;  // an odd count of doubles per row spreads the rows over the banks
    __shared__ double rows[ICP_THREADS][ROW];
    __shared__ double wsum[ICP_THREADS / 64][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) rows[threadIdx.x][i] = acc[i];
    __syncthreads();
    {
        const int col = lane >> 1, half = lane & 1;
        double v = 0.0;
        if (col < NS) {
            const double(*r)[ROW] = rows + wave * 64 + half;
#pragma unroll 8
            for (int j = 0; j < 32; ++j) v += r[2 * j][col];
        }
        v += __shfl_down(v, 1, 64);  // even rows + odd rows
        if (col < NS && half == 0) wsum[wave][col] = v;
    }
    __syncthreads();
    if (threadIdx.x < PCRS_ICP_NSUMS) {
        double v = 0.0;
        if ((int)threadIdx.x < NS) {
            for (int w = 0; w < ICP_THREADS / 64; ++w) v += wsum[w][threadIdx.x];
        }
        partials[(size_t)blockIdx.x * PCRS_ICP_NSUMS + threadIdx.x] = v;
    }
}

// What the host does between two iterations of the ICP loop (icp.rs:147-188), for one thread.
// exact_sum (nullable): the left-to-right f32 sum of dist*dist in source order (icp.rs:277-281);
// without it the rmse comes from the f64 tree sum, within ~1e-5 relative of the former.
__device__ __forceinline__ void icp_tail_step(pcrs_math::Work &w, int mode, const double *sums,
                                              float n_source, float tolerance, bool metrics_only,
                                              const float *exact_sum, IcpLoopState *st) {
    if (st->done) return;
    if (!metrics_only) st->num_iterations += 1u;
    if (!(sums[0] > 0.0)) {  // icp.rs:163-165
        st->done = 1u;
        return;
    }
    const float ncorr = (float)(size_t)sums[0];
    const float rmse = exact_sum ? sqrtf(*exact_sum / ncorr)            // icp.rs:281
                                 : (float)sqrt(sums[1] / sums[0]);
    const float fitness = ncorr / n_source;  // icp.rs:170
    st->last_rmse = rmse;
    st->last_fitness = fitness;
    st->last_ncorr = ncorr;
    st->last_live = metrics_only ? 1u : st->num_iterations;
    if (metrics_only) {  // icp.rs:190-197
        st->done = 1u;
        return;
    }
    if (fabsf(st->prev_rmse - rmse) < tolerance) {  // icp.rs:173-176
        st->converged = 1u;
        st->done = 1u;
        return;
    }
    st->prev_rmse = rmse;
    float r[9], t[3];
    if (mode == PCRS_ICP_POINT_TO_PLANE)
        pcrs_math::solve_plane(w, sums, r, t);
    else
        pcrs_math::solve_point(w, sums, r, t);
    for (int i = 0; i < 9; ++i) st->inc_r[i] = r[i];
    for (int i = 0; i < 3; ++i) st->inc_t[i] = t[i];
    float cr[9], ct[3], nr[9], nt[3];
    for (int i = 0; i < 9; ++i) cr[i] = st->cum_r[i];
    for (int i = 0; i < 3; ++i) ct[i] = st->cum_t[i];
    pcrs_math::compose(cr, ct, r, t, nr, nt);  // icp.rs:183
    for (int i = 0; i < 9; ++i) st->cum_r[i] = nr[i];
    for (int i = 0; i < 3; ++i) st->cum_t[i] = nt[i];
}


// Stage 2: 32 groups of threads fold the block partials (group g takes blocks g, g+32, ... in
// order), then 32 threads fold the groups in order: a fixed summation tree, a few microseconds.
// With tail_here, thread 0 then runs the tail of the iteration on the fresh sums.  (Folding this
// stage into the last block of stage 1 was tried: the device-scope fence every block then needs
// before it draws its ticket writes back the L2 of its XCD and cost 26 us per launch, three times
// what this launch costs.)
__global__ __launch_bounds__(1024) void icp_final_kernel(const double *partials, uint32_t nblocks,
                                                         double *sums, int mode, float n_source,
                                                         float tolerance, int metrics_only,
                                                         int tail_here, IcpLoopState *st) {
    if (st && st->done) return;
    __shared__ double grp[32][PCRS_ICP_NSUMS + 1];
    __shared__ double total[PCRS_ICP_NSUMS];
    __shared__ pcrs_math::Work work;  // the solver's arrays: LDS, so that the kernel needs no scratch
    const int i = threadIdx.x & 31, gidx = threadIdx.x >> 5;
    double v = 0.0;
    for (uint32_t b = gidx; b < nblocks; b += 32) v += partials[(size_t)b * PCRS_ICP_NSUMS + i];
    grp[gidx][i] = v;
    __syncthreads();
    if (threadIdx.x < PCRS_ICP_NSUMS) {
        double t = 0.0;
        for (int q = 0; q < 32; ++q) t += grp[q][threadIdx.x];
        total[threadIdx.x] = t;
        sums[threadIdx.x] = t;
    }
    __syncthreads();
    if (st && tail_here && threadIdx.x == 0)
        icp_tail_step(work, mode, total, n_source, tolerance, metrics_only != 0, nullptr, st);
}

}  // namespace pcrs

using namespace pcrs;

static unsigned stream_grid(pcrs_ctx *ctx, size_t n, int threads) {
    size_t blocks = (n + threads - 1) / threads;
    const size_t cap = (size_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

int pcrs_launch_mask(pcrs_ctx *ctx, const float *mean, size_t n, float thr, uint8_t *keep,
                     unsigned long long *d_count) {
    if (d_count) PCRS_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(*d_count), ctx->stream));
    if (n == 0) return PCRS_OK;
    // one block per CU at most: every block ends with an atomic on the same counter
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > (size_t)ctx->num_cus) blocks = (size_t)ctx->num_cus;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, mean, n, thr,
                       keep, d_count);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

__global__ __launch_bounds__(256) void scatter_perm_kernel(const float *in, const uint32_t *perm,
                                                           size_t n, float *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[perm[i]] = in[i];
}

// (point index, value) pairs bucketed by the rank that owns the index under the partition of
// [0, n_total) into `world` contiguous ranges (the first `rem` ranges one longer): the send side
// of an all-to-all that takes every result to the owner of its index range.
constexpr int ROUTE_MAX_WORLD = 64;
struct RouteSplit {
    uint32_t base, rem, world;  // range r = [r * base + min(r, rem), ...), length base + (r < rem)
};
__device__ __forceinline__ uint32_t route_dest(const RouteSplit &sp, uint32_t idx) {
    const uint32_t big = sp.rem * (sp.base + 1u);  // indices below this lie in the longer ranges
    return idx < big ? idx / (sp.base + 1u) : sp.rem + (idx - big) / (sp.base ? sp.base : 1u);
}
__global__ __launch_bounds__(256) void route_count_kernel(const uint32_t *point, size_t n,
                                                          RouteSplit sp, uint32_t *counts) {
    __shared__ uint32_t h[ROUTE_MAX_WORLD];
    if (threadIdx.x < ROUTE_MAX_WORLD) h[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        atomicAdd(&h[route_dest(sp, point[i])], 1u);
    __syncthreads();
    if (threadIdx.x < sp.world && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
// cursors[d] starts at the first send slot of destination d; a block reserves room for its
// pairs of every destination with one atomic each
__global__ __launch_bounds__(256) void route_scatter_kernel(const uint32_t *point, const float *value,
                                                            size_t n, RouteSplit sp, uint32_t *cursors,
                                                            uint32_t *send_point, float *send_value) {
    __shared__ uint32_t h[ROUTE_MAX_WORLD], at[ROUTE_MAX_WORLD];
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    if (threadIdx.x < ROUTE_MAX_WORLD) h[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = b + threadIdx.x; i < e; i += 256) atomicAdd(&h[route_dest(sp, point[i])], 1u);
    __syncthreads();
    if (threadIdx.x < sp.world) at[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], h[threadIdx.x]) : 0u;
    __syncthreads();
    for (size_t i = b + threadIdx.x; i < e; i += 256) {
        const uint32_t p = point[i];
        const uint32_t slot = atomicAdd(&at[route_dest(sp, p)], 1u);
        send_point[slot] = p;
        send_value[slot] = value[i];
    }
}
// out[point[i] - first] = value[i]
__global__ __launch_bounds__(256) void index_run_kernel(uint32_t *out, size_t n, uint32_t first) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = first + (uint32_t)i;
}

__global__ __launch_bounds__(256) void scatter_offset_kernel(const float *in, const uint32_t *point,
                                                             size_t n, uint32_t first, float *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[point[i] - first] = in[i];
}

// PointCloud::select by a mask (crates/core/src/cloud.rs:103-140 for x, y, z): the points whose
// mask byte is non-zero (or zero, with invert), in their order.  Count per block of 1024, scan,
// ordered write: no atomics, the order is the input's.
constexpr int SEL_PER_THREAD = 4;
constexpr int SEL_BLOCK = 256 * SEL_PER_THREAD;
__device__ __forceinline__ uint32_t select_flags(const uint8_t *mask, size_t n, size_t first, int invert) {
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < SEL_PER_THREAD; ++i)
        if (first + i < n && ((mask[first + i] != 0) != (invert != 0))) f |= 1u << i;
    return f;
}
__global__ __launch_bounds__(256) void select_count_kernel(const uint8_t *mask, size_t n, int invert,
                                                           uint32_t *block_count) {
    __shared__ uint32_t wsum[4];
    const size_t first = (size_t)blockIdx.x * SEL_BLOCK + threadIdx.x * SEL_PER_THREAD;
    uint32_t c = __popc(select_flags(mask, n, first, invert));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void select_write_kernel(const float *x, const float *y, const float *z,
                                                           const uint8_t *mask, size_t n, int invert,
                                                           const uint32_t *block_offset, float *ox,
                                                           float *oy, float *oz, uint32_t *oidx) {
    __shared__ uint32_t wsum[4];
    const size_t first = (size_t)blockIdx.x * SEL_BLOCK + threadIdx.x * SEL_PER_THREAD;
    const uint32_t f = select_flags(mask, n, first, invert);
    const uint32_t c = __popc(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t at = block_offset[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) at += wsum[w];
#pragma unroll
    for (int i = 0; i < SEL_PER_THREAD; ++i)
        if (f & (1u << i)) {
            ox[at] = x[first + i];
            oy[at] = y[first + i];
            oz[at] = z[first + i];
            if (oidx) oidx[at] = (uint32_t)(first + i);
            ++at;
        }
}

// A few words from device memory straight into pinned host memory: a kernel of one wave costs
// less than a copy command, and the host then only waits for the stream.
__global__ void publish_words_kernel(const uint32_t *src, uint32_t *host_dst, uint32_t nwords) {
    for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) host_dst[i] = src[i];
}

int pcrs_launch_publish(pcrs_ctx *ctx, const void *d_src, void *pinned_dst, size_t bytes) {
    if (bytes == 0) return PCRS_OK;
    if (bytes % 4 || bytes > 65536)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "publish: a small multiple of 4 bytes");
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       reinterpret_cast<const uint32_t *>(d_src),
                       reinterpret_cast<uint32_t *>(pinned_dst), (uint32_t)(bytes / 4));
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_scatter_perm(pcrs_ctx *ctx, const float *in, const uint32_t *perm, size_t n,
                             float *out) {
    if (n == 0) return PCRS_OK;
    hipLaunchKernelGGL(scatter_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, in, perm, n, out);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// d_total (nullable, device): receives the number of points selected
int pcrs_launch_select(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                       const uint8_t *mask, int invert, float *ox, float *oy, float *oz,
                       uint32_t *oidx, uint32_t *d_total) {
    if (n >= 0xFFFFFFFFull) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    DevBuf counts, sums, total;
    const uint32_t blocks = (uint32_t)((n + SEL_BLOCK - 1) / SEL_BLOCK);
    PCRS_HIP_CHECK(ctx, counts.alloc(ctx, ((size_t)blocks + 8) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, ((size_t)blocks / 1024 + 4) * sizeof(uint32_t)));
    if (!d_total) {
        PCRS_HIP_CHECK(ctx, total.alloc(ctx, sizeof(uint32_t)));
        d_total = total.as<uint32_t>();
    }
    if (n == 0) {
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(d_total, 0, sizeof(uint32_t), ctx->stream));
        return PCRS_OK;
    }
    hipLaunchKernelGGL(select_count_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mask, n, invert,
                       counts.as<uint32_t>());
    PCRS_TRY(pcrs_launch_scan_u32(ctx, counts.as<uint32_t>(), blocks, sums.as<uint32_t>(), d_total));
    hipLaunchKernelGGL(select_write_kernel, dim3(blocks), dim3(256), 0, ctx->stream, x, y, z, mask, n,
                       invert, counts.as<uint32_t>(), ox, oy, oz, oidx);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_scatter_offset(pcrs_ctx *ctx, const float *in, const uint32_t *point, size_t n,
                               uint32_t first, float *out) {
    if (n == 0) return PCRS_OK;
    hipLaunchKernelGGL(scatter_offset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, in, point, n, first, out);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_index_run(pcrs_ctx *ctx, uint32_t *out, size_t n, uint32_t first) {
    if (n == 0) return PCRS_OK;
    hipLaunchKernelGGL(index_run_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       out, n, first);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// Buckets n (point, value) pairs by the owner of the point's index range; counts[world] (host)
// receives the bucket sizes, the send arrays hold the buckets in rank order.  Waits for the
// stream once (the sizes are the split of the all-to-all that follows).
int pcrs_launch_route_pairs(pcrs_ctx *ctx, const uint32_t *point, const float *value, size_t n,
                            size_t n_total, uint32_t world, uint32_t *send_point, float *send_value,
                            size_t *counts) {
    if (world == 0 || world > ROUTE_MAX_WORLD)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "1 to %d ranks", ROUTE_MAX_WORLD);
    if (n_total >= 0xFFFFFFFFull || n > n_total)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    for (uint32_t r = 0; r < world; ++r) counts[r] = 0;
    if (n == 0) return PCRS_OK;
    RouteSplit sp;
    sp.world = world;
    sp.base = (uint32_t)(n_total / world);
    sp.rem = (uint32_t)(n_total % world);
    DevBuf d;
    PCRS_HIP_CHECK(ctx, d.alloc(ctx, 2 * ROUTE_MAX_WORLD * sizeof(uint32_t)));
    uint32_t *d_counts = d.as<uint32_t>(), *d_cursors = d_counts + ROUTE_MAX_WORLD;
    PCRS_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, ROUTE_MAX_WORLD * sizeof(uint32_t), ctx->stream));
    const unsigned blocks = stream_grid(ctx, n, 256);
    hipLaunchKernelGGL(route_count_kernel, dim3(blocks), dim3(256), 0, ctx->stream, point, n, sp, d_counts);
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, ROUTE_MAX_WORLD * sizeof(uint32_t), &pin));
    PCRS_TRY(pcrs_launch_publish(ctx, d_counts, pin, ROUTE_MAX_WORLD * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t *h = reinterpret_cast<const uint32_t *>(pin);
    uint32_t cursors[ROUTE_MAX_WORLD] = {0};
    uint32_t run = 0;
    for (uint32_t r = 0; r < world; ++r) {
        counts[r] = h[r];
        cursors[r] = run;
        run += h[r];
    }
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(d_cursors, cursors, ROUTE_MAX_WORLD * sizeof(uint32_t),
                                       hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(route_scatter_kernel, dim3(blocks), dim3(256), 0, ctx->stream, point, value, n,
                       sp, d_cursors, send_point, send_value);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_transform(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                          const float rot[9], const float t[3], float *ox, float *oy, float *oz) {
    if (n == 0) return PCRS_OK;
    Rigid T;
    for (int i = 0; i < 9; ++i) T.r[i] = rot[i];
    for (int i = 0; i < 3; ++i) T.t[i] = t[i];
    hipLaunchKernelGGL(transform_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0, ctx->stream,
                       x, y, z, n, T, ox, oy, oz);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_radius_count(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz,
                             size_t m, const float *qx, const float *qy, const float *qz, size_t n,
                             float radius, uint32_t *count) {
    if (n == 0) return PCRS_OK;
    if (m >= 0xFFFFFFFFull || n >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    // kdtree.rs:106-112: empty tree, radius <= 0 or non-finite -> no results
    if (m == 0 || !(radius > 0.0f) || !(radius - radius == 0.0f)) {
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(count, 0, n * sizeof(uint32_t), ctx->stream));
        return PCRS_OK;
    }
    const float r2 = radius * radius;  // kdtree.rs:114
    hipLaunchKernelGGL(radius_count_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, rx, ry, rz, (uint32_t)m, qx, qy, qz, (uint32_t)n, r2, count,
                       nullptr, nullptr);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// Second pass of a radius search by brute force: indices[offsets[i] ...] = the hits of query i,
// ascending.  Only for a usable radius (the count pass has already dealt with the other cases).
int pcrs_launch_radius_fill(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz,
                            size_t m, const float *qx, const float *qy, const float *qz, size_t n,
                            float radius, const uint32_t *offsets, uint32_t *indices) {
    if (n == 0 || m == 0) return PCRS_OK;
    hipLaunchKernelGGL(radius_count_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, rx, ry, rz, (uint32_t)m, qx, qy, qz, (uint32_t)n,
                       radius * radius, nullptr, offsets, indices);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

namespace pcrs {
// sum of n counts as a 64-bit total (a u32 scan cannot tell an overflow)
__global__ __launch_bounds__(256) void sum_u32_kernel(const uint32_t *v, size_t n,
                                                      unsigned long long *total) {
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        local += v[i];
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(total, local);
}
// indices[i] = low idx_bits bits of keys[i]
__global__ __launch_bounds__(256) void key_low_bits_kernel(const uint64_t *keys, size_t n,
                                                           int idx_bits, uint32_t *indices) {
    const uint64_t mask = (idx_bits >= 64) ? ~0ull : ((1ull << idx_bits) - 1ull);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        indices[i] = (uint32_t)(keys[i] & mask);
}
}  // namespace pcrs

int pcrs_launch_sum_u32(pcrs_ctx *ctx, const uint32_t *v, size_t n, unsigned long long *d_total) {
    PCRS_HIP_CHECK(ctx, hipMemsetAsync(d_total, 0, sizeof(*d_total), ctx->stream));
    if (n == 0) return PCRS_OK;
    hipLaunchKernelGGL(pcrs::sum_u32_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0,
                       ctx->stream, v, n, d_total);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_key_low_bits(pcrs_ctx *ctx, const uint64_t *keys, size_t n, int idx_bits,
                             uint32_t *indices) {
    if (n == 0) return PCRS_OK;
    hipLaunchKernelGGL(pcrs::key_low_bits_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0,
                       ctx->stream, keys, n, idx_bits, indices);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

namespace pcrs {
__global__ __launch_bounds__(256) void count_mask_kernel(const uint32_t *count, size_t n,
                                                         unsigned long long min_neighbors,
                                                         uint8_t *keep, unsigned long long *total) {
    unsigned local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const bool k = (unsigned long long)count[i] >= min_neighbors;
        keep[i] = k ? 1 : 0;
        local += k ? 1u : 0u;
    }
    block_count_add(local, total);
}
}  // namespace pcrs

int pcrs_launch_count_mask(pcrs_ctx *ctx, const uint32_t *count, size_t n, size_t min_neighbors,
                           uint8_t *keep, unsigned long long *d_count) {
    if (d_count) PCRS_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(*d_count), ctx->stream));
    if (n == 0) return PCRS_OK;
    size_t blocks = (n + 1023) / 1024;
    if (blocks > (size_t)ctx->num_cus) blocks = (size_t)ctx->num_cus;
    hipLaunchKernelGGL(pcrs::count_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       count, n, (unsigned long long)min_neighbors, keep, d_count);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// ---------------------------------------------------------------------------------------------
// Device-resident ICP loop: one thread does what the host does between two iterations.
namespace pcrs {

__global__ void icp_tail_kernel(int mode, const double *sums, float n_source, float tolerance,
                                int metrics_only, const float *exact_sum, IcpLoopState *st) {
    __shared__ pcrs_math::Work work;
    icp_tail_step(work, mode, sums, n_source, tolerance, metrics_only != 0, exact_sum, st);
}

// The same for a source split over ranks: the sums of the shards, gathered from all ranks, are
// added in rank order first, so that every rank continues from bit-identical totals.
__global__ __launch_bounds__(64) void icp_shard_tail_kernel(int mode, const double *sums_all,
                                                            uint32_t world, double *sums,
                                                            float n_source, float tolerance,
                                                            int metrics_only, const float *exact_sum,
                                                            IcpLoopState *st) {
    __shared__ pcrs_math::Work work;
    __shared__ double total[PCRS_ICP_NSUMS];
    if (st->done) return;
    if (threadIdx.x < PCRS_ICP_NSUMS) {
        double v = 0.0;
        for (uint32_t r = 0; r < world; ++r) v += sums_all[(size_t)r * PCRS_ICP_NSUMS + threadIdx.x];
        total[threadIdx.x] = v;
        sums[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        icp_tail_step(work, mode, total, n_source, tolerance, metrics_only != 0, exact_sum, st);
}

// After a loop whose rmse values came from the f64 sums (tolerance <= 0: the convergence test
// can never fire, so only the reported value matters): the reported rmse from the exact sum.
__global__ void icp_exact_rmse_kernel(IcpLoopState *st, const float *exact_sum) {
    if (st->last_live) st->last_rmse = sqrtf(*exact_sum / st->last_ncorr);
}

__global__ __launch_bounds__(256) void transform_state_kernel(float *x, float *y, float *z, size_t n,
                                                              const IcpLoopState *st) {
    if (st->done) return;
    float r[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = st->inc_r[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = st->inc_t[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const float px = x[i], py = y[i], pz = z[i];
        x[i] = ((r[0] * px + r[1] * py) + r[2] * pz) + t[0];
        y[i] = ((r[3] * px + r[4] * py) + r[5] * pz) + t[1];
        z[i] = ((r[6] * px + r[7] * py) + r[8] * pz) + t[2];
    }
}

}  // namespace pcrs

int pcrs_launch_icp_tail(pcrs_ctx *ctx, int mode, const double *d_sums, const IcpTail &tail) {
    const float *exact_sum =
        tail.exact ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(tail.seq_scalars) +
                                                     pcrs_seq_sum_result_offset())
                   : nullptr;
    hipLaunchKernelGGL(pcrs::icp_tail_kernel, dim3(1), dim3(1), 0, ctx->stream, mode, d_sums,
                       (float)tail.n_source, tail.tolerance, tail.metrics_only ? 1 : 0, exact_sum,
                       tail.state);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_icp_shard_tail(pcrs_ctx *ctx, int mode, const double *d_sums_all, uint32_t world,
                               double *d_sums, const IcpTail &tail) {
    const float *exact_sum =
        tail.exact ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(tail.seq_scalars) +
                                                     pcrs_seq_sum_result_offset())
                   : nullptr;
    hipLaunchKernelGGL(pcrs::icp_shard_tail_kernel, dim3(1), dim3(64), 0, ctx->stream, mode,
                       d_sums_all, world, d_sums, (float)tail.n_source, tail.tolerance,
                       tail.metrics_only ? 1 : 0, exact_sum, tail.state);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_icp_exact_rmse(pcrs_ctx *ctx, IcpLoopState *d_state, const void *d_seq_scalars) {
    hipLaunchKernelGGL(pcrs::icp_exact_rmse_kernel, dim3(1), dim3(1), 0, ctx->stream, d_state,
                       reinterpret_cast<const float *>(reinterpret_cast<const char *>(d_seq_scalars) +
                                                       pcrs_seq_sum_result_offset()));
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_transform_state(pcrs_ctx *ctx, float *x, float *y, float *z, size_t n,
                                const IcpLoopState *d_state) {
    if (n == 0) return PCRS_OK;
    hipLaunchKernelGGL(pcrs::transform_state_kernel, dim3(stream_grid(ctx, n, 256)), dim3(256), 0,
                       ctx->stream, x, y, z, n, d_state);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_icp_reduce(pcrs_ctx *ctx, int mode, const float *cx, const float *cy,
                           const float *cz, size_t n, const float *tx, const float *ty,
                           const float *tz, const float *tnx, const float *tny, const float *tnz,
                           const uint32_t *nn_idx, const float *nn_d2, float max_d,
                           uint32_t *tgt_idx_out, float *dist_out, double *d_sums,
                           const IcpTail &tail) {
    IcpLoopState *st = tail.state;
    if (!tail.sums_only && tail.exact && (!tail.d2seq || !tail.seq_scalars || !tail.seq_work))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "exact rmse needs its work buffers");
    if (n == 0) {
        PCRS_HIP_CHECK(ctx,
                       hipMemsetAsync(d_sums, 0, PCRS_ICP_NSUMS * sizeof(double), ctx->stream));
        if (st && !tail.sums_only) {
            IcpTail t2 = tail;
            t2.exact = false;  // no correspondences: the tail never looks at a sum
            return pcrs_launch_icp_tail(ctx, mode, d_sums, t2);
        }
        return PCRS_OK;
    }
    size_t blocks = (n + ICP_THREADS - 1) / ICP_THREADS;
    if (blocks > 256) blocks = 256;  // one per CU: fewer partials for the single block of stage 2
    void *scratch = nullptr;
    int rc = pcrs_ctx_scratch(ctx, blocks * PCRS_ICP_NSUMS * sizeof(double), &scratch);
    if (rc != PCRS_OK) return rc;
    double *partials = reinterpret_cast<double *>(scratch);
    if (mode == PCRS_ICP_POINT_TO_PLANE)
        hipLaunchKernelGGL(icp_partial_kernel<PCRS_ICP_POINT_TO_PLANE>, dim3((unsigned)blocks),
                           dim3(ICP_THREADS), 0, ctx->stream, cx, cy, cz, (uint32_t)n, tx, ty, tz,
                           tnx, tny, tnz, nn_idx, nn_d2, max_d, tgt_idx_out, dist_out, partials,
                           tail.d2seq, tail.perm, st);
    else
        hipLaunchKernelGGL(icp_partial_kernel<PCRS_ICP_POINT_TO_POINT>, dim3((unsigned)blocks),
                           dim3(ICP_THREADS), 0, ctx->stream, cx, cy, cz, (uint32_t)n, tx, ty, tz,
                           tnx, tny, tnz, nn_idx, nn_d2, max_d, tgt_idx_out, dist_out, partials,
                           tail.d2seq, tail.perm, st);
    hipLaunchKernelGGL(icp_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, partials,
                       (uint32_t)blocks, d_sums, mode, (float)tail.n_source, tail.tolerance,
                       tail.metrics_only ? 1 : 0, (st && !tail.exact && !tail.sums_only) ? 1 : 0, st);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    if (st && tail.exact && !tail.sums_only) {
        // the reference's rmse: dist*dist summed left to right in f32, in source order
        PCRS_TRY(pcrs_launch_seq_sum(ctx, tail.d2seq, n, tail.seq_scalars, tail.seq_work,
                                     tail.seq_work_bytes, &st->done));
        PCRS_TRY(pcrs_launch_icp_tail(ctx, mode, d_sums, tail));
    }
    return PCRS_OK;
}
