// Plane RANSAC after sample generation (crates/segmentation/src/ransac_plane.rs:56-138).  The
// reference draws its index triples from rand's StdRng (:75-79); the caller passes those triples,
// and everything behind them runs here: one plane per triple (:168-191), the inlier count of
// every plane over the whole cloud (:131-137), the reference's choice among them (:83-122) and
// the inlier mask of the winner (:125-128).
#include <cmath>

#include "knn_common.h"

#pragma clang fp contract(off)

namespace pcrs {

struct PlaneFit {
    float nx, ny, nz, d;
    uint32_t valid;  // 0: the three points are collinear or coincident (:181-184)
};

// fit_plane_from_three_points, operation by operation (:168-191)
__global__ __launch_bounds__(256) void plane_fit_kernel(const float *x, const float *y,
                                                        const float *z, uint32_t n,
                                                        const uint32_t *samples, uint32_t ns,
                                                        PlaneFit *fits) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= ns) return;
    PlaneFit f = {0.0f, 0.0f, 1.0f, 0.0f, 0u};
    const uint32_t i0 = samples[3 * s], i1 = samples[3 * s + 1], i2 = samples[3 * s + 2];
    if (i0 < n && i1 < n && i2 < n) {
        const float p0[3] = {x[i0], y[i0], z[i0]};
        const float v1[3] = {x[i1] - p0[0], y[i1] - p0[1], z[i1] - p0[2]};
        const float v2[3] = {x[i2] - p0[0], y[i2] - p0[1], z[i2] - p0[2]};
        const float nx = v1[1] * v2[2] - v1[2] * v2[1];
        const float ny = v1[2] * v2[0] - v1[0] * v2[2];
        const float nz = v1[0] * v2[1] - v1[1] * v2[0];
        const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
        if (!(len < 1e-10f)) {  // a NaN length passes, as in the reference
            f.nx = nx / len;
            f.ny = ny / len;
            f.nz = nz / len;
            f.d = -((f.nx * p0[0] + f.ny * p0[1]) + f.nz * p0[2]);
            f.valid = 1u;
        }
    }
    fits[s] = f;
}

// Inlier counts of up to 256 planes per block over one slice of the cloud: a lane owns a plane,
// the points of the slice pass through LDS and are read as broadcasts.  (:131-137, distance :17-20)
constexpr int RP_TILE = 1024;
__global__ __launch_bounds__(256) void plane_count_kernel(const float *x, const float *y,
                                                          const float *z, uint32_t n,
                                                          uint32_t slice, const PlaneFit *fits,
                                                          uint32_t ns, float thr,
                                                          unsigned long long *counts) {
    __shared__ float4 tile[RP_TILE];
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    PlaneFit f = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
    if (s < ns) f = fits[s];
    const uint32_t begin = blockIdx.y * slice;
    const uint32_t end = min(n, begin + slice);
    uint32_t c = 0;
    for (uint32_t base = begin; base < end; base += RP_TILE) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < RP_TILE; t += 256) {
            const uint32_t i = base + t;
            // past the slice: a point no plane accepts
            tile[t] = i < end ? make_float4(x[i], y[i], z[i], 0.0f)
                              : make_float4(__builtin_nanf(""), 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < RP_TILE; ++t) {
            const float4 p = tile[t];
            const float dist = fabsf(((f.nx * p.x + f.ny * p.y) + f.nz * p.z) + f.d);
            c += (dist <= thr) ? 1u : 0u;
        }
    }
    if (s < ns && f.valid && c) atomicAdd(&counts[s], (unsigned long long)c);
}

// What the search over the samples settles on: the plane, its inlier count, and the number of
// inliers the mask pass finds for it (the same number, unless no sample was valid).
struct PlanePick {
    PlaneFit best;
    uint32_t pad;
    unsigned long long count;
    unsigned long long inliers;
};

// ransac_plane.rs:83-94 on the device: reduce_with(|a, b| if a.1 >= b.1 { a } else { b }) over the
// valid samples keeps the earliest of the planes with the largest count; nothing valid -> the
// default model with count 0.  One block; also clears the inlier total for the mask pass.
__global__ __launch_bounds__(256) void plane_pick_kernel(const PlaneFit *fits,
                                                         const unsigned long long *counts,
                                                         uint32_t ns, PlanePick *pick) {
    __shared__ unsigned long long s_count[256];
    __shared__ uint32_t s_at[256];
    unsigned long long c = 0;
    uint32_t at = 0xFFFFFFFFu;  // no valid sample yet
    for (uint32_t s = threadIdx.x; s < ns; s += 256) {
        if (!fits[s].valid) continue;
        if (at == 0xFFFFFFFFu || counts[s] > c) {  // ascending s: an equal count keeps the earlier
            c = counts[s];
            at = s;
        }
    }
    s_count[threadIdx.x] = c;
    s_at[threadIdx.x] = at;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const unsigned long long oc = s_count[threadIdx.x + w];
            const uint32_t oa = s_at[threadIdx.x + w];
            const uint32_t ma = s_at[threadIdx.x];
            const bool take = oa != 0xFFFFFFFFu &&
                              (ma == 0xFFFFFFFFu || oc > s_count[threadIdx.x] ||
                               (oc == s_count[threadIdx.x] && oa < ma));
            if (take) {
                s_count[threadIdx.x] = oc;
                s_at[threadIdx.x] = oa;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t b = s_at[0];
        PlanePick p;
        p.best = b != 0xFFFFFFFFu ? fits[b] : PlaneFit{0.0f, 0.0f, 1.0f, 0.0f, 1u};
        p.pad = 0;
        p.count = b != 0xFFFFFFFFu ? s_count[0] : 0ull;
        p.inliers = 0;
        *pick = p;
    }
}

__global__ __launch_bounds__(256) void plane_mask_pick_kernel(const float *x, const float *y,
                                                              const float *z, size_t n,
                                                              PlanePick *pick, float thr,
                                                              uint8_t *mask) {
    const PlaneFit f = pick->best;
    unsigned local = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float dist = fabsf(((f.nx * x[i] + f.ny * y[i]) + f.nz * z[i]) + f.d);
        const bool in = dist <= thr;
        mask[i] = in ? 1 : 0;
        local += in ? 1u : 0u;
    }
    block_count_add(local, &pick->inliers);
}

__global__ __launch_bounds__(256) void plane_mask_kernel(const float *x, const float *y,
                                                         const float *z, size_t n, PlaneFit f,
                                                         float thr, uint8_t *mask,
                                                         unsigned long long *total) {
    unsigned local = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float dist = fabsf(((f.nx * x[i] + f.ny * y[i]) + f.nz * z[i]) + f.d);
        const bool in = dist <= thr;
        mask[i] = in ? 1 : 0;
        local += in ? 1u : 0u;
    }
    block_count_add(local, total);
}

}  // namespace pcrs

using namespace pcrs;

int pcrs_launch_ransac_plane(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                             size_t n, const uint32_t *d_samples, size_t ns, float thr,
                             float model_out[4], uint8_t *d_mask, size_t *n_inliers,
                             uint64_t *best_count_out, uint8_t *h_mask_out) {
    model_out[0] = 0.0f;  // PlaneModel::default (:23-30)
    model_out[1] = 0.0f;
    model_out[2] = 1.0f;
    model_out[3] = 0.0f;
    *n_inliers = 0;
    if (best_count_out) *best_count_out = 0;
    if (n >= 0xFFFFFFFFull || ns >= 0xFFFFFFFFull / 3)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "cloud or sample list too long");
    if (n < 3) {  // :64-66
        if (n) PCRS_HIP_CHECK(ctx, hipMemsetAsync(d_mask, 0, n, ctx->stream));
        if (n && h_mask_out) {
            PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h_mask_out, d_mask, n, hipMemcpyDeviceToHost, ctx->stream));
            PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        }
        return PCRS_OK;
    }
    PlaneFit best = {0.0f, 0.0f, 1.0f, 0.0f, 1u};
    unsigned long long best_count = 0;
    if (ns) {
        DevBuf fits, counts;
        PCRS_HIP_CHECK(ctx, fits.alloc(ctx, ns * sizeof(PlaneFit)));
        PCRS_HIP_CHECK(ctx, counts.alloc(ctx, ns * sizeof(unsigned long long)));
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(counts.p, 0, ns * sizeof(unsigned long long), ctx->stream));
        const unsigned groups = (unsigned)((ns + 255) / 256);
        hipLaunchKernelGGL(plane_fit_kernel, dim3(groups), dim3(256), 0, ctx->stream, x, y, z,
                           (uint32_t)n, d_samples, (uint32_t)ns, fits.as<PlaneFit>());
        // enough slices of the cloud to fill the GPU, each a multiple of the LDS tile
        size_t slices = ((size_t)ctx->num_cus * 8 + groups - 1) / groups;
        const size_t max_slices = (n + RP_TILE - 1) / RP_TILE;
        if (slices > max_slices) slices = max_slices;
        if (slices > 65535) slices = 65535;
        if (slices == 0) slices = 1;
        size_t slice = (n + slices - 1) / slices;
        slice = (slice + RP_TILE - 1) / RP_TILE * RP_TILE;
        slices = (n + slice - 1) / slice;
        hipLaunchKernelGGL(plane_count_kernel, dim3(groups, (unsigned)slices), dim3(256), 0,
                           ctx->stream, x, y, z, (uint32_t)n, (uint32_t)slice, fits.as<PlaneFit>(),
                           (uint32_t)ns, thr, counts.as<unsigned long long>());
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        if (n >= 10000 && ns >= 16) {  // :81, the parallel branch
            // The choice among the samples is a reduction with a fixed rule: it stays on the
            // device, the mask pass reads the chosen plane from memory, and the host waits once --
            // for the plane, its count and the inlier total, which kernels leave in pinned memory
            // -- instead of three times with two pageable copies of the per-sample results.
            void *pin = nullptr, *sc = nullptr;
            PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
            PCRS_TRY(pcrs_ctx_scratch(ctx, 64, &sc));
            static_assert(sizeof(PlanePick) <= 64 && sizeof(PlanePick) % 4 == 0, "fits the scratch words");
            PlanePick *dpick = reinterpret_cast<PlanePick *>(sc);
            hipLaunchKernelGGL(plane_pick_kernel, dim3(1), dim3(256), 0, ctx->stream,
                               fits.as<PlaneFit>(), counts.as<unsigned long long>(), (uint32_t)ns,
                               dpick);
            size_t mblocks = (n + 1023) / 1024;
            if (mblocks > (size_t)ctx->num_cus * 2) mblocks = (size_t)ctx->num_cus * 2;
            hipLaunchKernelGGL(plane_mask_pick_kernel, dim3((unsigned)mblocks), dim3(256), 0,
                               ctx->stream, x, y, z, n, dpick, thr, d_mask);
            PCRS_HIP_CHECK(ctx, hipGetLastError());
            PCRS_TRY(pcrs_launch_publish(ctx, dpick, pin, sizeof(PlanePick)));
            if (h_mask_out)
                PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h_mask_out, d_mask, n, hipMemcpyDeviceToHost,
                                                   ctx->stream));
            PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            const PlanePick *hp = reinterpret_cast<const PlanePick *>(pin);
            model_out[0] = hp->best.nx;
            model_out[1] = hp->best.ny;
            model_out[2] = hp->best.nz;
            model_out[3] = hp->best.d;
            if (best_count_out) *best_count_out = hp->count;
            *n_inliers = (size_t)hp->inliers;
            return PCRS_OK;
        }
        std::vector<PlaneFit> hf(ns);
        std::vector<unsigned long long> hc(ns);
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(hf.data(), fits.p, ns * sizeof(PlaneFit),
                                           hipMemcpyDeviceToHost, ctx->stream));
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), counts.p, ns * sizeof(unsigned long long),
                                           hipMemcpyDeviceToHost, ctx->stream));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const bool use_parallel = n >= 10000 && ns >= 16;  // :81
        bool have = false;
        if (use_parallel) {
            // :83-94: reduce_with(|a, b| if a.1 >= b.1 { a } else { b }) keeps the earliest of the
            // planes with the largest count; nothing valid -> default model
            for (size_t s = 0; s < ns; ++s) {
                if (!hf[s].valid) continue;
                if (!have || hc[s] > best_count) {
                    best = hf[s];
                    best_count = hc[s];
                    have = true;
                }
            }
            if (!have) best_count = 0;
        } else {
            // :96-122: strict improvement, adaptive early termination
            for (size_t s = 0; s < ns; ++s) {
                if (!hf[s].valid) continue;
                if (hc[s] > best_count) {
                    best_count = hc[s];
                    best = hf[s];
                    const double w = (double)best_count / (double)n;
                    if (w > 0.5) {
                        const double needed = std::log(1.0 - 0.999) / std::log(1.0 - w * w * w);
                        if ((double)s > needed) break;
                    }
                }
            }
        }
    }
    model_out[0] = best.nx;
    model_out[1] = best.ny;
    model_out[2] = best.nz;
    model_out[3] = best.d;
    if (best_count_out) *best_count_out = best_count;
    // :125-128 — the inliers of the chosen (possibly default) model
    void *pin = nullptr, *sc = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
    PCRS_TRY(pcrs_ctx_scratch(ctx, 64, &sc));
    unsigned long long *dtotal = reinterpret_cast<unsigned long long *>(sc);
    unsigned long long *htotal = reinterpret_cast<unsigned long long *>(pin);
    PCRS_HIP_CHECK(ctx, hipMemsetAsync(dtotal, 0, sizeof(*dtotal), ctx->stream));
    size_t blocks = (n + 1023) / 1024;
    if (blocks > (size_t)ctx->num_cus * 2) blocks = (size_t)ctx->num_cus * 2;
    hipLaunchKernelGGL(plane_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, y, z,
                       n, best, thr, d_mask, dtotal);
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(htotal, dtotal, sizeof(*htotal), hipMemcpyDeviceToHost,
                                       ctx->stream));
    if (h_mask_out)
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h_mask_out, d_mask, n, hipMemcpyDeviceToHost, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *n_inliers = (size_t)*htotal;
    return PCRS_OK;
}
