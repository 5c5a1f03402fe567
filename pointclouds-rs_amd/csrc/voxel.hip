// voxel_downsample (crates/filters/src/voxel_downsample.rs:12-65) on the device:
//   key      : (floor(p / voxel) as i32) per axis for every finite point        (:31-35)
//   pack     : the three keys relative to their minima, packed so that one unsigned compare is
//              the reference's lexicographic (i32, i32, i32) order               (:49-50)
//   sort     : stable radix sort of (packed key, point index)
//   centroid : one thread per voxel adds its points in ascending index order — the reference adds
//              them in input order (:37-41) — and divides by the count          (:56-62)
// Sums are f32 and sequential per voxel, so the centroids are bit-identical to the reference's.
#include <climits>

#include "knn_common.h"

#pragma clang fp contract(off)

namespace pcrs {

struct VoxelBounds {
    int lo[3], hi[3];
    unsigned int nvalid;
    unsigned int pad;
};

// Rust `f32 as i32`: saturating, NaN -> 0
__device__ __forceinline__ int sat_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT_MAX;
    if (v <= -2147483648.0f) return INT_MIN;
    return (int)v;
}

// multiply == false: floor(p / scale)  (voxel_downsample.rs:31-35)
// multiply == true : floor(p * scale)  (euclidean_cluster.rs:56-62, scale = 1 / threshold)
__global__ __launch_bounds__(256) void voxel_key_kernel(const float *x, const float *y,
                                                        const float *z, uint32_t n, float scale,
                                                        bool multiply, int *kx, int *ky, int *kz,
                                                        VoxelBounds *bounds) {
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    unsigned nv = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float px = x[i], py = y[i], pz = z[i];
        int k[3] = {0, 0, 0};
        if (finite3(px, py, pz)) {
            k[0] = sat_i32(floorf(multiply ? px * scale : px / scale));
            k[1] = sat_i32(floorf(multiply ? py * scale : py / scale));
            k[2] = sat_i32(floorf(multiply ? pz * scale : pz / scale));
            ++nv;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = min(lo[a], k[a]);
                hi[a] = max(hi[a], k[a]);
            }
        }
        kx[i] = k[0];
        ky[i] = k[1];
        kz[i] = k[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = min(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = max(hi[a], __shfl_down(hi[a], off, 64));
        }
    for (int off = 32; off > 0; off >>= 1) nv += __shfl_down(nv, off, 64);
    // one partial per block, reduced by the host (which waits for the bounds anyway): atomics of
    // every wave on the same seven words would serialise
    __shared__ int slo[3][4], shi[3][4];
    __shared__ unsigned snv[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            slo[a][wave] = lo[a];
            shi[a][wave] = hi[a];
        }
        snv[wave] = nv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        VoxelBounds b;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = min(min(slo[a][0], slo[a][1]), min(slo[a][2], slo[a][3]));
            b.hi[a] = max(max(shi[a][0], shi[a][1]), max(shi[a][2], shi[a][3]));
        }
        b.nvalid = snv[0] + snv[1] + snv[2] + snv[3];
        b.pad = 0;
        bounds[blockIdx.x] = b;
    }
}

// packed key of finite points; non-finite points get `invalid_key` (above every real key)
__global__ __launch_bounds__(256) void voxel_pack_kernel(const float *x, const float *y,
                                                         const float *z, uint32_t n, const int *kx,
                                                         const int *ky, const int *kz, int lox,
                                                         int loy, int loz, uint64_t ry,
                                                         uint64_t rz, uint64_t invalid_key,
                                                         uint64_t *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t key = invalid_key;
    if (finite3(x[i], y[i], z[i])) {
        const uint64_t dx = (uint64_t)((int64_t)kx[i] - (int64_t)lox);
        const uint64_t dy = (uint64_t)((int64_t)ky[i] - (int64_t)loy);
        const uint64_t dz = (uint64_t)((int64_t)kz[i] - (int64_t)loz);
        key = (dx * ry + dy) * rz + dz;
    }
    keys[i] = key;
    vals[i] = i;
}

// head[i] = flag[i] = 1 where sorted position i starts a voxel (positions >= nvalid hold the
// non-finite points).  Packed keys compare directly; the per-axis path compares the three keys of
// neighbouring points.
__global__ __launch_bounds__(256) void voxel_heads_kernel(const uint64_t *keys, uint32_t nvalid,
                                                          uint32_t *head, uint8_t *flag) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nvalid) return;
    const uint32_t h = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    head[i] = h;
    flag[i] = (uint8_t)h;
}

__global__ __launch_bounds__(256) void voxel_heads3_kernel(const uint32_t *order, uint32_t nvalid,
                                                           const int *kx, const int *ky,
                                                           const int *kz, uint32_t *head,
                                                           uint8_t *flag) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nvalid) return;
    uint32_t h = 1u;
    if (i != 0) {
        const uint32_t p = order[i], q = order[i - 1];
        h = (kx[p] != kx[q] || ky[p] != ky[q] || kz[p] != kz[q]) ? 1u : 0u;
    }
    head[i] = h;
    flag[i] = (uint8_t)h;
}

// key of one axis for the current order (per-axis path); non-finite points sort last
__global__ __launch_bounds__(256) void voxel_axis_key_kernel(const uint32_t *order, uint32_t n,
                                                             const int *kaxis, int lo,
                                                             uint64_t invalid_key, const float *x,
                                                             const float *y, const float *z,
                                                             uint64_t *keys) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = order[i];
    keys[i] = finite3(x[p], y[p], z[p]) ? (uint64_t)((int64_t)kaxis[p] - (int64_t)lo) : invalid_key;
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t *v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = i;
}

// One thread per voxel: adds its points in sorted (= ascending index) order.
__global__ __launch_bounds__(256) void voxel_centroid_kernel(const uint8_t *flag,
                                                             const uint32_t *order,
                                                             const uint32_t *voxel_id,
                                                             uint32_t nvalid, const float *x,
                                                             const float *y, const float *z,
                                                             float *ox, float *oy, float *oz) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nvalid || !flag[i]) return;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    uint32_t cnt = 0;
    uint32_t j = i;
    do {
        const uint32_t p = order[j];
        sx += x[p];
        sy += y[p];
        sz += z[p];
        ++cnt;
        ++j;
    } while (j < nvalid && !flag[j]);
    const float denom = (float)cnt;  // a.n as f32
    const uint32_t v = voxel_id[i];
    ox[v] = sx / denom;
    oy[v] = sy / denom;
    oz[v] = sz / denom;
}

}  // namespace pcrs

using namespace pcrs;

int pcrs_cell_sort(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                   float scale, bool multiply, PcrsCellSort *cs) {
    cs->nvalid = 0;
    cs->order = nullptr;
    if (n == 0) return PCRS_OK;
    if (n >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    const uint32_t un = (uint32_t)n;
    DevBuf dbounds;
    for (int a = 0; a < 3; ++a) PCRS_HIP_CHECK(ctx, cs->k[a].alloc(ctx, n * sizeof(int)));
    size_t blocks = (n + 255) / 256;
    if (blocks > (size_t)ctx->num_cus * 2) blocks = (size_t)ctx->num_cus * 2;
    PCRS_HIP_CHECK(ctx, dbounds.alloc(ctx, blocks * sizeof(VoxelBounds)));
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, blocks * sizeof(VoxelBounds) + 64, &pin));
    VoxelBounds *parts = reinterpret_cast<VoxelBounds *>(pin);
    hipLaunchKernelGGL(voxel_key_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, y, z,
                       un, scale, multiply, cs->k[0].as<int>(), cs->k[1].as<int>(),
                       cs->k[2].as<int>(), dbounds.as<VoxelBounds>());
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(parts, dbounds.p, blocks * sizeof(VoxelBounds),
                                       hipMemcpyDeviceToHost, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    VoxelBounds total;
    for (int a = 0; a < 3; ++a) {
        total.lo[a] = INT_MAX;
        total.hi[a] = INT_MIN;
    }
    total.nvalid = 0;
    for (size_t b = 0; b < blocks; ++b) {
        if (!parts[b].nvalid) continue;
        for (int a = 0; a < 3; ++a) {
            total.lo[a] = std::min(total.lo[a], parts[b].lo[a]);
            total.hi[a] = std::max(total.hi[a], parts[b].hi[a]);
        }
        total.nvalid += parts[b].nvalid;
    }
    const VoxelBounds *hb = &total;
    const uint32_t nvalid = hb->nvalid;
    cs->nvalid = nvalid;
    if (nvalid == 0) return PCRS_OK;
    for (int a = 0; a < 3; ++a) {
        cs->key_lo[a] = hb->lo[a];
        cs->key_hi[a] = hb->hi[a];
    }
    const int lo[3] = {hb->lo[0], hb->lo[1], hb->lo[2]};
    uint64_t range[3];
    for (int a = 0; a < 3; ++a) range[a] = (uint64_t)((int64_t)hb->hi[a] - (int64_t)lo[a]) + 1u;
    DevBuf &keys = cs->keys, &keys_alt = cs->keys_alt, &vals = cs->vals, &vals_alt = cs->vals_alt;
    DevBuf &head = cs->cell_id, &flag = cs->flag;
    DevBuf sums;
    PCRS_HIP_CHECK(ctx, keys.alloc(ctx, n * sizeof(uint64_t)));
    PCRS_HIP_CHECK(ctx, keys_alt.alloc(ctx, n * sizeof(uint64_t)));
    PCRS_HIP_CHECK(ctx, vals.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, vals_alt.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, head.alloc(ctx, ((size_t)nvalid + 8) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, flag.alloc(ctx, (size_t)nvalid + 8));
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, ((size_t)nvalid / 1024 + 4) * sizeof(uint32_t)));
    const unsigned pblocks = (unsigned)((n + 255) / 256);
    const unsigned vblocks = (nvalid + 255) / 256;
    auto bits_for = [](uint64_t max_value) {
        int b = 1;
        while (b < 64 && (max_value >> b) != 0) ++b;
        return b;
    };
    const uint32_t *order = nullptr;
    const long double cells = (long double)range[0] * (long double)range[1] * (long double)range[2];
    if (cells < 9.0e18L) {
        // usual case: the three keys pack into one 63-bit key (one value above the largest key
        // marks the non-finite points), one sort
        const uint64_t invalid_key = range[0] * range[1] * range[2];
        hipLaunchKernelGGL(voxel_pack_kernel, dim3(pblocks), dim3(256), 0, ctx->stream, x, y, z, un,
                           cs->k[0].as<int>(), cs->k[1].as<int>(), cs->k[2].as<int>(), lo[0], lo[1],
                           lo[2], range[1], range[2], invalid_key, keys.as<uint64_t>(),
                           vals.as<uint32_t>());
        bool in_alt = false;
        PCRS_TRY(pcrs_sort_pairs_u64(ctx, keys.as<uint64_t>(), vals.as<uint32_t>(),
                                     keys_alt.as<uint64_t>(), vals_alt.as<uint32_t>(), n,
                                     bits_for(invalid_key), &in_alt));
        const uint64_t *skeys = in_alt ? keys_alt.as<uint64_t>() : keys.as<uint64_t>();
        order = in_alt ? vals_alt.as<uint32_t>() : vals.as<uint32_t>();
        hipLaunchKernelGGL(voxel_heads_kernel, dim3(vblocks), dim3(256), 0, ctx->stream, skeys,
                           nvalid, head.as<uint32_t>(), flag.as<uint8_t>());
    } else {
        // keys spread over more than 2^63 cells (saturated coordinates): three stable sorts,
        // least significant axis first
        uint32_t *cur = vals.as<uint32_t>(), *alt = vals_alt.as<uint32_t>();
        hipLaunchKernelGGL(iota_kernel, dim3(pblocks), dim3(256), 0, ctx->stream, cur, un);
        for (int a = 2; a >= 0; --a) {
            hipLaunchKernelGGL(voxel_axis_key_kernel, dim3(pblocks), dim3(256), 0, ctx->stream, cur,
                               un, cs->k[a].as<int>(), lo[a], range[a], x, y, z,
                               keys.as<uint64_t>());
            bool in_alt = false;
            PCRS_TRY(pcrs_sort_pairs_u64(ctx, keys.as<uint64_t>(), cur, keys_alt.as<uint64_t>(),
                                         alt, n, bits_for(range[a]), &in_alt));
            if (in_alt) std::swap(cur, alt);
        }
        order = cur;
        hipLaunchKernelGGL(voxel_heads3_kernel, dim3(vblocks), dim3(256), 0, ctx->stream, order,
                           nvalid, cs->k[0].as<int>(), cs->k[1].as<int>(), cs->k[2].as<int>(),
                           head.as<uint32_t>(), flag.as<uint8_t>());
    }
    cs->order = order;
    cs->d_ncells = head.as<uint32_t>() + nvalid;
    PCRS_TRY(pcrs_launch_scan_u32(ctx, head.as<uint32_t>(), nvalid, sums.as<uint32_t>(),
                                  head.as<uint32_t>() + nvalid));
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_voxel_downsample(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                                 size_t n, float voxel_size, float *ox, float *oy, float *oz,
                                 size_t *n_out, double *bounds) {
    *n_out = 0;
    if (n == 0) return PCRS_OK;
    PcrsCellSort cs;
    PCRS_TRY(pcrs_cell_sort(ctx, x, y, z, n, voxel_size, false, &cs));
    if (cs.nvalid == 0) return PCRS_OK;  // voxel_downsample.rs:44-46
    if (bounds) {
        // The points of voxel k lie in [k, k + 1) x voxel_size up to the rounding of one division,
        // and so does their mean up to the rounding of its sums: a thousandth of a voxel and a
        // millionth of the coordinate on either side cover both.  Keys that saturate (coordinates
        // beyond 2^30 voxels) bound nothing.
        bool usable = true;
        for (int a = 0; a < 3; ++a)
            usable = usable && cs.key_lo[a] > -(1 << 30) && cs.key_hi[a] < (1 << 30);
        if (usable)
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)cs.key_lo[a] * (double)voxel_size;
                const double hi = ((double)cs.key_hi[a] + 1.0) * (double)voxel_size;
                const double slack = 1e-3 * (double)voxel_size + 1e-6 * std::max(std::fabs(lo), std::fabs(hi));
                bounds[a] = lo - slack;
                bounds[3 + a] = hi + slack;
            }
    }
    const uint32_t nvalid = cs.nvalid;
    hipLaunchKernelGGL(voxel_centroid_kernel, dim3((nvalid + 255) / 256), dim3(256), 0, ctx->stream,
                       cs.flag.as<uint8_t>(), cs.order, cs.cell_id.as<uint32_t>(), nvalid, x, y, z,
                       ox, oy, oz);
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
    uint32_t *h_total = reinterpret_cast<uint32_t *>(pin);
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h_total, cs.d_ncells, sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = *h_total;
    return PCRS_OK;
}
