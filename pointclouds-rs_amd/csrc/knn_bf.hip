// LDS-tiled brute-force k-nearest-neighbour kernel for gfx950.
//
// One query per lane, 256-thread workgroups.  The reference cloud streams from HBM once per
// workgroup as coalesced SoA column tiles staged in LDS; every lane then reads the same tile
// element (an LDS broadcast, conflict-free) and keeps its K best (d2, index) keys in registers.
// The per-query consumer (SOR mean, PCA normal, list / 1-NN output) runs in the same kernel.
//
// This is the exact replacement of kiddo's nearest_n behind KdTree::knn
// (crates/spatial/src/kdtree.rs:64-96) for clouds too small to benefit from the grid, the
// fallback for queries the grid kernels cannot certify, and the path for k > 64.
#include "knn_common.h"

#pragma clang fp contract(off)

namespace pcrs {

constexpr int BF_THREADS = 256;
constexpr int BF_TILE = 1024;  // points per LDS tile: 3 columns x 4 KB

struct BfArgs {
    const float *rx, *ry, *rz;
    uint32_t m;
    const float *qx, *qy, *qz;
    uint32_t nq;
    const uint32_t *qlist;        // nullable: positions of the queries to process
    const uint32_t *qlist_count;  // device count of qlist entries
    // raw key list output for the multi-pass path (nullable)
    uint64_t *list_keys;
    uint32_t list_stride;  // keys per query in list_keys
    uint32_t list_off;     // first slot this pass fills
    uint64_t *lo_io;       // per query: smallest admissible key (inclusive); updated per pass
};

template <int K>
__global__ __launch_bounds__(BF_THREADS) void knn_bf_kernel(BfArgs a, KnnOut out) {
    __shared__ __attribute__((aligned(16))) float tx[BF_TILE];
    __shared__ __attribute__((aligned(16))) float ty[BF_TILE];
    __shared__ __attribute__((aligned(16))) float tz[BF_TILE];

    const uint32_t tid = threadIdx.x;
    const uint32_t pos = blockIdx.x * BF_THREADS + tid;
    uint32_t total = a.nq;
    if (a.qlist) {
        total = *a.qlist_count;
        if (blockIdx.x * BF_THREADS >= total) return;  // uniform for the block
    }
    const bool active = pos < total;
    uint32_t qi = 0;
    if (active) qi = a.qlist ? a.qlist[pos] : pos;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (active) {
        qx = a.qx[qi];
        qy = a.qy[qi];
        qz = a.qz[qi];
    }
    const bool qvalid = active && finite3(qx, qy, qz);
    uint64_t lo = 0;
    if (a.lo_io && active) lo = a.lo_io[qi];

    TopK<K> top;
    top.init(qvalid);

    for (uint32_t base = 0; base < a.m; base += BF_TILE) {
        __syncthreads();
#pragma unroll
        for (int t = tid; t < BF_TILE; t += BF_THREADS) {
            const uint32_t g = base + t;
            float x = __builtin_nanf(""), y = 0.0f, z = 0.0f;
            if (g < a.m) {
                x = a.rx[g];
                y = a.ry[g];
                z = a.rz[g];
                if (!finite3(x, y, z)) x = __builtin_nanf("");  // never a neighbour
            }
            tx[t] = x;
            ty[t] = y;
            tz[t] = z;
        }
        __syncthreads();
        const uint32_t lim = min((uint32_t)BF_TILE, a.m - base);
        for (uint32_t j = 0; j < lim; j += 4) {
            const float4 X = *reinterpret_cast<const float4 *>(&tx[j]);
            const float4 Y = *reinterpret_cast<const float4 *>(&ty[j]);
            const float4 Z = *reinterpret_cast<const float4 *>(&tz[j]);
            const float xs[4] = {X.x, X.y, X.z, X.w};
            const float ys[4] = {Y.x, Y.y, Y.z, Y.w};
            const float zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d2 = sqdist(qx, qy, qz, xs[u], ys[u], zs[u]);
                if (d2 <= top.worst) {
                    const uint64_t nk = make_key(d2, base + j + u);
                    if (nk < top.key[K - 1] && nk >= lo) top.insert(nk);
                }
            }
        }
    }

    if (!active) return;
    if (a.list_keys) {
        uint64_t *dst = a.list_keys + (size_t)qi * a.list_stride + a.list_off;
        const uint32_t room = a.list_stride - a.list_off;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((uint32_t)j < room) dst[j] = top.key[j];
        // next pass admits only keys above the last one found here
        const uint64_t last = top.key[K - 1];
        a.lo_io[qi] = (last == PCRS_EMPTY_KEY) ? PCRS_EMPTY_KEY : last + 1;
        return;
    }
    consume(RegNbrs<K>(top.key, a.rx, a.ry, a.rz), top.count(), qvalid, (size_t)qi, qx, qy, qz,
            out);
}

// Consumers over key lists in global memory (multi-pass path).
__global__ __launch_bounds__(BF_THREADS) void consume_list_kernel(BfArgs a, KnnOut out) {
    const uint32_t pos = blockIdx.x * BF_THREADS + threadIdx.x;
    uint32_t total = a.nq;
    if (a.qlist) total = *a.qlist_count;
    if (pos >= total) return;
    const uint32_t qi = a.qlist ? a.qlist[pos] : pos;
    const float qx = a.qx[qi], qy = a.qy[qi], qz = a.qz[qi];
    const bool qvalid = finite3(qx, qy, qz);
    const uint64_t *keys = a.list_keys + (size_t)qi * a.list_stride;
    uint32_t cnt = 0;
    while (cnt < a.list_stride && keys[cnt] != PCRS_EMPTY_KEY) ++cnt;
    consume(ListNbrs{keys, a.rx, a.ry, a.rz}, cnt, qvalid, (size_t)qi, qx, qy, qz, out);
}

// Self-query ranges answered from the sorted grid array never see their non-finite points:
// this kernel writes the consumers' no-neighbour defaults for them.
__global__ __launch_bounds__(BF_THREADS) void invalid_defaults_kernel(BfArgs a, KnnOut out) {
    const uint32_t qi = blockIdx.x * BF_THREADS + threadIdx.x;
    if (qi >= a.nq) return;
    const float qx = a.qx[qi], qy = a.qy[qi], qz = a.qz[qi];
    if (finite3(qx, qy, qz)) return;
    uint64_t none[1] = {PCRS_EMPTY_KEY};
    consume(RegNbrs<1>(none, a.rx, a.ry, a.rz), 0u, false, (size_t)qi, qx, qy, qz, out);
}

__global__ void fill_u64_kernel(uint64_t *p, size_t n, uint64_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace pcrs

using namespace pcrs;

int pcrs_launch_knn_bf(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz, size_t m,
                       const KnnQuery &q, const KnnOut &out, uint32_t kneed, const uint32_t *qlist,
                       const uint32_t *qlist_count, size_t qlist_cap) {
    if (q.nq == 0) return PCRS_OK;
    if (m >= 0xFFFFFFFFull || q.nq >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    if (kneed == 0) kneed = 1;
    BfArgs a{};
    a.rx = rx;
    a.ry = ry;
    a.rz = rz;
    a.m = (uint32_t)m;
    a.qx = q.qx;
    a.qy = q.qy;
    a.qz = q.qz;
    a.nq = (uint32_t)q.nq;
    a.qlist = qlist;
    a.qlist_count = qlist_count;
    const size_t work = qlist ? qlist_cap : q.nq;
    if (work == 0) return PCRS_OK;
    const dim3 grid((unsigned)((work + BF_THREADS - 1) / BF_THREADS));
    const dim3 block(BF_THREADS);

    // A list never holds more than the m reference points.
    const uint32_t keff = (uint32_t)((size_t)kneed < m ? kneed : (m ? m : 1));
    if (keff <= PCRS_KMAX_REG) {
        const int kb = k_bucket(keff);
        pcrs_timing_begin(ctx);
#define PCRS_BF_CALL(KK) \
    hipLaunchKernelGGL(knn_bf_kernel<KK>, grid, block, 0, ctx->stream, a, out)
        PCRS_DISPATCH_K(kb, PCRS_BF_CALL)
#undef PCRS_BF_CALL
        pcrs_timing_end(ctx);
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        return PCRS_OK;
    }

    // k > 64: passes of 64 neighbours each, every pass admitting only keys above the previous
    // pass's last key; the consumers then read the assembled ascending list.
    const uint32_t stride = keff;
    const size_t list_bytes = (size_t)q.nq * stride * sizeof(uint64_t);
    const size_t lo_bytes = (size_t)q.nq * sizeof(uint64_t);
    DevBuf lists;  // private to this call: freed (after a stream sync) on return
    PCRS_HIP_CHECK(ctx, lists.alloc(list_bytes + lo_bytes));
    a.list_keys = lists.as<uint64_t>();
    a.lo_io = reinterpret_cast<uint64_t *>(lists.as<char>() + list_bytes);
    a.list_stride = stride;
    {
        const size_t nfill = (size_t)q.nq * stride;
        hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((nfill + 255) / 256)), dim3(256), 0,
                           ctx->stream, a.list_keys, nfill, (uint64_t)PCRS_EMPTY_KEY);
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(a.lo_io, 0, lo_bytes, ctx->stream));
    }
    for (uint32_t off = 0; off < stride; off += PCRS_KMAX_REG) {
        a.list_off = off;
        hipLaunchKernelGGL(knn_bf_kernel<PCRS_KMAX_REG>, grid, block, 0, ctx->stream, a, out);
    }
    hipLaunchKernelGGL(consume_list_kernel, grid, block, 0, ctx->stream, a, out);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return PCRS_OK;
}

int pcrs_launch_invalid_defaults(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz,
                                 const KnnQuery &q, const KnnOut &out) {
    if (q.nq == 0) return PCRS_OK;
    BfArgs a{};
    a.rx = rx;
    a.ry = ry;
    a.rz = rz;
    a.qx = q.qx;
    a.qy = q.qy;
    a.qz = q.qz;
    a.nq = (uint32_t)q.nq;
    hipLaunchKernelGGL(invalid_defaults_kernel, dim3((unsigned)((q.nq + BF_THREADS - 1) / BF_THREADS)),
                       dim3(BF_THREADS), 0, ctx->stream, a, out);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}
