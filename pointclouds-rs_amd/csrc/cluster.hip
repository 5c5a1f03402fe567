// euclidean_cluster (crates/segmentation/src/euclidean_cluster.rs:96-187) on the device.
//   cells   : finite points ordered by the reference's cell key floor(p / threshold) (:56-62,
//             :110-118), cells in lexicographic key order, points of a cell in index order
//   edges   : one thread per point walks its own cell (later points only) and the 13 forward
//             neighbour cells (:66-83, :127-160) and joins every pair within the threshold in a
//             lock-free union-find whose roots are the smallest index of their tree
//   labels  : label[i] = root of i; non-finite points stay singletons (:171-175)
//   clusters: stable sort by label (indices ascending inside a cluster, clusters by smallest
//             index), sizes, size filter, stable sort by descending size (:177-186), CSR output
// Connected components do not depend on the order of the unions, so the output equals the
// reference's exactly.
#include "knn_common.h"

#include <cstdlib>

#pragma clang fp contract(off)

namespace pcrs {

__device__ __forceinline__ uint32_t uf_load(const uint32_t *parent, uint32_t i) {
    return __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// root of x with path halving; indices only ever point to smaller indices
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    for (;;) {
        const uint32_t p = uf_load(parent, x);
        if (p == x) return x;
        const uint32_t gp = uf_load(parent, p);
        if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
    }
}

// joins the sets of a and b; returns the root of the joined set as seen by this thread
__device__ __forceinline__ uint32_t uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return a;
        if (a < b) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        // the larger root goes under the smaller one; a lost race starts over
        if (atomicCAS(parent + a, a, b) == a) return b;
    }
}

__global__ __launch_bounds__(256) void iota_u32_kernel(uint32_t *v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = i;
}

// per occupied cell: first sorted position and the three keys; first[ncells] = nvalid
__global__ __launch_bounds__(256) void cluster_cells_kernel(const uint8_t *flag,
                                                            const uint32_t *cell_id,
                                                            const uint32_t *order, uint32_t nvalid,
                                                            const int *kx, const int *ky,
                                                            const int *kz, uint32_t *first, int *ux,
                                                            int *uy, int *uz) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) first[cell_id[nvalid]] = nvalid;  // cell_id[nvalid] = number of cells
    if (i >= nvalid || !flag[i]) return;
    const uint32_t c = cell_id[i], p = order[i];
    first[c] = i;
    ux[c] = kx[p];
    uy[c] = ky[p];
    uz[c] = kz[p];
}

// first cell whose key is >= (x, y, z) in lexicographic order
__device__ __forceinline__ uint32_t cell_lower_bound(const int *ux, const int *uy, const int *uz,
                                                     uint32_t ncells, long long x, long long y,
                                                     long long z) {
    uint32_t lo = 0, hi = ncells;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        const long long mx = ux[mid], my = uy[mid], mz = uz[mid];
        const bool less = mx < x || (mx == x && (my < y || (my == y && mz < z)));
        if (less)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void cluster_union_kernel(
    const float *x, const float *y, const float *z, const uint8_t *flag, const uint32_t *cell_id,
    const uint32_t *order, uint32_t nvalid, const uint32_t *first, const int *ux, const int *uy,
    const int *uz, float r2, uint32_t *parent) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nvalid) return;
    const uint32_t ncells = cell_id[nvalid];
    const uint32_t cell = cell_id[i] - (flag[i] ? 0u : 1u);
    const uint32_t a = order[i];
    const float ax = x[a], ay = y[a], az = z[a];
    const long long cx = ux[cell], cy = uy[cell], cz = uz[cell];
    // In a dense cluster nearly every pair is already joined when it is looked at, and a find
    // is a chain of dependent device-scope loads.  The thread therefore remembers the root of its
    // own point and first reads the neighbour's parent with a plain load, four neighbours at a
    // time: parent[b] == root, even read from a stale line, proves that b and the root were once
    // in one set — sets only ever merge — so the pair needs no union.
    uint32_t root = a;
    auto join_range = [&](uint32_t from, uint32_t to) {
        if (to - from < 4u) {  // sparse cells: nothing to batch
            for (uint32_t j = from; j < to; ++j) {
                const uint32_t b = order[j];
                const float dx = ax - x[b], dy = ay - y[b], dz = az - z[b];
                if ((dx * dx + dy * dy) + dz * dz <= r2 && b != root)
                    root = uf_union(parent, a, b);
            }
            return;
        }
        for (uint32_t j0 = from; j0 < to; j0 += 4) {
            uint32_t b[4], pb[4];
            float d2[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t j = j0 + t < to ? j0 + t : to - 1;
                b[t] = order[j];
                const float dx = ax - x[b[t]], dy = ay - y[b[t]], dz = az - z[b[t]];
                d2[t] = (dx * dx + dy * dy) + dz * dz;
                pb[t] = parent[b[t]];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (j0 + t < to && d2[t] <= r2 && pb[t] != root && b[t] != root)
                    root = uf_union(parent, a, b[t]);  // :151-153
            }
        }
    };
    // own cell: the points after this one (:148)
    join_range(i + 1, first[cell + 1]);
    // (0, 0, +1) is the next cell of the list when it exists
    if (cell + 1 < ncells && ux[cell + 1] == cx && uy[cell + 1] == cy &&
        (long long)uz[cell + 1] == cz + 1)
        join_range(first[cell + 1], first[cell + 2]);
    // the four forward (dx, dy) columns, each with dz = -1, 0, +1 next to each other in the list
    const int ox[4] = {1, 1, 1, 0}, oy[4] = {0, 1, -1, 1};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long long tx = cx + ox[c], ty = cy + oy[c];
        uint32_t t = cell_lower_bound(ux, uy, uz, ncells, tx, ty, cz - 1);
        while (t < ncells && ux[t] == tx && uy[t] == ty && (long long)uz[t] <= cz + 1) {
            join_range(first[t], first[t + 1]);
            ++t;
        }
    }
}

// The same unions with ONE WAVE PER POINT, for small inputs (a few thousand obstacle points of a
// LiDAR frame in a handful of crowded cells): a thread per point is then a few dozen waves, each
// a serial loop over hundreds of candidates with a chain of dependent loads behind every union;
// here the 64 lanes of a wave share the point's candidates.  Which thread joins a pair, and in
// which order, does not change the components that come out.
__global__ __launch_bounds__(256) void cluster_union_wave_kernel(
    const float *x, const float *y, const float *z, const uint8_t *flag, const uint32_t *cell_id,
    const uint32_t *order, uint32_t nvalid, const uint32_t *first, const int *ux, const int *uy,
    const int *uz, float r2, uint32_t *parent) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per sorted position
    if (i >= nvalid) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ncells = cell_id[nvalid];
    const uint32_t cell = cell_id[i] - (flag[i] ? 0u : 1u);
    const uint32_t a = order[i];
    const float ax = x[a], ay = y[a], az = z[a];
    const long long cx = ux[cell], cy = uy[cell], cz = uz[cell];
    uint32_t root = a;  // what this lane last saw as the root of a (see cluster_union_kernel)
    // The up to 64 neighbours of a step are joined to a by the wave as a whole: all lanes calling
    // uf_union for the same point would queue up on the one root they all try to move (one
    // compare-and-swap wins per round, the others start over).  Here every lane finds the root of
    // its own neighbour, the smallest root of the step (a's included) is agreed on by shuffles, and
    // each distinct larger root is hung under it — different addresses, one round nearly always.
    auto join_range = [&](uint32_t from, uint32_t to) {
        for (uint32_t j0 = from; j0 < to; j0 += 64u) {
            const uint32_t j = j0 + lane;
            uint32_t b = a;
            bool hit = false;
            if (j < to) {
                b = order[j];
                const float dx = ax - x[b], dy = ay - y[b], dz = az - z[b];
                hit = (dx * dx + dy * dy) + dz * dz <= r2 && b != root && parent[b] != root;  // :151-153
            }
            while (__ballot(hit)) {
                const uint32_t ra = uf_find(parent, a);  // the same in every lane
                const uint32_t rb = hit ? uf_find(parent, b) : ra;
                uint32_t m = rb < ra ? rb : ra;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = __shfl_xor(m, off, 64);
                    m = o < m ? o : m;
                }
                // a larger root goes under the smallest one (indices only ever point downwards);
                // a lost race — the root was moved by someone else meanwhile — shows up as a
                // different root in the next round
                if (hit && rb != m) atomicCAS(parent + rb, rb, m);
                if (lane == 0u && ra != m) atomicCAS(parent + ra, ra, m);
                root = m;
                hit = hit && rb != ra;  // joined already when the two roots were one
            }
        }
    };
    join_range(i + 1, first[cell + 1]);  // own cell: the points after this one (:148)
    if (cell + 1 < ncells && ux[cell + 1] == cx && uy[cell + 1] == cy &&
        (long long)uz[cell + 1] == cz + 1)
        join_range(first[cell + 1], first[cell + 2]);
    const int ox[4] = {1, 1, 1, 0}, oy[4] = {0, 1, -1, 1};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long long tx = cx + ox[c], ty = cy + oy[c];
        uint32_t t = cell_lower_bound(ux, uy, uz, ncells, tx, ty, cz - 1);
        while (t < ncells && ux[t] == tx && uy[t] == ty && (long long)uz[t] <= cz + 1) {
            join_range(first[t], first[t + 1]);
            ++t;
        }
    }
}

// label = smallest index of the component, as the sort key
__global__ __launch_bounds__(256) void cluster_label_kernel(uint32_t *parent, uint32_t n,
                                                            uint64_t *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = uf_find(parent, i);
    vals[i] = i;
}

__global__ __launch_bounds__(256) void cluster_heads_kernel(const uint64_t *labels, uint32_t n,
                                                            uint32_t *head, uint8_t *flag) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = (i == 0 || labels[i] != labels[i - 1]) ? 1u : 0u;
    head[i] = h;
    flag[i] = (uint8_t)h;
}

// start[c] = first sorted position of cluster c; start[nc] = n
__global__ __launch_bounds__(256) void cluster_starts_kernel(const uint8_t *flag,
                                                             const uint32_t *cid, uint32_t n,
                                                             uint32_t *start) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) start[cid[n]] = n;
    if (i < n && flag[i]) start[cid[i]] = i;
}

// sort key of cluster c: n - size (ascending = size descending; stable keeps smallest index first)
__global__ __launch_bounds__(256) void cluster_size_keys_kernel(const uint32_t *start,
                                                                const uint32_t *d_nc, uint32_t n,
                                                                uint64_t *keys, uint32_t *vals) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= *d_nc) return;
    keys[c] = (uint64_t)(n - (start[c + 1] - start[c]));
    vals[c] = c;
}

// for the t-th cluster in output order: keep flag and the number of indices it contributes
__global__ __launch_bounds__(256) void cluster_filter_kernel(const uint32_t *sorted_cluster,
                                                             const uint32_t *start,
                                                             const uint32_t *d_nc,
                                                             unsigned long long min_size,
                                                             unsigned long long max_size,
                                                             uint32_t *keep, uint32_t *kept_size) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= *d_nc) return;
    const uint32_t c = sorted_cluster[t];
    const unsigned long long size = start[c + 1] - start[c];
    const bool k = size >= min_size && size <= max_size;  // :179
    keep[t] = k ? 1u : 0u;
    kept_size[t] = k ? (uint32_t)size : 0u;
}

// offsets of the kept clusters and, per cluster, where its indices go (NO_INDEX = dropped)
__global__ __launch_bounds__(256) void cluster_offsets_kernel(
    const uint32_t *sorted_cluster, const uint32_t *start, const uint32_t *d_nc,
    const uint32_t *keep_pos, const uint32_t *kept_off, unsigned long long min_size,
    unsigned long long max_size, uint32_t *offsets_out, uint32_t *dest) {
    const uint32_t nc = *d_nc;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) offsets_out[keep_pos[nc]] = kept_off[nc];  // totals close the CSR
    if (t >= nc) return;
    const uint32_t c = sorted_cluster[t];
    const unsigned long long size = start[c + 1] - start[c];
    const bool k = size >= min_size && size <= max_size;
    dest[c] = k ? kept_off[t] : PCRS_NO_INDEX;
    if (k) offsets_out[keep_pos[t]] = kept_off[t];
}

__global__ __launch_bounds__(256) void cluster_emit_kernel(const uint8_t *flag, const uint32_t *cid,
                                                           const uint32_t *sorted_index,
                                                           const uint32_t *start,
                                                           const uint32_t *dest, uint32_t n,
                                                           uint32_t *indices_out) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t c = cid[p] - (flag[p] ? 0u : 1u);
    const uint32_t d = dest[c];
    if (d != PCRS_NO_INDEX) indices_out[d + (p - start[c])] = sorted_index[p];
}

}  // namespace pcrs

using namespace pcrs;

int pcrs_launch_euclidean_cluster(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                                  size_t n, float threshold, size_t min_size, size_t max_size,
                                  uint32_t *d_offsets, uint32_t *d_indices, size_t *n_clusters,
                                  size_t *n_indices) {
    *n_clusters = 0;
    *n_indices = 0;
    // :102-104 (a NaN threshold passes, as in the reference)
    if (n == 0 || threshold <= 0.0f || min_size == 0) return PCRS_OK;
    if (n >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    const uint32_t un = (uint32_t)n;
    const unsigned nblocks = (un + 255) / 256;
    const float inv_r = 1.0f / threshold;      // :107
    const float r2 = threshold * threshold;    // :108

    DevBuf parent;
    PCRS_HIP_CHECK(ctx, parent.alloc(ctx, n * sizeof(uint32_t)));
    hipLaunchKernelGGL(iota_u32_kernel, dim3(nblocks), dim3(256), 0, ctx->stream,
                       parent.as<uint32_t>(), un);
    {
        PcrsCellSort cs;
        PCRS_TRY(pcrs_cell_sort(ctx, x, y, z, n, inv_r, true, &cs));
        if (cs.nvalid) {
            const uint32_t nv = cs.nvalid;
            DevBuf first, u[3];
            PCRS_HIP_CHECK(ctx, first.alloc(ctx, ((size_t)nv + 2) * sizeof(uint32_t)));
            for (int a = 0; a < 3; ++a) PCRS_HIP_CHECK(ctx, u[a].alloc(ctx, (size_t)nv * sizeof(int)));
            const unsigned vb = (nv + 255) / 256;
            hipLaunchKernelGGL(cluster_cells_kernel, dim3(vb), dim3(256), 0, ctx->stream,
                               cs.flag.as<uint8_t>(), cs.cell_id.as<uint32_t>(), cs.order, nv,
                               cs.k[0].as<int>(), cs.k[1].as<int>(), cs.k[2].as<int>(),
                               first.as<uint32_t>(), u[0].as<int>(), u[1].as<int>(), u[2].as<int>());
            static const uint32_t wave_max = [] {  // experiments only: PCRS_CLUSTER_WAVE_MAX
                const char *e = getenv("PCRS_CLUSTER_WAVE_MAX");
                return e ? (uint32_t)strtoul(e, nullptr, 10) : 16384u;
            }();
            // few points: a wave each (measured on crowded blobs: 0.36 against 0.54 ms at 5 K
            // points, even at 20 K, 2.4 against 1.0 ms at 60 K — more waves, more contention)
            if (nv <= wave_max)
                hipLaunchKernelGGL(cluster_union_wave_kernel, dim3((nv + 3) / 4), dim3(256), 0,
                                   ctx->stream, x, y, z, cs.flag.as<uint8_t>(),
                                   cs.cell_id.as<uint32_t>(), cs.order, nv, first.as<uint32_t>(),
                                   u[0].as<int>(), u[1].as<int>(), u[2].as<int>(), r2,
                                   parent.as<uint32_t>());
            else
                hipLaunchKernelGGL(cluster_union_kernel, dim3(vb), dim3(256), 0, ctx->stream, x, y, z,
                                   cs.flag.as<uint8_t>(), cs.cell_id.as<uint32_t>(), cs.order, nv,
                                   first.as<uint32_t>(), u[0].as<int>(), u[1].as<int>(),
                                   u[2].as<int>(), r2, parent.as<uint32_t>());
            PCRS_HIP_CHECK(ctx, hipGetLastError());
        }
    }
    // clusters: sort the points by label
    DevBuf keys, keys_alt, vals, vals_alt, cid, flag, sums, start;
    PCRS_HIP_CHECK(ctx, keys.alloc(ctx, n * sizeof(uint64_t)));
    PCRS_HIP_CHECK(ctx, keys_alt.alloc(ctx, n * sizeof(uint64_t)));
    PCRS_HIP_CHECK(ctx, vals.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, vals_alt.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, cid.alloc(ctx, (n + 8) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, flag.alloc(ctx, n + 8));
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, (n / 1024 + 4) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, start.alloc(ctx, (n + 2) * sizeof(uint32_t)));
    hipLaunchKernelGGL(cluster_label_kernel, dim3(nblocks), dim3(256), 0, ctx->stream,
                       parent.as<uint32_t>(), un, keys.as<uint64_t>(), vals.as<uint32_t>());
    int bits = 1;
    while (bits < 32 && ((uint64_t)(n - 1) >> bits) != 0) ++bits;
    bool in_alt = false;
    PCRS_TRY(pcrs_sort_pairs_u64(ctx, keys.as<uint64_t>(), vals.as<uint32_t>(),
                                 keys_alt.as<uint64_t>(), vals_alt.as<uint32_t>(), n, bits, &in_alt));
    const uint64_t *slabel = in_alt ? keys_alt.as<uint64_t>() : keys.as<uint64_t>();
    const uint32_t *sindex = in_alt ? vals_alt.as<uint32_t>() : vals.as<uint32_t>();
    // the other pair of sort buffers is free for the cluster-level sort
    uint64_t *ckeys = in_alt ? keys.as<uint64_t>() : keys_alt.as<uint64_t>();
    uint32_t *cvals = in_alt ? vals.as<uint32_t>() : vals_alt.as<uint32_t>();
    hipLaunchKernelGGL(cluster_heads_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, slabel, un,
                       cid.as<uint32_t>(), flag.as<uint8_t>());
    uint32_t *d_nc = cid.as<uint32_t>() + n;
    PCRS_TRY(pcrs_launch_scan_u32(ctx, cid.as<uint32_t>(), n, sums.as<uint32_t>(), d_nc));
    hipLaunchKernelGGL(cluster_starts_kernel, dim3(nblocks), dim3(256), 0, ctx->stream,
                       flag.as<uint8_t>(), cid.as<uint32_t>(), un, start.as<uint32_t>());
    // clusters by descending size: stable sort of (n - size, cluster) over at most n clusters
    DevBuf ckeys_alt, cvals_alt, keep, kept_size, dest, sums2;
    PCRS_HIP_CHECK(ctx, ckeys_alt.alloc(ctx, n * sizeof(uint64_t)));
    PCRS_HIP_CHECK(ctx, cvals_alt.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, keep.alloc(ctx, (n + 8) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, kept_size.alloc(ctx, (n + 8) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, dest.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, sums2.alloc(ctx, (n / 1024 + 4) * sizeof(uint32_t)));
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
    uint32_t *h = reinterpret_cast<uint32_t *>(pin);
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h, d_nc, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t nc = h[0];
    const unsigned cblocks = (nc + 255) / 256;
    hipLaunchKernelGGL(cluster_size_keys_kernel, dim3(cblocks), dim3(256), 0, ctx->stream,
                       start.as<uint32_t>(), d_nc, un, ckeys, cvals);
    int sbits = 1;
    while (sbits < 33 && ((uint64_t)n >> sbits) != 0) ++sbits;
    bool c_alt = false;
    PCRS_TRY(pcrs_sort_pairs_u64(ctx, ckeys, cvals, ckeys_alt.as<uint64_t>(),
                                 cvals_alt.as<uint32_t>(), nc, sbits, &c_alt));
    const uint32_t *sorted_cluster = c_alt ? cvals_alt.as<uint32_t>() : cvals;
    hipLaunchKernelGGL(cluster_filter_kernel, dim3(cblocks), dim3(256), 0, ctx->stream,
                       sorted_cluster, start.as<uint32_t>(), d_nc, (unsigned long long)min_size,
                       (unsigned long long)max_size, keep.as<uint32_t>(), kept_size.as<uint32_t>());
    PCRS_TRY(pcrs_launch_scan_u32(ctx, keep.as<uint32_t>(), nc, sums2.as<uint32_t>(),
                                  keep.as<uint32_t>() + nc));
    PCRS_TRY(pcrs_launch_scan_u32(ctx, kept_size.as<uint32_t>(), nc, sums2.as<uint32_t>(),
                                  kept_size.as<uint32_t>() + nc));
    hipLaunchKernelGGL(cluster_offsets_kernel, dim3(cblocks), dim3(256), 0, ctx->stream,
                       sorted_cluster, start.as<uint32_t>(), d_nc, keep.as<uint32_t>(),
                       kept_size.as<uint32_t>(), (unsigned long long)min_size,
                       (unsigned long long)max_size, d_offsets, dest.as<uint32_t>());
    hipLaunchKernelGGL(cluster_emit_kernel, dim3(nblocks), dim3(256), 0, ctx->stream,
                       flag.as<uint8_t>(), cid.as<uint32_t>(), sindex, start.as<uint32_t>(),
                       dest.as<uint32_t>(), un, d_indices);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h, keep.as<uint32_t>() + nc, sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h + 1, kept_size.as<uint32_t>() + nc, sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *n_clusters = h[0];
    *n_indices = h[1];
    return PCRS_OK;
}
