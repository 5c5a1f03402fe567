// Uniform-grid pruned exact k-nearest-neighbour search for gfx950.
//
// Build: the finite points are counting-sorted by cell (x-fastest cell order) into one AoS
// float4 array (x, y, z, original index), so the cells of one grid row are contiguous and a
// (2R+1)-cell row of a query's neighbourhood is a single slot range.
//
// Query: one query per lane.  Self queries run in sorted-slot order, so the 64 lanes of a wave
// sit in the same or adjacent cells and their candidate loads hit the same cache lines.  A lane
// walks the (2R+1)^3 cube of cells around its own cell and keeps the K smallest squared
// distances in registers with a branch-free v_med3 chain (no divergence); the result is exact
// iff the K-th distance is smaller than the distance from the query to the nearest face of the
// cube that is not a grid boundary.  Two kernels do this:
//   knn_tag_kernel   (neighbour identities needed, 2 <= K <= 24) one walk: the chain carries
//           32-bit keys of the distance's upper 22 bits and the candidate's ordinal in the walk,
//           so the sorted registers name the neighbours; pairs the truncation may have misordered
//           are put right by their full (d2, index) keys, crowded cubes and tie runs are handed
//           to the other kernel.
//   knn_cell_kernel  pass 1 keeps plain distances, which already yields the SOR mean; pass 2
//           (only when identities are needed) walks the cube again, appends the slots with
//           d2 <= K-th distance to a per-lane LDS list, then orders those few candidates by the
//           full (d2, index) key and runs the fused consumer.
// Queries that cannot be certified go to a compact list and are retried with a larger cube,
// and finally by the brute-force kernel, so the result is exact for every input.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <cstring>

#include "knn_common.h"

#pragma clang fp contract(off)

namespace pcrs {

constexpr int GB_THREADS = 256;

// ----------------------------------------------------------------------------- build kernels

// finite, and inside the box the index is built over (NaN fails every comparison)
__device__ __forceinline__ bool point_taken(const BoxFilter &b, float x, float y, float z) {
    return finite3(x, y, z) && x >= b.lo[0] && x < b.hi[0] && y >= b.lo[1] && y < b.hi[1] &&
           z >= b.lo[2] && z < b.hi[2];
}

struct BBoxPartial {
    float lo[3], hi[3];
    unsigned long long nvalid;
};

__global__ __launch_bounds__(GB_THREADS) void bbox_kernel(const float *x, const float *y,
                                                          const float *z, uint32_t m, BoxFilter box,
                                                          BBoxPartial *partials) {
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    unsigned nv = 0;
    for (uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x; i < m; i += gridDim.x * GB_THREADS) {
        const float p[3] = {x[i], y[i], z[i]};
        if (point_taken(box, p[0], p[1], p[2])) {
            ++nv;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], p[a]);
                hi[a] = fmaxf(hi[a], p[a]);
            }
        }
    }
    __shared__ float slo[3][GB_THREADS / 64], shi[3][GB_THREADS / 64];
    __shared__ unsigned snv[GB_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
        }
        if (lane == 0) {
            slo[a][wave] = lo[a];
            shi[a][wave] = hi[a];
        }
    }
    for (int off = 32; off > 0; off >>= 1) nv += __shfl_down(nv, off, 64);
    if (lane == 0) snv[wave] = nv;
    __syncthreads();
    if (threadIdx.x == 0) {
        BBoxPartial p;
        p.nvalid = 0;
        for (int a = 0; a < 3; ++a) {
            p.lo[a] = __builtin_inff();
            p.hi[a] = -__builtin_inff();
        }
        for (int w = 0; w < GB_THREADS / 64; ++w) {
            for (int a = 0; a < 3; ++a) {
                p.lo[a] = fminf(p.lo[a], slo[a][w]);
                p.hi[a] = fmaxf(p.hi[a], shi[a][w]);
            }
            p.nvalid += snv[w];
        }
        partials[blockIdx.x] = p;
    }
}

// Histogram of one coordinate of the finite points over SLAB_BINS equal bins of [lo, lo + bins/inv):
// where to cut a cloud into slabs of equal point counts.
constexpr int SLAB_BINS = 1024;
__global__ __launch_bounds__(GB_THREADS) void axis_hist_kernel(const float *x, const float *y,
                                                               const float *z, uint32_t m, int axis,
                                                               float lo, float inv,
                                                               uint32_t *hist) {
    __shared__ uint32_t h[SLAB_BINS];
    for (int t = threadIdx.x; t < SLAB_BINS; t += GB_THREADS) h[t] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x; i < m; i += gridDim.x * GB_THREADS) {
        const float p[3] = {x[i], y[i], z[i]};
        if (finite3(p[0], p[1], p[2])) {
            float f = floorf((p[axis] - lo) * inv);
            f = fminf(fmaxf(f, 0.0f), (float)(SLAB_BINS - 1));
            atomicAdd(&h[(int)f], 1u);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < SLAB_BINS; t += GB_THREADS)
        if (h[t]) atomicAdd(&hist[t], h[t]);
}

__device__ __forceinline__ int cell_coord(float p, float o, float inv, int n) {
    float f = floorf((p - o) * inv);
    f = fminf(fmaxf(f, 0.0f), (float)(n - 1));
    return (int)f;
}

__device__ __forceinline__ uint32_t cell_index(const GridGeom &g, float x, float y, float z) {
    const int cx = cell_coord(x, g.ox, g.inv, g.nx);
    const int cy = cell_coord(y, g.oy, g.inv, g.ny);
    const int cz = cell_coord(z, g.oz, g.inv, g.nz);
    return ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx + (uint32_t)cx;
}

// Distance from the grid-relative point u to the nearest face behind which points may lie that a
// search of the cube of half-width R around cell c has not seen: a face of the cube inside the
// grid, or -- where the cube reaches the end of the grid -- the face of the index's box, when the
// index left the points of the cloud beyond that face out (an open face).  +inf when there is no
// such face: the cube then holds every point a neighbour could be.
__device__ __forceinline__ float cube_margin(const GridGeom &g, int cx, int cy, int cz, float ux,
                                             float uy, float uz, int R) {
    const float u[3] = {ux, uy, uz};
    const int c[3] = {cx, cy, cz};
    const int n[3] = {g.nx, g.ny, g.nz};
    float margin = __builtin_inff();
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        margin = fminf(margin, c[ax] - R > 0 ? u[ax] - (float)(c[ax] - R) * g.cell
                                             : u[ax] - g.open_lo[ax]);
        margin = fminf(margin, c[ax] + R < n[ax] - 1 ? (float)(c[ax] + R + 1) * g.cell - u[ax]
                                                     : g.open_hi[ax] - u[ax]);
    }
    return margin;
}
// Distance from u to the nearest open face of the index's box (+inf: the index holds the whole
// cloud): a ball that reaches farther cannot be searched in this index alone.
__device__ __forceinline__ float open_face_distance(const GridGeom &g, float ux, float uy, float uz) {
    const float u[3] = {ux, uy, uz};
    float d = __builtin_inff();
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) d = fminf(d, fminf(u[ax] - g.open_lo[ax], g.open_hi[ax] - u[ax]));
    return d;
}

// cell id per point (PCRS_NO_INDEX for non-finite points), the cell histogram and the number of
// non-empty cells (the point that finds a cell empty counts it)
__global__ __launch_bounds__(GB_THREADS) void cell_count_kernel(const float *x, const float *y,
                                                                const float *z, uint32_t m,
                                                                GridGeom g, BoxFilter box,
                                                                uint32_t *cell_id, uint32_t *counts,
                                                                uint32_t *nonempty) {
    unsigned opened = 0;
    for (uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x; i < m; i += gridDim.x * GB_THREADS) {
        const float px = x[i], py = y[i], pz = z[i];
        uint32_t c = PCRS_NO_INDEX;
        if (point_taken(box, px, py, pz)) {
            c = cell_index(g, px, py, pz);
            opened += (atomicAdd(&counts[c], 1u) == 0u) ? 1u : 0u;
        }
        cell_id[i] = c;
    }
    for (int off = 32; off > 0; off >>= 1) opened += __shfl_down(opened, off, 64);
    if ((threadIdx.x & 63) == 0 && opened) atomicAdd(nonempty, opened);
}

// Exclusive scan of counts[0..n) in three launches: per-block scan of 1024 elements, scan of the
// block totals, add-back.  counts[n] receives the grand total.
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_BLOCK = GB_THREADS * SCAN_ITEMS;

// Exclusive scan of a short array by ONE block (1024 threads, consecutive items per thread): the
// three-launch scan below costs three launch latencies whatever the length, and the tables of the
// sorts of a small cloud (a LiDAR frame after voxel down-sampling) have a few thousand entries.
constexpr uint32_t SCAN_SMALL_MAX = 32768;
__global__ __launch_bounds__(1024) void scan_small_kernel(uint32_t *data, uint32_t n,
                                                          uint32_t *total_out) {
    __shared__ uint32_t wsum[16];
    constexpr int PER_MAX = SCAN_SMALL_MAX / 1024;
    const uint32_t per = (n + 1023u) / 1024u;  // <= PER_MAX
    const uint32_t b = threadIdx.x * per;
    // a thread's items in registers, every load requested before the first is used (one block
    // has nothing else to hide a chain of dependent loads behind: 17 items took 17 round trips)
    uint32_t v[PER_MAX];
#pragma unroll
    for (int i = 0; i < PER_MAX; ++i) v[i] = ((uint32_t)i < per && b + i < n) ? data[b + i] : 0u;
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < PER_MAX; ++i) t += v[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = t;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - t;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < PER_MAX; ++i) {
        if ((uint32_t)i < per && b + i < n) data[b + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 1023 && total_out) *total_out = run;
}

__global__ __launch_bounds__(GB_THREADS) void scan_block_kernel(uint32_t *data, uint32_t n,
                                                                uint32_t *block_sums) {
    __shared__ uint32_t wsum[GB_THREADS / 64];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], t = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? data[base + i] : 0u;
        t += v[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = t;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wsum[w];
    uint32_t run = wave_off + incl - t;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == GB_THREADS - 1) block_sums[blockIdx.x] = wave_off + incl;
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t *block_sums, uint32_t nblocks,
                                                         uint32_t *total_out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < nblocks) ? block_sums[i] : 0u;
        uint32_t incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += wsum[w];
        const uint32_t carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(GB_THREADS) void scan_add_kernel(uint32_t *data, uint32_t n,
                                                              const uint32_t *block_sums) {
    const uint32_t off = block_sums[blockIdx.x];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) data[base + i] += off;
}

// counting-sort scatter: fill[] starts as a copy of cell_start and is advanced atomically
__global__ __launch_bounds__(GB_THREADS) void scatter_kernel(const float *x, const float *y,
                                                             const float *z, uint32_t m,
                                                             const uint32_t *cell_id,
                                                             uint32_t *fill, float4 *spts) {
    for (uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x; i < m; i += gridDim.x * GB_THREADS) {
        const uint32_t c = cell_id[i];
        if (c == PCRS_NO_INDEX) continue;
        const uint32_t s = atomicAdd(&fill[c], 1u);
        spts[s] = make_float4(x[i], y[i], z[i], __uint_as_float(i));
    }
}

// ---- two-level counting sort by cell (no global atomics)
// Level 1 splits the cloud into <= 8192 (16128 for the very largest grids) coarse buckets of
// 2^shift consecutive cells with per-block
// LDS histograms (bucket-major table of per-block counts -> exclusive scan -> per-block write
// cursors).  Level 2 gives every bucket to one block, which histograms its 2^shift cells in LDS,
// scans them into cell_start[] and places the points.  Both levels use LDS atomics only; the
// global-memory version above needs one device-scope atomic per point and pass.
struct SortPlan {
    GridGeom g;
    BoxFilter box;     // points outside it are left out, like the non-finite ones
    uint32_t shift;    // log2(cells per coarse bucket), <= FINE_BIG_SHIFT
    uint32_t nb;       // coarse buckets
    uint32_t nblk;     // level-1 blocks
    uint32_t per_blk;  // points per level-1 block
    uint32_t ncell;
};
constexpr uint32_t FINE_MAX_SHIFT = 12;
// Grids of more cells than 8192 buckets of 2^12 (a flat scene of a hundred million points: most
// of its layers of cells are empty) get buckets of up to 2^15 cells: 128 KB of counters, one
// block per CU, no staging array -- still far cheaper than the one-level sort's device-scope
// atomic per point and pass.
constexpr uint32_t FINE_BIG_SHIFT = 15;

__global__ __launch_bounds__(GB_THREADS) void coarse_hist_kernel(const float *x, const float *y,
                                                                 const float *z, uint32_t m,
                                                                 SortPlan sp, uint32_t *table,
                                                                 uint32_t *zero_word,
                                                                 uint32_t *zero_block) {
    // the first kernel of a sort clears what the later ones accumulate into: the non-empty-cell
    // counter and the sum of squared cell counts behind it (fine_sort_kernel) and, for an index,
    // the 32 stage counters of its searches
    if (blockIdx.x == 0) {
        if (zero_word && threadIdx.x < 4) zero_word[threadIdx.x] = 0u;  // count + sum of squares
        if (zero_block && threadIdx.x < 32) zero_block[threadIdx.x] = 0u;
    }
    extern __shared__ uint32_t hist[];
    for (uint32_t t = threadIdx.x; t < sp.nb; t += GB_THREADS) hist[t] = 0;
    __syncthreads();
    const uint32_t begin = blockIdx.x * sp.per_blk;
    const uint32_t end = min(m, begin + sp.per_blk);
    // four points per thread in flight: a block is a few waves working through thousands of points
    for (uint32_t i0 = begin + threadIdx.x; i0 < end; i0 += 4 * GB_THREADS) {
        float px[4], py[4], pz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = i0 + j * GB_THREADS;
            px[j] = py[j] = pz[j] = __builtin_nanf("");
            if (i < end) {
                px[j] = x[i];
                py[j] = y[i];
                pz[j] = z[i];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (point_taken(sp.box, px[j], py[j], pz[j]))
                atomicAdd(&hist[cell_index(sp.g, px[j], py[j], pz[j]) >> sp.shift], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < sp.nb; t += GB_THREADS)
        table[(size_t)t * sp.nblk + blockIdx.x] = hist[t];
}

__global__ __launch_bounds__(GB_THREADS) void coarse_scatter_kernel(const float *x, const float *y,
                                                                    const float *z, uint32_t m,
                                                                    SortPlan sp,
                                                                    const uint32_t *table,
                                                                    float4 *tmp) {
    extern __shared__ uint32_t cursor[];
    for (uint32_t t = threadIdx.x; t < sp.nb; t += GB_THREADS)
        cursor[t] = table[(size_t)t * sp.nblk + blockIdx.x];
    __syncthreads();
    const uint32_t begin = blockIdx.x * sp.per_blk;
    const uint32_t end = min(m, begin + sp.per_blk);
    for (uint32_t i0 = begin + threadIdx.x; i0 < end; i0 += 4 * GB_THREADS) {
        float px[4], py[4], pz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = i0 + j * GB_THREADS;
            px[j] = py[j] = pz[j] = __builtin_nanf("");
            if (i < end) {
                px[j] = x[i];
                py[j] = y[i];
                pz[j] = z[i];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (point_taken(sp.box, px[j], py[j], pz[j])) {
                const uint32_t at =
                    atomicAdd(&cursor[cell_index(sp.g, px[j], py[j], pz[j]) >> sp.shift], 1u);
                tmp[at] = make_float4(px[j], py[j], pz[j], __uint_as_float(i0 + j * GB_THREADS));
            }
        }
    }
}

template <uint32_t MAXSHIFT, uint32_t T>
__global__ __launch_bounds__(T) void fine_sort_kernel(const float4 *tmp, SortPlan sp,
                                                               const uint32_t *table,
                                                               uint32_t *cell_start, float4 *spts,
                                                               float *spts3, uint32_t *nonempty) {
    // spts3 (nullable): the same points without their index, 12 bytes apart (what the
    // lane-per-query search kernel reads)
    constexpr bool STAGED = MAXSHIFT <= FINE_MAX_SHIFT;
    __shared__ uint32_t cnt[1u << MAXSHIFT];
    __shared__ uint32_t wsum[T / 64];
    const uint32_t b = blockIdx.x;
    const uint32_t cpb = 1u << sp.shift;
    const uint32_t cell0 = b << sp.shift;
    const uint32_t ncl = min(cpb, sp.ncell - cell0);
    const uint32_t begin = table[(size_t)b * sp.nblk];
    const uint32_t end = table[(size_t)(b + 1) * sp.nblk];  // the scan's total closes the table
    for (uint32_t t = threadIdx.x; t < cpb; t += T) cnt[t] = 0;
    // the first FINE_HELD points of every thread stay in registers between the counting and the
    // placing pass (all their loads are in flight together); longer buckets are read twice
    constexpr int FINE_HELD = 8;
    float4 held[FINE_HELD];
    uint32_t held_cell[FINE_HELD];
#pragma unroll
    for (int h = 0; h < FINE_HELD; ++h) {
        const uint32_t i = begin + threadIdx.x + (uint32_t)h * T;
        if (i < end) held[h] = tmp[i];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < FINE_HELD; ++h) {
        const uint32_t i = begin + threadIdx.x + (uint32_t)h * T;
        if (i < end) {
            held_cell[h] = cell_index(sp.g, held[h].x, held[h].y, held[h].z) - cell0;
            atomicAdd(&cnt[held_cell[h]], 1u);
        }
    }
    for (uint32_t i = begin + threadIdx.x + FINE_HELD * T; i < end; i += T) {
        const float4 p = tmp[i];
        atomicAdd(&cnt[cell_index(sp.g, p.x, p.y, p.z) - cell0], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0..cpb): each thread owns cpb/256 consecutive cells
    const uint32_t ipt = (cpb + T - 1) / T;
    const uint32_t f0 = threadIdx.x * ipt;
    uint32_t local = 0, opened = 0;
    unsigned long long squares = 0;  // sum of squared cell counts = points x occupancy seen by a point
    for (uint32_t j = 0; j < ipt; ++j) {
        const uint32_t f = f0 + j;
        if (f < cpb) {
            local += cnt[f];
            opened += cnt[f] ? 1u : 0u;
            squares += (unsigned long long)cnt[f] * cnt[f];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = local;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    __shared__ uint32_t wopen[T / 64];
    __shared__ unsigned long long wsq[T / 64];
    if (lane == 63) wsum[wave] = incl;
    for (int off = 32; off > 0; off >>= 1) {
        opened += __shfl_down(opened, off, 64);
        squares += __shfl_down(squares, off, 64);
    }
    if (lane == 0) {
        wopen[wave] = opened;
        wsq[wave] = squares;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic each per block; nonempty + 2 is 8-byte aligned
        uint32_t o = 0;
        unsigned long long sq = 0;
        for (uint32_t w = 0; w < T / 64; ++w) {
            o += wopen[w];
            sq += wsq[w];
        }
        if (o) {
            atomicAdd(nonempty, o);
            atomicAdd(reinterpret_cast<unsigned long long *>(nonempty + 2), sq);
        }
    }
    uint32_t run = begin + incl - local;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (uint32_t j = 0; j < ipt; ++j) {
        const uint32_t f = f0 + j;
        if (f < cpb) {
            const uint32_t c = cnt[f];
            cnt[f] = run;  // becomes the write cursor of the cell
            run += c;
        }
    }
    if (b == sp.nb - 1 && threadIdx.x == 0) cell_start[sp.ncell] = end;
    __syncthreads();
    // the cell starts leave through consecutive lanes (each thread scanned consecutive cells)
    for (uint32_t t = threadIdx.x; t < ncl; t += T) cell_start[cell0 + t] = cnt[t];
    __syncthreads();
    // A bucket that fits the staging array is put in order in LDS and leaves as one linear copy:
    // thousands of blocks each scattering 16-byte points over their own window of the output
    // overflow the L2 once the windows together exceed it, and every partial line costs a
    // read-modify-write.
    __shared__ float4 stage[STAGED ? FINE_HELD * T : 1];
    if (STAGED && end - begin <= FINE_HELD * T) {
#pragma unroll
        for (int h = 0; h < FINE_HELD; ++h) {
            const uint32_t i = begin + threadIdx.x + (uint32_t)h * T;
            if (i < end) stage[atomicAdd(&cnt[held_cell[h]], 1u) - begin] = held[h];
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < end - begin; t += T) spts[begin + t] = stage[t];
        if (spts3) {
            const float *sf = reinterpret_cast<const float *>(stage);
            float *o3 = spts3 + (size_t)begin * 3;
            for (uint32_t t = threadIdx.x; t < 3u * (end - begin); t += T)
                o3[t] = sf[(t / 3u) * 4u + t % 3u];
        }
        return;
    }
    auto put = [&](uint32_t at, const float4 &p) {
        spts[at] = p;
        if (spts3) {
            spts3[(size_t)at * 3] = p.x;
            spts3[(size_t)at * 3 + 1] = p.y;
            spts3[(size_t)at * 3 + 2] = p.z;
        }
    };
#pragma unroll
    for (int h = 0; h < FINE_HELD; ++h) {
        const uint32_t i = begin + threadIdx.x + (uint32_t)h * T;
        if (i < end) put(atomicAdd(&cnt[held_cell[h]], 1u), held[h]);
    }
    for (uint32_t i = begin + threadIdx.x + FINE_HELD * T; i < end; i += T) {
        const float4 p = tmp[i];
        put(atomicAdd(&cnt[cell_index(sp.g, p.x, p.y, p.z) - cell0], 1u), p);
    }
}

// ----------------------------------------------------------------------------- query kernel

// bit 31 of a retry-list entry: "not tried yet, just too crowded for one lane"
constexpr uint32_t PCRS_HEAVY_FLAG = 0x80000000u;

// Which sorted slots are queries: original index in [q_begin, q_begin + q_len) and, with boxes,
// inside `in` but inside none of the n_ex boxes `ex` (a slab of a sharded cloud; a level of a
// refined index without the parts its finer grids answer).
// handed (nullable): a byte per point of the columns; a point whose byte equals `level` is a query
// too, whatever the boxes say (a level of a refined index: what the level before it passed on).
// no_own: only those.
struct SlotChoice {
    uint32_t q_begin, q_len;
    bool boxes;
    BoxFilter in;
    BoxFilter ex[PCRS_MAX_KIDS];
    int n_ex;
    const uint8_t *handed;
    uint8_t level;
    bool no_own;
    // (nullable) the same bytes, writable: a pass that only takes what it is handed and runs
    // before everybody has marked their queries strikes out (0xFF, no level's number) what it
    // takes, so that the pass after the late-comers does not take it again
    uint8_t *consume;
};
// The test itself, for one sorted slot.
__device__ __forceinline__ bool slot_is_query(const SlotChoice &ch, const float4 &p) {
    const uint32_t orig = __float_as_uint(p.w);
    bool own = (orig - ch.q_begin) < ch.q_len && !ch.no_own;
    if (ch.boxes) {
        own = own && point_taken(ch.in, p.x, p.y, p.z);
        for (int j = 0; j < ch.n_ex; ++j) own = own && !point_taken(ch.ex[j], p.x, p.y, p.z);
    }
    if (ch.handed) {
        // a grid with finer grids under it is only handed points of their boxes: a pass that
        // takes nothing else need not look up the byte of any other point
        bool maybe = true;
        if (ch.no_own && ch.boxes && ch.n_ex > 0) {
            maybe = false;
            for (int j = 0; j < ch.n_ex; ++j) maybe = maybe || point_taken(ch.ex[j], p.x, p.y, p.z);
        }
        if (maybe) own = own || ch.handed[orig] == ch.level;
    }
    return own;
}

struct CellArgs {
    GridGeom g;
    const uint32_t *cell_start;
    const float4 *spts;
    const float *spts3;         // x, y, z of the same slots, 12 bytes apart (lane-per-query kernel)
    uint32_t nvalid;
    const float *rx, *ry, *rz;  // original columns (consumer gathers)
    // query addressing:
    //   qmode 0: position p is sorted slot p (p < nq); query = spts[p], output slot = idx - q_begin
    //   qmode 1: position p is query p of (qx, qy, qz)
    //   qmode 2: query qlist[p] of (qx, qy, qz), p < *qcount
    //   qmode 3: sorted slot qlist[p], p < *qcount (self queries of a sub-range, compacted)
    //   qmode 4: entry p of qpts (external queries sorted by grid cell: x, y, z, query index),
    //            p < *qcount
    //   qmode 5: like 0, but only the slots that pass *choice (device memory) are queries: a level
    //            of a refined index that answers most of its points needs no compacted list
    int qmode;
    const SlotChoice *choice;
    const float4 *qpts;
    const float *qx, *qy, *qz;
    uint32_t nq;
    uint32_t q_begin, q_len;
    const uint32_t *qlist;
    const uint32_t *qcount;
    int R;                  // cube half-width in cells (per-lane kernels)
    int wave_radii[5];      // cube half-widths the wave-per-query kernel starts from; entry 0
                            // is only for queries flagged PCRS_HEAVY_FLAG in the list
    int n_radii;            // entries of wave_radii in use
    float useful_d2;        // 1-NN only: neighbours beyond this squared distance are of no use
                            // to the caller (+inf = no such bound)
    uint32_t heavy_only;    // wave-per-query kernel: answer only the entries flagged PCRS_HEAVY_FLAG
                            // and pass the others on to fail_list untouched (a finer level of a
                            // refined index: what its cubes cannot certify is too sparse for its
                            // cells and is better served by the next coarser level's lanes)
    uint32_t heavy;         // per-lane kernels hand cubes with more points than this to a wave
    // per-lane kernels: a query they cannot certify whose cube held fewer points than
    // sparse_below (0: never) sits in a sparse spot of the cloud -- its neighbourhood spans many of
    // this grid's cells and few of a coarser grid's --
    // and is marked for the lanes of the next coarser grid (hand_flags[query] = hand_level, a byte
    // per point of the cloud) instead of joining fail_list and this grid's wave-per-query kernel.
    // The wave-per-query kernel marks what it cannot certify the same way when hand_flags is set.
    uint32_t sparse_below;
    uint8_t *hand_flags;
    uint8_t hand_level;
    uint32_t *fail_list;    // queries that could not be certified (as indices into qx/qy/qz)
    uint32_t *fail_count;
    float *fail_bound;      // lane-per-query kernels, beside fail_list (nullable): an upper bound of
                            // the query's kcert-th squared distance (+inf = none) -- the kcert-th
                            // distance among the points the kernel did see
    const float *qbound;    // wave-per-query kernel, beside qlist (nullable): those bounds
    uint32_t *alt_list;     // one-walk kernel: queries it hands to the two-walk kernel (a cube of
    uint32_t *alt_count;    // too many candidates for its tags, a run of tied distances)
    uint32_t *stage1_fail;  // 1-NN kernel (it retries in place): counts what its first stage left
    uint32_t kcert;         // neighbours that must be certified (<= K)
    uint32_t kcollect;      // neighbours whose identity is needed (pass 2), 0 = none
    uint32_t *zero_me;      // first-stage kernels: 16 words to clear (the next search's counters)
    const uint32_t *skip;   // nullable device word: non-zero makes the launch a no-op
    // 1-NN kernel, qmode 1: a rigid increment applied to each query as it is loaded and written
    // back through wx/wy/wz (see KnnQuery::xform); null = queries are used as they are
    const IcpLoopState *xform;
    float *wx, *wy, *wz;
};

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Queries come in
// cell order, so consecutive blocks read neighbouring rows of the sorted array: this bijection
// hands every XCD one contiguous eighth of the blocks, and the rows a block shares with the
// blocks one row or one layer of cells away are found in its own XCD's L2.
__device__ __forceinline__ uint32_t xcd_contiguous_block(uint32_t b, uint32_t nblocks) {
    const uint32_t q = nblocks / 8u, r = nblocks % 8u, x = b % 8u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + b / 8u;
}

// Calls f(point, address of the point in the sorted array) for every point of the (2R+1)^3 cube
// of cells centred on (cx, cy, cz).
template <class F>
__device__ __forceinline__ void walk_cube(const CellArgs &a, int cx, int cy, int cz, F f) {
    const GridGeom &g = a.g;
    const int R = a.R;
    const int x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
    for (int dz = -R; dz <= R; ++dz) {
        const int zz = cz + dz;
        if (zz < 0 || zz >= g.nz) continue;
        for (int dy = -R; dy <= R; ++dy) {
            const int yy = cy + dy;
            if (yy < 0 || yy >= g.ny) continue;
            const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
            const uint32_t s0 = a.cell_start[row + x0];
            const uint32_t s1 = a.cell_start[row + x1 + 1];
            // four loads in flight per iteration, addressed off one running pointer so that
            // the loads use immediate offsets instead of four 64-bit address computations
            const float4 *pp = a.spts + s0;
            const float4 *const pe = a.spts + s1;
            for (; pp + 3 < pe; pp += 4) {
                const float4 p0 = pp[0];
                const float4 p1 = pp[1];
                const float4 p2 = pp[2];
                const float4 p3 = pp[3];
                f(p0, pp);
                f(p1, pp + 1);
                f(p2, pp + 2);
                f(p3, pp + 3);
            }
            for (; pp < pe; ++pp) f(*pp, pp);
        }
    }
}

// The same walk over the 27-cell cube (R = 1) reading 12-byte points, four at a time:
// f4(d2[4], first slot) for every full group of four consecutive points of a row, f1(point, slot,
// live) for the one to three that are left (a call with live == false carries no point and must
// change nothing).  The search is bound by vector-instruction issue, then by what its per-lane
// candidate loads move through the L1 (TA busy 75 % of the kernel): a candidate without its index
// is a quarter smaller, and four of them are three 16-byte loads whose dwords pair up for packed
// arithmetic as they are (sqdist_pk4).
struct pcrs_p3 {
    float x, y, z;
};
struct EveryRow {
    __device__ __forceinline__ bool operator()(uint32_t) const { return true; }
};
// row(points in the row) is asked before each row; false ends the walk there.
template <int R = 1, class F4, class F1, class FR = EveryRow>
__device__ __forceinline__ void walk_cube3(const CellArgs &a, int cx, int cy, int cz,
                                           const pcrs_q3 &q, F4 f4, F1 f1, FR row_ok = FR()) {
    const GridGeom &g = a.g;
    const int x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
    const pcrs_p3 *base = reinterpret_cast<const pcrs_p3 *>(a.spts3);
    for (int dz = -R; dz <= R; ++dz) {
        const int zz = cz + dz;
        if (zz < 0 || zz >= g.nz) continue;
        for (int dy = -R; dy <= R; ++dy) {
            const int yy = cy + dy;
            if (yy < 0 || yy >= g.ny) continue;
            const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
            const uint32_t s0 = a.cell_start[row + x0];
            const uint32_t s1 = a.cell_start[row + x1 + 1];
            if (!row_ok(s1 - s0)) return;
            const pcrs_p3 *pp = base + s0;
            const pcrs_p3 *const pe = base + s1;
            uint32_t slot = s0;  // counted beside the pointer: (pp - base) is a division by 12
            for (; pp + 3 < pe; pp += 4, slot += 4) {
                const pcrs_f4u *v = reinterpret_cast<const pcrs_f4u *>(pp);
                const pcrs_f4u va = v[0], vb = v[1], vc = v[2];
                float d2[4];
                sqdist_pk4(q, va, vb, vc, d2);
                f4(d2, slot);
            }
            // The one to three points that are left, in a single step with the missing ones
            // switched off (their loads repeat the first point, so they stay inside the row): some
            // lane of the wave nearly always has three left, and three trips through a
            // one-point loop cost half as much again.
            const uint32_t rem = (uint32_t)(pe - pp);
            if (rem) {
                const pcrs_p3 p0 = pp[0];
                const pcrs_p3 p1 = pp[rem > 1u ? 1 : 0];
                const pcrs_p3 p2 = pp[rem > 2u ? 2 : 0];
                f1(p0, slot, true);
                f1(p1, slot + 1, rem > 1u);
                f1(p2, slot + 2, rem > 2u);
            }
        }
    }
}

#ifndef PCRS_WPE
#define PCRS_WPE 5
#endif
// Waves per SIMD the register allocator must leave room for: the walk is latency-bound, so
// occupancy is worth more than a few spilled or rematerialised values.
constexpr int cell_min_waves(int k, bool collect) {
    return k <= 17 ? PCRS_WPE : 1;
}

template <int K, bool COLLECT>
__global__ __launch_bounds__(GB_THREADS) __attribute__((amdgpu_waves_per_eu(cell_min_waves(K, COLLECT))))
void knn_cell_kernel(CellArgs a, KnnOut out) {
    // per-lane neighbour list for pass 2; ties at the K-th distance beyond it retry per wave
    constexpr int CAP = K + 3 < 64 ? K + 3 : 64;
    __shared__ uint32_t lst[COLLECT ? CAP : 1][GB_THREADS];  // slots of the hits, walk order
    __shared__ uint8_t ord[COLLECT ? CAP : 1][GB_THREADS];   // ord[r] = list position of rank r

    if (a.zero_me && blockIdx.x == 0 && threadIdx.x < 16) a.zero_me[threadIdx.x] = 0u;
    if (a.skip && *a.skip) return;
    uint32_t total = a.nq;
    if (a.qmode >= 2 && a.qmode != 5) total = *a.qcount;
    const GridGeom &g = a.g;

    for (uint32_t pos = xcd_contiguous_block(blockIdx.x, gridDim.x) * GB_THREADS + threadIdx.x;
         pos < total; pos += gridDim.x * GB_THREADS) {
        uint32_t qi;
        float qx, qy, qz;
        if (a.qmode == 0 || a.qmode == 3 || a.qmode == 5) {
            const float4 p = a.spts[a.qmode == 3 ? a.qlist[pos] : pos];
            const uint32_t orig = __float_as_uint(p.w);
            if (orig - a.q_begin >= a.q_len) continue;  // not a query of this call
            if (a.qmode == 5 && !slot_is_query(*a.choice, p)) continue;
            qi = orig - a.q_begin;
            qx = p.x;
            qy = p.y;
            qz = p.z;
        } else if (a.qmode == 4) {
            const float4 p = a.qpts[pos];
            qi = __float_as_uint(p.w);
            qx = p.x;
            qy = p.y;
            qz = p.z;
        } else {
            qi = (a.qmode == 2) ? a.qlist[pos] : pos;
            if (a.qmode == 2 && (qi & PCRS_HEAVY_FLAG) && a.heavy != 0xFFFFFFFFu) {
                // a cube an earlier stage found too crowded for a lane: on to the wave kernel
                if (a.fail_list) {
                    const uint32_t f = atomicAdd(a.fail_count, 1u);
                    a.fail_list[f] = qi;
                    if (a.fail_bound) a.fail_bound[f] = __builtin_inff();
                }
                continue;
            }
            qx = a.qx[qi];
            qy = a.qy[qi];
            qz = a.qz[qi];
        }
        const bool qvalid = finite3(qx, qy, qz);
        if (!qvalid) {  // no neighbours: the consumers write their defaults
            uint64_t none[1] = {PCRS_EMPTY_KEY};
            consume<true>(RegNbrs<1>(none, a.rx, a.ry, a.rz), 0u, false, (size_t)qi, qx, qy, qz,
                          out);
            continue;
        }
        const float ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
        const int cx = cell_coord(qx, g.ox, g.inv, g.nx);
        const int cy = cell_coord(qy, g.oy, g.inv, g.ny);
        const int cz = cell_coord(qz, g.oz, g.inv, g.nz);

        // ---- pass 1: K smallest squared distances, ascending, branch-free
        float D[K];
#pragma unroll
        for (int i = 0; i < K; ++i) D[i] = __builtin_inff();
        const pcrs_f2 qxy = {qx, qy};
        const pcrs_q3 q3 = make_q3(qx, qy, qz);
        auto insert = [&](float d2) {
#pragma unroll
            for (int s = K - 1; s > 0; --s) D[s] = __builtin_amdgcn_fmed3f(d2, D[s - 1], D[s]);
            D[0] = fminf(D[0], d2);
        };
        // A cube of thousands of points is no work for a single lane (a blob of a hundred
        // thousand points in a few cells would keep every lane of its waves walking all of it):
        // the nine row ranges tell before anything is walked.  Such a query first looks at the
        // start of its own cell: where the cloud repeats one point over and over (returns a sensor
        // parks at the origin, padding) the list is full of zero distances after a handful of
        // candidates, and a list of zeros cannot get any better -- the query is answered here and
        // now.  Otherwise it goes, flagged, to the wave-per-query kernel, which samples crowded
        // regions instead of walking them.
        uint32_t probe0 = 0, probe1 = 0;  // the slots the probe looked at (probe1 > probe0: answered by it)
        uint32_t cube_pts = 0;
        {
            constexpr int RR = 1;  // cube half-width of this kernel
            const int x0 = max(cx - RR, 0), x1 = min(cx + RR, g.nx - 1);
            constexpr int SIDE = 2 * RR + 1;
#pragma unroll
            for (int i = 0; i < SIDE * SIDE; ++i) {
                const int yy = cy + i % SIDE - RR, zz = cz + i / SIDE - RR;
                if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                    const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
                    cube_pts += a.cell_start[row + x1 + 1] - a.cell_start[row + x0];
                }
            }
            if (cube_pts > a.heavy) {
                const uint32_t own = ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx + (uint32_t)cx;
                const uint32_t c0 = a.cell_start[own], c1 = a.cell_start[own + 1];
                const uint32_t pe = c1 - c0 > 64u ? c0 + 64u : c1;
                const pcrs_p3 *base = reinterpret_cast<const pcrs_p3 *>(a.spts3);
                for (uint32_t sl = c0; sl < pe; ++sl) {
                    const pcrs_p3 p = base[sl];
                    insert(sqdist(qx, qy, qz, p.x, p.y, p.z));
                }
                float t0 = __builtin_inff();
#pragma unroll
                for (int i = 0; i < K; ++i)
                    if ((uint32_t)i == a.kcert - 1) t0 = D[i];
                // (neighbour lists name their neighbours: which of the repeated points those are
                // is the wave kernel's to find out)
                const bool lists = COLLECT && (out.idx || out.dist || out.count);
                if (t0 == 0.0f && !lists) {
                    probe0 = c0;
                    probe1 = pe;
                } else {
                    if (a.fail_list) {
                        const uint32_t f = atomicAdd(a.fail_count, 1u);
                        a.fail_list[f] = qi | PCRS_HEAVY_FLAG;
                        if (a.fail_bound) a.fail_bound[f] = __builtin_inff();
                    }
                    continue;
                }
            }
        }
        const bool probed = probe1 > probe0;
        if (!probed)
            walk_cube3(
                a, cx, cy, cz, q3,
                [&](const float(&d2)[4], uint32_t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) insert(d2[j]);
                },
                [&](const pcrs_p3 &p, uint32_t, bool live) {
                    const float d2 = sqdist_pk(qxy, qz, make_float4(p.x, p.y, p.z, 0.0f));
                    insert(live ? d2 : __builtin_inff());  // +inf passes through without a trace
                });

        // ---- certification: every point outside the cube is farther than `margin`
        constexpr int R = 1;
        // (a list of zero distances is final: a point at distance zero lies where the query lies,
        // in the query's own cell)
        const float margin = cube_margin(g, cx, cy, cz, ux, uy, uz, R);
        const bool whole = probed || !(margin < __builtin_inff());
        bool certified = whole;
        float tau_cert = __builtin_inff();
#pragma unroll
        for (int i = 0; i < K; ++i)
            if ((uint32_t)i == a.kcert - 1) tau_cert = D[i];
        if (!whole) {
            // slack covers the f32 rounding of the cell assignment and of the offsets above
            const float amax = fmaxf(fmaxf(fabsf(ux), fabsf(uy)), fabsf(uz));
            const float m = (margin - (g.slack + 2e-6f * amax)) * (1.0f - 2e-6f);
            certified = (m > 0.0f) && (tau_cert < (m * m) * (1.0f - 2e-6f));
        }
        if (!certified) {
            if (cube_pts < a.sparse_below) {
                a.hand_flags[qi] = a.hand_level;
            } else if (a.fail_list) {
                const uint32_t f = atomicAdd(a.fail_count, 1u);
                a.fail_list[f] = qi;
                if (a.fail_bound) a.fail_bound[f] = tau_cert;
            }
            continue;
        }

        // ---- SOR mean straight from the sorted distances (statistical_outlier.rs:28-37)
        KnnOut o2 = out;
        if (out.mean) {
            const uint32_t kk = out.k_sor + 1u;
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) c += ((uint32_t)i < kk && D[i] < __builtin_inff()) ? 1u : 0u;
            const uint32_t first = c > 1u ? 1u : 0u;
            float mean = __builtin_inff();
            if (c - first > 0u) {
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < K; ++i)
                    if ((uint32_t)i >= first && (uint32_t)i < c) sum += sqrtf(D[i]);
                mean = sum / (float)(c - first);
            }
            store_mean(o2, qi, mean);
        }

        if constexpr (COLLECT) {
            // ---- pass 2: identities of the kcollect nearest.  The second walk only appends
            // the slots of the points within the kcollect-th distance to a per-lane LDS list (a
            // short body: nearly every step of the walk has a hit in some lane of the wave).
            float tau = __builtin_inff();
#pragma unroll
            for (int i = 0; i < K; ++i)
                if ((uint32_t)i == a.kcollect - 1) tau = D[i];
            uint32_t hits = 0;
            auto take = [&](float d2, uint32_t slot) {
                if (d2 <= tau) {
                    if (hits < (uint32_t)CAP) lst[hits][threadIdx.x] = slot;
                    ++hits;
                }
            };
            if (probed) {
                // the neighbours all lie where the query lies, so the covariance is the same
                // whichever of them are taken: the first kcollect the probe met
                const pcrs_p3 *base = reinterpret_cast<const pcrs_p3 *>(a.spts3);
                for (uint32_t sl = probe0; sl < probe1 && hits < a.kcollect; ++sl) {
                    const pcrs_p3 p = base[sl];
                    take(sqdist(qx, qy, qz, p.x, p.y, p.z), sl);
                }
            } else
            walk_cube3(
                a, cx, cy, cz, q3,
                [&](const float(&d2)[4], uint32_t slot) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) take(d2[j], slot + j);
                },
                [&](const pcrs_p3 &p, uint32_t slot, bool live) {
                    const float d2 = sqdist_pk(qxy, qz, make_float4(p.x, p.y, p.z, 0.0f));
                    if (live) take(d2, slot);
                });
            if (hits > (uint32_t)CAP) {  // too many ties at the K-th distance for the list
                if (a.fail_list) {
                    const uint32_t f = atomicAdd(a.fail_count, 1u);
                    a.fail_list[f] = qi;
                    if (a.fail_bound) a.fail_bound[f] = tau_cert;
                }
                continue;
            }
            // A hit's place in the neighbour order is the number of pass-1 distances below its
            // own: ord[place] = position in the list, no sorting network and no key registers.
            // Hits at equal distance get the same place; a bitmap of taken places moves the later
            // one up (the places of a tie group are consecutive and used by nobody else), and an
            // insertion pass over ord[] then puts tie groups into index order.
            uint64_t used = 0;
            bool ties = false;
            const pcrs_p3 *p3 = reinterpret_cast<const pcrs_p3 *>(a.spts3);
            auto place_hit = [&](uint32_t j, const pcrs_p3 &p) {
                const float d2 = sqdist(qx, qy, qz, p.x, p.y, p.z);
                uint32_t rank = 0;
#pragma unroll
                for (int i = 0; i < K; ++i) rank += (D[i] < d2) ? 1u : 0u;
                const uint32_t place = rank + (uint32_t)__builtin_ctzll(~(used >> rank));
                ties |= place != rank;
                used |= 1ull << place;
                ord[place][threadIdx.x] = (uint8_t)j;
            };
            {
                uint32_t j = 0;
                for (; j + 3 < hits; j += 4) {
                    const pcrs_p3 h0 = p3[lst[j][threadIdx.x]];
                    const pcrs_p3 h1 = p3[lst[j + 1][threadIdx.x]];
                    const pcrs_p3 h2 = p3[lst[j + 2][threadIdx.x]];
                    const pcrs_p3 h3 = p3[lst[j + 3][threadIdx.x]];
                    place_hit(j, h0);
                    place_hit(j + 1, h1);
                    place_hit(j + 2, h2);
                    place_hit(j + 3, h3);
                }
                for (; j < hits; ++j) place_hit(j, p3[lst[j][threadIdx.x]]);
            }
            if (ties) {
                auto key_at = [&](uint32_t o) {
                    const float4 p = a.spts[lst[o][threadIdx.x]];
                    return make_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
                };
                for (uint32_t j = 1; j < hits; ++j) {
                    const uint32_t oj = ord[j][threadIdx.x];
                    const uint64_t kj = key_at(oj);
                    uint32_t i = j;
                    while (i > 0) {
                        const uint32_t oi = ord[i - 1][threadIdx.x];
                        if (key_at(oi) <= kj) break;
                        ord[i][threadIdx.x] = (uint8_t)oi;
                        --i;
                    }
                    ord[i][threadIdx.x] = (uint8_t)oj;
                }
            }
            o2.mean = nullptr;  // already found from D[]
            const uint32_t have = hits < a.kcollect ? hits : a.kcollect;
            consume<true>(LdsNbrs<CAP, GB_THREADS>(lst, ord, a.spts, qx, qy, qz), have, true,
                          (size_t)qi, qx, qy, qz, o2);
        }
    }
}

// ------------------------------------------------------- query kernel, one walk with tagged keys
//
// knn_cell_kernel walks the cube twice when neighbour identities are needed: the insertion chain
// of the first walk carries distances only, the second walk finds whom they belong to.  Here the
// chain carries 32-bit keys made of the distance's upper 22 bits and, in the low 10, the
// candidate's ordinal in the walk, so one walk yields the K nearest and who they are: no second
// walk (a fifth of the vector instructions and two fifths of the per-lane loads of the two-walk
// kernel), no lists in LDS.  The ordinals become sorted slots by going over the nine row ranges
// once more (cell_start only).
// Truncating the distance can misorder two neighbours whose distances agree in the upper 22 bits
// (about one pair in 300): such a pair is put right by comparing the full (distance, index) keys
// of the two; a run of three or more, or a pair across the K-th place -- where a candidate the
// chain has already dropped could belong to the answer -- sends the query to the retry list, as
// does a cube of more than 1024 candidates.  The SOR mean and the covariance are formed from
// the neighbours' own coordinates afterwards, so their values do not depend on the truncation.
constexpr uint32_t TAG_MASK = 1023u;
constexpr uint32_t TAG_INF = 0x7F800000u;  // keys at or above +inf's pattern are "no neighbour"

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Neighbours by sorted slot, in registers; keys are recomputed from the point.
template <int K>
struct SlotRegNbrs {
    const uint32_t (&slot)[K];
    const float4 *spts;
    float qx, qy, qz;
    __device__ __forceinline__ SlotRegNbrs(const uint32_t (&s)[K], const float4 *sp, float x,
                                           float y, float z)
        : slot(s), spts(sp), qx(x), qy(y), qz(z) {}
    __device__ __forceinline__ uint64_t key_of(const float4 &p) const {
        return make_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
    }
    template <class F>
    __device__ __forceinline__ void for_each(uint32_t c, F f) const {
        if (c == 0u) return;
#pragma unroll
        for (int r = 0; r < K; r += 4) {
            uint32_t s[4];
            float4 p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)  // ranks past the end repeat the first (a valid slot)
                if (r + i < K) s[i] = (uint32_t)(r + i) < c ? slot[r + i] : slot[0];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (r + i < K) p[i] = spts[s[i]];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (r + i < K)
                    if ((uint32_t)(r + i) < c) f((uint32_t)(r + i), key_of(p[i]), s[i]);
        }
    }
    __device__ __forceinline__ void fetch(uint64_t, uint32_t sl, float &x, float &y, float &z) const {
        const float4 p = spts[sl];
        x = p.x;
        y = p.y;
        z = p.z;
    }
};

#ifndef PCRS_TAG_WPE
#define PCRS_TAG_WPE 5
#endif
template <int K>
__global__ __launch_bounds__(GB_THREADS) __attribute__((amdgpu_waves_per_eu(K <= 17 ? PCRS_TAG_WPE : 1)))
void knn_tag_kernel(CellArgs a, KnnOut out) {
    constexpr int KC = K + 2;  // two more than asked for: the neighbours behind the last place
    if (a.zero_me && blockIdx.x == 0 && threadIdx.x < 16) a.zero_me[threadIdx.x] = 0u;
    if (a.skip && *a.skip) return;
    uint32_t total = a.nq;
    if (a.qmode >= 2 && a.qmode != 5) total = *a.qcount;
    const GridGeom &g = a.g;
    const uint32_t keff = a.kcert;

    for (uint32_t pos = xcd_contiguous_block(blockIdx.x, gridDim.x) * GB_THREADS + threadIdx.x;
         pos < total; pos += gridDim.x * GB_THREADS) {
        uint32_t qi;
        float qx, qy, qz;
        if (a.qmode == 0 || a.qmode == 3 || a.qmode == 5) {
            const float4 p = a.spts[a.qmode == 3 ? a.qlist[pos] : pos];
            const uint32_t orig = __float_as_uint(p.w);
            if (orig - a.q_begin >= a.q_len) continue;  // not a query of this call
            if (a.qmode == 5 && !slot_is_query(*a.choice, p)) continue;
            qi = orig - a.q_begin;
            qx = p.x;
            qy = p.y;
            qz = p.z;
        } else if (a.qmode == 4) {
            const float4 p = a.qpts[pos];
            qi = __float_as_uint(p.w);
            qx = p.x;
            qy = p.y;
            qz = p.z;
        } else {
            qi = (a.qmode == 2) ? a.qlist[pos] : pos;
            qx = a.qx[qi];
            qy = a.qy[qi];
            qz = a.qz[qi];
        }
        if (!finite3(qx, qy, qz)) {  // no neighbours: the consumers write their defaults
            uint64_t none[1] = {PCRS_EMPTY_KEY};
            consume<true>(RegNbrs<1>(none, a.rx, a.ry, a.rz), 0u, false, (size_t)qi, qx, qy, qz,
                          out);
            continue;
        }
        const float ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
        const int cx = cell_coord(qx, g.ox, g.inv, g.nx);
        const int cy = cell_coord(qy, g.oy, g.inv, g.ny);
        const int cz = cell_coord(qz, g.oz, g.inv, g.nz);

        // ---- the walk: K + 2 smallest tagged keys, ascending, branch-free
        uint32_t D[KC];
#pragma unroll
        for (int i = 0; i < KC; ++i) D[i] = 0xFFFFFFFFu;
        const pcrs_f2 qxy = {qx, qy};
        const pcrs_q3 q3 = make_q3(qx, qy, qz);
        auto insert = [&](uint32_t key) {
#pragma unroll
            for (int s = KC - 1; s > 0; --s) D[s] = umed3(key, D[s - 1], D[s]);
            D[0] = D[0] < key ? D[0] : key;
        };
        uint32_t ord = 0;  // candidates met so far
        bool crowded = false;  // more candidates than the tags can name
        walk_cube3(
            a, cx, cy, cz, q3,
            [&](const float(&d2)[4], uint32_t) {
#pragma unroll
                for (int j = 0; j < 4; ++j)  // ordinals stay below 1024: the row gate sees to it
                    insert((__float_as_uint(d2[j]) & ~TAG_MASK) | (ord + (uint32_t)j));
                ord += 4u;
            },
            [&](const pcrs_p3 &p, uint32_t, bool live) {
                const float d2 = sqdist_pk(qxy, qz, make_float4(p.x, p.y, p.z, 0.0f));
                insert(live ? ((__float_as_uint(d2) & ~TAG_MASK) | ord) : 0xFFFFFFFFu);
                ord += live ? 1u : 0u;
            },
            [&](uint32_t len) {  // known before the row is walked: a crowded cube is left at once
                crowded = ord + len > TAG_MASK + 1u;
                return !crowded;
            });
        uint32_t nv = 0;  // neighbours found, at most K
#pragma unroll
        for (int i = 0; i < K; ++i) nv += D[i] < TAG_INF ? 1u : 0u;

        // ---- pairs the truncation may have put in the wrong order (bit i: ranks i and i + 1),
        // up to the pair behind the last place
        uint32_t near = 0;
#pragma unroll
        for (int i = 0; i <= K; ++i)
            if ((uint32_t)i <= keff && D[i + 1] < TAG_INF && (D[i] ^ D[i + 1]) <= TAG_MASK)
                near |= 1u << i;
        // A crowded cube, or three or more neighbours in a row that the truncation cannot tell
        // apart, is not for this kernel.  The lanes of a wave that say so append together: where
        // many do (a crowded region, a lattice), they go as a run of neighbouring queries to the
        // list of the two-walk kernel, which takes such cubes and ties in its stride; a stray few
        // join the retry list (any launch that finds work costs a wave's lifetime, and the retry
        // kernel is launched anyway).
        if (crowded || (near & (near >> 1))) {
            const unsigned long long m = __ballot(1);
            const uint32_t cnt = (uint32_t)__popcll(m);
            const bool alt = a.alt_list && cnt >= 8u;
            uint32_t *const list = alt ? a.alt_list : a.fail_list;
            uint32_t *const count = alt ? a.alt_count : a.fail_count;
            if (list) {
                const int leader = __builtin_ctzll(m);
                const uint32_t lane = threadIdx.x & 63u;
                uint32_t first = 0;
                if ((int)lane == leader) first = atomicAdd(count, cnt);
                first = __shfl(first, leader, 64);
                const uint32_t at = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                list[at] = qi;
                if (!alt && a.fail_bound) a.fail_bound[at] = __builtin_inff();  // nothing known
            }
            continue;
        }
        near &= (1u << keff) - 1u;  // the pairs that matter: up to (last place, the one behind it)

        // ---- certification: every point outside the cube is farther than `margin`
        const int R = a.R;
        const float margin = cube_margin(g, cx, cy, cz, ux, uy, uz, R);
        const bool whole = !(margin < __builtin_inff());
        bool certified = whole;
        // the keff-th distance, rounded up to the key's resolution (+inf when there is none)
        uint32_t tb = TAG_INF;
#pragma unroll
        for (int i = 0; i < K; ++i)
            if ((uint32_t)i == keff - 1 && D[i] < TAG_INF) tb = D[i] | TAG_MASK;
        const float tau_cert = __uint_as_float(tb);
        if (!whole) {
            const float amax = fmaxf(fmaxf(fabsf(ux), fabsf(uy)), fabsf(uz));
            const float m = (margin - (g.slack + 2e-6f * amax)) * (1.0f - 2e-6f);
            certified = (m > 0.0f) && (tau_cert < (m * m) * (1.0f - 2e-6f));
        }
        if (!certified) {  // the retry starts from what this walk has seen
            if (ord < a.sparse_below) {
                a.hand_flags[qi] = a.hand_level;
            } else if (a.fail_list) {
                const uint32_t f = atomicAdd(a.fail_count, 1u);
                a.fail_list[f] = qi;
                if (a.fail_bound) a.fail_bound[f] = tau_cert;
            }
            continue;
        }

        // ---- ordinals to sorted slots: the rows again, in the order of the walk
        // (a candidate of row t has ordinal base_t + its offset in the row: its slot is the
        // ordinal plus the sum of the steps (s0_t - base_t) - (s0_(t-1) - base_(t-1)) over the
        // rows that start at or below it)
        uint32_t slot[K + 1];
#pragma unroll
        for (int i = 0; i <= K; ++i) slot[i] = D[i] & TAG_MASK;
        {
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
            uint32_t base = 0, shift = 0;
            for (int dz = -1; dz <= 1; ++dz) {
                const int zz = cz + dz;
                if (zz < 0 || zz >= g.nz) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int yy = cy + dy;
                    if (yy < 0 || yy >= g.ny) continue;
                    const uint32_t row =
                        ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
                    const uint32_t s0 = a.cell_start[row + x0];
                    const uint32_t len = a.cell_start[row + x1 + 1] - s0;
                    const uint32_t step = (s0 - base) - shift;
#pragma unroll
                    for (int i = 0; i <= K; ++i)
                        slot[i] += (D[i] & TAG_MASK) >= base ? step : 0u;
                    shift += step;
                    base += len;
                }
            }
        }
        // an isolated pair: compare the full keys of the two
        if (near) {
            auto key_at = [&](uint32_t sl) {
                const float4 p = a.spts[sl];
                return make_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
            };
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if ((near >> i) & 1u) {
                    if (key_at(slot[i + 1]) < key_at(slot[i])) {
                        const uint32_t t = slot[i];
                        slot[i] = slot[i + 1];
                        slot[i + 1] = t;
                    }
                }
            }
        }
        const uint32_t have = nv < keff ? nv : keff;
        consume<true>(SlotRegNbrs<K + 1>(slot, a.spts, qx, qy, qz), have, true, (size_t)qi, qx, qy,
                      qz, out);
    }
}

// Wave-cooperative walk over the rows of a box of cells: f(p, valid, slot) is called by the whole
// wave for every 64-point piece of the box.  The lanes fetch the row ranges in parallel (one
// memory latency for up to 64 rows) and a scan of the row lengths turns the rows into one list;
// every lane finds the row of its list position with a binary search over the lanes' prefixes
// (ds_bpermute), so a box of many short rows — a surface seen through a large cube — costs a few
// full-width steps instead of one step per row.  Four steps are requested before any is used: a
// retried query is a single wave, so nothing else hides its load latency.
// clip (r >= 0): only the cells that reach into the ball of radius r around the grid-relative
// point (ux, uy, uz) — rows farther from it than r in y and z are dropped, the others cut in x to
// the ball's chord; r must include the allowance for f32 cell assignment.  stride: every stride-th
// point of the list only (a sample of the box).
struct BallClip {
    float ux, uy, uz, r;
};
// cells [xa, xb] of the row (yy, zz) of a box spanning x0 .. x1 that the clip leaves (xa > xb: none)
__device__ __forceinline__ void clipped_row(const GridGeom &g, const BallClip &clip, int x0, int x1,
                                            int yy, int zz, int *xa, int *xb) {
    *xa = x0;
    *xb = x1;
    if (clip.r < 0.0f) return;
    // distance from the point to the row's slab of cells in y and in z
    const float gy = fmaxf(fmaxf((float)yy * g.cell - clip.uy, clip.uy - (float)(yy + 1) * g.cell), 0.0f);
    const float gz = fmaxf(fmaxf((float)zz * g.cell - clip.uz, clip.uz - (float)(zz + 1) * g.cell), 0.0f);
    const float rem = clip.r * clip.r - (gy * gy + gz * gz);
    if (rem >= 0.0f) {
        const float xr = sqrtf(rem);
        *xa = max(x0, (int)fmaxf(floorf((clip.ux - xr) * g.inv), -1.0f));  // (int) of a huge float is undefined
        *xb = min(x1, (int)fminf(floorf((clip.ux + xr) * g.inv), 1e9f));
    } else {
        *xb = x0 - 1;
    }
}
template <int DEPTH_ALL = 8, int DEPTH = 4, class F>
__device__ __forceinline__ void wave_walk(const CellArgs &a, int x0, int x1, int y0, int y1, int z0,
                                          int z1, int lane, F f,
                                          BallClip clip = BallClip{0.0f, 0.0f, 0.0f, -1.0f},
                                          uint32_t stride = 1u) {
    const GridGeom &g = a.g;
    const int ny = y1 - y0 + 1;
    const int nrows = ny * (z1 - z0 + 1);
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (clip.r < 0.0f && x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.nx - 1 && y1 == g.ny - 1 &&
        z1 == g.nz - 1) {
        // the whole grid is the whole sorted array: a linear scan, eight steps in flight
        const uint32_t ns = (a.nvalid + stride - 1u) / stride;
        for (uint32_t s0 = 0; s0 < ns; s0 += DEPTH_ALL * 64) {
            float4 p[DEPTH_ALL];
#pragma unroll
            for (int j = 0; j < DEPTH_ALL; ++j) {
                const uint32_t sidx = s0 + j * 64 + lane;
                p[j] = sidx < ns ? a.spts[sidx * stride] : zero;
            }
#pragma unroll
            for (int j = 0; j < DEPTH_ALL; ++j)
                if (s0 + j * 64 < ns) f(p[j], s0 + j * 64 + lane < ns, (s0 + j * 64 + lane) * stride);
        }
        return;
    }
    for (int rbase = 0; rbase < nrows; rbase += 64) {
        const int r = rbase + lane;
        uint32_t ms0 = 0, ms1 = 0;
        if (r < nrows) {
            const int zz = z0 + r / ny, yy = y0 + r % ny;
            const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
            int xa, xb;
            clipped_row(g, clip, x0, x1, yy, zz, &xa, &xb);
            if (xa <= xb) {
                ms0 = a.cell_start[row + xa];
                ms1 = a.cell_start[row + xb + 1];
            }
        }
        const uint32_t len = ms1 - ms0;
        uint32_t incl = len;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const uint32_t pre = incl - len;  // list position of this lane's row
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t ns = (total + stride - 1u) / stride;  // list positions 0, stride, 2 stride, ...
        for (uint32_t s0 = 0; s0 < ns; s0 += DEPTH * 64) {
            uint32_t slot[DEPTH];
            bool v[DEPTH];
            float4 p[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const uint32_t sidx = s0 + j * 64 + lane;
                const uint32_t t = sidx * stride;
                int lo = 0;  // the last row whose list position is <= t (empty rows share theirs)
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const uint32_t pc = __shfl(pre, lo + step, 64);
                    if (pc <= t) lo += step;
                }
                slot[j] = __shfl(ms0, lo, 64) + (t - __shfl(pre, lo, 64));
                v[j] = sidx < ns;
                p[j] = v[j] ? a.spts[slot[j]] : zero;
            }
#pragma unroll
            for (int j = 0; j < DEPTH; ++j)
                if (s0 + j * 64 < ns) f(p[j], v[j], slot[j]);
        }
    }
}

// Number of points in a box of cells (in the part of it that the clip leaves), from the row ranges
// alone; the same value in every lane.
__device__ __forceinline__ uint32_t wave_box_count(const CellArgs &a, int x0, int x1, int y0, int y1,
                                                   int z0, int z1, int lane,
                                                   BallClip clip = BallClip{0.0f, 0.0f, 0.0f, -1.0f}) {
    const GridGeom &g = a.g;
    const int ny = y1 - y0 + 1;
    const int nrows = ny * (z1 - z0 + 1);
    uint32_t c = 0;
    for (int r = lane; r < nrows; r += 64) {
        const int zz = z0 + r / ny, yy = y0 + r % ny;
        const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
        int xa, xb;
        clipped_row(g, clip, x0, x1, yy, zz, &xa, &xb);
        if (xa <= xb) c += a.cell_start[row + xb + 1] - a.cell_start[row + xa];
    }
    for (int off = 1; off < 64; off <<= 1) c += __shfl_xor(c, off, 64);
    return c;
}

// Half-width of the cube a wave tries after the one of half-width R left its query uncertified.
// tau: the K-th squared distance found inside that cube, +inf when it held fewer than K points.
// A finite tau bounds the true K-th distance from above, so the cube that reaches sqrt(tau) beyond
// the query's own cell on every side is certain to certify it: the wave goes straight to that one
// (a sparse spot of a scan whose cells were sized for its dense parts may need a dozen cells)
// instead of doubling its way there, and no further than needed.  Without a bound it takes the
// next entry of the list, and doubles on when the list ends with the whole grid (a cube that has
// outgrown the grid is the whole grid).  Never beyond the last entry, always beyond R.
__device__ __forceinline__ int next_half_width(const CellArgs &a, int n_radii, int R, float tau) {
    const int r_last = a.wave_radii[n_radii - 1];
    int next = 2 * R;  // past the list (whose last entry, when it stands for the whole grid, is huge)
    for (int j = n_radii - 1; j >= 0; --j)
        if (a.wave_radii[j] > R && a.wave_radii[j] < (1 << 20)) next = a.wave_radii[j];
    if (tau < __builtin_inff()) {
        const float cells = ceilf(sqrtf(tau) * a.g.inv * 1.0001f) + 1.0f;
        next = cells < (float)r_last ? (int)cells : r_last;
    }
    return min(max(next, R + 1), r_last);
}

// Retry kernel: ONE WAVE PER QUERY.  The queries that the per-lane kernel could not certify are
// few and scattered, so a lane-per-query retry would be a handful of long serial walks.  Here the
// 64 lanes of a wave share one query: they stride over each row of the (2R+1)^3 cube (coalesced
// float4 loads), every lane keeps the K smallest distances of its share, and a 6-step xor
// butterfly over ds_bpermute merges the 64 sorted lists so that all lanes end with the global K
// smallest.  Identities, when needed, are gathered by a second walk with ballot-compacted appends
// into a per-wave LDS list, ranked across lanes, and consumed by lane 0.
template <int K, bool COLLECT>
__global__ __launch_bounds__(GB_THREADS) void knn_wave_kernel(CellArgs a, KnnOut out) {
    constexpr int WAVES = GB_THREADS / 64;
    __shared__ uint32_t wl_slot[WAVES][64];
    __shared__ uint64_t wl_key[WAVES][64];
    __shared__ float wl_px[WAVES][64], wl_py[WAVES][64], wl_pz[WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (a.skip && *a.skip) return;
    const uint32_t total = *a.qcount;
    const GridGeom &g = a.g;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int n_radii = a.n_radii;

    for (uint32_t pos = blockIdx.x * WAVES + w; pos < total; pos += gridDim.x * WAVES) {
        const uint32_t entry = a.qlist[pos];
        const uint32_t qi = entry & ~PCRS_HEAVY_FLAG;
        const int first_radius = (entry & PCRS_HEAVY_FLAG) ? 0 : 1;
        if (a.heavy_only && !(entry & PCRS_HEAVY_FLAG)) {
            if (lane == 0) {
                if (a.hand_flags) a.hand_flags[qi] = a.hand_level;
                else if (a.fail_list) a.fail_list[atomicAdd(a.fail_count, 1u)] = qi;
            }
            continue;
        }
        const float qx = a.qx[qi], qy = a.qy[qi], qz = a.qz[qi];  // finite: stage 1 filtered
        const float ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
        const int cx = cell_coord(qx, g.ox, g.inv, g.nx);
        const int cy = cell_coord(qy, g.oy, g.inv, g.ny);
        const int cz = cell_coord(qz, g.oz, g.inv, g.nz);
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0, z0 = 0, z1 = 0;
        float D[K];
        uint64_t best1 = PCRS_EMPTY_KEY;  // K == 1: key of the nearest point
        bool certified = false;
        const float amax = fmaxf(fmaxf(fabsf(ux), fabsf(uy)), fabsf(uz));
        const float allow = g.slack + 2e-6f * amax;  // f32 rounding of cell assignment and offsets
        // An index over a box of the cloud: the ball of a result must end before the box's open
        // faces, or points the index does not hold could belong to the answer.
        const float open_dist = open_face_distance(g, ux, uy, uz);
        auto fits = [&](float d2) { return sqrtf(d2) * (1.0f + 4e-6f) + 3.0f * allow < open_dist; };
        // K smallest distances of a box of cells (all of it, every stride-th point of it, or the
        // part of it inside a ball), merged so that every lane holds them
        auto search_box = [&](const BallClip &clip, uint32_t stride) {
#pragma unroll
            for (int i = 0; i < K; ++i) D[i] = __builtin_inff();
            best1 = PCRS_EMPTY_KEY;
            wave_walk(
                a, x0, x1, y0, y1, z0, z1, lane,
                [&](const float4 &p, bool valid, uint32_t) {
                    if (valid) {
                        const float d2 = sqdist(qx, qy, qz, p.x, p.y, p.z);
                        if constexpr (K == 1) {  // one neighbour: carry its identity along
                            if (d2 <= D[0]) {
                                const uint64_t k = make_key(d2, __float_as_uint(p.w));
                                if (k < best1) {
                                    best1 = k;
                                    D[0] = d2;
                                }
                            }
                        } else {
#pragma unroll
                            for (int t = K - 1; t > 0; --t)
                                D[t] = __builtin_amdgcn_fmed3f(d2, D[t - 1], D[t]);
                            D[0] = fminf(D[0], d2);
                        }
                    }
                },
                clip, stride);
            if constexpr (K == 1) {
                for (int r = 1; r < 64; r <<= 1) {
                    const uint32_t lo = __shfl_xor((uint32_t)best1, r, 64);
                    const uint32_t hi = __shfl_xor((uint32_t)(best1 >> 32), r, 64);
                    const uint64_t other = ((uint64_t)hi << 32) | lo;
                    if (other < best1) best1 = other;
                }
                D[0] = key_d2(best1);
            } else {
                // butterfly merge of the 64 per-lane lists
#pragma unroll 1
                for (int r = 1; r < 64; r <<= 1) {
                    float T[K];
#pragma unroll
                    for (int i = 0; i < K; ++i) T[i] = D[i];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const float v = __shfl_xor(T[j], r, 64);
#pragma unroll
                        for (int t = K - 1; t > 0; --t)
                            D[t] = __builtin_amdgcn_fmed3f(v, D[t - 1], D[t]);
                        D[0] = fminf(D[0], v);
                    }
                }
            }
        };
        auto kth = [&](uint32_t kk) {  // D[kk - 1]
            float v = __builtin_inff();
#pragma unroll
            for (int i = 0; i < K; ++i)
                if ((uint32_t)i == kk - 1) v = D[i];
            return v;
        };
        // Larger and larger cubes until one holds enough points (counted from the row ranges
        // alone: no point is read, nothing merged).  That cube is searched, and what it gives is
        // either certified by the distance to the cube's faces or an upper bound of the K-th
        // distance; the ball of that radius then holds the answer, whatever cube it needs, and only
        // the cells that reach into it are searched.  A region of more than 8192 points is sampled
        // first (every n-th point of its list): the K-th distance of the sample is a bound too, and
        // its ball is a much smaller region.  A stray point above a dense surface thus costs a
        // sample and the few rows under it instead of the square of surface that a cube around it
        // would cover, and a stray far from a large cloud a few samples instead of the cloud.
        // (The wave is uniform in the query, so none of this diverges.)
        const int r_last = a.wave_radii[n_radii - 1];
        int R = a.wave_radii[first_radius];
        BallClip clip = {ux, uy, uz, -1.0f};  // the region is the box, cut to this ball once r >= 0
        uint32_t n_region = 0;
        bool counted = false;
        float bound = __builtin_inff();  // smallest upper bound of the kcert-th distance so far
        // The kernel that handed the query on has seen its 27-cell cube: the kcert-th distance
        // among those points is a bound already, and the search starts with its ball instead of
        // counting and searching whole cubes (a query whose cube fell a little short of the
        // certificate costs a step or two of the walk instead of the 125 cells around it).
        if (a.qbound && K > 1) {
            const float hint = a.qbound[pos];
            if (hint < __builtin_inff()) {
                const float rb = sqrtf(hint) * (1.0f + 4e-6f) + 2.0f * allow;
                const float cells = ceilf(rb * g.inv);
                if (cells <= (float)r_last) {
                    bound = hint;
                    R = max((int)cells, 1);
                    clip.r = rb;
                }
            }
        }
#pragma unroll 1
        for (;;) {
            x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
            y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
            z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
            const bool whole = (x0 == 0) && (x1 == g.nx - 1) && (y0 == 0) && (y1 == g.ny - 1) &&
                               (z0 == 0) && (z1 == g.nz - 1);
            const bool closed_whole = whole && !(open_dist < __builtin_inff());
            if (!counted)
                n_region = (whole && clip.r < 0.0f) ? a.nvalid
                                                    : wave_box_count(a, x0, x1, y0, y1, z0, z1, lane, clip);
            counted = false;
            if (clip.r < 0.0f && !whole && n_region < a.kcert) {  // too few points to certify any
                if (R >= r_last) break;
                R = next_half_width(a, n_radii, R, __builtin_inff());
                continue;
            }
            uint32_t stride = n_region > 8192u ? n_region / 2048u : 1u;
            search_box(clip, stride);
            float tau_cert = kth(a.kcert);
            if (stride > 1u && !(fminf(bound, tau_cert) < __builtin_inff())) {
                // a sample without a finite distance (coordinates whose squares overflow) bounds
                // nothing: the region is searched as it is
                stride = 1u;
                search_box(clip, 1u);
                tau_cert = kth(a.kcert);
            }
            if (stride == 1u) {
                // every point of the region was seen: a ball around a bound holds the answer (the
                // comparison can only fail if the ball missed a cell it should have had: such a
                // query is left to the brute-force stage rather than answered wrongly)
                certified = closed_whole ||
                            (clip.r >= 0.0f && (K == 1 || tau_cert <= bound) &&
                             fits(K == 1 ? fminf(tau_cert, bound) : tau_cert));
                if (!certified && clip.r >= 0.0f) break;
                if (!certified) {
                    const float margin = cube_margin(g, cx, cy, cz, ux, uy, uz, R);
                    const float m = (margin - allow) * (1.0f - 2e-6f);
                    certified = (m > 0.0f) && (tau_cert < (m * m) * (1.0f - 2e-6f) ||
                                               (K == 1 && a.useful_d2 < (m * m) * (1.0f - 2e-6f)));
                }
                if (certified) break;
            }
            // at least kcert points lie within `bound` of the query (for one neighbour: or nothing
            // beyond it is of use), so the kcert nearest are inside the ball of that radius
            bound = fminf(bound, K == 1 ? fminf(tau_cert, a.useful_d2) : tau_cert);
            if (!(bound < __builtin_inff())) {
                if (R >= r_last) break;
                R = next_half_width(a, n_radii, R, bound);
                continue;
            }
            const float rb = sqrtf(bound) * (1.0f + 4e-6f) + 2.0f * allow;
            const float cells = ceilf(rb * g.inv);
            if (cells <= (float)r_last) {
                R = max((int)cells, 1);
            } else {
                if (r_last < (1 << 20)) break;  // beyond what this launch may search
                R = r_last;                     // the whole grid, cut to the ball
            }
            clip.r = rb;
            // a sample whose ball is hardly smaller than the region it came from (points at equal
            // distances all around) is not tried twice: the region is then searched as it is
            const int bx0 = max(cx - R, 0), bx1 = min(cx + R, g.nx - 1);
            const int by0 = max(cy - R, 0), by1 = min(cy + R, g.ny - 1);
            const int bz0 = max(cz - R, 0), bz1 = min(cz + R, g.nz - 1);
            const uint32_t n_ball = wave_box_count(a, bx0, bx1, by0, by1, bz0, bz1, lane, clip);
            counted = true;
            if (stride > 1u && n_ball > n_region / 2u) {
                x0 = bx0, x1 = bx1, y0 = by0, y1 = by1, z0 = bz0, z1 = bz1;
                search_box(clip, 1u);
                const float t = kth(a.kcert);
                certified = fits(K == 1 ? fminf(t, bound) : t);
                break;
            }
            n_region = n_ball;
        }
        bool failed = !certified;

        uint32_t have = 0;
        if (!failed && COLLECT && K == 1) {
            // beyond the caller's bound: not a usable neighbour (and possibly not the nearest)
            if (D[0] > a.useful_d2) best1 = PCRS_EMPTY_KEY;
            if (lane == 0) wl_key[w][0] = best1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            have = best1 != PCRS_EMPTY_KEY ? 1u : 0u;
        } else if (!failed && COLLECT) {
            float tau = __builtin_inff();
#pragma unroll
            for (int i = 0; i < K; ++i)
                if ((uint32_t)i == a.kcollect - 1) tau = D[i];
            uint32_t base = 0;
            // every hit lies in the ball of the collected distance: only its rows are walked
            const BallClip ball = {ux, uy, uz,
                                   tau < __builtin_inff()
                                       ? sqrtf(tau) * (1.0f + 4e-6f) + 2.0f * allow
                                       : -1.0f};
            wave_walk(
                a, x0, x1, y0, y1, z0, z1, lane,
                [&](const float4 &p, bool valid, uint32_t s) {
                    const bool hit = valid && sqdist(qx, qy, qz, p.x, p.y, p.z) <= tau;
                    const unsigned long long mask = __ballot(hit);
                    if (hit) {
                        const uint32_t at = base + (uint32_t)__popcll(mask & lt_mask);
                        if (at < 64u) wl_slot[w][at] = s;
                    }
                    base += (uint32_t)__popcll(mask);
                },
                ball, 1u);
            if (base > 64u) {
                failed = true;  // more ties at the K-th distance than the list holds
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                uint64_t key = PCRS_EMPTY_KEY;
                if ((uint32_t)lane < base) {
                    const float4 p = a.spts[wl_slot[w][lane]];
                    key = make_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
                }
                uint32_t rank = 0;
                for (uint32_t i = 0; i < base; ++i) {
                    const uint32_t lo = __shfl((uint32_t)key, (int)i, 64);
                    const uint32_t hi = __shfl((uint32_t)(key >> 32), (int)i, 64);
                    const uint64_t other = ((uint64_t)hi << 32) | lo;
                    rank += (other < key) ? 1u : 0u;
                }
                if ((uint32_t)lane < base) wl_key[w][rank] = key;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                have = base < a.kcollect ? base : a.kcollect;
            }
        }
        if (failed) {
            if (lane == 0) {
                if (a.hand_flags) {
                    a.hand_flags[qi] = a.hand_level;
                } else if (a.fail_list) {
                    const uint32_t f = atomicAdd(a.fail_count, 1u);
                    a.fail_list[f] = qi;
                }
            }
            continue;
        }
        // The neighbours' coordinates, one per lane and all in flight together: the lane that
        // consumes the list would otherwise fetch them one dependent load after the other
        if (COLLECT && out.nx) {
            if ((uint32_t)lane < have) {
                const uint32_t id = key_idx(wl_key[w][lane]);
                wl_px[w][lane] = a.rx[id];
                wl_py[w][lane] = a.ry[id];
                wl_pz[w][lane] = a.rz[id];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (lane == 0) {
            KnnOut o2 = out;
            if (out.mean) {
                const uint32_t kk = out.k_sor + 1u;
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < K; ++i)
                    c += ((uint32_t)i < kk && D[i] < __builtin_inff()) ? 1u : 0u;
                const uint32_t first = c > 1u ? 1u : 0u;
                float mean = __builtin_inff();
                if (c - first > 0u) {
                    float sum = 0.0f;
#pragma unroll
                    for (int i = 0; i < K; ++i)
                        if ((uint32_t)i >= first && (uint32_t)i < c) sum += sqrtf(D[i]);
                    mean = sum / (float)(c - first);
                }
                store_mean(o2, qi, mean);
            }
            if (COLLECT) {
                o2.mean = nullptr;
                consume(StagedListNbrs{&wl_key[w][0], &wl_px[w][0], &wl_py[w][0], &wl_pz[w][0]}, have,
                        true, (size_t)qi, qx, qy, qz, o2);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // lists are reused next round
    }
}

// total[0] += *entries, total[3] += *left (nullable): the stage counters of one level of a refined
// index joining the search's totals.
__global__ void add_counts_kernel(uint32_t *total, const uint32_t *entries, const uint32_t *left) {
    if (threadIdx.x == 0) {
        atomicAdd(&total[0], *entries);
        if (left) atomicAdd(&total[3], *left);
    }
}

// Sorted slots whose original index lies in [q_begin, q_begin + q_len), in slot order, so that a
// sub-range of a cloud is still processed with full waves of neighbouring queries.  Count per
// block of 1024 slots, exclusive scan, ordered write: no atomics (one per wave on a single counter
// made this step cost more than the search at 8 M points).
constexpr int OWN_PER_THREAD = 4;
constexpr int OWN_BLOCK = GB_THREADS * OWN_PER_THREAD;

__device__ __forceinline__ uint32_t owned_flags(const float4 *spts, uint32_t nvalid, uint32_t first,
                                                const SlotChoice &ch) {
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < OWN_PER_THREAD; ++i) {
        const uint32_t s = first + i;
        if (s < nvalid) {
            if (slot_is_query(ch, spts[s])) f |= 1u << i;
        }
    }
    return f;
}

__global__ __launch_bounds__(GB_THREADS) void owned_count_kernel(const float4 *spts,
                                                                 uint32_t nvalid, SlotChoice ch,
                                                                 uint32_t *block_count) {
    __shared__ uint32_t wsum[GB_THREADS / 64];
    const uint32_t first = blockIdx.x * OWN_BLOCK + threadIdx.x * OWN_PER_THREAD;
    uint32_t c = __popc(owned_flags(spts, nvalid, first, ch));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(GB_THREADS) void owned_write_kernel(const float4 *spts,
                                                                 uint32_t nvalid, SlotChoice ch,
                                                                 const uint32_t *block_offset,
                                                                 uint32_t *list) {
    __shared__ uint32_t wsum[GB_THREADS / 64];
    const uint32_t first = blockIdx.x * OWN_BLOCK + threadIdx.x * OWN_PER_THREAD;
    const uint32_t f = owned_flags(spts, nvalid, first, ch);
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
    for (int i = 0; i < OWN_PER_THREAD; ++i)
        if (f & (1u << i)) {
            list[at++] = first + i;
            if (ch.consume) ch.consume[__float_as_uint(spts[first + i].w)] = 0xFFu;
        }
}

// 1-NN retry of ONE query by a whole wave (the K = 1 case of knn_wave_kernel as a function, see
// there): larger and larger cubes until one holds a point, a search of that cube (of a sample of
// it when it is large), then the ball of the distance found.  Lane 0 stores the answer; a query
// that stays uncertified goes to a.fail_list for the brute-force stage.  entry: query index,
// PCRS_HEAVY_FLAG set when the 27-cell cube itself is still to be searched.
__device__ __forceinline__ void nn1_wave_retry(const CellArgs &a, const KnnOut &out, uint32_t entry,
                                               float qx, float qy, float qz, int lane) {
    const GridGeom &g = a.g;
    const uint32_t qi = entry & ~PCRS_HEAVY_FLAG;
    const float ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
    const int cx = cell_coord(qx, g.ox, g.inv, g.nx);
    const int cy = cell_coord(qy, g.oy, g.inv, g.ny);
    const int cz = cell_coord(qz, g.oz, g.inv, g.nz);
    const float amax = fmaxf(fmaxf(fabsf(ux), fabsf(uy)), fabsf(uz));
    const float allow = g.slack + 2e-6f * amax;
    const float open_dist = open_face_distance(g, ux, uy, uz);  // see knn_wave_kernel
    auto fits = [&](float d2) { return sqrtf(d2) * (1.0f + 4e-6f) + 3.0f * allow < open_dist; };
    uint64_t best = PCRS_EMPTY_KEY;
    float best_d2 = __builtin_inff();
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0, z0 = 0, z1 = 0;
    auto search_box = [&](const BallClip &clip, uint32_t stride) {
        best = PCRS_EMPTY_KEY;
        best_d2 = __builtin_inff();
        wave_walk<4, 2>(
            a, x0, x1, y0, y1, z0, z1, lane,
            [&](const float4 &p, bool valid, uint32_t) {
                if (valid) {
                    const float d2 = sqdist(qx, qy, qz, p.x, p.y, p.z);
                    if (d2 <= best_d2) {
                        const uint64_t k = make_key(d2, __float_as_uint(p.w));
                        if (k < best) {
                            best = k;
                            best_d2 = d2;
                        }
                    }
                }
            },
            clip, stride);
        for (int r = 1; r < 64; r <<= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)best, r, 64);
            const uint32_t hi = __shfl_xor((uint32_t)(best >> 32), r, 64);
            const uint64_t other = ((uint64_t)hi << 32) | lo;
            if (other < best) best = other;
        }
        best_d2 = key_d2(best);
    };
    bool certified = false;
    const int r_last = a.wave_radii[a.n_radii - 1];
    int R = a.wave_radii[(entry & PCRS_HEAVY_FLAG) ? 0 : 1];
    BallClip clip = {ux, uy, uz, -1.0f};
    uint32_t n_region = 0;
    bool counted = false;
    float bound = __builtin_inff();  // nothing farther than this can be the answer
#pragma unroll 1
    for (;;) {
        x0 = max(cx - R, 0), x1 = min(cx + R, g.nx - 1);
        y0 = max(cy - R, 0), y1 = min(cy + R, g.ny - 1);
        z0 = max(cz - R, 0), z1 = min(cz + R, g.nz - 1);
        const bool whole = (x0 == 0) && (x1 == g.nx - 1) && (y0 == 0) && (y1 == g.ny - 1) &&
                           (z0 == 0) && (z1 == g.nz - 1);
        const bool closed_whole = whole && !(open_dist < __builtin_inff());
        if (!counted)
            n_region = (whole && clip.r < 0.0f) ? a.nvalid
                                                : wave_box_count(a, x0, x1, y0, y1, z0, z1, lane, clip);
        counted = false;
        if (clip.r < 0.0f && !whole && n_region == 0u) {  // an empty cube
            // ... which still settles the query when nothing outside it can be of use
            const float margin = cube_margin(g, cx, cy, cz, ux, uy, uz, R);
            const float m = (margin - allow) * (1.0f - 2e-6f);
            if (m > 0.0f && a.useful_d2 < (m * m) * (1.0f - 2e-6f)) {
                best = PCRS_EMPTY_KEY;
                best_d2 = __builtin_inff();
                certified = true;
                break;
            }
            if (R >= r_last) break;
            R = next_half_width(a, a.n_radii, R, a.useful_d2);
            continue;
        }
        uint32_t stride = n_region > 8192u ? n_region / 2048u : 1u;
        search_box(clip, stride);
        if (stride > 1u && !(fminf(bound, fminf(best_d2, a.useful_d2)) < __builtin_inff())) {
            // a sample without a finite distance (coordinates whose squares overflow) bounds
            // nothing: the region is searched as it is
            stride = 1u;
            search_box(clip, 1u);
        }
        if (stride == 1u) {
            // (a ball that was searched in full settles the query when it ends before the open
            // faces of the index's box; one that does not leaves it to the next stage)
            certified = closed_whole ||
                        (clip.r >= 0.0f && fits(fminf(fminf(best_d2, a.useful_d2), bound)));
            if (!certified && clip.r >= 0.0f) break;
            if (!certified) {
                const float margin = cube_margin(g, cx, cy, cz, ux, uy, uz, R);
                const float m = (margin - allow) * (1.0f - 2e-6f);
                certified = (m > 0.0f) && (best_d2 < (m * m) * (1.0f - 2e-6f) ||
                                           a.useful_d2 < (m * m) * (1.0f - 2e-6f));
            }
            if (certified) break;
        }
        bound = fminf(bound, fminf(best_d2, a.useful_d2));
        if (!(bound < __builtin_inff())) {
            if (R >= r_last) break;
            R = next_half_width(a, a.n_radii, R, bound);
            continue;
        }
        const float rb = sqrtf(bound) * (1.0f + 4e-6f) + 2.0f * allow;
        const float cells = ceilf(rb * g.inv);
        if (cells <= (float)r_last) {
            R = max((int)cells, 1);
        } else {
            if (r_last < (1 << 20)) break;  // beyond what this launch may search
            R = r_last;                     // the whole grid, cut to the ball
        }
        clip.r = rb;
        const int bx0 = max(cx - R, 0), bx1 = min(cx + R, g.nx - 1);
        const int by0 = max(cy - R, 0), by1 = min(cy + R, g.ny - 1);
        const int bz0 = max(cz - R, 0), bz1 = min(cz + R, g.nz - 1);
        const uint32_t n_ball = wave_box_count(a, bx0, bx1, by0, by1, bz0, bz1, lane, clip);
        counted = true;
        if (stride > 1u && n_ball > n_region / 2u) {  // a sample that did not narrow the region
            x0 = bx0, x1 = bx1, y0 = by0, y1 = by1, z0 = bz0, z1 = bz1;
            search_box(clip, 1u);
            certified = fits(fminf(fminf(best_d2, a.useful_d2), bound));
            break;
        }
        n_region = n_ball;
    }
    if (lane != 0) return;
    if (!certified) {
        const uint32_t f = atomicAdd(a.fail_count, 1u);
        a.fail_list[f] = qi;
        return;
    }
    // beyond the caller's bound: not a usable neighbour (and possibly not the nearest)
    const bool none = best == PCRS_EMPTY_KEY || best_d2 > a.useful_d2;
    out.nn_idx[qi] = none ? PCRS_NO_INDEX : key_idx(best);
    if (out.nn_d2) out.nn_d2[qi] = none ? __builtin_inff() : best_d2;
}

// 1-NN variant: a single pass tracking the best (d2, index) key, EIGHT LANES PER QUERY.  The
// clouds this serves (ICP correspondences, ~1e5 queries) give a lane-per-query kernel barely one
// wave per SIMD, each a long chain of dependent loads; eight lanes share the query's candidate
// list (coalesced 128-byte reads), which makes eight times the waves and an eighth of the chain.
constexpr int NN_SUB = 8;

__global__ __launch_bounds__(GB_THREADS) __attribute__((amdgpu_waves_per_eu(5)))
void nn_cell_kernel(CellArgs a, KnnOut out) {
    if (a.zero_me && blockIdx.x == 0 && threadIdx.x < 16) a.zero_me[threadIdx.x] = 0u;
    if (a.skip && *a.skip) return;
    uint32_t total = a.nq;
    if (a.qmode == 2 || a.qmode == 4) total = *a.qcount;
    const GridGeom &g = a.g;
    const uint32_t sub = threadIdx.x & (NN_SUB - 1);
    const uint32_t groups_per_grid = gridDim.x * (GB_THREADS / NN_SUB);
    // the loop bound is the same for all lanes of a wave: the shuffles below need them all
    for (uint32_t q0 = (blockIdx.x * GB_THREADS + (threadIdx.x & ~63u)) / NN_SUB; q0 < total;
         q0 += groups_per_grid) {
        const uint32_t qpos = q0 + ((threadIdx.x & 63u) / NN_SUB);
        const bool active = qpos < total;
        uint32_t qi = 0u;
        float qx = 0.0f, qy = 0.0f, qz = 0.0f;
        if (active && a.qmode == 4) {  // queries sorted by cell: (x, y, z, query index)
            const float4 p = a.qpts[qpos];
            qi = __float_as_uint(p.w);
            qx = p.x;
            qy = p.y;
            qz = p.z;
        } else if (active) {
            qi = (a.qmode == 2) ? a.qlist[qpos] : qpos;
            qx = a.qx[qi];
            qy = a.qy[qi];
            qz = a.qz[qi];
            if (a.xform) {
                // the increment of the previous iteration (icp.rs:39-47,186), applied here instead
                // of by a kernel of its own; every lane of the query computes the same point, one
                // stores it for the kernels that follow
                const float *r = a.xform->inc_r, *t = a.xform->inc_t;
                const float px = qx, py = qy, pz = qz;
                qx = ((r[0] * px + r[1] * py) + r[2] * pz) + t[0];
                qy = ((r[3] * px + r[4] * py) + r[5] * pz) + t[1];
                qz = ((r[6] * px + r[7] * py) + r[8] * pz) + t[2];
                if (sub == 0) {
                    a.wx[qi] = qx;
                    a.wy[qi] = qy;
                    a.wz[qi] = qz;
                }
            }
        }
        const bool qvalid = active && finite3(qx, qy, qz);
        const float ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
        const int cx = cell_coord(qx, g.ox, g.inv, g.nx);
        const int cy = cell_coord(qy, g.oy, g.inv, g.ny);
        const int cz = cell_coord(qz, g.oz, g.inv, g.nz);
        // the nine row ranges of the cube, all requested together
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        uint32_t rb[9], re[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int yy = cy + i % 3 - 1, zz = cz + i / 3 - 1;
            rb[i] = 0;
            re[i] = 0;
            if (qvalid && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
                rb[i] = a.cell_start[row + x0];
                re[i] = a.cell_start[row + x1 + 1];
            }
        }
        // A crowded cube (a dense spot of the cloud) goes to the wave-per-query kernel, flagged
        // so that that kernel starts with this same cube.
        uint32_t cube_pts = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) cube_pts += re[i] - rb[i];
        const bool diverted = qvalid && cube_pts > a.heavy && a.fail_list != nullptr;

        uint64_t best = PCRS_EMPTY_KEY;
        float best_d2 = __builtin_inff();
        const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto visit = [&](const float4 &p) {
            const float d2 = sqdist(qx, qy, qz, p.x, p.y, p.z);
            if (d2 <= best_d2) {
                const uint64_t k = make_key(d2, __float_as_uint(p.w));
                if (k < best) {
                    best = k;
                    best_d2 = d2;
                }
            }
        };
        auto share = [&]() {  // the best key of the query's eight lanes, in all of them
#pragma unroll
            for (int r = 1; r < NN_SUB; r <<= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)best, r, 64);
                const uint32_t hi = __shfl_xor((uint32_t)(best >> 32), r, 64);
                const uint64_t other = ((uint64_t)hi << 32) | lo;
                if (other < best) best = other;
            }
            best_d2 = key_d2(best);
        };
        // The query's own row first (rows are indexed 3 (dz + 1) + (dy + 1): the centre is 4).
        // Where the cloud is dense the nearest neighbour is almost always there, and every
        // other row lies farther away than it in y and z alone.
        const uint32_t n_centre = diverted ? 0u : re[4] - rb[4];
        for (uint32_t t0 = sub; t0 < n_centre; t0 += 4 * NN_SUB) {
            float4 p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t t = t0 + j * NN_SUB;
                p[j] = (t < n_centre) ? a.spts[rb[4] + t] : zero;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t0 + j * NN_SUB < n_centre) visit(p[j]);
        }
        share();
        // The other rows as one list, without those that cannot hold anything nearer: the gap to
        // a row is at least the gap to the faces of the query's own cell in y and z, shortened by
        // the same allowance for f32 cell assignment as the certification below.  A row at exactly
        // the best distance stays (ties go to the larger index, wherever it lies).
        const float amax_q = fmaxf(fmaxf(fabsf(ux), fabsf(uy)), fabsf(uz));
        const float allow = g.slack + 2e-6f * amax_q;
        const float gy[3] = {fmaxf(uy - (float)cy * g.cell - allow, 0.0f), 0.0f,
                             fmaxf((float)(cy + 1) * g.cell - uy - allow, 0.0f)};
        const float gz[3] = {fmaxf(uz - (float)cz * g.cell - allow, 0.0f), 0.0f,
                             fmaxf((float)(cz + 1) * g.cell - uz - allow, 0.0f)};
        // start[i] = list position of row i, off[i] maps a list position inside row i to its slot
        uint32_t off[9], start[9], total_pts = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float gap2 = (gy[i % 3] * gy[i % 3] + gz[i / 3] * gz[i / 3]) * (1.0f - 8e-6f);
            const bool skip = i == 4 || diverted || gap2 > best_d2;
            start[i] = total_pts;
            off[i] = rb[i] - total_pts;
            total_pts += skip ? 0u : re[i] - rb[i];
        }
        for (uint32_t t0 = sub; t0 < total_pts; t0 += 4 * NN_SUB) {
            float4 p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t t = t0 + j * NN_SUB;
                uint32_t o = off[0];
#pragma unroll
                for (int i = 1; i < 9; ++i) o = (t >= start[i]) ? off[i] : o;
                p[j] = (t < total_pts) ? a.spts[t + o] : zero;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t0 + j * NN_SUB < total_pts) visit(p[j]);
        }
        share();
        // after the shuffles: one lane finishes the query; what it cannot finish is retried below
        // by the whole wave, one query at a time
        uint32_t retry = 0xFFFFFFFFu;  // entry for nn1_wave_retry, or none
        if (sub == 0 && active) {
            if (!qvalid) {
                out.nn_idx[qi] = PCRS_NO_INDEX;
                if (out.nn_d2) out.nn_d2[qi] = __builtin_inff();
            } else if (diverted) {
                retry = qi | PCRS_HEAVY_FLAG;
            } else {
                const int R = 1;
                const float margin = cube_margin(g, cx, cy, cz, ux, uy, uz, R);
                const bool whole = !(margin < __builtin_inff());
                bool certified = whole;
                if (!whole) {
                    const float amax = fmaxf(fmaxf(fabsf(ux), fabsf(uy)), fabsf(uz));
                    const float m = (margin - (g.slack + 2e-6f * amax)) * (1.0f - 2e-6f);
                    // either the nearest point is certainly inside the cube, or everything outside
                    // it is farther than the caller cares about
                    certified = (m > 0.0f) && (best_d2 < (m * m) * (1.0f - 2e-6f) ||
                                               a.useful_d2 < (m * m) * (1.0f - 2e-6f));
                }
                if (!certified) {
                    if (a.fail_list) retry = qi;
                } else {
                    // A query certified only because everything outside the cube is beyond the
                    // caller's bound may hold a point of the cube that is neither within the bound
                    // nor the nearest one: it has no neighbour the caller can use, and is stored
                    // as such.
                    const bool none = best == PCRS_EMPTY_KEY || best_d2 > a.useful_d2;
                    out.nn_idx[qi] = none ? PCRS_NO_INDEX : key_idx(best);
                    if (out.nn_d2) out.nn_d2[qi] = none ? __builtin_inff() : best_d2;
                }
            }
        }
        // Larger cubes for the queries of this wave that were not certified (or were too crowded
        // for eight lanes), here and now: a retry kernel behind this one costs a launch and a
        // wave lifetime in every ICP iteration, for a handful of queries.
        unsigned long long todo = __ballot(retry != 0xFFFFFFFFu);
        if (todo && a.stage1_fail && (threadIdx.x & 63u) == 0)
            atomicAdd(a.stage1_fail, (uint32_t)__popcll(todo));
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            // the query is the same in every lane: as scalars it costs the retry no registers
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)retry, src);
            const float fx = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qx), src));
            const float fy = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qy), src));
            const float fz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qz), src));
            nn1_wave_retry(a, out, e, fx, fy, fz, (int)(threadIdx.x & 63u));
        }
    }
}

// Number of indexed points within sqrt(r2) of each query (kdtree.rs:114-129: d2 <= r2, the query
// itself included when it is in the cloud): the cube of a.R cells around the query's cell covers
// the ball, so one walk counts exactly.  qmode 0 (every sorted slot, output at the point's
// original index) or 1 (explicit queries).
// FILL: instead of counting, the hits are written as keys (query << idx_bits | point index) at
// keys[offsets[query] ...] in walk order (a sort by key then orders every list by index).
template <bool FILL>
__global__ __launch_bounds__(GB_THREADS) void radius_cell_kernel(CellArgs a, float r2,
                                                                 uint32_t *count,
                                                                 const uint32_t *offsets,
                                                                 uint64_t *keys, int idx_bits) {
    const GridGeom &g = a.g;
    for (uint32_t pos = blockIdx.x * GB_THREADS + threadIdx.x; pos < a.nq;
         pos += gridDim.x * GB_THREADS) {
        uint32_t qi = pos;
        float qx, qy, qz;
        if (a.qmode == 0) {
            const float4 p = a.spts[pos];
            qi = __float_as_uint(p.w);
            qx = p.x;
            qy = p.y;
            qz = p.z;
        } else {
            qx = a.qx[qi];
            qy = a.qy[qi];
            qz = a.qz[qi];
        }
        uint32_t c = 0;
        uint32_t at = FILL ? offsets[qi] : 0u;
        if (finite3(qx, qy, qz)) {
            const int cx = cell_coord(qx, g.ox, g.inv, g.nx);
            const int cy = cell_coord(qy, g.oy, g.inv, g.ny);
            const int cz = cell_coord(qz, g.oz, g.inv, g.nz);
            walk_cube(a, cx, cy, cz, [&](const float4 &p, const float4 *) {
                const bool hit = sqdist(qx, qy, qz, p.x, p.y, p.z) <= r2;
                if (FILL) {
                    if (hit) keys[at++] = ((uint64_t)qi << idx_bits) | __float_as_uint(p.w);
                } else {
                    c += hit ? 1u : 0u;
                }
            });
        }
        if (!FILL) count[qi] = c;
    }
}


// Where a grid is crowded: the box (in cells) of the cells that hold more than `threshold` points,
// as six maxima (the low sides stored as n - 1 - coordinate), their number and their points.
// And where it is too sparse for its cells: short_points = the points whose own 27-cell cube holds
// fewer than k_short points (no search of k_short neighbours can end in that cube).
struct DenseScan {
    uint32_t hi[3], lo_rev[3];
    uint32_t cells, pad;
    unsigned long long points;
    unsigned long long short_points;
};
__global__ __launch_bounds__(GB_THREADS) void dense_scan_kernel(const uint32_t *cell_start,
                                                                GridGeom g, uint32_t ncell,
                                                                uint32_t threshold, uint32_t k_short,
                                                                DenseScan *out) {
    __shared__ uint32_t smax[6];
    __shared__ uint32_t scells;
    __shared__ unsigned long long spoints, sshort;
    if (threadIdx.x < 6) smax[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        scells = 0;
        spoints = 0;
        sshort = 0;
    }
    __syncthreads();
    for (uint32_t c = blockIdx.x * GB_THREADS + threadIdx.x; c < ncell; c += gridDim.x * GB_THREADS) {
        const uint32_t cnt = cell_start[c + 1] - cell_start[c];
        if (cnt > 0u && cnt < k_short) {
            const int cx = (int)(c % (uint32_t)g.nx), cy = (int)((c / (uint32_t)g.nx) % (uint32_t)g.ny);
            const int cz = (int)(c / ((uint32_t)g.nx * (uint32_t)g.ny));
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
            uint32_t cube = 0;
            for (int i = 0; i < 9 && cube < k_short; ++i) {
                const int yy = cy + i % 3 - 1, zz = cz + i / 3 - 1;
                if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                    const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
                    cube += cell_start[row + x1 + 1] - cell_start[row + x0];
                }
            }
            if (cube < k_short) atomicAdd(&sshort, (unsigned long long)cnt);
        }
        if (cnt > threshold) {
            const uint32_t cx = c % (uint32_t)g.nx, cy = (c / (uint32_t)g.nx) % (uint32_t)g.ny;
            const uint32_t cz = c / ((uint32_t)g.nx * (uint32_t)g.ny);
            atomicMax(&smax[0], cx);
            atomicMax(&smax[1], cy);
            atomicMax(&smax[2], cz);
            atomicMax(&smax[3], (uint32_t)g.nx - 1u - cx);
            atomicMax(&smax[4], (uint32_t)g.ny - 1u - cy);
            atomicMax(&smax[5], (uint32_t)g.nz - 1u - cz);
            atomicAdd(&scells, 1u);
            atomicAdd(&spoints, (unsigned long long)cnt);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 && scells) atomicMax(&out->hi[threadIdx.x], smax[threadIdx.x]);
    if (threadIdx.x >= 3 && threadIdx.x < 6 && scells) atomicMax(&out->lo_rev[threadIdx.x - 3], smax[threadIdx.x]);
    if (threadIdx.x == 0 && scells) {
        atomicAdd(&out->cells, scells);
        atomicAdd(&out->points, spoints);
    }
    if (threadIdx.x == 0 && sshort) atomicAdd(&out->short_points, sshort);
}

// The same look at the crowded cells, kept apart by region: the grid is divided into DENSE_B^3
// blocks of cells, and every block gets the box of its crowded cells (low sides stored reversed,
// as above), their number and their points.  The host joins neighbouring blocks into regions.
constexpr int DENSE_B = 16;
constexpr int DENSE_T = 3;  // thresholds looked at in one pass: threshold, 2x, 4x
struct DenseBlock {
    uint32_t hi[3], lo_rev[3];
    uint32_t cells, pad;
    unsigned long long points;
    unsigned long long heavy_points;  // of them, in cells whose cube holds more than heavy_cube points
    unsigned long long visits;        // candidates their queries read: points x cube x walks
};
__global__ __launch_bounds__(GB_THREADS) void dense_blocks_kernel(const uint32_t *cell_start,
                                                                  GridGeom g, uint32_t ncell,
                                                                  uint32_t threshold,
                                                                  uint32_t heavy_cube, DenseBlock *out) {
    for (uint32_t c = blockIdx.x * GB_THREADS + threadIdx.x; c < ncell; c += gridDim.x * GB_THREADS) {
        const uint32_t cnt = cell_start[c + 1] - cell_start[c];
        if (cnt <= threshold) continue;
        const uint32_t cx = c % (uint32_t)g.nx, cy = (c / (uint32_t)g.nx) % (uint32_t)g.ny;
        const uint32_t cz = c / ((uint32_t)g.nx * (uint32_t)g.ny);
        // the cell's 3 x 3 x 3 cube, as the search kernels count it: nine runs of up to three cells
        uint32_t cube = 0;
        {
            const int x0 = cx ? (int)cx - 1 : 0, x1 = (int)cx + 1 < g.nx ? (int)cx + 1 : g.nx - 1;
            for (int zz = (int)cz - 1; zz <= (int)cz + 1; ++zz) {
                if (zz < 0 || zz >= g.nz) continue;
                for (int yy = (int)cy - 1; yy <= (int)cy + 1; ++yy) {
                    if (yy < 0 || yy >= g.ny) continue;
                    const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
                    cube += cell_start[row + x1 + 1] - cell_start[row + x0];
                }
            }
        }
        const uint32_t bx = cx * DENSE_B / (uint32_t)g.nx, by = cy * DENSE_B / (uint32_t)g.ny;
        const uint32_t bz = cz * DENSE_B / (uint32_t)g.nz;
        for (int t = 0; t < DENSE_T; ++t) {  // out[t]: the cells above threshold << t
            if (cnt <= (threshold << t)) break;
            DenseBlock *b = out + (size_t)t * DENSE_B * DENSE_B * DENSE_B + (bz * DENSE_B + by) * DENSE_B + bx;
            atomicMax(&b->hi[0], cx);
            atomicMax(&b->hi[1], cy);
            atomicMax(&b->hi[2], cz);
            atomicMax(&b->lo_rev[0], (uint32_t)g.nx - 1u - cx);
            atomicMax(&b->lo_rev[1], (uint32_t)g.ny - 1u - cy);
            atomicMax(&b->lo_rev[2], (uint32_t)g.nz - 1u - cz);
            atomicAdd(&b->cells, 1u);
            atomicAdd(&b->points, (unsigned long long)cnt);
            if (cube > heavy_cube) atomicAdd(&b->heavy_points, (unsigned long long)cnt);
            else atomicAdd(&b->visits, (unsigned long long)cnt * cube * (cube > 1024u ? 2u : 1u));
        }
    }
}

// Every stride-th point of the columns as columns of their own (x, y, z one after the other).
__global__ __launch_bounds__(GB_THREADS) void strided_sample_kernel(const float *x, const float *y, const float *z,
                                                                    uint32_t stride, uint32_t ms, float *out) {
    const uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x;
    if (i >= ms) return;
    const size_t src = (size_t)i * stride;
    out[i] = x[src];
    out[(size_t)ms + i] = y[src];
    out[2 * (size_t)ms + i] = z[src];
}

// The points of a box of cells of a sorted grid as columns of their own (x, y, z and the index in
// the cloud's columns), row by row: rows[r] = first slot of row r's run of cells, offs[r] = where
// its points go (an exclusive scan of the run lengths, offs[nrows] = their sum).
__global__ __launch_bounds__(GB_THREADS) void box_row_len_kernel(const uint32_t *cell_start, GridGeom g,
                                                                 int x0, int x1, int y0, int ny, int z0,
                                                                 uint32_t nrows, uint32_t *rows,
                                                                 uint32_t *lens) {
    const uint32_t r = blockIdx.x * GB_THREADS + threadIdx.x;
    if (r >= nrows) return;
    const int yy = y0 + (int)(r % (uint32_t)ny), zz = z0 + (int)(r / (uint32_t)ny);
    const uint32_t row = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx;
    const uint32_t s0 = cell_start[row + x0];
    rows[r] = s0;
    lens[r] = cell_start[row + x1 + 1] - s0;
}

__global__ __launch_bounds__(64) void box_gather_kernel(const float4 *spts, const uint32_t *rows,
                                                        const uint32_t *offs, uint32_t nrows,
                                                        uint32_t total, float *ox, float *oy, float *oz,
                                                        uint32_t *oidx) {
    const uint32_t r = blockIdx.x;
    if (r >= nrows) return;
    const uint32_t s0 = rows[r], o0 = offs[r];
    const uint32_t len = (r + 1 < nrows ? offs[r + 1] : total) - o0;
    for (uint32_t t = threadIdx.x; t < len; t += 64) {
        const float4 p = spts[s0 + t];
        ox[o0 + t] = p.x;
        oy[o0 + t] = p.y;
        oz[o0 + t] = p.z;
        oidx[o0 + t] = __float_as_uint(p.w);
    }
}

// spts[i].w (an index into gathered columns) -> the index in the cloud's columns
__global__ __launch_bounds__(GB_THREADS) void remap_index_kernel(float4 *spts, uint32_t n,
                                                                 const uint32_t *index) {
    const uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x;
    if (i < n) spts[i].w = __uint_as_float(index[__float_as_uint(spts[i].w)]);
}

// Puts the points of every cell into ascending original-index order (one thread per cell; the
// scatter's atomics leave them in arrival order).  Used where a reproducible point order matters.
__global__ __launch_bounds__(GB_THREADS) void sort_cells_kernel(const uint32_t *cell_start,
                                                                uint32_t ncell, float4 *spts) {
    const uint32_t c = blockIdx.x * GB_THREADS + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t s0 = cell_start[c], s1 = cell_start[c + 1];
    for (uint32_t j = s0 + 1; j < s1; ++j) {
        const float4 pj = spts[j];
        const uint32_t kj = __float_as_uint(pj.w);
        uint32_t i = j;
        while (i > s0) {
            const float4 pi = spts[i - 1];
            if (__float_as_uint(pi.w) <= kj) break;
            spts[i] = pi;
            --i;
        }
        spts[i] = pj;
    }
}

// Columns of the cell-sorted points plus the permutation (sorted position -> original index).
__global__ __launch_bounds__(GB_THREADS) void unpack_sorted_kernel(const float4 *spts, uint32_t n,
                                                                   float *ox, float *oy, float *oz,
                                                                   uint32_t *perm) {
    const uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 p = spts[i];
    ox[i] = p.x;
    oy[i] = p.y;
    oz[i] = p.z;
    perm[i] = __float_as_uint(p.w);
}

// The same for a list of slots: columns and original indices of the listed points, in list order.
__global__ __launch_bounds__(GB_THREADS) void gather_slots_kernel(const float4 *spts,
                                                                  const uint32_t *list, uint32_t n,
                                                                  float *ox, float *oy, float *oz,
                                                                  uint32_t *perm) {
    const uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 p = spts[list[i]];
    ox[i] = p.x;
    oy[i] = p.y;
    oz[i] = p.z;
    perm[i] = __float_as_uint(p.w);
}

// x, y, z of the sorted points as 12-byte records, once per build
__global__ __launch_bounds__(GB_THREADS) void pack_xyz_kernel(const float4 *spts, uint32_t n,
                                                              float *spts3) {
    const uint32_t i = blockIdx.x * GB_THREADS + threadIdx.x;
    if (i >= n) return;
    const float4 p = spts[i];
    spts3[3 * (size_t)i] = p.x;
    spts3[3 * (size_t)i + 1] = p.y;
    spts3[3 * (size_t)i + 2] = p.z;
}

// Second half of the normal estimation: one thread per query turns the covariance sums left by
// the search kernels into the oriented unit normal.
__global__ __launch_bounds__(GB_THREADS) void normals_from_cov_kernel(
    uint32_t nq, const float *rec, const float *qx, const float *qy, const float *qz, KnnOut out) {
    const uint32_t qi = blockIdx.x * GB_THREADS + threadIdx.x;
    if (qi >= nq) return;
    const float4 r0 = *reinterpret_cast<const float4 *>(rec + (size_t)qi * PCRS_REC_FLOATS);
    const float4 r1 = *reinterpret_cast<const float4 *>(rec + (size_t)qi * PCRS_REC_FLOATS + 4);
    float nnx = 0.0f, nny = 0.0f, nnz = 1.0f;
    const float c00 = r0.x;
    if (c00 >= 0.0f || c00 != c00)
        normal_from_cov(c00, r0.y, r0.z, r0.w, r1.x, r1.y, qx[qi], qy[qi], qz[qi], out.vp, nnx,
                        nny, nnz);
    out.nx[qi] = nnx;
    out.ny[qi] = nny;
    out.nz[qi] = nnz;
    if (out.mean) out.mean[qi] = r1.z;
}

}  // namespace pcrs

using namespace pcrs;

// ----------------------------------------------------------------------------- host: build

namespace {

unsigned grid_for(pcrs_ctx *ctx, size_t n, int threads) {
    size_t blocks = (n + threads - 1) / threads;
    const size_t cap = (size_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

// Finer grids over crowded regions (pcrs_grid_refine); experiments may switch them off:
// PCRS_GRID_LEVELS=0.
static bool levels_enabled() {
    static const bool on = [] {
        const char *e = getenv("PCRS_GRID_LEVELS");
        return !(e && e[0] == '0');
    }();
    return on;
}
// how many times fuller than the average occupied cell the cell of the average point must be
// for a grid to leave its crowded part to a finer level (experiments: PCRS_GRID_LEVEL_RATIO)
static double level_ratio() {
    static const double r = [] {
        const char *e = getenv("PCRS_GRID_LEVEL_RATIO");
        const double v = e ? atof(e) : 0.0;
        return v >= 1.2 && v <= 100.0 ? v : 2.0;
    }();
    return r;
}
constexpr size_t kLevelMinPoints = 65536;  // smaller clouds: the launches of a level cost more than it saves
constexpr int kLevelMaxDepth = 6;

// Points per cell the grid aims for: the 27-cell cube then holds ~27 P candidates and the ball
// of radius one cell ~4.2 P >= 2 K points, so the first stage certifies almost every query.
double target_occupancy(uint32_t k_hint) {
    // experiments only: PCRS_GRID_OCC overrides the points-per-cell target
    if (const char *e = getenv("PCRS_GRID_OCC")) {
        const double v = atof(e);
        if (v >= 0.5 && v <= 256.0) return v;
    }
    double p = 0.42 * (double)(k_hint ? k_hint : 16);
    if (p < 2.0) p = 2.0;
    if (p > 64.0) p = 64.0;
    return p;
}

// Where the sort's four statistics words live behind cell_start[0 .. ncell]: the number of
// non-empty cells, a spare word and the 64-bit sum of squared cell counts (8-byte aligned).
static inline size_t nonempty_slot(uint32_t ncell) { return ((size_t)ncell + 2) & ~(size_t)1; }

// Cells per axis at most (experiments may override it: PCRS_GRID_MAXDIM).  The positions are f32:
// the certification margins allow 2e-6 of the extent for rounding, 1 % of a cell at 4096 cells per
// axis.  (A limit of 1024 made the cells of a flat scene of a hundred million points -- a survey
// flight, 2 km x 2 km -- four times too wide: 454 ms against 208 ms for the search.)
static const double kMaxDim = [] {
    const char *e = getenv("PCRS_GRID_MAXDIM");
    const double v = e ? atof(e) : 0.0;
    return v >= 16.0 && v <= 65536.0 ? v : 4096.0;
}();

// Geometry for a requested cell edge: dims per axis (capped), cell enlarged until the cell count
// fits the budget and every capped axis still covers its extent.
void make_geometry(const double lo[3], const double ext[3], double emax, double cell,
                   double max_cells, GridGeom *g, uint32_t *ncell, double *cell_used) {
    int dims[3];
    for (;;) {
        double total = 1.0;
        for (int a = 0; a < 3; ++a) {
            double d = floor(ext[a] / cell) + 1.0;
            if (d > kMaxDim) d = kMaxDim;
            dims[a] = (int)d;
            total *= d;
        }
        if (total <= max_cells) break;
        cell *= 1.26;
    }
    for (int a = 0; a < 3; ++a)
        if (ext[a] / cell >= (double)dims[a]) cell = ext[a] / (double)dims[a] * (1.0 + 1e-6);
    g->ox = (float)lo[0];
    g->oy = (float)lo[1];
    g->oz = (float)lo[2];
    g->cell = (float)cell;
    g->inv = 1.0f / g->cell;
    g->nx = dims[0];
    g->ny = dims[1];
    g->nz = dims[2];
    g->slack = (float)(2e-6 * (emax + cell));
    for (int a = 0; a < 3; ++a) {  // faces with nothing behind them (the caller opens the others)
        g->open_lo[a] = -__builtin_inff();
        g->open_hi[a] = __builtin_inff();
    }
    *ncell = (uint32_t)dims[0] * (uint32_t)dims[1] * (uint32_t)dims[2];
    *cell_used = cell;
}

// Shape of the two-level sort for this grid; false when the grid has more cells than 16128 coarse
// buckets of 2^15 (the global-atomic counting sort handles those).
// m: points in the columns; n_taken: how many of them the sort will place (fewer when the index
// covers a box of the cloud): the buckets are sized for those, the blocks of level 1 share out m.
bool plan_sort(const GridGeom &g, uint32_t ncell, uint32_t m, SortPlan *sp, uint32_t n_taken = 0) {
    if (n_taken == 0 || n_taken > m) n_taken = m;
    // experiments only: PCRS_GRID_SORT=atomic keeps the one-level sort
    if (const char *e = getenv("PCRS_GRID_SORT"))
        if (e[0] == 'a') return false;
    // Buckets of about 2000 points: the level-2 block then holds its bucket in registers and
    // LDS (one read, one linear write).  Few large buckets leave most of the chip idle and turn
    // level 2 into long serial loops; the table of per-block counts bounds their number.
    uint32_t nb_max = 1024u;
    if (n_taken / 1536u > nb_max) nb_max = n_taken / 1536u < 8192u ? n_taken / 1536u : 8192u;
    if (ncell > (nb_max << FINE_MAX_SHIFT)) nb_max = 8192u;
    // (beyond 8192 buckets of 2^15 cells: 16128 buckets, 63 of the 64 KB of histogram a level-1
    // block can have without asking for more)
    if (ncell > (8192u << FINE_BIG_SHIFT)) nb_max = 16128u;
    uint32_t shift = 0;
    while (((ncell + (1u << shift) - 1) >> shift) > nb_max) ++shift;
    if (shift > FINE_BIG_SHIFT) return false;
    sp->g = g;
    sp->box = pcrs_no_box();
    sp->shift = shift;
    sp->nb = (ncell + (1u << shift) - 1) >> shift;
    sp->ncell = ncell;
    uint32_t nblk = (m + 4095) / 4096;
    // at most about 4 M table entries (nb x nblk), never more than 2048 level-1 blocks
    const uint32_t cap = ((4u << 20) / sp->nb) > 256u ? (4u << 20) / sp->nb : 256u;
    if (nblk > cap) nblk = cap;
    if (nblk > 2048) nblk = 2048;
    if (nblk == 0) nblk = 1;
    uint32_t per = (m + nblk - 1) / nblk;
    per = (per + GB_THREADS - 1) / GB_THREADS * GB_THREADS;
    sp->nblk = (m + per - 1) / per;
    sp->per_blk = per;
    return true;
}

}  // namespace

// The two-level sort of (x, y, z)[0..m) under plan sp: points by cell into `sorted` (x, y, z,
// index), slot ranges into cell_start[0..ncell], occupied-cell count added to *d_nonempty.
// table needs nb * nblk + 1 words (its last word receives the number of finite points),
// table_sums the scan's block sums, tmp one float4 per finite point.
static int run_two_level_sort(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                              uint32_t m, const SortPlan &sp, uint32_t *table, uint32_t *table_sums,
                              float4 *tmp, float4 *sorted, uint32_t *cell_start,
                              uint32_t *d_nonempty, float *sorted3 = nullptr,
                              uint32_t *zero_block = nullptr) {
    // *d_nonempty needs no clearing by the caller: the first kernel does it
    const size_t entries = (size_t)sp.nb * sp.nblk;
    hipLaunchKernelGGL(coarse_hist_kernel, dim3(sp.nblk), dim3(GB_THREADS), sp.nb * sizeof(uint32_t),
                       ctx->stream, x, y, z, m, sp, table, d_nonempty, zero_block);
    if (entries <= SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, table,
                           (uint32_t)entries, table + entries);
    } else {
        const uint32_t tblocks = (uint32_t)((entries + SCAN_BLOCK - 1) / SCAN_BLOCK);
        hipLaunchKernelGGL(scan_block_kernel, dim3(tblocks), dim3(GB_THREADS), 0, ctx->stream, table,
                           (uint32_t)entries, table_sums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, table_sums,
                           tblocks, table + entries);
        hipLaunchKernelGGL(scan_add_kernel, dim3(tblocks), dim3(GB_THREADS), 0, ctx->stream, table,
                           (uint32_t)entries, table_sums);
    }
    hipLaunchKernelGGL(coarse_scatter_kernel, dim3(sp.nblk), dim3(GB_THREADS),
                       sp.nb * sizeof(uint32_t), ctx->stream, x, y, z, m, sp, table, tmp);
    if (sp.shift <= FINE_MAX_SHIFT)
        hipLaunchKernelGGL((fine_sort_kernel<FINE_MAX_SHIFT, GB_THREADS>), dim3(sp.nb), dim3(GB_THREADS), 0,
                           ctx->stream, tmp, sp, table, cell_start, sorted, sorted3, d_nonempty);
    else  // one block per CU (its counters fill the LDS): sixteen waves instead of four
        hipLaunchKernelGGL((fine_sort_kernel<FINE_BIG_SHIFT, 1024>), dim3(sp.nb), dim3(1024), 0, ctx->stream,
                           tmp, sp, table, cell_start, sorted, sorted3, d_nonempty);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_grid_build(pcrs_index *ix, uint32_t k_hint, float cell_request) {
    pcrs_ctx *ctx = ix->ctx;
    // the points to sort: the cloud's columns, or a gathered part of them (see pcrs_index::src_cols)
    const bool from_src = ix->src_n > 0;
    const uint32_t m = (uint32_t)(from_src ? ix->src_n : ix->m);
    const float *bx = from_src ? ix->src_cols.as<float>() : ix->x;
    const float *by = from_src ? ix->src_cols.as<float>() + ix->src_n : ix->y;
    const float *bz = from_src ? ix->src_cols.as<float>() + 2 * ix->src_n : ix->z;
    ix->algo = PCRS_ALGO_BRUTE;
    if (m == 0) return PCRS_OK;

    // 1. bounding box of the finite points
    const unsigned nb = grid_for(ctx, m, GB_THREADS);
    // (the blocks write their partial boxes straight into pinned host memory: no copy command)
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, std::max<size_t>(nb * sizeof(BBoxPartial), 512) + 64, &pin));
    BBoxPartial *hp = reinterpret_cast<BBoxPartial *>(pin);
    const BoxFilter box = ix->box;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    size_t nvalid = 0;
    // A caller that has just measured the cloud (the slab cut of a sharded run) brings bounds that
    // hold every point the index takes, and an upper limit of their number: no pass over the
    // columns for them, the exact count comes back with the first sort's statistics.
    bool count_exact = !ix->have_bounds || ix->bounds_count_exact;
    if (ix->have_bounds) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = ix->bounds_lo[a];
            hi[a] = ix->bounds_hi[a];
        }
        nvalid = ix->bounds_n_upper;
    } else {
        hipLaunchKernelGGL(bbox_kernel, dim3(nb), dim3(GB_THREADS), 0, ctx->stream, bx, by, bz, m, box, hp);
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (unsigned b = 0; b < nb; ++b) {
            const BBoxPartial &p = hp[b];
            nvalid += p.nvalid;
            for (int a = 0; a < 3; ++a) {
                if (p.lo[a] < lo[a]) lo[a] = p.lo[a];
                if (p.hi[a] > hi[a]) hi[a] = p.hi[a];
            }
        }
    }
    ix->nvalid = nvalid;
    if (nvalid < 64) return PCRS_OK;  // brute force
    double ext[3], emax = 0.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = hi[a] - lo[a];
        if (ext[a] > emax) emax = ext[a];
    }
    if (!(emax < 1e30)) return PCRS_OK;  // coordinates whose differences overflow: brute force
    // All finite points in one place: a grid of a single cell.  Every query then sees the whole
    // cloud as its cube, and the searches' shortcut for a list full of zero distances answers the
    // cloud's own points after a handful of candidates (the tiled brute-force kernel would compare
    // every point with every other).
    const bool coincident = !(emax > 0.0);

    // 2. cell size: start from the bounding-box density, then shrink while the occupied cells
    //    are much fuller than the target (surfaces and clusters leave most of the box empty).
    //    Every attempt takes the full histogram with cell ids, so the accepted one is reused.
    const double P = target_occupancy(k_hint);
    static const bool weighted_occupancy = [] {  // experiments only: PCRS_GRID_WEIGHTED=0
        const char *e = getenv("PCRS_GRID_WEIGHTED");
        return !(e && e[0] == '0');
    }();
    double eff[3];
    for (int a = 0; a < 3; ++a) eff[a] = ext[a] > 1e-3 * emax ? ext[a] : 1e-3 * emax;
    double cell = cbrt(eff[0] * eff[1] * eff[2] * P / (double)nvalid);
    // a radius search wants cells of the radius (one attempt, no occupancy feedback)
    const bool fixed_cell = coincident || (cell_request > 0.0f && cell_request < 1e30f);
    if (fixed_cell) cell = coincident ? 1.0 : (double)cell_request;
    // a cloud like the one before it (same list length, size and extent within 30 %) starts from
    // the cell that one ended with, scaled by the change in density a surface would show
    static const bool use_hint = [] {  // experiments only: PCRS_GRID_HINT=0
        const char *e = getenv("PCRS_GRID_HINT");
        return !(e && e[0] == '0');
    }();
    pcrs_ctx::GridHint &gh = ix->level_key ? ctx->level_hints[ix->level_key] : ctx->grid_hint;
    bool hinted = false;
    // (Only for clouds that arrive in host memory: the ranks of a sharded run build their indices
    // over device columns and must all derive the same geometry from the cloud alone, whatever
    // each of them has built before.)
    if (use_hint && (ix->own_x.p || ix->use_hint) && !fixed_cell && gh.cell > 0.0 && gh.k_hint == k_hint &&
        (double)nvalid > 0.7 * (double)gh.n && (double)nvalid < 1.3 * (double)gh.n &&
        emax > 0.7 * gh.emax && emax < 1.3 * gh.emax) {
        cell = gh.cell * pow((double)gh.n / (double)nvalid, 1.0 / gh.dim);
        hinted = true;
    }
    // A cloud of tens of millions of points is sized on a sample first.  The start above takes
    // the cloud for a volume that fills its box; a survey flight does not, and finding that out
    // costs a sort of all its points (8 ms at 100 M).  Every 64th point, gathered into columns of
    // their own, goes through this very function (two or three sorts of 1/64 of the points,
    // a few hundred microseconds); the cell and the dimension it ends with, scaled to the full
    // density, are where the cloud itself starts.  Only a starting point, like the remembered
    // geometry: the build still checks the occupancy it gets.
    double start_dim = hinted ? gh.dim : 3.0;
    static const bool probe_on = [] {  // experiments only: PCRS_GRID_PROBE=0
        const char *e = getenv("PCRS_GRID_PROBE");
        return !(e && e[0] == '0');
    }();
    constexpr uint32_t kProbeKey = 0xFFFFFFFEu, kProbeStride = 64;
    if (probe_on && !hinted && !fixed_cell && !from_src && ix->level_key != kProbeKey && nvalid >= (16u << 20)) {
        pcrs_index probe;
        probe.ctx = ctx;
        probe.x = ix->x;
        probe.y = ix->y;
        probe.z = ix->z;
        probe.m = ix->m;
        probe.k_hint = k_hint;
        probe.box = box;
        probe.level_key = kProbeKey;
        const uint32_t ms = m / kProbeStride;
        PCRS_HIP_CHECK(ctx, probe.src_cols.alloc(ctx, (size_t)ms * 3 * sizeof(float)));
        probe.src_n = ms;
        hipLaunchKernelGGL(strided_sample_kernel, dim3((ms + GB_THREADS - 1) / GB_THREADS), dim3(GB_THREADS), 0,
                           ctx->stream, bx, by, bz, kProbeStride, ms, probe.src_cols.as<float>());
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        pcrs_ctx::GridHint &ph = ctx->level_hints[kProbeKey];
        ph = pcrs_ctx::GridHint{};
        const int prc = pcrs_grid_build(&probe, k_hint, 0.0f);
        if (prc == PCRS_OK && probe.algo == PCRS_ALGO_GRID && ph.cell > 0.0 && ph.n >= 4096) {
            cell = ph.cell * pow((double)ph.n / (double)nvalid, 1.0 / ph.dim);
            start_dim = ph.dim;
            hinted = true;
        }
        // (its pinned read-back area is this build's too: nothing of it is needed any more)
    }
    // cell budget: 4 cells per point; experiments may override the factor
    double cells_per_point = 4.0;
    if (const char *e = getenv("PCRS_GRID_CELLS_PER_POINT")) {
        const double v = atof(e);
        if (v >= 1.0 && v <= 64.0) cells_per_point = v;
    }
    const double max_cells = cells_per_point * (double)nvalid + 4096.0;
    GridGeom g{};
    uint32_t ncell = 0;
    DevBuf counts, cell_id, table, table_sums, tmp, sorted, sorted3;
    // searches for more than one neighbour read an index-free copy of the sorted points: the
    // sort leaves it too (a 1-NN index never needs it)
    const bool want3 = k_hint > 1;
    // sized for the largest grid the budget allows: no reallocation between attempts
    PCRS_HIP_CHECK(ctx, counts.alloc(ctx, ((size_t)max_cells + 8) * sizeof(uint32_t)));
    uint32_t *h_nonempty = reinterpret_cast<uint32_t *>(pin);
    bool sorted_done = false;  // the accepted attempt already left cell_start[] and the points
    // lists of 8 neighbours and more: the aim follows the dimension of the cloud (below)
    static const bool dim_switch = [] {  // experiments only: PCRS_GRID_DIM=0
        const char *e = getenv("PCRS_GRID_DIM");
        return !(e && e[0] == '0');
    }();
    const bool dim_aware = dim_switch && k_hint >= 8 && !getenv("PCRS_GRID_OCC");
    ix->aim = P;
    double P_surface = 0.75 * (double)k_hint;
    P_surface = P_surface < P ? P : (P_surface > 64.0 ? 64.0 : P_surface);
    double dim = start_dim, prev_cell = 0.0, prev_occ = 0.0;
    bool prev_is_seen = false;
    for (int attempt = 0;; ++attempt) {
        double used = cell;
        make_geometry(lo, ext, emax, cell, max_cells, &g, &ncell, &used);
        cell = used;
        // a bounded side of the box is an open face: the cloud may go on behind it
        for (int a = 0; a < 3; ++a) {
            const float o = a == 0 ? g.ox : (a == 1 ? g.oy : g.oz);
            if (box.lo[a] > -__builtin_inff()) g.open_lo[a] = (float)((double)box.lo[a] - (double)o);
            if (box.hi[a] < __builtin_inff()) g.open_hi[a] = (float)((double)box.hi[a] - (double)o);
        }
        uint32_t *d_nonempty = counts.as<uint32_t>() + nonempty_slot(ncell);
        SortPlan sp{};
        const bool two_level = plan_sort(g, ncell, m, &sp, (uint32_t)nvalid);
        sp.box = box;
        if (two_level) {
            // the whole sort is cheaper than one pass of global atomics, so every attempt sorts
            const size_t entries = (size_t)sp.nb * sp.nblk;  // differs between attempts
            if (table.bytes < (entries + 8) * sizeof(uint32_t)) {
                PCRS_HIP_CHECK(ctx, table.alloc(ctx, (entries + entries / 4 + 8) * sizeof(uint32_t)));
                PCRS_HIP_CHECK(ctx, table_sums.alloc(
                                        ctx, ((entries + entries / 4) / SCAN_BLOCK + 2) * sizeof(uint32_t)));
            }
            if (!tmp.p) {
                PCRS_HIP_CHECK(ctx, tmp.alloc(ctx, nvalid * sizeof(float4)));
                PCRS_HIP_CHECK(ctx, sorted.alloc(ctx, nvalid * sizeof(float4)));
                if (want3) PCRS_HIP_CHECK(ctx, sorted3.alloc(ctx, nvalid * 3 * sizeof(float)));
            }
            if (!ix->counters.p) PCRS_HIP_CHECK(ctx, ix->counters.alloc(ctx, 32 * sizeof(uint32_t)));
            ix->counter_phase = 0;
            PCRS_TRY(run_two_level_sort(ctx, bx, by, bz, m, sp, table.as<uint32_t>(),
                                        table_sums.as<uint32_t>(), tmp.as<float4>(),
                                        sorted.as<float4>(), counts.as<uint32_t>(), d_nonempty,
                                        sorted3.as<float>(), ix->counters.as<uint32_t>()));
        } else {
            if (!cell_id.p) PCRS_HIP_CHECK(ctx, cell_id.alloc(ctx, (size_t)m * sizeof(uint32_t)));
            PCRS_HIP_CHECK(ctx, hipMemsetAsync(counts.p, 0, ((size_t)ncell + 6) * sizeof(uint32_t),
                                               ctx->stream));
            hipLaunchKernelGGL(cell_count_kernel, dim3(nb), dim3(GB_THREADS), 0, ctx->stream, bx,
                               by, bz, m, g, box, cell_id.as<uint32_t>(), counts.as<uint32_t>(),
                               d_nonempty);
        }
        sorted_done = two_level;
        if (!count_exact) {  // the number of points the sort placed, behind its table
            if (!two_level)
                return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "grid too large for a build from bounds");
            PCRS_TRY(pcrs_launch_publish(ctx, table.as<uint32_t>() + (size_t)sp.nb * sp.nblk,
                                         h_nonempty + 8, sizeof(uint32_t)));
        }
        if ((attempt == 3 || fixed_cell) && count_exact) break;
        PCRS_TRY(pcrs_launch_publish(ctx, d_nonempty, h_nonempty, 4 * sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!count_exact) {
            nvalid = h_nonempty[8];
            ix->nvalid = nvalid;
            count_exact = true;
            if (nvalid < 64) return PCRS_OK;  // brute force (the sort's buffers go back to the pool)
        }
        if (attempt == 3 || fixed_cell) break;
        const uint32_t nonempty = *h_nonempty;
        double occ = (double)nvalid / (double)(nonempty ? nonempty : 1);
        unsigned long long squares_seen = 0;
        if (two_level) memcpy(&squares_seen, h_nonempty + 2, sizeof(squares_seen));
        // A cloud whose points mostly sit in cells many times fuller than the average occupied
        // cell is not served by one cell size: this grid is then sized for its sparse part (the
        // mean over the occupied cells) and the crowded part gets finer grids of its own
        // (pcrs_grid_refine).
        if (two_level) {  // what the accepted geometry looks like, for pcrs_grid_refine
            ix->occ_mean = occ;
            ix->occ_seen = (double)squares_seen / (double)nvalid - 1.0;
        }
        bool occ_is_seen = false;  // which of the two measures `occ` holds in this attempt
        const bool leave_crowded_to_levels =
            two_level && levels_enabled() && k_hint > 1 && nvalid >= kLevelMinPoints &&
            (double)squares_seen / (double)nvalid - 1.0 > level_ratio() * occ;
        if (two_level && weighted_occupancy && !leave_crowded_to_levels) {
            // What a search pays for is the occupancy a *point* sees, sum(c^2) / n (uniform
            // clouds: one more than the mean).  A scan that is dense near the sensor and sparse
            // far out has a modest mean over its occupied cells and hundreds of points in the
            // cells where most of its points are; the sparse part then takes the retry stage,
            // which is the cheaper of the two.
            unsigned long long squares;
            memcpy(&squares, h_nonempty + 2, sizeof(squares));
            const double seen = (double)squares / (double)nvalid - 1.0;
            if (seen > occ) {
                occ = seen;
                occ_is_seen = true;
            }
        }
        static const bool debug = getenv("PCRS_GRID_DEBUG") != nullptr;  // experiments only
        if (debug)
            fprintf(stderr, "pcrs grid attempt %d: cell %.5g, %d x %d x %d cells, %u occupied, "
                            "mean %.2f points per occupied cell, %.2f seen by a point, aim %.2f (dimension "
                    "%.2f before this attempt)\n",
                    attempt, cell, g.nx, g.ny, g.nz, nonempty, (double)nvalid / (nonempty ? nonempty : 1),
                    occ, P, dim);
        double next;
        if (dim_aware) {
            // How the occupancy answers a change of the cell tells what the cloud is: a volume
            // goes with cell^3, a surface with cell^2.  On a surface the K-th neighbour is
            // farther away for the same number of points per cell (K = pi r^2 rho instead of
            // 4/3 pi r^3 rho), so a grid aimed at the volume figure leaves a surface with cells
            // its 27-cell cubes cannot certify (a flat slab came out at 3.8 points per cell and
            // half of its queries in the retry stage), while the cube of a surface holds a third
            // of the cells: it can afford fuller ones.
            // (both attempts by the same measure: the mean of one against what a point sees in the
            // other says nothing about the cloud)
            if (prev_cell > 0.0 && prev_occ > 0.0 && occ > 0.0 && prev_is_seen == occ_is_seen &&
                (prev_cell >= 1.25 * cell || cell >= 1.25 * prev_cell)) {
                dim = log(prev_occ / occ) / log(prev_cell / cell);
                dim = dim < 2.0 ? 2.0 : (dim > 3.0 ? 3.0 : dim);
            } else if (attempt == 0 && !hinted) {
                // nothing to compare with yet: a grid one or two cells thick, or one whose points
                // leave two thirds of the cells empty, is taken for a surface
                const bool thin = std::min(g.nx, std::min(g.ny, g.nz)) <= 2;
                dim = thin || (double)nonempty < 0.35 * (double)ncell ? 2.0 : 3.0;
            }
            prev_cell = cell;
            prev_occ = occ;
            prev_is_seen = occ_is_seen;
            const double aim = P + (3.0 - dim) * (P_surface - P);
            ix->aim = aim;
            if (occ >= 0.75 * aim && occ <= 1.6 * aim) break;
            double f = pow(aim / occ, 1.0 / dim);
            f = f < 0.2 ? 0.2 : (f > 2.5 ? 2.5 : f);
            next = cell * f;
        } else {
            if (occ <= 1.6 * P) break;
            // occupied volume ~ nonempty * cell^3: re-aim at P points per occupied cell.  A
            // surface responds with cell^2 rather than cell^3, hence the repeated attempts.
            double shrink = cbrt(P / occ);
            if (shrink < 0.4) shrink = 0.4;
            next = cell * shrink;
        }
        if (next < emax / kMaxDim) next = emax / kMaxDim;
        GridGeom g2{};
        uint32_t ncell2 = 0;
        double used2 = next;
        make_geometry(lo, ext, emax, next, max_cells, &g2, &ncell2, &used2);
        // budget or dimension cap reached: keep this geometry
        if (next < cell ? used2 > 0.95 * cell : used2 < 1.05 * cell) break;
        cell = next;
    }

    if (sorted_done) {
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        ix->spts.release();
        ix->spts.p = sorted.p;
        ix->spts.bytes = sorted.bytes;
        ix->spts.owner = sorted.owner;
        sorted.p = nullptr;
        sorted.bytes = 0;
        sorted.owner = nullptr;
        ix->spts3.release();
        ix->spts3.p = sorted3.p;
        ix->spts3.bytes = sorted3.bytes;
        ix->spts3.owner = sorted3.owner;
        sorted3.p = nullptr;
        sorted3.bytes = 0;
        sorted3.owner = nullptr;
    } else {
        ix->spts3.release();  // made by the first search that wants it (pcrs_launch_knn_grid)
        // scan the histogram into slot ranges and scatter the points.  All buffers come from the
        // context's pool, so the temporaries are released without waiting for the stream.
        DevBuf scan_sums, fill;
        const uint32_t nblocks = (ncell + SCAN_BLOCK - 1) / SCAN_BLOCK;
        PCRS_HIP_CHECK(ctx, scan_sums.alloc(ctx, ((size_t)nblocks + 1) * sizeof(uint32_t)));
        hipLaunchKernelGGL(scan_block_kernel, dim3(nblocks), dim3(GB_THREADS), 0, ctx->stream,
                           counts.as<uint32_t>(), ncell, scan_sums.as<uint32_t>());
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                           scan_sums.as<uint32_t>(), nblocks, counts.as<uint32_t>() + ncell);
        hipLaunchKernelGGL(scan_add_kernel, dim3(nblocks), dim3(GB_THREADS), 0, ctx->stream,
                           counts.as<uint32_t>(), ncell, scan_sums.as<uint32_t>());
        PCRS_HIP_CHECK(ctx, fill.alloc(ctx, (size_t)ncell * sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(fill.p, counts.p, (size_t)ncell * sizeof(uint32_t),
                                           hipMemcpyDeviceToDevice, ctx->stream));
        PCRS_HIP_CHECK(ctx, ix->spts.alloc(ctx, nvalid * sizeof(float4)));
        hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(GB_THREADS), 0, ctx->stream, bx, by,
                           bz, m, cell_id.as<uint32_t>(), fill.as<uint32_t>(),
                           ix->spts.as<float4>());
        PCRS_HIP_CHECK(ctx, hipGetLastError());
    }
    ix->cell_start.release();
    ix->cell_start.p = counts.p;  // take ownership of the scanned histogram
    ix->cell_start.bytes = counts.bytes;
    ix->cell_start.owner = counts.owner;
    counts.p = nullptr;
    counts.bytes = 0;
    counts.owner = nullptr;
    ix->g = g;
    ix->ncell = ncell;
    if (!fixed_cell) {
        gh.cell = cell;
        gh.emax = emax;
        gh.dim = dim;
        gh.n = nvalid;
        gh.k_hint = k_hint;
    }
    if (from_src) {  // the sort numbered the gathered points: back to the cloud's own indices
        if (ix->src_index.p)  // (a sample taken to size a grid has none: nobody searches it)
            hipLaunchKernelGGL(remap_index_kernel, dim3((unsigned)((nvalid + GB_THREADS - 1) / GB_THREADS)),
                               dim3(GB_THREADS), 0, ctx->stream, ix->spts.as<float4>(), (uint32_t)nvalid,
                               ix->src_index.as<uint32_t>());
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        ix->src_cols.release();
        ix->src_index.release();
        ix->src_n = 0;
    }
    ix->algo = PCRS_ALGO_GRID;
    return PCRS_OK;
}

// Cuts a cloud into `parts` slabs of (nearly) equal point counts along its longest axis: slab r
// holds the finite points with cuts[r] <= p[axis] < cuts[r + 1] (cuts[0] = -inf, cuts[parts] =
// +inf).  A pure function of the columns, so every process that holds the same cloud derives the
// same cuts.  halo: how far beyond its slab an index should reach so that the neighbourhoods of
// nearly all of the slab's points lie inside it (a few times the distance within which a point of
// a cloud of this density finds k neighbours).
int pcrs_slab_cuts(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t m,
                   uint32_t parts, uint32_t k_hint, int *axis_out, float *cuts, float *halo,
                   size_t *n_finite, SlabMeasure *measure) {
    for (uint32_t r = 0; r <= parts; ++r) cuts[r] = r == 0 ? -INFINITY : INFINITY;
    *axis_out = 0;
    *halo = 0.0f;
    *n_finite = 0;
    if (m == 0 || parts == 0) return PCRS_OK;
    if (m >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    const unsigned nb = grid_for(ctx, m, GB_THREADS);
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, std::max<size_t>(nb * sizeof(BBoxPartial), SLAB_BINS * sizeof(uint32_t)) + 64, &pin));
    BBoxPartial *hp = reinterpret_cast<BBoxPartial *>(pin);
    hipLaunchKernelGGL(bbox_kernel, dim3(nb), dim3(GB_THREADS), 0, ctx->stream, x, y, z, (uint32_t)m,
                       pcrs_no_box(), hp);
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    size_t nvalid = 0;
    for (unsigned b = 0; b < nb; ++b) {
        nvalid += hp[b].nvalid;
        for (int a = 0; a < 3; ++a) {
            if (hp[b].lo[a] < lo[a]) lo[a] = hp[b].lo[a];
            if (hp[b].hi[a] > hi[a]) hi[a] = hp[b].hi[a];
        }
    }
    *n_finite = nvalid;
    if (nvalid == 0) return PCRS_OK;
    int axis = 0;
    double ext[3], emax = 0.0, vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = hi[a] - lo[a];
        if (ext[a] > emax) {
            emax = ext[a];
            axis = a;
        }
    }
    *axis_out = axis;
    if (!(emax > 0.0) || !(emax < 1e30)) return PCRS_OK;  // one slab holds everything
    for (int a = 0; a < 3; ++a) vol *= ext[a] > 1e-3 * emax ? ext[a] : 1e-3 * emax;
    const double kk = (double)(k_hint ? k_hint : 16);
    *halo = (float)(2.0 * cbrt(vol * kk / (double)nvalid));
    DevBuf hist;
    PCRS_HIP_CHECK(ctx, hist.alloc(ctx, SLAB_BINS * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipMemsetAsync(hist.p, 0, SLAB_BINS * sizeof(uint32_t), ctx->stream));
    const float flo = (float)lo[axis];
    const float width = (float)(emax / SLAB_BINS * (1.0 + 1e-6));
    const float inv = 1.0f / width;
    hipLaunchKernelGGL(axis_hist_kernel, dim3(std::min(nb, 512u)), dim3(GB_THREADS), 0, ctx->stream, x,
                       y, z, (uint32_t)m, axis, flo, inv, hist.as<uint32_t>());
    PCRS_TRY(pcrs_launch_publish(ctx, hist.p, pin, SLAB_BINS * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t *h = reinterpret_cast<const uint32_t *>(pin);
    if (measure) {
        for (int a = 0; a < 3; ++a) {
            measure->lo[a] = lo[a];
            measure->hi[a] = hi[a];
        }
        measure->hist_lo = flo;
        measure->hist_width = width;
        measure->hist.assign(h, h + SLAB_BINS);
    }
    // cut r at the lower edge of the first bin whose running count reaches r * n / parts
    size_t run = 0;
    uint32_t r = 1;
    for (int b = 0; b < SLAB_BINS && r < parts; ++b) {
        while (r < parts && run >= (size_t)(((unsigned long long)r * nvalid) / parts)) {
            cuts[r] = flo + (float)b * width;
            ++r;
        }
        run += h[b];
    }
    for (; r < parts; ++r) cuts[r] = INFINITY;  // (empty slabs at the end of a degenerate cloud)
    for (uint32_t q = 1; q < parts; ++q)  // monotone, whatever the rounding of the edges did
        if (cuts[q] < cuts[q - 1]) cuts[q] = cuts[q - 1];
    return PCRS_OK;
}

// Where the cells of a grid are crowded and where its cubes are short of points: one pass over the
// cells, one read-back.
static int scan_cells(pcrs_index *ix, uint32_t threshold, uint32_t k_short, DenseScan *out) {
    pcrs_ctx *ctx = ix->ctx;
    DevBuf d;
    PCRS_HIP_CHECK(ctx, d.alloc(ctx, sizeof(DenseScan)));
    PCRS_HIP_CHECK(ctx, hipMemsetAsync(d.p, 0, sizeof(DenseScan), ctx->stream));
    hipLaunchKernelGGL(dense_scan_kernel, dim3(grid_for(ctx, ix->ncell, GB_THREADS)), dim3(GB_THREADS), 0,
                       ctx->stream, ix->cell_start.as<uint32_t>(), ix->g, ix->ncell, threshold, k_short,
                       d.as<DenseScan>());
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, sizeof(DenseScan) + 64, &pin));
    PCRS_TRY(pcrs_launch_publish(ctx, d.p, pin, sizeof(DenseScan)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = *reinterpret_cast<const DenseScan *>(pin);
    return PCRS_OK;
}

// ---- a grid of cells three times as wide, out of the cells of the grid it stands behind
// Same origin, so a wide cell is 3 x 3 x 3 of the fine ones and its points are nine runs of up to
// three consecutive fine cells in the sorted array: count, scan, copy -- no sort.  (A sort of the
// same points by cells 27 times as full leaves a few buckets with most of the points: 0.2 ms for
// the coarser grid of a ring scan of a million points.)  A point's wide cell is the one its fine
// cell lies in, which can differ from the cell its coordinates compute to by the rounding of one
// multiplication: the same slack that every certificate already allows for a point's own cell.
__global__ __launch_bounds__(GB_THREADS) void wide_count_kernel(const uint32_t *cell_start, GridGeom g,
                                                                GridGeom w, uint32_t wcells, uint32_t *count) {
    const uint32_t c = blockIdx.x * GB_THREADS + threadIdx.x;
    if (c >= wcells) return;
    const int X = (int)(c % (uint32_t)w.nx), Y = (int)((c / (uint32_t)w.nx) % (uint32_t)w.ny);
    const int Z = (int)(c / ((uint32_t)w.nx * (uint32_t)w.ny));
    uint32_t total = 0;
    const int x0 = 3 * X, x1 = min(3 * X + 2, g.nx - 1);
    if (x0 < g.nx)
        for (int z = 3 * Z; z <= min(3 * Z + 2, g.nz - 1); ++z)
            for (int y = 3 * Y; y <= min(3 * Y + 2, g.ny - 1); ++y) {
                const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
                total += cell_start[row + x1 + 1] - cell_start[row + x0];
            }
    count[c] = total;
}

// One thread per slot of the wide grid's sorted array: its wide cell by bisection of the slot
// ranges, then the run of fine cells it falls into.
__global__ __launch_bounds__(GB_THREADS) void wide_gather_kernel(const uint32_t *cell_start, GridGeom g,
                                                                 const float4 *spts, GridGeom w, uint32_t wcells,
                                                                 const uint32_t *wstart, uint32_t n,
                                                                 float4 *out, float *out3) {
    const uint32_t s = blockIdx.x * GB_THREADS + threadIdx.x;
    if (s >= n) return;
    uint32_t lo = 0, hi = wcells;  // the last cell with wstart[c] <= s
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (wstart[mid] <= s) lo = mid; else hi = mid;
    }
    const uint32_t c = lo;
    uint32_t local = s - wstart[c];
    const int X = (int)(c % (uint32_t)w.nx), Y = (int)((c / (uint32_t)w.nx) % (uint32_t)w.ny);
    const int Z = (int)(c / ((uint32_t)w.nx * (uint32_t)w.ny));
    const int x0 = 3 * X, x1 = min(3 * X + 2, g.nx - 1);
    uint32_t src = 0xFFFFFFFFu;
    for (int z = 3 * Z; z <= min(3 * Z + 2, g.nz - 1) && src == 0xFFFFFFFFu; ++z)
        for (int y = 3 * Y; y <= min(3 * Y + 2, g.ny - 1); ++y) {
            const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
            const uint32_t b = cell_start[row + x0], len = cell_start[row + x1 + 1] - b;
            if (local < len) {
                src = b + local;
                break;
            }
            local -= len;
        }
    if (src == 0xFFFFFFFFu) return;  // cannot happen: the counts are the sums of these runs
    const float4 p = spts[src];
    out[s] = p;
    if (out3) {
        out3[(size_t)s * 3] = p.x;
        out3[(size_t)s * 3 + 1] = p.y;
        out3[(size_t)s * 3 + 2] = p.z;
    }
}

// Fills `c` (bounds, box and columns set by the caller) with the grid of cells three times as wide
// as ix's.  *done = false when the geometry does not come out as 3 x 3 x 3 of ix's cells (the
// caller then sorts).
static int wide_grid_from_cells(pcrs_index *ix, pcrs_index *c, bool *done) {
    pcrs_ctx *ctx = ix->ctx;
    const GridGeom &g = ix->g;
    *done = false;
    double lo[3], ext[3], emax = 0.0;
    for (int a = 0; a < 3; ++a) {
        lo[a] = c->bounds_lo[a];
        ext[a] = c->bounds_hi[a] - c->bounds_lo[a];
        emax = std::max(emax, ext[a]);
    }
    GridGeom w{};
    uint32_t wcells = 0;
    const double want = (double)(3.0f * g.cell);
    double used = want;
    make_geometry(lo, ext, emax, want, 4.0 * (double)ix->nvalid + 4096.0, &w, &wcells, &used);
    if (used != want || w.ox != g.ox || w.oy != g.oy || w.oz != g.oz || 3 * w.nx < g.nx || 3 * w.ny < g.ny ||
        3 * w.nz < g.nz || ix->nvalid >= 0xFFFFFFFFull)
        return PCRS_OK;
    for (int a = 0; a < 3; ++a) {  // the same origin: the same faces are open
        w.open_lo[a] = g.open_lo[a];
        w.open_hi[a] = g.open_hi[a];
    }
    const uint32_t n = (uint32_t)ix->nvalid;
    DevBuf sums;
    PCRS_HIP_CHECK(ctx, c->cell_start.alloc(ctx, (nonempty_slot(wcells) + 8) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, ((size_t)wcells / 1024 + 4) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, c->spts.alloc(ctx, (size_t)n * sizeof(float4)));
    PCRS_HIP_CHECK(ctx, c->spts3.alloc(ctx, (size_t)n * 3 * sizeof(float)));
    uint32_t *wstart = c->cell_start.as<uint32_t>();
    hipLaunchKernelGGL(wide_count_kernel, dim3((wcells + GB_THREADS - 1) / GB_THREADS), dim3(GB_THREADS), 0,
                       ctx->stream, ix->cell_start.as<uint32_t>(), g, w, wcells, wstart);
    PCRS_TRY(pcrs_launch_scan_u32(ctx, wstart, wcells, sums.as<uint32_t>(), wstart + wcells));
    hipLaunchKernelGGL(wide_gather_kernel, dim3((n + GB_THREADS - 1) / GB_THREADS), dim3(GB_THREADS), 0,
                       ctx->stream, ix->cell_start.as<uint32_t>(), g, ix->spts.as<float4>(), w, wcells, wstart, n,
                       c->spts.as<float4>(), c->spts3.as<float>());
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    c->g = w;
    c->ncell = wcells;
    c->nvalid = ix->nvalid;
    c->aim = ix->aim;
    c->algo = PCRS_ALGO_GRID;
    *done = true;
    return PCRS_OK;
}

constexpr int kCoarseMaxDepth = 3;
static bool coarse_enabled() {
    static const bool on = [] {  // experiments only: PCRS_GRID_COARSE=0
        const char *e = getenv("PCRS_GRID_COARSE");
        return !(e && e[0] == '0');
    }();
    return on;
}

// A coarser grid behind `ix` when enough of its points sit in cubes that cannot hold their
// neighbours (short_points of scan_cells), and another behind that one while the same holds there.
static int grid_coarsen(pcrs_index *ix, uint32_t k_hint, unsigned long long short_points, int depth) {
    pcrs_ctx *ctx = ix->ctx;
    const GridGeom &g = ix->g;
    if (!coarse_enabled() || depth >= kCoarseMaxDepth) return PCRS_OK;
    static const unsigned long long coarse_min = [] {  // experiments only: PCRS_GRID_COARSE_MIN=<points>
        const char *e = getenv("PCRS_GRID_COARSE_MIN");
        return e ? (unsigned long long)atoll(e) : 8192ull;
    }();
    static const unsigned long long coarse_part = [] {  // experiments only: PCRS_GRID_COARSE_PART=<1/part of the points>
        const char *e = getenv("PCRS_GRID_COARSE_PART");
        return e && atoll(e) > 0 ? (unsigned long long)atoll(e) : 64ull;
    }();
    if (short_points < std::max<unsigned long long>(coarse_min, ix->nvalid / coarse_part)) return PCRS_OK;
    if (std::max(g.nx, std::max(g.ny, g.nz)) < 12) return PCRS_OK;  // a handful of cells already
    pcrs_index *c = new (std::nothrow) pcrs_index();
    if (!c) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    c->ctx = ctx;
    c->x = ix->x;
    c->y = ix->y;
    c->z = ix->z;
    c->m = ix->m;
    c->k_hint = k_hint;
    c->box = ix->box;
    // the same points: their bounds and number are known (no pass over the columns, no read-back)
    const double o[3] = {g.ox, g.oy, g.oz};
    const int n[3] = {g.nx, g.ny, g.nz};
    c->have_bounds = true;
    c->bounds_count_exact = true;
    c->bounds_n_upper = ix->nvalid;
    for (int a = 0; a < 3; ++a) {
        c->bounds_lo[a] = o[a];
        c->bounds_hi[a] = o[a] + (double)n[a] * (double)g.cell;
    }
    static const float coarse_ratio = [] {  // experiments only: PCRS_GRID_COARSE_RATIO
        const char *e = getenv("PCRS_GRID_COARSE_RATIO");
        const float v = e ? (float)atof(e) : 0.0f;
        return v >= 1.6f && v <= 8.0f ? v : 3.0f;
    }();
    bool from_cells = false;
    static const bool wide_from_cells = [] {  // experiments only: PCRS_GRID_COARSE_SORT=1 sorts instead
        const char *e = getenv("PCRS_GRID_COARSE_SORT");
        return !(e && e[0] == '1');
    }();
    int rc = PCRS_OK;
    if (coarse_ratio == 3.0f && wide_from_cells) rc = wide_grid_from_cells(ix, c, &from_cells);
    if (rc == PCRS_OK && !from_cells) rc = pcrs_grid_build(c, k_hint, coarse_ratio * g.cell);
    static const bool debug = getenv("PCRS_GRID_DEBUG") != nullptr;  // experiments only
    if (debug)
        fprintf(stderr, "pcrs grid coarse level %d: %llu of %zu points in cubes of fewer than %u points; "
                        "coarser grid with cell %.5g (this level %.5g)\n",
                depth + 1, short_points, ix->nvalid, k_hint, (double)c->g.cell, (double)g.cell);
    if (rc != PCRS_OK || c->algo != PCRS_ALGO_GRID || !(c->g.cell > 1.5f * g.cell)) {
        delete c;
        return rc;
    }
    ix->coarse = c;
    DenseScan ds{};
    PCRS_TRY(scan_cells(c, 0xFFFFFFFFu, k_hint, &ds));
    return grid_coarsen(c, k_hint, ds.short_points, depth + 1);
}

int pcrs_grid_refine(pcrs_index *ix, uint32_t k_hint, int depth) {
    pcrs_ctx *ctx = ix->ctx;
    if (!levels_enabled() || ix->algo != PCRS_ALGO_GRID || k_hint <= 1 || depth >= kLevelMaxDepth ||
        ix->nvalid < kLevelMinPoints || !(ix->aim > 0.0))
        return PCRS_OK;
    // Cells of much the same fullness (what a point sees is close to the mean over the occupied
    // cells, as in any evenly sampled volume or surface) hold no crowded region worth a level: not
    // even a look at the cells.  (Both figures are those of the accepted sort; a build that never
    // read them back has none and is looked at.)
    if (ix->occ_mean > 0.0 && ix->occ_seen < 2.0 * ix->occ_mean + 2.0) return PCRS_OK;
    const GridGeom &g = ix->g;
    const uint32_t threshold = (uint32_t)std::ceil(4.0 * ix->aim);
    const bool top = depth == 0 && ix->box.unbounded();
    DenseScan ds{};
    PCRS_TRY(scan_cells(ix, threshold, top ? k_hint : 0u, &ds));
    if (top) PCRS_TRY(grid_coarsen(ix, k_hint, ds.short_points, 0));
    static const bool debug = getenv("PCRS_GRID_DEBUG") != nullptr;  // experiments only
    const size_t enough = std::max<size_t>(16384, ix->nvalid / 20);
    if (ds.cells == 0 || ds.points < enough) return PCRS_OK;
    const int n[3] = {g.nx, g.ny, g.nz};
    struct Region {
        int lo[3], hi[3];
        unsigned long long points, heavy_points, visits;
        uint32_t cells;
    };
    std::vector<Region> regions;
    double whole = 1.0;
    for (int a = 0; a < 3; ++a) whole *= (double)n[a];
    auto volume = [](const Region &r) {
        double v = 1.0;
        for (int a = 0; a < 3; ++a) v *= (double)(r.hi[a] - r.lo[a] + 1);
        return v;
    };
    {
        Region one{};
        for (int a = 0; a < 3; ++a) {
            one.hi[a] = (int)ds.hi[a];
            one.lo[a] = n[a] - 1 - (int)ds.lo_rev[a];
        }
        one.points = ds.points;
        one.cells = ds.cells;
        // one box, if it is a small part of the grid and the crowded cells are not mere specks in
        // it (four clusters in four corners of a field: eighteen cells in a box of twelve thousand)
        if (volume(one) <= 0.3 * whole && (volume(one) <= 64.0 || (double)one.cells >= 0.01 * volume(one)))
            regions.push_back(one);
    }
    if (regions.empty()) {
        // Crowded cells all over the grid (the roofs and trees of a survey, the cars of a street):
        // one box around them is the grid again.  They are looked at block by block, neighbouring
        // blocks with crowded cells are joined into regions, and every region that is worth a
        // grid of its own gets one.
        // Where the lowest bar makes one region of everything (flat patches of a terrain are
        // crowded by it too) a bar two or four times as high is tried: all three are counted in
        // the one pass.
        constexpr size_t nblk = (size_t)DENSE_B * DENSE_B * DENSE_B;
        DevBuf d;
        PCRS_HIP_CHECK(ctx, d.alloc(ctx, DENSE_T * nblk * sizeof(DenseBlock)));
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(d.p, 0, DENSE_T * nblk * sizeof(DenseBlock), ctx->stream));
        hipLaunchKernelGGL(dense_blocks_kernel, dim3(grid_for(ctx, ix->ncell, GB_THREADS)), dim3(GB_THREADS), 0,
                           ctx->stream, ix->cell_start.as<uint32_t>(), g, ix->ncell, threshold, 8192u,
                           d.as<DenseBlock>());
        std::vector<DenseBlock> hall(DENSE_T * nblk);
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(hall.data(), d.p, DENSE_T * nblk * sizeof(DenseBlock),
                                           hipMemcpyDeviceToHost, ctx->stream));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (int t = 0; t < DENSE_T && regions.empty(); ++t) {
            const DenseBlock *hb = hall.data() + (size_t)t * nblk;
            std::vector<int> label(nblk, -1);
            std::vector<size_t> stack;
            for (size_t b0 = 0; b0 < nblk; ++b0) {
                if (!hb[b0].cells || label[b0] >= 0) continue;
                Region r{};
                for (int a = 0; a < 3; ++a) {
                    r.lo[a] = n[a];
                    r.hi[a] = -1;
                }
                label[b0] = (int)regions.size();
                stack.assign(1, b0);
                while (!stack.empty()) {
                    const size_t b = stack.back();
                    stack.pop_back();
                    const DenseBlock &k = hb[b];
                    for (int a = 0; a < 3; ++a) {
                        r.lo[a] = std::min(r.lo[a], n[a] - 1 - (int)k.lo_rev[a]);
                        r.hi[a] = std::max(r.hi[a], (int)k.hi[a]);
                    }
                    r.points += k.points;
                    r.heavy_points += k.heavy_points;
                    r.visits += k.visits;
                    r.cells += k.cells;
                    const int bx = (int)(b % DENSE_B), by = (int)((b / DENSE_B) % DENSE_B),
                              bz = (int)(b / (DENSE_B * DENSE_B));
                    for (int dz = -1; dz <= 1; ++dz)
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                const int x2 = bx + dx, y2 = by + dy, z2 = bz + dz;
                                if (x2 < 0 || y2 < 0 || z2 < 0 || x2 >= DENSE_B || y2 >= DENSE_B || z2 >= DENSE_B)
                                    continue;
                                const size_t nb2 = ((size_t)z2 * DENSE_B + y2) * DENSE_B + x2;
                                if (hb[nb2].cells && label[nb2] < 0) {
                                    label[nb2] = label[b0];
                                    stack.push_back(nb2);
                                }
                            }
                }
                regions.push_back(r);
            }
            // the boxes of two regions may still overlap (an L around a corner): such regions are one
            for (bool merged = true; merged;) {
                merged = false;
                for (size_t i = 0; i < regions.size() && !merged; ++i)
                    for (size_t j = i + 1; j < regions.size() && !merged; ++j) {
                        bool apart = false;
                        for (int a = 0; a < 3; ++a)
                            apart = apart || regions[i].hi[a] + 1 < regions[j].lo[a] - 1 ||
                                    regions[j].hi[a] + 1 < regions[i].lo[a] - 1;
                        if (apart) continue;
                        for (int a = 0; a < 3; ++a) {
                            regions[i].lo[a] = std::min(regions[i].lo[a], regions[j].lo[a]);
                            regions[i].hi[a] = std::max(regions[i].hi[a], regions[j].hi[a]);
                        }
                        regions[i].points += regions[j].points;
                        regions[i].heavy_points += regions[j].heavy_points;
                        regions[i].visits += regions[j].visits;
                        regions[i].cells += regions[j].cells;
                        regions.erase(regions.begin() + (long)j);
                        merged = true;
                    }
            }
            // A grid of its own costs a region about half a millisecond (a gather, a sort or two,
            // five read-backs).  It pays where the region's cubes are too crowded for a lane (more
            // than 8192 points: every query there takes the wave-per-query kernel, some 30 ns
            // each), or where the candidates the lanes would read here (1.6 ps each: 0.3 ns for
            // the 190 of an evenly filled cube; cubes of more than 1024 are walked twice) cost
            // three quarters of a millisecond more than in a grid of the region's own.  The
            // largest first; together a small part of the grid.
            static const double region_ms = [] {  // experiments only: PCRS_GRID_REGION_MS=<ms>
                const char *e = getenv("PCRS_GRID_REGION_MS");
                return e ? atof(e) : 0.75;
            }();
            static const unsigned long long region_heavy = [] {  // experiments only
                const char *e = getenv("PCRS_GRID_REGION_HEAVY");
                return e ? (unsigned long long)atoll(e) : 8192ull;
            }();
            for (size_t i = 0; i < regions.size();) {
                const Region &q = regions[i];
                const double lane_ms = (double)q.visits * 1.6e-9 - (double)q.points * 0.3e-6;
                const bool pays = q.points >= 16384 && (q.heavy_points >= region_heavy || lane_ms >= region_ms);
                if (pays) ++i; else regions.erase(regions.begin() + (long)i);
            }
            std::sort(regions.begin(), regions.end(),
                      [](const Region &a, const Region &b) { return a.points > b.points; });
            while (regions.size() > (size_t)PCRS_MAX_KIDS) regions.pop_back();
            unsigned long long sum = 0;
            double vsum = 0.0;
            for (const Region &r : regions) {
                sum += r.points;
                vsum += volume(r);
            }
            if (regions.empty() || vsum > 0.3 * whole) regions.clear();
            (void)sum;
        }
        if (regions.empty()) return PCRS_OK;
    }
    const double o[3] = {g.ox, g.oy, g.oz};
    int rc_all = PCRS_OK;
    for (const Region &r : regions) {
        BoxFilter qbox = pcrs_no_box(), pbox = ix->box;
        for (int a = 0; a < 3; ++a) {
            // a side on the grid's own end has nothing of this grid behind it
            if (r.lo[a] > 0) {
                qbox.lo[a] = (float)(o[a] + (double)r.lo[a] * (double)g.cell);
                pbox.lo[a] = (float)(o[a] + (double)(r.lo[a] - 1) * (double)g.cell);
            }
            if (r.hi[a] < n[a] - 1) {
                qbox.hi[a] = (float)(o[a] + (double)(r.hi[a] + 1) * (double)g.cell);
                pbox.hi[a] = (float)(o[a] + (double)(r.hi[a] + 2) * (double)g.cell);
            }
        }
        pcrs_index *child = new (std::nothrow) pcrs_index();
        if (!child) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
        child->ctx = ctx;
        child->x = ix->x;
        child->y = ix->y;
        child->z = ix->z;
        child->m = ix->m;
        child->k_hint = k_hint;
        child->box = pbox;
        // a finer grid of a cloud like the one before it starts from the cell that one's grid at
        // this place in the tree ended with (only a starting point, as for the index itself)
        child->level_key = ix->level_key * 31u + (uint32_t)(&r - regions.data()) + 1u;
        child->use_hint = true;
        // (only where the region is a small part of the cloud: gathering costs a read-back of its
        // own, and three passes over a cloud of a million points cost less than that)
        if ((double)r.points * 8.0 < (double)ix->m) {
            // the child's points are gathered from this grid's cells around the box (two cells
            // more than the box on every side: the box's own margin of one, and one for points
            // whose cell the rounding of the cell assignment put next door); the child's sort then
            // reads those instead of passing over the whole cloud three times an attempt
            const int x0 = std::max(r.lo[0] - 2, 0), x1 = std::min(r.hi[0] + 2, n[0] - 1);
            const int y0 = std::max(r.lo[1] - 2, 0), y1 = std::min(r.hi[1] + 2, n[1] - 1);
            const int z0 = std::max(r.lo[2] - 2, 0), z1 = std::min(r.hi[2] + 2, n[2] - 1);
            const uint32_t nrows = (uint32_t)(y1 - y0 + 1) * (uint32_t)(z1 - z0 + 1);
            DevBuf rows, lens, sums;
            PCRS_HIP_CHECK(ctx, rows.alloc(ctx, ((size_t)nrows + 2) * sizeof(uint32_t)));
            PCRS_HIP_CHECK(ctx, lens.alloc(ctx, ((size_t)nrows + 2) * sizeof(uint32_t)));
            PCRS_HIP_CHECK(ctx, sums.alloc(ctx, ((size_t)nrows / 1024 + 4) * sizeof(uint32_t)));
            hipLaunchKernelGGL(box_row_len_kernel, dim3((nrows + GB_THREADS - 1) / GB_THREADS), dim3(GB_THREADS), 0,
                               ctx->stream, ix->cell_start.as<uint32_t>(), g, x0, x1, y0, y1 - y0 + 1, z0, nrows,
                               rows.as<uint32_t>(), lens.as<uint32_t>());
            PCRS_TRY(pcrs_launch_scan_u32(ctx, lens.as<uint32_t>(), nrows, sums.as<uint32_t>(),
                                          lens.as<uint32_t>() + nrows));
            void *pin = nullptr;
            PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
            PCRS_TRY(pcrs_launch_publish(ctx, lens.as<uint32_t>() + nrows, pin, sizeof(uint32_t)));
            PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            const uint32_t total = *reinterpret_cast<const uint32_t *>(pin);
            if (total < 16384) {
                delete child;
                continue;
            }
            PCRS_HIP_CHECK(ctx, child->src_cols.alloc(ctx, (size_t)total * 3 * sizeof(float)));
            PCRS_HIP_CHECK(ctx, child->src_index.alloc(ctx, (size_t)total * sizeof(uint32_t)));
            child->src_n = total;
            hipLaunchKernelGGL(box_gather_kernel, dim3(nrows), dim3(64), 0, ctx->stream, ix->spts.as<float4>(),
                               rows.as<uint32_t>(), lens.as<uint32_t>(), nrows, total,
                               child->src_cols.as<float>(), child->src_cols.as<float>() + total,
                               child->src_cols.as<float>() + 2 * (size_t)total, child->src_index.as<uint32_t>());
            PCRS_HIP_CHECK(ctx, hipGetLastError());
        }
        const int rc = pcrs_grid_build(child, k_hint);
        if (debug)
            fprintf(stderr, "pcrs grid level %d: %u crowded cells (> %u points) with %llu points in a box of "
                            "%d x %d x %d cells; finer grid over %zu points, cell %.5g (this level %.5g)\n",
                    depth + 1, r.cells, threshold, r.points, r.hi[0] - r.lo[0] + 1, r.hi[1] - r.lo[1] + 1,
                    r.hi[2] - r.lo[2] + 1, child->nvalid, (double)child->g.cell, (double)g.cell);
        // a level must be a real step: enough points, markedly smaller cells
        if (rc != PCRS_OK || child->algo != PCRS_ALGO_GRID || child->nvalid < 16384 ||
            !(child->g.cell < 0.8f * g.cell)) {
            delete child;
            if (rc != PCRS_OK) rc_all = rc;
            continue;
        }
        ix->kids.push_back(child);
        ix->kid_qbox.push_back(qbox);
        PCRS_TRY(pcrs_grid_refine(child, k_hint, depth + 1));
    }
    return rc_all;
}

int pcrs_grid_sorted_columns(pcrs_index *ix, float *ox, float *oy, float *oz, uint32_t *perm) {
    pcrs_ctx *ctx = ix->ctx;
    if (ix->algo != PCRS_ALGO_GRID || ix->nvalid == 0)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "index has no grid");
    ix->spts3.release();  // the order inside the cells changes: the copy is rebuilt when needed
    hipLaunchKernelGGL(sort_cells_kernel, dim3((ix->ncell + GB_THREADS - 1) / GB_THREADS),
                       dim3(GB_THREADS), 0, ctx->stream, ix->cell_start.as<uint32_t>(), ix->ncell,
                       ix->spts.as<float4>());
    const uint32_t n = (uint32_t)ix->nvalid;
    hipLaunchKernelGGL(unpack_sorted_kernel, dim3((n + GB_THREADS - 1) / GB_THREADS),
                       dim3(GB_THREADS), 0, ctx->stream, ix->spts.as<float4>(), n, ox, oy, oz, perm);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// bounds[r] = first slot of the first cell that starts at or behind slot r * nvalid / parts:
// a split of the sorted slots into `parts` runs of whole cells (the order of the points inside
// a cell is not reproducible between two builds, the cell boundaries are)
__global__ void slot_bounds_kernel(const uint32_t *cell_start, uint32_t ncell, uint32_t nvalid,
                                   uint32_t parts, uint32_t *bounds) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > parts) return;
    const uint32_t target = (uint32_t)(((unsigned long long)r * nvalid) / parts);
    uint32_t lo = 0, hi = ncell;  // smallest c with cell_start[c] >= target; cell_start[ncell] = nvalid
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (cell_start[mid] >= target) hi = mid; else lo = mid + 1;
    }
    bounds[r] = r == parts ? nvalid : cell_start[lo];
}

int pcrs_grid_slot_bounds(pcrs_index *ix, uint32_t parts, size_t *bounds) {
    pcrs_ctx *ctx = ix->ctx;
    if (ix->algo != PCRS_ALGO_GRID || ix->nvalid == 0)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "index has no grid");
    DevBuf d;
    PCRS_HIP_CHECK(ctx, d.alloc(ctx, ((size_t)parts + 1) * sizeof(uint32_t)));
    hipLaunchKernelGGL(slot_bounds_kernel, dim3((parts + 64) / 64), dim3(64), 0, ctx->stream,
                       ix->cell_start.as<uint32_t>(), ix->ncell, (uint32_t)ix->nvalid, parts,
                       d.as<uint32_t>());
    std::vector<uint32_t> h((size_t)parts + 1);
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), d.p, h.size() * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t r = 0; r <= parts; ++r) bounds[r] = h[r];
    return PCRS_OK;
}

int pcrs_grid_slot_slice(pcrs_index *ix, size_t begin, size_t count, float *ox, float *oy,
                         float *oz, uint32_t *perm) {
    pcrs_ctx *ctx = ix->ctx;
    if (ix->algo != PCRS_ALGO_GRID || begin > ix->nvalid || count > ix->nvalid - begin)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "slot range outside the grid");
    if (count == 0) return PCRS_OK;
    hipLaunchKernelGGL(unpack_sorted_kernel, dim3((unsigned)((count + GB_THREADS - 1) / GB_THREADS)),
                       dim3(GB_THREADS), 0, ctx->stream, ix->spts.as<float4>() + begin,
                       (uint32_t)count, ox, oy, oz, perm);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// Cube half-width (cells) that is certain to cover a ball of the given radius around any query,
// with the same allowance for f32 cell assignment as the neighbour-search certification.
static int radius_cells(const pcrs_index *ix, float radius) {
    const GridGeom &g = ix->g;
    const double extent = (double)std::max(g.nx, std::max(g.ny, g.nz)) * (double)g.cell;
    const double need = (double)radius * (1.0 + 1e-5) + (double)g.slack + 4e-6 * extent;
    const double rc = std::ceil(need / (double)g.cell);
    return rc > 1e6 ? 1000000 : (int)rc;
}

bool pcrs_grid_radius_ok(const pcrs_index *ix, float radius) {
    return ix->algo == PCRS_ALGO_GRID && ix->nvalid > 0 && radius_cells(ix, radius) <= 3;
}

int pcrs_launch_radius_count_grid(pcrs_index *ix, const KnnQuery &q, float radius,
                                  uint32_t *count) {
    pcrs_ctx *ctx = ix->ctx;
    if (q.nq == 0) return PCRS_OK;
    CellArgs a{};
    a.g = ix->g;
    a.cell_start = ix->cell_start.as<uint32_t>();
    a.spts = ix->spts.as<float4>();
    a.nvalid = (uint32_t)ix->nvalid;
    a.R = radius_cells(ix, radius);
    const float r2 = radius * radius;  // kdtree.rs:114
    uint32_t nq = (uint32_t)q.nq;
    if (q.self) {  // the whole cloud, in sorted-slot order; non-finite points count nothing
        if (ix->nvalid < ix->m)
            PCRS_HIP_CHECK(ctx, hipMemsetAsync(count, 0, ix->m * sizeof(uint32_t), ctx->stream));
        a.qmode = 0;
        nq = (uint32_t)ix->nvalid;
    } else {
        a.qmode = 1;
        a.qx = q.qx;
        a.qy = q.qy;
        a.qz = q.qz;
    }
    a.nq = nq;
    hipLaunchKernelGGL(radius_cell_kernel<false>, dim3((nq + GB_THREADS - 1) / GB_THREADS),
                       dim3(GB_THREADS), 0, ctx->stream, a, r2, count, nullptr, nullptr, 0);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// The second pass of a radius search over explicit queries: the hits of query i as keys
// (i << idx_bits | point index) at keys[offsets[i] ...], in walk order.
int pcrs_launch_radius_fill_grid(pcrs_index *ix, const KnnQuery &q, float radius,
                                 const uint32_t *offsets, uint64_t *keys, int idx_bits) {
    pcrs_ctx *ctx = ix->ctx;
    if (q.nq == 0) return PCRS_OK;
    CellArgs a{};
    a.g = ix->g;
    a.cell_start = ix->cell_start.as<uint32_t>();
    a.spts = ix->spts.as<float4>();
    a.nvalid = (uint32_t)ix->nvalid;
    a.R = radius_cells(ix, radius);
    a.qmode = 1;
    a.qx = q.qx;
    a.qy = q.qy;
    a.qz = q.qz;
    a.nq = (uint32_t)q.nq;
    hipLaunchKernelGGL(radius_cell_kernel<true>, dim3((a.nq + GB_THREADS - 1) / GB_THREADS),
                       dim3(GB_THREADS), 0, ctx->stream, a, radius * radius, nullptr, offsets, keys,
                       idx_bits);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

int pcrs_launch_normals_from_cov(pcrs_ctx *ctx, size_t nq, const float *cov, const float *qx,
                                 const float *qy, const float *qz, const KnnOut &out) {
    if (nq == 0) return PCRS_OK;
    hipLaunchKernelGGL(normals_from_cov_kernel, dim3((unsigned)((nq + GB_THREADS - 1) / GB_THREADS)),
                       dim3(GB_THREADS), 0, ctx->stream, (uint32_t)nq, cov, qx, qy, qz, out);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

pcrs_index::~pcrs_index() {
    delete whole;
    for (pcrs_index *k : kids) delete k;
    delete coarse;
}

int pcrs_launch_scan_u32(pcrs_ctx *ctx, uint32_t *data, size_t n, uint32_t *sums,
                         uint32_t *total_out) {
    if (n == 0) {
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(total_out, 0, sizeof(uint32_t), ctx->stream));
        return PCRS_OK;
    }
    if (n >= 0xFFFFFFFFull) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "scan too long");
    if (n <= SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, data, (uint32_t)n,
                           total_out);
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        return PCRS_OK;
    }
    const uint32_t nblocks = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    hipLaunchKernelGGL(scan_block_kernel, dim3(nblocks), dim3(GB_THREADS), 0, ctx->stream, data,
                       (uint32_t)n, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, sums, nblocks,
                       total_out);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nblocks), dim3(GB_THREADS), 0, ctx->stream, data,
                       (uint32_t)n, sums);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// ----------------------------------------------------------------------------- host: query

// identities of 2 to 24 neighbours: one walk with tagged keys (knn_tag_kernel)
static bool lane_kernel_is_one_walk(int kb, bool collect) {
    static const bool use_tag = [] {  // experiments only: PCRS_ONE_WALK=0
        const char *e = getenv("PCRS_ONE_WALK");
        return !(e && e[0] == '0');
    }();
    return use_tag && collect && kb >= 2 && kb <= 24;
}

static void launch_two_walk_kernel(pcrs_ctx *ctx, const CellArgs &a, const KnnOut &out, int kb,
                                   unsigned blocks) {
#define PCRS_CELL2_CALL(KK)                                                                    \
    hipLaunchKernelGGL((knn_cell_kernel<KK, true>), dim3(blocks), dim3(GB_THREADS), 0, ctx->stream, a, out)
    PCRS_DISPATCH_K(kb, PCRS_CELL2_CALL)
#undef PCRS_CELL2_CALL
}

// the lane-per-query kernel of a search for more than one neighbour
static void launch_lane_kernel(pcrs_ctx *ctx, const CellArgs &a, const KnnOut &out, int kb,
                               bool collect, unsigned blocks) {
#define PCRS_CELL_CALL(KK)                                                                      \
    do {                                                                                        \
        if (collect)                                                                            \
            hipLaunchKernelGGL((knn_cell_kernel<KK, true>), dim3(blocks), dim3(GB_THREADS), 0,  \
                               ctx->stream, a, out);                                            \
        else                                                                                    \
            hipLaunchKernelGGL((knn_cell_kernel<KK, false>), dim3(blocks), dim3(GB_THREADS), 0, \
                               ctx->stream, a, out);                                            \
    } while (0)
    if (lane_kernel_is_one_walk(kb, collect)) {
#define PCRS_TAG_CALL(KK)                                                                      \
    hipLaunchKernelGGL((knn_tag_kernel<KK>), dim3(blocks), dim3(GB_THREADS), 0, ctx->stream, a, out)
        switch (kb) {
            case 2: PCRS_TAG_CALL(2); break;
            case 4: PCRS_TAG_CALL(4); break;
            case 8: PCRS_TAG_CALL(8); break;
            case 12: PCRS_TAG_CALL(12); break;
            case 16: PCRS_TAG_CALL(16); break;
            case 17: PCRS_TAG_CALL(17); break;
            case 21: PCRS_TAG_CALL(21); break;
            default: PCRS_TAG_CALL(24); break;
        }
#undef PCRS_TAG_CALL
        return;
    }
    PCRS_DISPATCH_K(kb, PCRS_CELL_CALL)
#undef PCRS_CELL_CALL
}

static void launch_wave_kernel(pcrs_ctx *ctx, const CellArgs &a, const KnnOut &out, int kb, bool coll,
                               unsigned blocks) {
#define PCRS_WAVE_CALL(KK)                                                                      \
    do {                                                                                        \
        if (coll)                                                                               \
            hipLaunchKernelGGL((knn_wave_kernel<KK, true>), dim3(blocks), dim3(GB_THREADS), 0,  \
                               ctx->stream, a, out);                                            \
        else                                                                                    \
            hipLaunchKernelGGL((knn_wave_kernel<KK, false>), dim3(blocks), dim3(GB_THREADS), 0, \
                               ctx->stream, a, out);                                            \
    } while (0)
    PCRS_DISPATCH_K(kb, PCRS_WAVE_CALL)
#undef PCRS_WAVE_CALL
}

// The sorted slots that `ch` picks, in slot order, into list (room for nvalid entries); their
// number into *d_count (device).  Count per block of 1024 slots, scan, ordered write.
static int choose_slots(pcrs_index *ix, const SlotChoice &ch, uint32_t *list, uint32_t *d_count) {
    pcrs_ctx *ctx = ix->ctx;
    // (kept with the index: a buffer given back here could be handed, while these kernels are
    // still queued, to a grid of the same index that runs on another stream)
    DevBuf &own_counts = ix->slot_counts, &own_sums = ix->slot_sums;
    const uint32_t oblocks = (uint32_t)((ix->nvalid + OWN_BLOCK - 1) / OWN_BLOCK);
    if (own_counts.bytes < ((size_t)oblocks + 8) * sizeof(uint32_t))
        PCRS_HIP_CHECK(ctx, own_counts.alloc(ctx, ((size_t)oblocks + 8) * sizeof(uint32_t)));
    if (own_sums.bytes < ((size_t)oblocks / 1024 + 4) * sizeof(uint32_t))
        PCRS_HIP_CHECK(ctx, own_sums.alloc(ctx, ((size_t)oblocks / 1024 + 4) * sizeof(uint32_t)));
    hipLaunchKernelGGL(owned_count_kernel, dim3(oblocks), dim3(GB_THREADS), 0, ctx->stream,
                       ix->spts.as<float4>(), (uint32_t)ix->nvalid, ch, own_counts.as<uint32_t>());
    PCRS_TRY(pcrs_launch_scan_u32(ctx, own_counts.as<uint32_t>(), oblocks, own_sums.as<uint32_t>(),
                                  d_count));
    hipLaunchKernelGGL(owned_write_kernel, dim3(oblocks), dim3(GB_THREADS), 0, ctx->stream,
                       ix->spts.as<float4>(), (uint32_t)ix->nvalid, ch, own_counts.as<uint32_t>(),
                       list);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// The indexed points inside a box as a query set (the queries a rank owns of an index over its
// slab plus a halo): their slots are kept in the index, their number comes back to the host.
int pcrs_grid_select_box(pcrs_index *ix, const BoxFilter &qbox, size_t *count) {
    pcrs_ctx *ctx = ix->ctx;
    *count = 0;
    if (ix->algo != PCRS_ALGO_GRID || ix->nvalid == 0)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "index has no grid");
    PCRS_HIP_CHECK(ctx, ix->sel_list.alloc(ctx, (ix->nvalid + 2) * sizeof(uint32_t)));
    SlotChoice ch{};
    ch.q_begin = 0;
    ch.q_len = 0xFFFFFFFFu;
    ch.boxes = true;
    ch.in = qbox;
    ch.n_ex = 0;
    uint32_t *d_count = ix->sel_list.as<uint32_t>() + ix->nvalid;
    PCRS_TRY(choose_slots(ix, ch, ix->sel_list.as<uint32_t>(), d_count));
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
    PCRS_TRY(pcrs_launch_publish(ctx, d_count, pin, sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ix->sel_count = *reinterpret_cast<uint32_t *>(pin);
    *count = ix->sel_count;
    return PCRS_OK;
}

// Columns and original indices of the selected points, in cell order.
int pcrs_grid_gather_selected(pcrs_index *ix, float *ox, float *oy, float *oz, uint32_t *perm) {
    pcrs_ctx *ctx = ix->ctx;
    if (!ix->sel_list.p) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "no queries selected");
    const uint32_t n = ix->sel_count;
    if (n == 0) return PCRS_OK;
    hipLaunchKernelGGL(gather_slots_kernel, dim3((n + GB_THREADS - 1) / GB_THREADS), dim3(GB_THREADS),
                       0, ctx->stream, ix->spts.as<float4>(), ix->sel_list.as<uint32_t>(), n, ox, oy,
                       oz, perm);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// A search of the cloud's own points over an index with further grids (pcrs_index::kids, the
// finer ones under a grid, and pcrs_index::coarse).  Every point is a query of the finest grid
// whose query box holds it; every grid runs after the finer ones under it, each with the stages of
// a single grid (lane-per-query kernel, two-walk kernel for what that hands over, wave-per-query
// kernel for what neither certifies), and what a grid cannot certify -- a neighbourhood that
// reaches across an open face of its box -- joins the queries of the grid above it.  The outputs
// are indexed by point, so the grids simply fill the same arrays.
static int launch_knn_levels(pcrs_index *top, const KnnQuery &q, const KnnOut &out_in,
                             uint32_t kneed) {
    pcrs_ctx *ctx = top->ctx;
    KnnOut out = out_in;
    const uint32_t m = (uint32_t)top->m;
    // the order the grids run in: every grid after the finer ones under it (finest first), the
    // index itself after all of them, the coarser ones last
    struct Level {
        pcrs_index *ix;
        bool own;            // has queries of its own (a coarser level only takes what it is handed)
        BoxFilter in;        // its queries: inside `in` ...
        std::vector<BoxFilter> ex;  // ... and outside the boxes its finer grids answer
        bool pass_sparse;    // sparse failures skip this level's wave kernel for the next grid's lanes
        int next;            // where it hands on to (position in the order), -1: nowhere
    };
    std::vector<Level> order;
    {
        struct Visit {
            pcrs_index *ix;
            BoxFilter in;
            size_t kid;  // next of its kids to go down into
            std::vector<int> kid_pos;
        };
        std::vector<Visit> stack;
        stack.push_back(Visit{top, pcrs_no_box(), 0, {}});
        while (!stack.empty()) {
            Visit &v = stack.back();
            if (v.kid < v.ix->kids.size()) {
                const size_t j = v.kid++;
                stack.push_back(Visit{v.ix->kids[j], v.ix->kid_qbox[j], 0, {}});
                continue;
            }
            Level l{};
            l.ix = v.ix;
            l.own = true;
            l.in = v.in;
            l.ex = v.ix->kid_qbox;
            l.pass_sparse = true;
            l.next = -1;
            const int me = (int)order.size();
            for (int kp : v.kid_pos) order[(size_t)kp].next = me;
            order.push_back(l);
            stack.pop_back();
            if (!stack.empty()) stack.back().kid_pos.push_back(me);
        }
        for (pcrs_index *c = top->coarse; c; c = c->coarse) {
            Level v{};
            v.ix = c;
            v.own = false;
            v.pass_sparse = true;
            v.next = -1;
            order.back().next = (int)order.size();
            order.push_back(v);
        }
        // a sparse spot of one level is only handed to the next when that one is the next size up
        // (a millimetre blob in a scene of metres has levels a thousand times apart: the rim of
        // the blob, sparse by the blob's measure, would meet the whole blob in one cell there)
        for (Level &l : order)
            if (l.next < 0 || order[(size_t)l.next].ix->g.cell > 4.5f * l.ix->g.cell) l.pass_sparse = false;
    }
    const bool collect = out.idx || out.dist || out.count || out.nx;
    const uint32_t keff = kneed;  // every level holds far more points than a register list
    uint32_t kc = 0;
    if (out.idx || out.dist || out.count) kc = out.k;
    if (out.nx && out.k_nrm > kc) kc = out.k_nrm;
    if (kc > keff) kc = keff;
    const int kb = k_bucket(keff);
    const bool one_walk = lane_kernel_is_one_walk(kb, collect);
    DevBuf cov;
    if (out.nx && !out.cov) {
        PCRS_HIP_CHECK(ctx, cov.alloc(ctx, PCRS_REC_FLOATS * (size_t)m * sizeof(float)));
        out.cov = cov.as<float>();
    }
    const bool timing_was = ctx->timing;
    // What a level passes on is marked in a byte per point (the number of the level it is meant
    // for): the next level picks the marked points up together with its own queries, in the order
    // of its own cells -- neighbouring lanes then walk neighbouring cubes, which a list in the
    // order of the failures would not give them.
    if (order.size() > 255) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "too many grid levels");
    if (top->hand_flags.bytes < (size_t)m + 64) PCRS_HIP_CHECK(ctx, top->hand_flags.alloc(ctx, (size_t)m + 64));
    uint8_t *hand = top->hand_flags.as<uint8_t>();
    if (order.size() > 1) PCRS_HIP_CHECK(ctx, hipMemsetAsync(hand, 0, m, ctx->stream));
    std::vector<bool> is_target(order.size(), false);  // some level hands on to it
    for (const Level &l : order)
        if (l.next >= 0) is_target[(size_t)l.next] = true;
    size_t top_pos = 0;
    while (order[top_pos].ix != top) ++top_pos;
    std::vector<SlotChoice> choices(2 * order.size());
    for (size_t li = 0; li < order.size(); ++li) {
        const Level &lv = order[li];
        SlotChoice ch{};
        ch.q_begin = 0;
        ch.q_len = m;
        ch.boxes = true;
        ch.in = lv.in;
        ch.n_ex = (int)std::min<size_t>(lv.ex.size(), (size_t)PCRS_MAX_KIDS);
        for (int j = 0; j < ch.n_ex; ++j) ch.ex[j] = lv.ex[(size_t)j];
        ch.no_own = !lv.own;
        ch.handed = is_target[li] ? hand : nullptr;
        ch.level = (uint8_t)li;
        choices[li] = ch;
    }
    for (size_t li = 0; li < order.size(); ++li) {  // every grid without what it is handed
        choices[order.size() + li] = choices[li];
        choices[order.size() + li].handed = nullptr;
    }
    if (!top->level_choices.p) {  // the levels of an index never change: uploaded once
        PCRS_HIP_CHECK(ctx, top->level_choices.alloc(ctx, choices.size() * sizeof(SlotChoice)));
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(top->level_choices.p, choices.data(), choices.size() * sizeof(SlotChoice),
                                           hipMemcpyHostToDevice, ctx->stream));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    // Grids that do not depend on each other -- the finer grids under one grid, each with the
    // tail of its slowest lane and a wave-kernel lifetime -- run on side streams: a grid on the
    // stream of the first finer grid under it (or, with none, on the next stream in turn), behind
    // the events of the others; the index's own grid and the coarser ones on the context's stream.
    static const bool streams_on = [] {  // experiments only: PCRS_GRID_STREAMS=0
        const char *e = getenv("PCRS_GRID_STREAMS");
        return !(e && e[0] == '0');
    }();
    const bool side = streams_on && top_pos >= 2;
    // The index's own grid answers most of the cloud and waits for nobody with its own queries:
    // those go first, on the context's stream, while the finer grids run beside them; what the
    // finer grids hand to it is a second, small pass behind them.  (Not when the caller wants the
    // uncertified queries back as one list and this grid is the last: two passes, two lists.)
    // (The second pass costs a scan of this grid's slots: 100 M points 43.5 -> 42.9 ms with it,
    // 48 M 27.0 -> 25.7 ms, 24 M 15.1 -> 14.2 ms; the 200 K-point blob, whose chain of finer grids
    // is hidden behind this grid's own queries: 4.8 -> 3.45 ms.)
    static const size_t split_max = [] {  // experiments only: PCRS_GRID_SPLIT_MAX=<points>
        const char *e = getenv("PCRS_GRID_SPLIT_MAX");
        return e ? (size_t)atoll(e) : ~(size_t)0;
    }();
    const bool split = side && !(order[top_pos].next < 0 && q.no_brute) && top->nvalid <= split_max;
    std::vector<int> lane_of(order.size(), 0);  // 0: the context's stream, 1..: a side stream
    std::vector<std::vector<int>> under(order.size());
    hipStream_t const main_stream = ctx->stream;
    struct Rejoin {  // whatever happens, no side stream is left with work the caller cannot see
        pcrs_ctx *ctx;
        hipStream_t main_stream;
        bool armed;
        ~Rejoin() {
            ctx->stream = main_stream;
            if (armed)
                for (hipStream_t st : ctx->side_streams)
                    if (st) (void)hipStreamSynchronize(st);
        }
    } rejoin{ctx, main_stream, false};
    bool lane_forked[pcrs_ctx::kSideStreams + 1] = {true, false, false, false};
    if (side) {
        for (int i = 0; i < pcrs_ctx::kSideStreams; ++i)
            if (!ctx->side_streams[i]) {
                    // ahead of the context's stream where the two compete: the finer grids are small,
                    // and the grid above them waits for them
                    int least = 0, greatest = 0;
                    PCRS_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
                    PCRS_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->side_streams[i], hipStreamNonBlocking,
                                                                    greatest));
                }
        if (!ctx->fork_event) PCRS_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
        while (ctx->level_events.size() < top_pos) {
            hipEvent_t e = nullptr;
            PCRS_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->level_events.push_back(e);
        }
        for (size_t li = 0; li < top_pos; ++li) under[(size_t)order[li].next].push_back((int)li);
        int turn = 0;
        for (size_t li = 0; li < top_pos; ++li)
            lane_of[li] = !under[li].empty() ? lane_of[(size_t)under[li][0]]
                          : split            ? 1 + (turn++) % pcrs_ctx::kSideStreams
                                             : (turn++) % (pcrs_ctx::kSideStreams + 1);
        rejoin.armed = true;
    }
    auto stream_of = [&](size_t li) {
        return lane_of[li] == 0 ? main_stream : ctx->side_streams[lane_of[li] - 1];
    };
    // What the searches report (pcrs_index_last_stats): the entries of every level's wave kernel
    // added up in word 0, what was left for the brute-force kernel in word 3.
    if (!top->level_stats.p) PCRS_HIP_CHECK(ctx, top->level_stats.alloc(ctx, 16 * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipMemsetAsync(top->level_stats.p, 0, 16 * sizeof(uint32_t), main_stream));
    if (side) PCRS_HIP_CHECK(ctx, hipEventRecord(ctx->fork_event, main_stream));
    uint32_t *level_stats = top->level_stats.as<uint32_t>();
    const uint32_t *esc_list = nullptr, *esc_count = nullptr;  // what the last level left
    // One level's stages.  mode 0: its own queries and what it is handed; 1: its own only (waits
    // for nobody); 2: only what it is handed; 3: what it has been handed so far, struck out as it
    // is taken (another pass of the level follows).
    auto run_level = [&](size_t li, int mode) -> int {
        const Level &lv = order[li];
        pcrs_index *ix = lv.ix;
        const bool last_level = lv.next < 0;
        if (side) {
            ctx->stream = stream_of(li);
            if (!lane_forked[lane_of[li]]) {
                PCRS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->fork_event, 0));
                lane_forked[lane_of[li]] = true;
            }
            if (mode != 1)
                for (int d : under[li])
                    if (lane_of[(size_t)d] != lane_of[li])
                        PCRS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->level_events[(size_t)d], 0));
        }
        // every query of a level is one of its own points, so its lists are as long as the level
        // (not as the cloud: two dozen levels of a hundred million points would hold 50 GB)
        const size_t list_elems = ix->nvalid + 64;
        const size_t need = (5 * list_elems + 64) * sizeof(uint32_t);
        if (ix->fail_lists.bytes < need) PCRS_HIP_CHECK(ctx, ix->fail_lists.alloc(ctx, need));
        uint32_t *listA = ix->fail_lists.as<uint32_t>();
        uint32_t *listB = listA + list_elems;
        uint32_t *listC = listB + list_elems;
        float *boundA = reinterpret_cast<float *>(listC + list_elems);
        uint32_t *listS = listC + 2 * list_elems;  // the level's own queries (sorted slots)
        if (!ix->counters.p) {
            PCRS_HIP_CHECK(ctx, ix->counters.alloc(ctx, 32 * sizeof(uint32_t)));
            PCRS_HIP_CHECK(ctx, hipMemsetAsync(ix->counters.p, 0, 32 * sizeof(uint32_t), ctx->stream));
            ix->counter_phase = 0;
        }
        uint32_t *counters = ix->counters.as<uint32_t>() + 16 * ix->counter_phase;
        uint32_t *counters_next = ix->counters.as<uint32_t>() + 16 * (ix->counter_phase ^ 1);
        if (!ix->spts3.p) {
            PCRS_HIP_CHECK(ctx, ix->spts3.alloc(ctx, ix->nvalid * 3 * sizeof(float)));
            hipLaunchKernelGGL(pack_xyz_kernel, dim3((unsigned)((ix->nvalid + GB_THREADS - 1) / GB_THREADS)),
                               dim3(GB_THREADS), 0, ctx->stream, ix->spts.as<float4>(),
                               (uint32_t)ix->nvalid, ix->spts3.as<float>());
        }
        // the level's queries: its own (inside its query box, outside the next finer level's)
        // and what the level before it marked for it.  A level with queries of its own answers
        // most of its points: its lanes go over all its slots and test each (three launches over
        // the level's points less, and no list to read); one that only takes what it is handed
        // gets a compacted list.
        const bool filtered = lv.own && mode != 2;
        if (!filtered) {
            SlotChoice ch = choices[li];
            if (mode == 2) ch.no_own = true;
            if (mode == 3) ch.consume = hand;
            PCRS_TRY(choose_slots(ix, ch, listS, &counters[4]));
        }

        CellArgs a{};
        a.g = ix->g;
        a.cell_start = ix->cell_start.as<uint32_t>();
        a.spts = ix->spts.as<float4>();
        a.spts3 = ix->spts3.as<float>();
        a.nvalid = (uint32_t)ix->nvalid;
        a.rx = ix->x;
        a.ry = ix->y;
        a.rz = ix->z;
        a.qx = ix->x;
        a.qy = ix->y;
        a.qz = ix->z;
        a.nq = m;
        a.q_begin = 0;
        a.q_len = m;
        a.kcert = keff;
        a.kcollect = collect ? (kc ? kc : 1) : 0;
        a.useful_d2 = __builtin_inff();
        a.R = 1;
        // (The few queries the finer grids hand to a second pass of their own sit at the rims of
        // crowded regions: a lane walks a cube of thousands of points for milliseconds, and with
        // nothing else in the launch to hide behind that is the launch.  A wave each serves them.)
        a.heavy = m >= PCRS_HEAVY_FLAG ? 0xFFFFFFFFu : (mode == 2 ? 1024u : 8192u);
        // a cube with fewer than three lists' worth of points that certifies nothing: a sparse spot
        a.sparse_below = (lv.pass_sparse && !last_level) ? 3u * keff : 0u;
        a.hand_flags = last_level ? nullptr : hand;
        a.hand_level = (uint8_t)(lv.next < 0 ? 0 : lv.next);
        a.wave_radii[0] = 1;
        a.wave_radii[1] = 2;
        a.wave_radii[2] = 4;
        a.wave_radii[3] = 8;
        a.wave_radii[4] = 1 << 20;
        a.n_radii = 5;
        a.skip = q.skip_flag;
        a.zero_me = counters_next;
        a.fail_list = listA;
        a.fail_count = &counters[0];
        a.fail_bound = boundA;
        a.alt_list = listC;
        a.alt_count = &counters[5];
        const unsigned lane_blocks = std::min((unsigned)((ix->nvalid + GB_THREADS - 1) / GB_THREADS),
                                              (unsigned)ctx->num_cus * 64u);
        // stage 1 over the level's queries
        if (filtered) {
            a.qmode = 5;
            a.nq = (uint32_t)ix->nvalid;
            a.choice = top->level_choices.as<SlotChoice>() + (mode == 1 ? order.size() + li : li);
        } else {
            a.qmode = 3;
            a.qlist = listS;
            a.qcount = &counters[4];
        }
        pcrs_timing_begin(ctx);
        launch_lane_kernel(ctx, a, out, kb, collect,
                           filtered ? lane_blocks : std::min(lane_blocks, (unsigned)ctx->num_cus * 16u));
        pcrs_timing_end(ctx);
        ix->counter_phase ^= 1;
        a.zero_me = nullptr;
        ctx->timing = false;
        a.alt_list = nullptr;
        a.alt_count = nullptr;
        if (one_walk) {
            a.qmode = 2;
            a.qlist = listC;
            a.qcount = &counters[5];
            launch_two_walk_kernel(ctx, a, out, kb, std::min(lane_blocks, (unsigned)ctx->num_cus * 8u));
        }
        // larger cubes, samples and balls for what is left, a wave per query
        a.qmode = 2;
        a.qlist = listA;
        a.qcount = &counters[0];
        a.qbound = boundA;
        a.fail_list = listB;
        a.fail_count = &counters[3];
        a.fail_bound = nullptr;
        static const bool pass_up = [] {  // experiments only: PCRS_GRID_PASS_UP=1
            const char *e = getenv("PCRS_GRID_PASS_UP");
            return e && e[0] == '1';
        }();
        a.heavy_only = (pass_up && lv.own && lv.ix != top) ? 1u : 0u;
        launch_wave_kernel(ctx, a, out, kb, collect,
                           (unsigned)std::min<size_t>(((size_t)m + 3) / 4, (size_t)ctx->num_cus * 8));
        hipLaunchKernelGGL(add_counts_kernel, dim3(1), dim3(64), 0, ctx->stream, level_stats, &counters[0],
                           last_level ? &counters[3] : nullptr);
        PCRS_HIP_CHECK(ctx, hipGetLastError());
        ix->left_list = listB;
        ix->left_count = &counters[3];
        esc_list = listB;
        esc_count = &counters[3];
        if ((mode == 1 || mode == 3) && last_level) {
            // nothing comes after this grid: what its own queries leave is answered here and now,
            // before the second pass takes the lists
            PCRS_TRY(pcrs_launch_knn_bf(ctx, top->x, top->y, top->z, top->m, q, out, kneed, esc_list, esc_count,
                                        m));
        }
        ctx->timing = timing_was;
        if (side && mode != 1 && li < top_pos && lane_of[(size_t)lv.next] != lane_of[li])
            PCRS_HIP_CHECK(ctx, hipEventRecord(ctx->level_events[li], ctx->stream));
        static const bool debug = getenv("PCRS_GRID_DEBUG") != nullptr;  // experiments only
        if (debug) {
            uint32_t c[8];
            (void)hipMemcpyAsync(c, counters, sizeof(c), hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            fprintf(stderr, "pcrs search level %zu (cell %.5g, %zu points): %u queries (its own and what the "
                            "level before marked for it), %u handed to the two-walk kernel, %u left by the "
                            "lanes to the wave kernel, %u left for brute force (last level)\n",
                    li, (double)ix->g.cell, ix->nvalid, c[4], c[5], c[0], c[3]);
        }
        return PCRS_OK;
    };
    if (split) PCRS_TRY(run_level(top_pos, 1));
    // ... and the coarser grid behind it takes what those leave it right away, beside the finer
    // grids, instead of after everything (what the second pass hands it is a pass of its own)
    const bool early_coarse = split && order[top_pos].next >= 0;
    if (early_coarse) PCRS_TRY(run_level((size_t)order[top_pos].next, 3));
    // (The same one level down -- every grid with finer grids under it running its own queries
    // beside them -- was tried: the blob's chain of three 3.1 -> 2.8 ms, but the ring scan
    // 1.56 -> 1.68 ms and the survey scenes 1-3 % slower: a second pass is a compaction and a
    // wave-kernel lifetime more per grid.)
    std::vector<bool> two_passes(order.size(), false);
    if (split) two_passes[top_pos] = true;
    for (size_t li = 0; li < order.size(); ++li) PCRS_TRY(run_level(li, two_passes[li] ? 2 : 0));
    // what even the coarsest level (no open faces) could not certify: more ties than a wave lists
    ctx->stream = main_stream;  // every side stream has been waited for by the grid above its grids
    rejoin.armed = false;
    ctx->timing = false;
    const int rc_bf = q.no_brute ? PCRS_OK
                                 : pcrs_launch_knn_bf(ctx, top->x, top->y, top->z, top->m, q, out, kneed,
                                                      esc_list, esc_count, m);
    if (cov.p && rc_bf == PCRS_OK)
        (void)pcrs_launch_normals_from_cov(ctx, m, cov.as<float>(), top->x, top->y, top->z, out);
    ctx->timing = timing_was;
    PCRS_TRY(rc_bf);
    if (top->nvalid < top->m)
        PCRS_TRY(pcrs_launch_invalid_defaults(ctx, top->x, top->y, top->z, q, out_in));
    top->left_list = esc_list;
    top->left_count = esc_count;
    top->stats_dev = level_stats;
    top->stats_nq = m;
    return PCRS_OK;
}

int pcrs_launch_knn_grid(pcrs_index *ix, const KnnQuery &q, const KnnOut &out_in, uint32_t kneed) {
    pcrs_ctx *ctx = ix->ctx;
    KnnOut out = out_in;
    if (q.nq == 0) return PCRS_OK;
    if (q.nq >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    if (kneed == 0) kneed = 1;
    const uint32_t keff = (uint32_t)((size_t)kneed < ix->nvalid ? kneed : ix->nvalid);
    // Lists longer than the register budget go to the brute-force kernel, which handles any k.
    if (keff > PCRS_KMAX_REG || keff == 0) {
        ix->stats[2] = q.nq;
        q.xform_fused = false;
        if (q.xform)  // a pending increment of the queries: the brute-force kernel does not apply it
            PCRS_TRY(pcrs_launch_transform_state(ctx, q.wx, q.wy, q.wz, q.nq,
                                                 reinterpret_cast<const IcpLoopState *>(q.xform)));
        return pcrs_launch_knn_bf(ctx, ix->x, ix->y, ix->z, ix->m, q, out, kneed, nullptr, nullptr,
                                  0);
    }
    const bool nn_only = out.nn_idx && !out.idx && !out.dist && !out.count && !out.mean && !out.nx;
    const uint32_t nq = (uint32_t)q.nq;
    if ((!ix->kids.empty() || ix->coarse) && q.self && q.out_offset == 0 && q.nq == ix->m && !nn_only && !q.only_list &&
        !q.xform && keff == kneed)
        return launch_knn_levels(ix, q, out_in, kneed);
    // A pending increment of the queries (KnnQuery::xform) is applied by the 1-NN kernel as it
    // loads them, when that kernel reads the caller's columns directly; otherwise here, first.
    const bool fuse_xform = q.xform && nn_only && !q.self && (q.coherent || nq < 32768u);
    q.xform_fused = fuse_xform;
    if (q.xform && !fuse_xform)
        PCRS_TRY(pcrs_launch_transform_state(ctx, q.wx, q.wy, q.wz, q.nq,
                                             reinterpret_cast<const IcpLoopState *>(q.xform)));

    // two retry lists (ping-pong) + stage counters
    const size_t list_elems = (size_t)nq + 64;
    const size_t need = (4 * list_elems + 64) * sizeof(uint32_t);
    if (ix->fail_lists.bytes < need) {
        PCRS_HIP_CHECK(ctx, ix->fail_lists.alloc(ctx, need));
    }
    uint32_t *listA = ix->fail_lists.as<uint32_t>();
    uint32_t *listB = listA + list_elems;
    uint32_t *listC = listB + list_elems;  // handed from the one-walk to the two-walk kernel
    float *boundA = reinterpret_cast<float *>(listC + list_elems);  // beside listA (fail_bound)
    if (!ix->counters.p) {
        PCRS_HIP_CHECK(ctx, ix->counters.alloc(ctx, 32 * sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(ix->counters.p, 0, 32 * sizeof(uint32_t), ctx->stream));
        ix->counter_phase = 0;
    }
    // [0..3] failures after each cube stage, [4] owned, [5] handed over; this search's set is zero (cleared by the
    // search before it, or just now), and its first kernel clears the other set
    uint32_t *counters = ix->counters.as<uint32_t>() + 16 * ix->counter_phase;
    uint32_t *counters_next = ix->counters.as<uint32_t>() + 16 * (ix->counter_phase ^ 1);
    // (the phase flips once the first stage, which clears the other set, has been queued: a call
    // that fails before then must leave the next search on this still-clean set)

    CellArgs a{};
    a.g = ix->g;
    a.cell_start = ix->cell_start.as<uint32_t>();
    a.spts = ix->spts.as<float4>();
    a.nvalid = (uint32_t)ix->nvalid;
    const bool nn_early = out_in.nn_idx && !out_in.idx && !out_in.dist && !out_in.count &&
                          !out_in.mean && !out_in.nx;
    if (!nn_early && !ix->spts3.p) {  // the index-free copy of the sorted points for k > 1 searches
        PCRS_HIP_CHECK(ctx, ix->spts3.alloc(ctx, ix->nvalid * 3 * sizeof(float)));
        hipLaunchKernelGGL(pack_xyz_kernel, dim3((unsigned)((ix->nvalid + GB_THREADS - 1) / GB_THREADS)),
                           dim3(GB_THREADS), 0, ctx->stream, ix->spts.as<float4>(), (uint32_t)ix->nvalid,
                           ix->spts3.as<float>());
    }
    a.spts3 = ix->spts3.as<float>();
    a.rx = ix->x;
    a.ry = ix->y;
    a.rz = ix->z;
    a.qx = q.qx;
    a.qy = q.qy;
    a.qz = q.qz;
    a.kcert = keff;
    a.zero_me = counters_next;
    a.skip = q.skip_flag;
    a.useful_d2 = __builtin_inff();
    if (nn_only && q.useful_distance < __builtin_inff()) {
        const float u = q.useful_distance * (1.0f + 1e-5f);  // sqrt and square round trip
        a.useful_d2 = u * u;
    }
    // neighbour identities are needed for everything except the SOR mean
    const bool collect = out.idx || out.dist || out.count || out.nx;
    uint32_t kc = 0;
    if (out.idx || out.dist || out.count) kc = out.k;
    if (out.nx && out.k_nrm > kc) kc = out.k_nrm;
    if (kc > keff) kc = keff;
    a.kcollect = (collect || nn_only) ? (kc ? kc : 1) : 0;
    const int kb = k_bucket(keff);

    const bool one_walk = !nn_only && lane_kernel_is_one_walk(kb, collect);
    auto launch_two_walk = [&](unsigned blocks) { launch_two_walk_kernel(ctx, a, out, kb, blocks); };
    auto launch = [&](unsigned blocks) {
        if (nn_only) {  // eight lanes per query
            hipLaunchKernelGGL(nn_cell_kernel, dim3(blocks * NN_SUB), dim3(GB_THREADS), 0,
                               ctx->stream, a, out);
            return;
        }
        launch_lane_kernel(ctx, a, out, kb, collect, blocks);
    };

    // stage 1: the 27-cell cube over every query
    a.R = 1;
    // cubes beyond this many points leave the lane-per-query kernels for the wave-per-query one
    // (2048 for eight lanes per query, 8192 for one); the flag needs bit 31 of the list entry
    a.heavy = nq >= PCRS_HEAVY_FLAG ? 0xFFFFFFFFu : (nn_only ? 2048u : 8192u);
    a.fail_list = listA;
    a.fail_count = &counters[0];
    a.fail_bound = nn_only ? nullptr : boundA;
    unsigned blocks1 = (unsigned)((nq + GB_THREADS - 1) / GB_THREADS);
    DevBuf q_table, q_sums, q_tmp, q_sorted, q_cells;
    bool sorted_queries = false;
    if (q.self && !nn_only) {
        a.q_begin = (uint32_t)q.out_offset;
        a.q_len = nq;
        if (q.out_offset == 0 && q.nq == ix->m) {
            a.qmode = 0;  // the whole cloud, in sorted-slot order
            a.nq = (uint32_t)ix->nvalid;
            blocks1 = (unsigned)((ix->nvalid + GB_THREADS - 1) / GB_THREADS);
        } else {
            // a sub-range: compact its sorted slots first so that every wave is full
            SlotChoice ch{};
            ch.q_begin = a.q_begin;
            ch.q_len = a.q_len;
            PCRS_TRY(choose_slots(ix, ch, listB, &counters[4]));
            a.qmode = 3;
            a.qlist = listB;
            a.qcount = &counters[4];
        }
    } else if (q.only_list) {
        // a list of the explicit queries (what another index could not certify)
        a.qmode = 2;
        a.nq = nq;
        a.qlist = q.only_list;
        a.qcount = q.only_count;
        blocks1 = std::min(blocks1, (unsigned)ctx->num_cus * 8u);
    } else {
        a.qmode = 1;
        a.nq = nq;
        // A large batch of explicit queries arrives in the caller's order: neighbouring lanes
        // would walk unrelated cells.  Sorting the queries by the cell they fall into (the same
        // two-level sort as the index build, under the index's own geometry) restores that.
        SortPlan sp{};
        if (!q.coherent && nq >= 32768 && plan_sort(ix->g, ix->ncell, nq, &sp)) {
            PCRS_HIP_CHECK(ctx, q_table.alloc(ctx, ((size_t)sp.nb * sp.nblk + 8) * sizeof(uint32_t)));
            PCRS_HIP_CHECK(ctx, q_sums.alloc(ctx, (((size_t)sp.nb * sp.nblk) / SCAN_BLOCK + 2) *
                                                      sizeof(uint32_t)));
            PCRS_HIP_CHECK(ctx, q_tmp.alloc(ctx, (size_t)nq * sizeof(float4)));
            PCRS_HIP_CHECK(ctx, q_sorted.alloc(ctx, (size_t)nq * sizeof(float4)));
            PCRS_HIP_CHECK(ctx, q_cells.alloc(ctx, ((size_t)ix->ncell + 8) * sizeof(uint32_t)));
            uint32_t *q_nonempty = q_cells.as<uint32_t>() + nonempty_slot(ix->ncell);
            PCRS_TRY(run_two_level_sort(ctx, q.qx, q.qy, q.qz, nq, sp, q_table.as<uint32_t>(),
                                        q_sums.as<uint32_t>(), q_tmp.as<float4>(),
                                        q_sorted.as<float4>(), q_cells.as<uint32_t>(), q_nonempty));
            a.qmode = 4;
            a.qpts = q_sorted.as<float4>();
            a.qcount = q_table.as<uint32_t>() + (size_t)sp.nb * sp.nblk;  // finite queries
            sorted_queries = true;
        }
    }
    // normals: the search kernels leave covariance sums, a light kernel solves them afterwards
    DevBuf cov;
    if (out.nx && !out.cov) {
        PCRS_HIP_CHECK(ctx, cov.alloc(ctx, PCRS_REC_FLOATS * (size_t)nq * sizeof(float)));
        out.cov = cov.as<float>();
    }
    // the timing bracket covers the dominant kernel only (stage 1 over every query), so that it
    // can be set against the same kernel's duration in a profiler trace
    if (fuse_xform) {
        if (a.qmode != 1) return pcrs_set_err(ctx, PCRS_ERR_HIP, "internal: fused transform needs direct queries");
        a.xform = reinterpret_cast<const IcpLoopState *>(q.xform);
        a.wx = q.wx;
        a.wy = q.wy;
        a.wz = q.wz;
    }
    // one launch grows the cube of each failed query from these half-widths (next_half_width)
    a.wave_radii[0] = 1;
    a.wave_radii[1] = 2;
    a.wave_radii[2] = 4;
    a.wave_radii[3] = 8;
    a.n_radii = 4;
    // The list ends with the whole grid as the last "cube": for the queries that are left (points
    // far from the cloud) a wave costs far less than the tiled brute-force kernel, where a single
    // lane would work through the whole cloud.  The retries sample large regions and then search
    // a ball, so neither the size of the cloud nor the number of such queries matters to them.
    // 1-NN searches need no brute-force launch at all; longer lists keep it for more ties at the
    // K-th distance than a wave lists.
    const bool wave_does_all = nn_only;  // one neighbour: no list can overflow
    a.wave_radii[4] = 1 << 20;
    a.n_radii = 5;
    if (nn_only) {
        // the 1-NN kernel retries in place (a wave per failed query, right behind the wave's own
        // first stage): what it still cannot certify goes straight to the brute-force list
        a.fail_list = listB;
        a.fail_count = &counters[3];
        a.stage1_fail = &counters[0];
    }
    a.alt_list = listC;
    a.alt_count = &counters[5];
    pcrs_timing_begin(ctx);
    launch(blocks1);
    pcrs_timing_end(ctx);
    ix->counter_phase ^= 1;
    a.xform = nullptr;
    a.zero_me = nullptr;
    a.alt_list = nullptr;
    a.alt_count = nullptr;
    if (one_walk) {
        // what the one-walk kernel handed over (nothing, on most clouds: the launch then ends at
        // once) is answered by the two-walk kernel; its failures join the same retry list
        a.qmode = 2;
        a.qlist = listC;
        a.qcount = &counters[5];
        launch_two_walk(std::min(blocks1, (unsigned)ctx->num_cus * 8u));
    }
    const bool timing_was = ctx->timing;
    ctx->timing = false;  // the retry stages below are not separate measurements

    // stages 2-4: larger cubes over the failures, one wave per query (their number is known only
    // on the device: fixed grid with a grid-stride loop)
    auto launch_wave = [&](unsigned blocks) {
        launch_wave_kernel(ctx, a, out, kb, collect || nn_only, blocks);
    };
    if (!nn_only) {
        unsigned retry_blocks = (unsigned)std::min<size_t>((nq + 3) / 4, (size_t)ctx->num_cus * 8);
        a.qmode = 2;
        a.qlist = listA;
        a.qcount = &counters[0];
        a.qbound = boundA;
        a.fail_list = listB;
        a.fail_count = &counters[3];
        a.fail_bound = nullptr;
        launch_wave(retry_blocks);
    }
    uint32_t *src = listB;
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    // stage 4: brute force over what is left
    ix->left_list = src;  // what no stage of this index certified (answered below, unless the
    ix->left_count = &counters[3];  // caller has another index to ask)
    const int rc_bf = (wave_does_all || q.no_brute) ? PCRS_OK
                                    : pcrs_launch_knn_bf(ctx, ix->x, ix->y, ix->z, ix->m, q, out,
                                                         kneed, src, &counters[3], nq);
    if (cov.p && rc_bf == PCRS_OK) {  // (a caller that brings its own records solves them itself)
        const float *cqx = q.self ? ix->x + q.out_offset : q.qx;
        const float *cqy = q.self ? ix->y + q.out_offset : q.qy;
        const float *cqz = q.self ? ix->z + q.out_offset : q.qz;
        (void)pcrs_launch_normals_from_cov(ctx, nq, cov.as<float>(), cqx, cqy, cqz, out);
    }
    ctx->timing = timing_was;
    PCRS_TRY(rc_bf);
    // queries answered from a sorted array never saw the non-finite ones among them
    if (sorted_queries) PCRS_TRY(pcrs_launch_invalid_defaults(ctx, ix->x, ix->y, ix->z, q, out_in));
    if (q.self && !nn_only && ix->nvalid < ix->m)
        PCRS_TRY(pcrs_launch_invalid_defaults(ctx, ix->x, ix->y, ix->z, q, out_in));  // no cov
    ix->stats_dev = counters;
    ix->stats_nq = nq;
    return PCRS_OK;
}
