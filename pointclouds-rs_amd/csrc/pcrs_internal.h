// Internal declarations shared by the kernels and the C-ABI layer of libpcrs_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <map>
#include <vector>

#include "../../include/pcrs_hip.h"

// All device arithmetic must keep the reference's f32 operation order: never contract a*b+c.
#pragma clang fp contract(off)

// (d2 = +inf, index = none): larger than every real key, and decodes to d2 = +inf
#define PCRS_EMPTY_KEY 0x7F800000FFFFFFFFull
#define PCRS_KMAX_REG 64  // largest neighbour list kept in registers
#define PCRS_REC_FLOATS_HOST 8  // floats per query of the covariance records (knn_common.h)

struct pcrs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char err[512] = {0};  // fixed storage: reporting an error never allocates or throws
    // pinned host staging for small scalar read-backs
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    // a second pinned area for small inputs and results a call moves itself (the kernels'
    // read-backs land in the first at any time): copies from and to it are queued without the
    // staging and the wait a copy from pageable memory costs
    void *pinned_io = nullptr;
    size_t pinned_io_bytes = 0;
    // reusable device scratch (grown on demand, never shrunk)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int num_cus = 256;
    // this device's copy of the library's static scratch array (stats.hip), resolved once at
    // context creation; the entry point that must not allocate works in it
    void *static_scratch = nullptr;
    size_t static_scratch_bytes = 0;
    // second stream for result copies that may overlap later kernels of the same call
    // (created on first use); copy_event orders it behind the producing kernels
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_event = nullptr;
    // Side streams for the grids of a refined index that do not depend on each other (grid.hip,
    // launch_knn_levels; created on first use).  fork_event orders them behind the context's
    // stream, level_events order a grid behind the finer ones under it; all of them have joined
    // the context's stream again when the search returns.
    static constexpr int kSideStreams = 3;
    hipStream_t side_streams[kSideStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t fork_event = nullptr;
    std::vector<hipEvent_t> level_events;
    // optional HIP-event bracketing of the neighbour-search kernel (benchmark instrumentation)
    bool timing = false;
    std::vector<hipEvent_t> ev_start, ev_stop;  // one pair per bracketed launch
    size_t ev_used = 0;
    int timing_depth = 0;  // nested brackets collapse into the outermost one
    // Free device buffers kept for reuse.  Everything a context does runs on its one stream, so
    // a buffer released by one operation can be handed to the next without synchronising.
    struct PoolEntry {
        void *p;
        size_t bytes;
    };
    std::vector<PoolEntry> pool;
    size_t pool_mallocs = 0, pool_drops = 0;  // how often the pool went to the device (experiments)
    // The grid geometry the last index build settled on (grid.hip): a caller that feeds one cloud
    // after another of the same kind -- frames of a sensor -- starts the next build from it and
    // usually needs one sizing attempt instead of three.  Only a starting point: every build
    // still checks the occupancy it gets.
    struct GridHint {
        double cell = 0.0, emax = 0.0, dim = 3.0;
        size_t n = 0;
        uint32_t k_hint = 0;
    } grid_hint;
    // The same for the finer grids of a refined index, by their place in the tree (pcrs_index::
    // level_key): the second scan of a sensor's stream builds every level in one sizing attempt,
    // where the first needed two or three.  (Kept apart from grid_hint, which a finer grid's build
    // would otherwise overwrite with a geometry no whole cloud can use.)
    std::map<uint32_t, GridHint> level_hints;
};

hipError_t pcrs_pool_alloc(pcrs_ctx *ctx, size_t bytes, void **p, size_t *cap);
void pcrs_pool_release(pcrs_ctx *ctx, void *p, size_t cap);

// Brackets one launch of the dominant (neighbour-search) kernel with HIP events on the context's
// stream when timing is enabled; no-ops otherwise.
void pcrs_timing_begin(pcrs_ctx *ctx);
void pcrs_timing_end(pcrs_ctx *ctx);

int pcrs_set_err(pcrs_ctx *ctx, int code, const char *fmt, ...);
int pcrs_on_exception(pcrs_ctx *ctx) noexcept;
// grows the context's device scratch / pinned staging; returns PCRS_OK or an error code
int pcrs_ctx_scratch(pcrs_ctx *ctx, size_t bytes, void **out);
int pcrs_ctx_pinned(pcrs_ctx *ctx, size_t bytes, void **out);
int pcrs_ctx_pinned_io(pcrs_ctx *ctx, size_t bytes, void **out);

#define PCRS_TRY(expr)                  \
    do {                                \
        int _rc = (expr);               \
        if (_rc != PCRS_OK) return _rc; \
    } while (0)

#define PCRS_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (void)hipGetLastError(); /* the runtime keeps it: a later check must not see it */ \
            return pcrs_set_err((ctx), _e == hipErrorOutOfMemory ? PCRS_ERR_OOM : PCRS_ERR_HIP, \
                                "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),          \
                                __FILE__, __LINE__);                                           \
        }                                                                                      \
    } while (0)

// Owning device buffer; pooled when allocated through a context.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    pcrs_ctx *owner = nullptr;
    hipError_t alloc(size_t b) {  // plain hipMalloc
        release();
        if (b == 0) b = 16;
        hipError_t e = hipMalloc(&p, b);
        if (e == hipSuccess) bytes = b;
        else (void)hipGetLastError();  // do not leave the failure for a later check to find
        return e;
    }
    hipError_t alloc(pcrs_ctx *ctx, size_t b) {  // from the context's pool
        release();
        if (b == 0) b = 16;
        hipError_t e = pcrs_pool_alloc(ctx, b, &p, &bytes);
        if (e == hipSuccess) owner = ctx;
        return e;
    }
    void release() {
        if (p) {
            if (owner)
                pcrs_pool_release(owner, p, bytes);
            else
                (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
        owner = nullptr;
    }
    template <class T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// Uniform grid over the finite points of a cloud (counting-sorted copy).
struct GridGeom {
    float ox, oy, oz;   // origin (min corner of finite points)
    float cell, inv;    // cell edge and its reciprocal
    int nx, ny, nz;     // cells per axis
    float slack;        // absolute safety margin covering f32 rounding of cell assignment
    // An index over a BOX of a cloud (one GPU's slab of a sharded cloud, a dense region given a
    // finer grid of its own) leaves the points beyond the box out.  Where the cloud goes on behind
    // a face of the box that face is "open": open_lo / open_hi hold its grid-relative coordinate
    // (-inf / +inf for a face with nothing behind it), and no search may certify a result whose
    // ball reaches across it.
    float open_lo[3], open_hi[3];
};

// Which points of the columns an index takes in: finite, and lo[a] <= p[a] < hi[a] on every axis
// (-inf / +inf: no limit on that side).
struct BoxFilter {
    float lo[3], hi[3];
    __host__ __device__ bool unbounded() const {
        return lo[0] == -__builtin_inff() && lo[1] == -__builtin_inff() && lo[2] == -__builtin_inff() &&
               hi[0] == __builtin_inff() && hi[1] == __builtin_inff() && hi[2] == __builtin_inff();
    }
};
inline BoxFilter pcrs_no_box() {
    BoxFilter b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = -__builtin_inff();
        b.hi[a] = __builtin_inff();
    }
    return b;
}

constexpr int PCRS_MAX_KIDS = 16;  // finer grids directly under one grid

struct pcrs_index {
    ~pcrs_index();
    pcrs_ctx *ctx = nullptr;
    const float *x = nullptr, *y = nullptr, *z = nullptr;  // device columns (original order)
    size_t m = 0;
    DevBuf own_x, own_y, own_z;  // set when built from host memory
    int algo = PCRS_ALGO_BRUTE;  // resolved strategy (BRUTE or GRID)
    uint32_t k_hint = 0;         // the list length the grid was sized for
    // the part of the columns the grid holds (everything finite, by default)
    BoxFilter box = pcrs_no_box();
    // bounds that hold every point the grid takes and an upper limit of their number, from a
    // caller that has just measured the cloud: the build then skips its own pass for them
    bool use_hint = false;  // start from the context's last geometry although the columns are device memory
    uint32_t level_key = 0;  // place in the tree of a refined index: 0 the index itself, else parent * 31 + ordinal + 1
    bool have_bounds = false;
    double bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
    size_t bounds_n_upper = 0;
    bool bounds_count_exact = false;  // bounds_n_upper is the number of points, not a limit of it
    // grid (algo == GRID)
    GridGeom g{};
    size_t nvalid = 0;   // points in the grid: finite and inside the box
    uint32_t ncell = 0;
    DevBuf cell_start;   // uint32 [ncell + 1] slot ranges of the cells (x-fastest cell order)
    DevBuf spts;         // float4 [nvalid] points sorted by cell: (x, y, z, original index bits)
    DevBuf spts3;        // float [3 * nvalid] the same points without the index (12-byte candidates:
                         // the lane-per-query search is bound by the bytes its per-lane loads move)
    DevBuf fail_lists;   // two uint32 [nq] retry lists, grown on demand
    // Two sets of 16 stage counters.  A search uses one set and its first kernel clears the
    // other for the search after it, so no search starts with a fill command of its own.
    DevBuf counters;
    int counter_phase = 0;
    // queries picked by pcrs_grid_select_box: their sorted slots (+ the count behind them)
    DevBuf sel_list;
    uint32_t sel_count = 0;
    // what the last search left uncertified (device list of query positions and its length);
    // valid until the next search on this index
    const uint32_t *left_list = nullptr;
    const uint32_t *left_count = nullptr;
    // Finer grids over the boxes in which this grid's cells are crowded (a scan that is dense near
    // the sensor, a blob in a sparse scene, the roofs and trees of an aerial survey), each refined
    // in turn where it is crowded: a search of the cloud's own points answers every point at the
    // finest grid whose query box holds it.  kid_qbox: the part of space a kid answers; the kid
    // indexes a margin of one of this grid's cells more.
    // A finer grid is sorted from a gathered copy of the parent's points around its box (columns
    // of src_n points and their indices in the cloud's columns) instead of from passes over the
    // whole cloud; only the build reads them.
    DevBuf src_cols, src_index;
    size_t src_n = 0;
    std::vector<pcrs_index *> kids;   // owned
    std::vector<BoxFilter> kid_qbox;  // the part of space each of them answers (disjoint boxes)
    // A coarser grid over the same points (cells three times as wide, itself with a coarser one
    // where that is still too fine) for the sparse parts of a cloud of very uneven density: what
    // this grid's lanes cannot certify because the query's cube held hardly any points goes to the
    // coarser grid's lanes instead of this grid's wave-per-query kernel.
    pcrs_index *coarse = nullptr;
    // a byte per point of the columns, for a search over the chain of grids: the level a query
    // that one level could not answer is meant for next
    DevBuf hand_flags;
    DevBuf slot_counts, slot_sums;  // scratch of the compaction of this grid's query slots
    DevBuf level_stats;    // what the last search over the levels reports (16 words)
    DevBuf level_choices;  // the query test of every level (grid.hip: SlotChoice), in the order they run
    double aim = 0.0;  // points per occupied cell the build aimed at
    // of the accepted sort, when it read them back: mean points per occupied cell, and the
    // fullness of the cell the average point sits in (sum of squared counts / points - 1)
    double occ_mean = 0.0, occ_seen = 0.0;
    // an index over the whole cloud, built on first need, for the queries an index over a box
    // cannot certify (their neighbourhood reaches across an open face)
    pcrs_index *whole = nullptr;
    // counters of the last query: read back lazily by pcrs_index_last_stats
    uint64_t stats[3] = {0, 0, 0};
    const uint32_t *stats_dev = nullptr;  // device counters [4]: failures after cube stages 1-4
    uint64_t stats_nq = 0;
};

// Output selection of a neighbour search.  Every pointer is nullable.
struct KnnOut {
    // KNN list output
    uint32_t *idx = nullptr;    // [nq * k]
    float *dist = nullptr;      // [nq * k]
    uint32_t *count = nullptr;  // [nq]
    uint32_t k = 0;             // neighbours requested for the list output
    // statistical-outlier mean distance (ranks 1..k_sor of a (k_sor+1)-NN list)
    float *mean = nullptr;  // [nq]
    uint32_t k_sor = 0;
    // PCA normals (ranks 0..k_nrm-1)
    float *nx = nullptr, *ny = nullptr, *nz = nullptr;  // [nq]
    uint32_t k_nrm = 0;
    float vp[3] = {0, 0, 0};
    // When set, the consumers store the six covariance sums and the SOR mean in a record of eight
    // floats per query (c00 = -1 for "no neighbours") instead of solving for the normal: the f64
    // eigen solver needs ~120 registers and would cap the occupancy of the search kernels.
    float *cov = nullptr;
    float rec_mean = 0.0f;  // per-thread: the query's mean on its way into the record
    // 1-NN correspondence
    uint32_t *nn_idx = nullptr;  // [nq]
    float *nn_d2 = nullptr;      // [nq] squared distance, +inf when none
};

// Query description: either explicit query columns or a self-query range of the indexed cloud.
struct KnnQuery {
    const float *qx = nullptr, *qy = nullptr, *qz = nullptr;
    size_t nq = 0;
    size_t out_offset = 0;  // self queries: first query is point out_offset of the cloud
    bool self = false;
    bool coherent = false;  // explicit queries already in a spatial order: do not sort them again
    // 1-NN searches whose caller drops neighbours farther than this (correspondence.rs:30): a
    // query whose searched cube already covers that distance is finished, found or not.  The
    // grid kernels then store "no neighbour" (PCRS_NO_INDEX, +inf) for a query whose nearest
    // point lies beyond the bound; the brute-force kernel stores the true nearest point.
    float useful_distance = __builtin_inff();
    // device word that, when non-zero, turns the whole search into a no-op (an ICP iteration
    // queued behind the end of its loop)
    const uint32_t *skip_flag = nullptr;
    // explicit queries only: answer just the listed ones (device list of positions in qx/qy/qz
    // and its length), e.g. those another index left uncertified
    const uint32_t *only_list = nullptr, *only_count = nullptr;
    // leave what the grid stages cannot certify in the index's left_list instead of sending it to
    // the brute-force kernel (the caller asks another index)
    bool no_brute = false;
    // 1-NN searches of an ICP loop: `xform` (device, an IcpLoopState) holds a rigid increment that
    // is still to be applied to the queries; a search kernel that can do so applies it as it
    // loads each query, writes the moved point back to wx/wy/wz (the query columns, writable) and
    // reports it in xform_fused -- one launch less per iteration.
    const void *xform = nullptr;
    float *wx = nullptr, *wy = nullptr, *wz = nullptr;
    mutable bool xform_fused = false;
};

// ---- brute force (knn_bf.hip)
// Searches all m reference points for each query.  qlist (nullable) restricts the work to the
// listed query positions; its length is read from *qlist_count on the device.
int pcrs_launch_knn_bf(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz, size_t m,
                       const KnnQuery &q, const KnnOut &out, uint32_t kneed,
                       const uint32_t *qlist, const uint32_t *qlist_count, size_t qlist_cap);

// ---- grid (grid.hip)
// cell_request > 0 fixes the cell edge (enlarged only if the cell budget demands it)
int pcrs_grid_build(pcrs_index *ix, uint32_t k_hint, float cell_request = 0.0f);
// finer grids over the crowded part of a grid, level by level (see pcrs_index::kids)
int pcrs_grid_refine(pcrs_index *ix, uint32_t k_hint, int depth = 0);
// radius count over the grid: possible when the ball fits a cube of at most 7^3 cells
bool pcrs_grid_radius_ok(const pcrs_index *ix, float radius);
int pcrs_launch_radius_count_grid(pcrs_index *ix, const KnnQuery &q, float radius,
                                  uint32_t *count);
// keep[i] = count[i] >= min_neighbors (radius_outlier.rs:13)
// Second pass of a radius search (after the counts have been scanned into offsets): brute force
// writes ascending indices, the grid writes keys (query << idx_bits | index) in walk order.
int pcrs_launch_radius_fill(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz,
                            size_t m, const float *qx, const float *qy, const float *qz, size_t n,
                            float radius, const uint32_t *offsets, uint32_t *indices);
int pcrs_launch_radius_fill_grid(pcrs_index *ix, const KnnQuery &q, float radius,
                                 const uint32_t *offsets, uint64_t *keys, int idx_bits);
int pcrs_launch_sum_u32(pcrs_ctx *ctx, const uint32_t *v, size_t n, unsigned long long *d_total);
int pcrs_launch_key_low_bits(pcrs_ctx *ctx, const uint64_t *keys, size_t n, int idx_bits,
                             uint32_t *indices);
int pcrs_launch_count_mask(pcrs_ctx *ctx, const uint32_t *count, size_t n, size_t min_neighbors,
                           uint8_t *keep, unsigned long long *d_count);
// In-place exclusive scan of data[0..n) (u32); *total_out (device) receives the grand total.
// sums must hold n / 1024 + 2 words.
int pcrs_launch_scan_u32(pcrs_ctx *ctx, uint32_t *data, size_t n, uint32_t *sums,
                         uint32_t *total_out);
// Stable LSD radix sort of (u64 key, u32 value) pairs on the low key_bits bits of the key
// (sort.hip).  Both buffer pairs are used; *in_alt tells whether the result is in the alt pair.
// vals and vals_alt may both be null: the keys are then sorted alone.
int pcrs_sort_pairs_u64(pcrs_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt,
                        uint32_t *vals_alt, size_t n, int key_bits, bool *in_alt);
// Points ordered by an integer cell key per axis (voxel.hip): key = floor(p / scale) or
// floor(p * scale) as a saturating i32, finite points only, cells in ascending (x, y, z) key
// order, points of a cell in ascending index order.
struct PcrsCellSort {
    DevBuf k[3];                  // int [n]: the three keys of every point (0 for non-finite)
    DevBuf keys, keys_alt, vals, vals_alt;  // sort buffers; `order` points into vals or vals_alt
    DevBuf cell_id;               // u32 [nvalid + 1]: exclusive scan of the cell-start flags
    DevBuf flag;                  // u8 [nvalid]: 1 where a sorted position starts a cell
    const uint32_t *order = nullptr;  // [nvalid] point index at each sorted position
    const uint32_t *d_ncells = nullptr;  // device: number of occupied cells
    uint32_t nvalid = 0;          // finite points
    int key_lo[3] = {0, 0, 0}, key_hi[3] = {0, 0, 0};  // smallest and largest key per axis (nvalid > 0)
};
int pcrs_cell_sort(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                   float scale, bool multiply, PcrsCellSort *cs);
// voxel_downsample on device columns (voxel.hip); ox/oy/oz hold up to n centroids.
// bounds (nullable, [6]: lo x y z, hi x y z): a box that holds every centroid, from the voxel keys
// the sort has to know anyway; left untouched (NaN in, NaN out) where the keys saturate.
int pcrs_launch_voxel_downsample(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                                 size_t n, float voxel_size, float *ox, float *oy, float *oz,
                                 size_t *n_out, double *bounds = nullptr);
// Plane RANSAC behind its sample generation (ransac.hip): d_samples holds ns index triples,
// d_mask receives the inlier mask of the chosen plane.  The call waits for the stream (its results
// are host values); h_mask_out (nullable, host memory) receives a copy of the mask before that
// wait, so that a caller with host buffers does not wait a second time.
int pcrs_launch_ransac_plane(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                             size_t n, const uint32_t *d_samples, size_t ns, float thr,
                             float model_out[4], uint8_t *d_mask, size_t *n_inliers,
                             uint64_t *best_count_out, uint8_t *h_mask_out = nullptr);
// euclidean_cluster on device columns (cluster.hip): CSR output, d_offsets holds up to n + 1
// entries and d_indices up to n.
int pcrs_launch_euclidean_cluster(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                                  size_t n, float threshold, size_t min_size, size_t max_size,
                                  uint32_t *d_offsets, uint32_t *d_indices, size_t *n_clusters,
                                  size_t *n_indices);
// Writes the points in cell order (ties inside a cell by original index) as columns, with the
// permutation sorted position -> original index.  Needs algo == GRID; all nvalid points.
int pcrs_grid_sorted_columns(pcrs_index *ix, float *ox, float *oy, float *oz, uint32_t *perm);
// Split of the sorted slots into `parts` runs of whole cells (bounds[parts + 1], host memory;
// one read-back) and the columns + original indices of a run of slots.
int pcrs_grid_slot_bounds(pcrs_index *ix, uint32_t parts, size_t *bounds);
int pcrs_grid_slot_slice(pcrs_index *ix, size_t begin, size_t count, float *ox, float *oy,
                         float *oz, uint32_t *perm);
int pcrs_launch_knn_grid(pcrs_index *ix, const KnnQuery &q, const KnnOut &out, uint32_t kneed);
int pcrs_grid_select_box(pcrs_index *ix, const BoxFilter &qbox, size_t *count);
// what the cut measured on its way (nullable): the cloud's bounding box and the histogram of its
// points along the cut axis, bin b = [hist_lo + b * hist_width, hist_lo + (b + 1) * hist_width)
struct SlabMeasure {
    double lo[3], hi[3];
    float hist_lo = 0.0f, hist_width = 0.0f;
    std::vector<uint32_t> hist;
};
int pcrs_slab_cuts(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t m,
                   uint32_t parts, uint32_t k_hint, int *axis_out, float *cuts, float *halo,
                   size_t *n_finite, SlabMeasure *measure = nullptr);
int pcrs_grid_gather_selected(pcrs_index *ix, float *ox, float *oy, float *oz, uint32_t *perm);
// one thread per query: the oriented normal from the query's covariance record (and its mean)
int pcrs_launch_normals_from_cov(pcrs_ctx *ctx, size_t nq, const float *cov, const float *qx,
                                 const float *qy, const float *qz, const KnnOut &out);
// consumer defaults for the non-finite queries of a self-query range (knn_bf.hip)
int pcrs_launch_invalid_defaults(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz,
                                 const KnnQuery &q, const KnnOut &out);

// ---- misc kernels (misc.hip)
int pcrs_launch_mask(pcrs_ctx *ctx, const float *mean, size_t n, float thr, uint8_t *keep,
                     unsigned long long *d_count);
// out[perm[i]] = in[i]
// Copies a few words (a multiple of 4 bytes, at most 64 KB) from device memory into pinned host
// memory with a one-wave kernel; visible to the host after the stream has been synchronised.
int pcrs_launch_publish(pcrs_ctx *ctx, const void *d_src, void *pinned_dst, size_t bytes);
int pcrs_launch_scatter_perm(pcrs_ctx *ctx, const float *in, const uint32_t *perm, size_t n,
                             float *out);
// PointCloud::select by a mask (cloud.rs:103-140), x / y / z only: the points whose mask byte is
// non-zero (zero with invert) in input order; oidx (nullable) their indices, d_total (nullable,
// device) their number.  The outputs need room for n points.
int pcrs_launch_select(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                       const uint8_t *mask, int invert, float *ox, float *oy, float *oz,
                       uint32_t *oidx, uint32_t *d_total);
// out[point[i] - first] = in[i]
int pcrs_launch_scatter_offset(pcrs_ctx *ctx, const float *in, const uint32_t *point, size_t n,
                               uint32_t first, float *out);
// out[i] = first + i
int pcrs_launch_index_run(pcrs_ctx *ctx, uint32_t *out, size_t n, uint32_t first);
// (point, value) pairs bucketed by the owner of the point's index range (see misc.hip)
int pcrs_launch_route_pairs(pcrs_ctx *ctx, const uint32_t *point, const float *value, size_t n,
                            size_t n_total, uint32_t world, uint32_t *send_point, float *send_value,
                            size_t *counts);
int pcrs_launch_transform(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                          const float rot[9], const float t[3], float *ox, float *oy, float *oz);
int pcrs_launch_radius_count(pcrs_ctx *ctx, const float *rx, const float *ry, const float *rz,
                             size_t m, const float *qx, const float *qy, const float *qz, size_t n,
                             float radius, uint32_t *count);
// ICP consumer reduction over dense correspondences (nn_idx / nn_d2 per source point).
// partials: device scratch of PCRS_ICP_NSUMS doubles per block; result in d_sums[PCRS_ICP_NSUMS].
// Loop state of an ICP run that lives on the device, so that several iterations can be queued
// without a host round trip (icp.rs:147-188 is the loop it carries).
struct IcpLoopState {
    float cum_r[9], cum_t[3];  // accumulated transform
    float inc_r[9], inc_t[3];  // increment solved by the last live iteration
    float prev_rmse, last_rmse, last_fitness;
    uint32_t converged;
    uint32_t done;  // loop has ended (converged or no correspondences): later iterations are no-ops
    uint32_t num_iterations;
    float last_ncorr;    // correspondence count behind last_rmse, as f32 (icp.rs:281)
    uint32_t last_live;  // 1-based iteration that produced last_rmse (0 = none yet)
};
// What one iteration of the device-resident loop needs beyond the reduction itself.
struct IcpTail {
    IcpLoopState *state = nullptr;  // null: sums only, no loop
    size_t n_source = 0;
    float tolerance = 0.0f;
    bool metrics_only = false;  // max_iterations == 0: one pass for rmse and fitness (icp.rs:190-197)
    // d2seq (nullable): receives dist*dist of every correspondence (0 where there is none) at the
    // point's ORIGINAL source index (perm undoes the spatial order of the working copy)
    float *d2seq = nullptr;
    const uint32_t *perm = nullptr;
    // exact: the rmse of this iteration is the reference's left-to-right f32 sum over d2seq
    // (icp.rs:277-281) instead of the f64 tree sum; needs seq_scalars / seq_work
    bool exact = false;
    void *seq_scalars = nullptr, *seq_work = nullptr;
    size_t seq_work_bytes = 0;
    // sums_only: the reduction stops at d_sums (skipped once the loop state says done); the tail
    // follows later, on the sums of all shards (pcrs_launch_icp_shard_tail)
    bool sums_only = false;
};
// After the sums of an iteration are in d_sums: metrics, convergence test, solve, compose
// (pcrs_launch_icp_reduce does the same inside its last kernel when given a loop state).
int pcrs_launch_icp_tail(pcrs_ctx *ctx, int mode, const double *d_sums, const IcpTail &tail);
// The tail over the sums of `world` shards (world * PCRS_ICP_NSUMS doubles, added in rank order
// into d_sums first).
int pcrs_launch_icp_shard_tail(pcrs_ctx *ctx, int mode, const double *d_sums_all, uint32_t world,
                               double *d_sums, const IcpTail &tail);
// last_rmse of the state from the exact sum left by pcrs_launch_seq_sum in d_seq_scalars
int pcrs_launch_icp_exact_rmse(pcrs_ctx *ctx, IcpLoopState *d_state, const void *d_seq_scalars);
// cur = inc(cur) with the increment read from the device state; no-op once the loop is done.
int pcrs_launch_transform_state(pcrs_ctx *ctx, float *x, float *y, float *z, size_t n,
                                const IcpLoopState *d_state);
int pcrs_launch_icp_reduce(pcrs_ctx *ctx, int mode, const float *cx, const float *cy,
                           const float *cz, size_t n, const float *tx, const float *ty,
                           const float *tz, const float *tnx, const float *tny, const float *tnz,
                           const uint32_t *nn_idx, const float *nn_d2, float max_d,
                           uint32_t *tgt_idx_out, float *dist_out, double *d_sums,
                           const IcpTail &tail);
// SOR global statistics with the reference's left-to-right f32 summation, on the device (stats.hip)
size_t pcrs_sor_stats_workspace(size_t n);
// Left-to-right f32 sum of n non-negative values with the same machinery; the sum is the float at
// byte pcrs_seq_sum_result_offset() of d_scalars.  Work space as for the statistics.  skip
// (nullable device word): non-zero turns every kernel of the sum into a no-op.
int pcrs_launch_seq_sum(pcrs_ctx *ctx, const float *values, size_t n, void *d_scalars, void *d_work,
                        size_t work_bytes, const uint32_t *skip = nullptr);
size_t pcrs_seq_sum_result_offset();
size_t pcrs_sor_stats_scalars_bytes();
// clear_bytes: how much memory from d_scalars on is to start at zero (the scalars, and whatever
// the caller keeps behind them, e.g. the mask's counter); the run's first kernel clears it.
int pcrs_launch_sor_stats(pcrs_ctx *ctx, const float *dmean, size_t n, float std_mul,
                          void *d_scalars, void *d_work, size_t work_bytes, size_t clear_bytes = 0);
// count_is_clear: *d_count is already zero (cleared together with the scalars)
int pcrs_launch_mask_dev_thr(pcrs_ctx *ctx, const float *mean, size_t n, const void *d_scalars,
                             uint8_t *keep, unsigned long long *d_count,
                             bool count_is_clear = false);
void pcrs_sor_stats_decode(const void *host_copy, float *mu, float *thr, int *has_finite);
int pcrs_sor_threshold_impl(pcrs_ctx *ctx, const float *dmean, size_t n, float std_mul,
                            float *thr, int *has_finite);
// Device address and size of the static scratch array on the current device (stats.hip); resolved
// once per device.  Call with the device current.
int pcrs_static_scratch(pcrs_ctx *ctx, void **ptr, size_t *bytes);
// {thr, has_finite, n_keep} from the device scalars and the mask count into *d_result
int pcrs_launch_select_result(pcrs_ctx *ctx, const void *d_scalars,
                              const unsigned long long *d_count, pcrs_sor_select_result *d_result);

// pcrs_index_build_dev for a caller that feeds one cloud after another of the same kind from
// device memory (the frames of a sensor): the build starts from the geometry the context's last
// build ended with (api.cpp)
// bounds (nullable, [6] as above, all finite): a box that holds all m points, every one of them
// finite -- the build then skips its pass over the columns and the read-back behind it.
int pcrs_index_build_hinted_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                                size_t m, uint32_t k_hint, pcrs_index **out,
                                const double *bounds = nullptr);

// what the multi-GPU driver (dist.cpp) needs to know about an ICP handle (api.cpp)
pcrs_ctx *pcrs_icp_context(pcrs_icp *icp);
void pcrs_icp_lengths(const pcrs_icp *icp, size_t *n_source, size_t *n_target, size_t *n_normals);

// ---- host math (host_math.cpp)
void pcrs_host_compose(const float ra[9], const float ta[3], const float rb[9], const float tb[3],
                       float ro[9], float to[3]);
void pcrs_host_solve_plane(const double *sums, float rot[9], float t[3]);
void pcrs_host_solve_point(const double *sums, float rot[9], float t[3]);
