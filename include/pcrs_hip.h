/*
 * pcrs_hip.h — C-ABI of libpcrs_hip.so, the MI355X (gfx950) implementation of
 * the pointclouds-rs per-point KNN hot path.
 *
 * The reference has no FFI of its own: the path sits behind safe-Rust `pub fn`s
 * in crates that `#![forbid(unsafe_code)]`.  Each entry point below cites the
 * Rust function (file:line, relative to the reference checkout) it stands in
 * for; a `pointclouds-hip-sys` crate binds exactly these symbols.
 *
 * Conventions
 *  - Every function returns PCRS_OK (0) or a negative PCRS_ERR_* code and never
 *    unwinds across the boundary.  pcrs_last_error() gives the message of the
 *    last failure on that context.
 *  - Point clouds are structure-of-arrays f32 columns (x[], y[], z[]), exactly
 *    `PointCloud { x, y, z: Vec<f32> }` (crates/core/src/cloud.rs:3-11), so a
 *    Rust caller passes `Vec::as_ptr()` with no repacking.
 *  - Functions WITHOUT the `_dev` suffix take HOST pointers, copy in/out and
 *    return when the outputs are complete.  The caller owns every buffer; the
 *    library keeps no host pointer after return.
 *  - Functions WITH the `_dev` suffix take DEVICE pointers (hipMalloc'd memory
 *    on the context's device), run on the context's stream and synchronise
 *    that stream only where a scalar is returned to the host.
 *  - A context is bound to one device and one stream and is not thread-safe;
 *    use one context per GPU (one per process in multi-GPU jobs).
 *  - Neighbour order is ascending squared distance, then DESCENDING point
 *    index among equidistant points (kiddo's own tie order is traversal
 *    dependent; this is the order under which the reference test
 *    icp.rs:345-368 passes as written); a point with a non-finite coordinate is
 *    never a neighbour; a query with a non-finite coordinate has no neighbours
 *    (kdtree.rs:65).
 */
#ifndef PCRS_HIP_H
#define PCRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCRS_OK 0
#define PCRS_ERR_INVALID_ARG (-1)
#define PCRS_ERR_HIP (-2)
#define PCRS_ERR_OOM (-3)
/* IcpPlaneError::NormalsMismatch, crates/registration/src/icp_plane.rs:27-32,101-107 */
#define PCRS_ERR_NORMALS_MISMATCH (-4)

#define PCRS_NO_INDEX 0xFFFFFFFFu

#define PCRS_ICP_POINT_TO_POINT 0
#define PCRS_ICP_POINT_TO_PLANE 1

/* search strategy of an index (pcrs_index_build*): */
#define PCRS_ALGO_AUTO 0  /* uniform grid when the cloud is large enough, else brute force */
#define PCRS_ALGO_BRUTE 1 /* LDS-tiled brute force over the whole cloud */
#define PCRS_ALGO_GRID 2  /* uniform-grid pruned tiles with exact fallback */

typedef struct pcrs_ctx pcrs_ctx;
typedef struct pcrs_index pcrs_index;
typedef struct pcrs_icp pcrs_icp;

/* IcpResult + RigidTransform, crates/registration/src/icp.rs:7-11,111-118 */
typedef struct pcrs_icp_result {
    float rotation[9]; /* row-major 3x3 */
    float translation[3];
    float fitness;
    float rmse;
    int32_t converged;
    uint32_t num_iterations;
} pcrs_icp_result;

/* Number of doubles pcrs_icp_accumulate* produces (layout documented there). */
#define PCRS_ICP_NSUMS 32

/* ---------------------------------------------------------------- context */

/* Creates a context on HIP device `device` with its own stream. */
int pcrs_ctx_create(int device, pcrs_ctx **out);
/* Uses `hip_stream` (a hipStream_t, e.g. torch's current stream) from now on. */
int pcrs_ctx_set_stream(pcrs_ctx *ctx, void *hip_stream);
void pcrs_ctx_destroy(pcrs_ctx *ctx);
const char *pcrs_last_error(const pcrs_ctx *ctx);
int pcrs_ctx_synchronize(pcrs_ctx *ctx);
/* Library version string and the GPU architecture the kernels were built for. */
const char *pcrs_version(void);

/* ------------------------------------------------- spatial index (KdTree) */

/* KdTree::build — crates/spatial/src/kdtree.rs:25-44.
 * Builds the device-resident search structure over m points.  `k_hint` is the
 * neighbour count the index will mostly be asked for (sizes the grid cells;
 * 0 = default).  The _dev form keeps the column pointers, which must stay
 * valid and unchanged until pcrs_index_destroy; the host form uploads a copy. */
int pcrs_index_build(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t m,
                     uint32_t k_hint, int algo, pcrs_index **out);
int pcrs_index_build_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                         size_t m, uint32_t k_hint, int algo, pcrs_index **out);
void pcrs_index_destroy(pcrs_index *index);
/* KdTree::len — kdtree.rs:47-49 */
size_t pcrs_index_len(const pcrs_index *index);

/* Batched KdTree::knn / knn_indices — kdtree.rs:64-96.
 * For each of the n queries writes up to k neighbours, ascending:
 *   idx  [n*k] point indices, PCRS_NO_INDEX padded
 *   dist [n*k] Euclidean distances sqrt(d2) (kdtree.rs:76), NaN padded; nullable
 *   count[n]   neighbours found = min(k, finite points), 0 for a non-finite
 *              query, k == 0 or an empty index; nullable */
int pcrs_knn_batch(pcrs_index *index, const float *qx, const float *qy, const float *qz, size_t n,
                   uint32_t k, uint32_t *idx, float *dist, uint32_t *count);
int pcrs_knn_batch_dev(pcrs_index *index, const float *dqx, const float *dqy, const float *dqz,
                       size_t n, uint32_t k, uint32_t *didx, float *ddist, uint32_t *dcount);

/* Batched count of KdTree::radius_search results — kdtree.rs:105-135 (points
 * with d2 <= radius*radius, the query itself included when it is in the cloud).
 * count[n]; a non-positive or non-finite radius or a non-finite query gives 0. */
int pcrs_radius_count(pcrs_index *index, const float *qx, const float *qy, const float *qz,
                      size_t n, float radius, uint32_t *count);
int pcrs_radius_count_dev(pcrs_index *index, const float *dqx, const float *dqy, const float *dqz,
                          size_t n, float radius, uint32_t *dcount);

/* KdTree::radius_search for a batch of queries — kdtree.rs:105-135.  CSR result:
 * indices[offsets[i] .. offsets[i+1]) = the points with d2 <= radius*radius of query i in
 * ascending index order (kdtree.rs:131-132); offsets[n+1].  *total receives the number of
 * results.  When they do not fit `capacity` (or indices is NULL) the call returns
 * PCRS_ERR_INVALID_ARG with offsets and *total filled in, so that the caller can size the list
 * and call again.  More than 2^32-2 results in one call are refused with the same code: *total
 * still holds their number, but 32-bit offsets cannot describe them and are all zero then (search
 * fewer queries at a time). */
int pcrs_radius_search(pcrs_index *index, const float *qx, const float *qy, const float *qz,
                       size_t n, float radius, uint32_t *offsets, uint32_t *indices,
                       size_t capacity, size_t *total);
int pcrs_radius_search_dev(pcrs_index *index, const float *dqx, const float *dqy,
                           const float *dqz, size_t n, float radius, uint32_t *doffsets,
                           uint32_t *dindices, size_t capacity, size_t *total);

/* radius_outlier_removal — crates/filters/src/radius_outlier.rs:4-18: keep[i] = 1
 * when at least min_neighbors points of the cloud (point i itself included) lie
 * within `radius` of point i.  As in KdTree::radius_search (kdtree.rs:105-135) a
 * non-positive or non-finite radius, or a non-finite point, finds nothing, so such
 * points are kept only when min_neighbors == 0.  count (nullable) receives the
 * per-point neighbour counts; the caller gathers the kept rows (cloud.rs:103-162).
 * The _dev variant takes device pointers and synchronises only to return n_keep. */
int pcrs_radius_outlier(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                        float radius, size_t min_neighbors, uint8_t *keep, uint32_t *count,
                        size_t *n_keep);
int pcrs_radius_outlier_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                            size_t n, float radius, size_t min_neighbors, uint8_t *dkeep,
                            uint32_t *dcount, size_t *n_keep);

/* voxel_downsample — crates/filters/src/voxel_downsample.rs:12-65: one centroid per
 * occupied voxel of edge voxel_size, in ascending (x, y, z) voxel-key order, each the
 * f32 sum of the voxel's finite points taken in input order divided by their count
 * (bit-identical to the reference).  out_x/out_y/out_z must hold n values; *n_out
 * receives the number of voxels.  voxel_size must be > 0 and finite (the reference
 * asserts it): PCRS_ERR_INVALID_ARG otherwise.  The _dev variant takes device pointers
 * and returns after the centroids are complete. */
int pcrs_voxel_downsample(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                          float voxel_size, float *out_x, float *out_y, float *out_z,
                          size_t *n_out);
int pcrs_voxel_downsample_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                              size_t n, float voxel_size, float *dout_x, float *dout_y,
                              float *dout_z, size_t *n_out);

/* euclidean_cluster — crates/segmentation/src/euclidean_cluster.rs:96-187: connected
 * components of "distance <= distance_threshold" (pairs are examined, as in the
 * reference, when their cells floor(p / threshold) are equal or adjacent), clusters
 * with min_size <= size <= max_size only, largest first, ties by smallest index,
 * indices ascending inside a cluster; non-finite points are clusters of one.  The
 * result is in CSR form: cluster c holds indices[offsets[c] .. offsets[c+1]).
 * offsets must hold n + 1 entries, indices n.  An empty cloud, a threshold <= 0 or
 * min_size == 0 gives no clusters (:102-104). */
int pcrs_euclidean_cluster(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                           float distance_threshold, size_t min_size, size_t max_size,
                           uint32_t *offsets, uint32_t *indices, size_t *n_clusters);
int pcrs_euclidean_cluster_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                               size_t n, float distance_threshold, size_t min_size,
                               size_t max_size, uint32_t *doffsets, uint32_t *dindices,
                               size_t *n_clusters, size_t *n_indices);

/* ransac_plane_seeded behind its sample generation —
 * crates/segmentation/src/ransac_plane.rs:56-138.  The reference draws n_samples index
 * triples from rand's StdRng (:75-79, sample_three_distinct :141-165) and the caller
 * passes exactly those (samples[3*s .. 3*s+2], the triples that sampling returned, in
 * order).  The library fits one plane per triple (:168-191), counts the inliers of
 * every plane over the cloud (|n.p + d| <= distance_threshold, :17-20,131-137), picks
 * the plane the reference picks — earliest plane with the largest count when
 * n >= 10000 and n_samples >= 16 (:81-94), otherwise the sequential scan with its
 * adaptive early stop (:96-122) — and returns model = {nx, ny, nz, d} and the inlier
 * mask of that plane (:125-128; the caller turns it into ascending indices).
 * n < 3, or no usable triple, gives the default plane (0,0,1,0) (:64-66, :23-30).
 * best_count (nullable) receives the winning inlier count. */
int pcrs_ransac_plane_samples(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                              size_t n, const uint32_t *samples, size_t n_samples,
                              float distance_threshold, float model[4], uint8_t *inlier_mask,
                              size_t *n_inliers, uint64_t *best_count);
int pcrs_ransac_plane_samples_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                                  size_t n, const uint32_t *dsamples, size_t n_samples,
                                  float distance_threshold, float model[4], uint8_t *dinlier_mask,
                                  size_t *n_inliers, uint64_t *best_count);

/* PointCloud::select / select_inverse by a mask, x / y / z columns on the device —
 * crates/core/src/cloud.rs:103-162: the points whose mask byte is non-zero (zero with invert != 0)
 * in their order, so that the stages of a pipeline hand device columns to each other.  dindex
 * (nullable) receives their indices; the outputs need room for n points.  *n_out (nullable)
 * receives their number; passing it makes the call wait for the stream. */
int pcrs_select_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz, size_t n,
                    const uint8_t *dmask, int invert, float *dox, float *doy, float *doz,
                    uint32_t *dindex, size_t *n_out);

/* The index triples ransac_plane_seeded would sample (ransac_plane.rs:75-79,141-165: three
 * distinct indices below n, up to 100 redraws each, a triple is dropped when they fail) from a
 * SplitMix64 stream seeded with `seed`, each draw (next * n) >> 64.  The reference draws from
 * rand's StdRng, which cannot be reproduced without that crate: these are the triples of the
 * Python mirror (pointclouds_rs.sample_triples), for callers that want that stream.  triples holds
 * 3 * iterations indices; *n_triples receives how many triples were made.  Host code only. */
int pcrs_sample_triples(size_t n, size_t iterations, uint64_t seed, uint32_t *triples,
                        size_t *n_triples);

/* The obstacle pipeline of the reference's example as one call —
 * examples/python/kitti_obstacle_detection.py:87-124: voxel_downsample(voxel_size) ->
 * statistical_outlier_removal(sor_k, sor_std_mul) -> ransac_plane_seeded(plane_threshold,
 * plane_iterations, triples of pcrs_sample_triples(kept points, plane_iterations, plane_seed)) ->
 * the points off the plane (select_inverse) -> euclidean_cluster(cluster_threshold, min, max).
 * Every stage computes what its own entry point computes; the cloud is uploaded once and the
 * intermediate clouds stay on the device.  Host buffers: obst_x/y/z (nullable, room for n) receive
 * the obstacle cloud the clusters index into; cluster_offsets (n + 1) and cluster_indices (n) the
 * clusters in CSR form as pcrs_euclidean_cluster gives them. */
typedef struct pcrs_obstacle_params {
    float voxel_size;
    uint32_t sor_k;
    float sor_std_mul;
    float plane_threshold;
    uint32_t plane_iterations;
    uint64_t plane_seed;
    float cluster_threshold;
    size_t cluster_min_size, cluster_max_size;
} pcrs_obstacle_params;
typedef struct pcrs_obstacle_result {
    size_t n_voxels, n_kept, n_ground, n_obstacles, n_clusters, n_cluster_points;
    float plane[4];      /* nx, ny, nz, d */
    float sor_threshold; /* NaN when none was computed */
} pcrs_obstacle_result;
int pcrs_obstacle_pipeline(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                           const pcrs_obstacle_params *params, pcrs_obstacle_result *result,
                           float *obst_x, float *obst_y, float *obst_z, uint32_t *cluster_offsets,
                           uint32_t *cluster_indices);

/* ------------------------------------------- statistical_outlier_removal */

/* The per-point loop of statistical_outlier_removal —
 * crates/filters/src/statistical_outlier.rs:19-39: mean Euclidean distance to
 * the k nearest other points ((k+1)-NN, first dropped); +inf for a non-finite
 * point.  Self-query over points [q_begin, q_begin+q_count) of the indexed
 * cloud; mean[q_count].  Requires k >= 1. */
int pcrs_sor_mean_dist_dev(pcrs_index *index, size_t q_begin, size_t q_count, uint32_t k,
                           float *dmean);

/* Global threshold mu + std_mul*sigma over the finite means, with the
 * reference's sequential-f32 summation order — statistical_outlier.rs:43-62.
 * *has_finite = 0 (and *thr = NaN) when no mean is finite (:49-51). */
int pcrs_sor_threshold_dev(pcrs_ctx *ctx, const float *dmean, size_t n, float std_mul, float *thr,
                           int *has_finite);

/* keep[i] = mean[i] <= thr — statistical_outlier.rs:64-66.  *n_keep nullable. */
int pcrs_sor_mask_dev(pcrs_ctx *ctx, const float *dmean, size_t n, float thr, uint8_t *dkeep,
                      size_t *n_keep);

/* Threshold and mask in one call — statistical_outlier.rs:43-66: the threshold is
 * taken over all n_all means (a multi-GPU caller passes the gathered means of the
 * whole cloud), the mask covers the n means at dmean (its own range).  One read-back
 * at the end returns thr, has_finite and n_keep (each nullable); with no finite mean
 * every keep[i] is 0 (:49-51). */
int pcrs_sor_select_dev(pcrs_ctx *ctx, const float *dmean_all, size_t n_all, float std_mul,
                        const float *dmean, size_t n, uint8_t *dkeep, float *thr, int *has_finite,
                        size_t *n_keep);

/* pcrs_sor_select_dev for callers that cannot allocate (no counterpart in the reference):
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The same
 * kernels work in a static device array of the library (PCRS_SOR_SELECT_STATIC_BYTES, held by
 * every GPU the library runs on until the process ends) instead of the context's scratch, and the
 * results stay on the device: *dresult (device memory, written on the context's stream) receives
 * thr (NaN when no mean is finite), has_finite and n_keep.  The call allocates nothing, frees
 * nothing and does not synchronise.  n_all is limited to PCRS_SOR_SELECT_STATIC_MAX means; beyond
 * it the call fails with PCRS_ERR_INVALID_ARG.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams, two contexts, two threads) race. */
#define PCRS_SOR_SELECT_STATIC_MAX ((size_t)1 << 22)
#define PCRS_SOR_SELECT_STATIC_BYTES ((size_t)1536 * 1024)
typedef struct pcrs_sor_select_result {
    float thr;
    int32_t has_finite;
    uint64_t n_keep;
} pcrs_sor_select_result;
int pcrs_sor_select_static_dev(pcrs_ctx *ctx, const float *dmean_all, size_t n_all, float std_mul,
                               const float *dmean, size_t n, uint8_t *dkeep,
                               pcrs_sor_select_result *dresult);

/* statistical_outlier_removal(cloud, k, std_mul) — statistical_outlier.rs:4-69.
 * keep[n] is the inlier mask (the caller applies PointCloud::select,
 * crates/core/src/cloud.rs:103-140); *n_keep the survivor count; *thr_out
 * (nullable) the threshold, NaN when none was computed.  Edge cases as the
 * reference: n == 0 or k == 0 keeps nothing, n == 1 keeps the point.
 * mean_out (nullable) receives the n per-point mean distances when computed. */
int pcrs_sor(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n, uint32_t k,
             float std_mul, uint8_t *keep, size_t *n_keep, float *thr_out, float *mean_out);

/* -------------------------------------------------------- estimate_normals */

/* estimate_normals_with_viewpoint — crates/normals/src/estimate.rs:19-124,
 * eigen-solve :139-238.  Self-query over points [q_begin, q_begin+q_count);
 * nx/ny/nz[q_count].  Requires k >= 1. */
int pcrs_normals_dev(pcrs_index *index, size_t q_begin, size_t q_count, uint32_t k,
                     const float viewpoint[3], float *dnx, float *dny, float *dnz);

/* One neighbour search feeding both consumers: SOR means from ranks 1..k_sor of
 * the (k_sor+1)-NN list, normals from ranks 0..k_normals-1.  Either output
 * group may be NULL to skip it. */
int pcrs_sor_normals_dev(pcrs_index *index, size_t q_begin, size_t q_count, uint32_t k_sor,
                         uint32_t k_normals, const float viewpoint[3], float *dmean, float *dnx,
                         float *dny, float *dnz);

/* The same fused search over a run of the index's cell-ordered slots instead of
 * an index range (no counterpart in the reference: the unit of work for
 * sharding one cloud over several GPUs by space rather than by input order).
 * pcrs_index_slot_bounds splits the slots into `parts` runs of whole cells of
 * nearly equal length: run r is [bounds[r], bounds[r+1]) (host array of
 * parts + 1 entries; identical on every process that built an index over the
 * same points).  All bounds are 0 when the index has no cell order to split
 * (brute-force index, or a cloud with non-finite points).
 * pcrs_sor_normals_slots_dev answers the queries of one run: dpoint[i] is the
 * original index of the point in slot slot_begin + i, and dmean / dnx / dny /
 * dnz[i] are that point's results — the values pcrs_sor_normals_dev writes at
 * index dpoint[i].  pcrs_scatter_f32_dev is the way back to index order:
 * dout[dpoint[i]] = dvalues[i]. */
int pcrs_index_slot_bounds(pcrs_index *index, uint32_t parts, size_t *bounds);
int pcrs_sor_normals_slots_dev(pcrs_index *index, size_t slot_begin, size_t slot_count,
                               uint32_t k_sor, uint32_t k_normals, const float viewpoint[3],
                               uint32_t *dpoint, float *dmean, float *dnx, float *dny, float *dnz);
int pcrs_scatter_f32_dev(pcrs_ctx *ctx, const float *dvalues, const uint32_t *dpoint, size_t n,
                         float *dout);

/* One cloud over several GPUs by SPACE (no counterpart in the reference; what
 * pointclouds_rs/dist.py runs with ownership="slabs").  Every process holds the
 * whole cloud in device columns, but indexes only its own slab of it plus a halo,
 * and answers the slab's points:
 *
 *   pcrs_slab_cuts_dev    cuts the cloud along its longest axis into `parts` slabs of
 *       nearly equal point counts: slab r = the finite points with cuts[r] <= p[axis]
 *       < cuts[r + 1] (cuts[0] = -INFINITY, cuts[parts] = +INFINITY; host array of
 *       parts + 1 floats).  A pure function of the columns: every process derives the
 *       same cuts without communication.  *halo: how far an index should reach beyond
 *       its slab so that nearly every neighbourhood of the slab lies inside it;
 *       *n_finite: finite points of the cloud (points with a non-finite coordinate
 *       belong to no slab: a cloud that has some is sharded by index ranges instead).
 *   pcrs_index_build_box_dev   an index over the points with box_lo[a] <= p[a] <
 *       box_hi[a] only (-INFINITY / +INFINITY: no limit).  The cloud may go on behind a
 *       limited side of the box, so a search never certifies a result whose ball reaches
 *       across such a side.  The columns are kept, as by pcrs_index_build_dev.
 *   pcrs_index_select_box      picks the indexed points inside [q_lo, q_hi) (a part of
 *       the index's box: the slab without its halo) as the queries of the calls that
 *       follow; *count = their number.  Fails with PCRS_ERR_INVALID_ARG when the box
 *       holds too few points for a grid (shard by index ranges then).
 *   pcrs_sor_normals_selected_dev   the fused search of pcrs_sor_normals_dev over the
 *       selected points, in cell order: dpoint[i] = index of the i-th of them in the
 *       columns, dmean / dnx / dny / dnz[i] = its results (either group may be NULL).
 *       A query whose neighbourhood reaches across a limited side of the box is answered
 *       from an index over the whole cloud, built inside this index on first need;
 *       *n_escalated (nullable) = how many took that road.  Waits for the stream once.
 *   pcrs_slab_index_build_dev   the first three calls in one, for slab `part` of `parts`
 *       (the cloud is measured once: the index takes its bounds from the cut): the index
 *       over the slab and its halo with the slab's points selected, *n_queries of them;
 *       *axis and slab[2] (nullable) = the cut axis and the slab's limits on it.
 *       *out = NULL with PCRS_OK when the cloud has no slabs: points with a non-finite
 *       coordinate, no extent, or too few points in this slab for a grid. */
int pcrs_slab_cuts_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz, size_t m,
                       uint32_t parts, uint32_t k_hint, int *axis, float *cuts, float *halo,
                       size_t *n_finite);
int pcrs_index_build_box_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                             size_t m, uint32_t k_hint, const float box_lo[3],
                             const float box_hi[3], pcrs_index **out);
int pcrs_index_select_box(pcrs_index *index, const float q_lo[3], const float q_hi[3],
                          size_t *count);
int pcrs_sor_normals_selected_dev(pcrs_index *index, uint32_t k_sor, uint32_t k_normals,
                                  const float viewpoint[3], uint32_t *dpoint, float *dmean,
                                  float *dnx, float *dny, float *dnz, size_t *n_escalated);
int pcrs_slab_index_build_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                              size_t m, uint32_t k_hint, uint32_t parts, uint32_t part,
                              pcrs_index **out, size_t *n_queries, int *axis, float slab[2]);

/* The way back to index order for results a process computed for arbitrary points of a sharded
 * cloud (no counterpart in the reference): the index range [0, n_total) is divided among `world`
 * processes as contiguous ranges (the first n_total % world one longer, as pointclouds_rs.dist
 * shard_range has it).  pcrs_route_pairs_dev buckets n (point index, value) pairs by the owner
 * of the index: dsend_point / dsend_values hold the buckets in rank order, counts[world] (host)
 * their sizes -- the send side of an all-to-all; waits for the stream once.
 * pcrs_scatter_range_f32_dev drops received pairs into the owner's range:
 * dout[dpoint[i] - first] = dvalues[i]. */
int pcrs_route_pairs_dev(pcrs_ctx *ctx, const uint32_t *dpoint, const float *dvalues, size_t n,
                         size_t n_total, uint32_t world, uint32_t *dsend_point, float *dsend_values,
                         size_t *counts);
int pcrs_scatter_range_f32_dev(pcrs_ctx *ctx, const float *dvalues, const uint32_t *dpoint,
                               size_t n, size_t first, float *dout);

/* estimate_normals_with_viewpoint on host buffers.  k == 0 or n == 0 writes
 * nothing (the reference returns empty vectors, estimate.rs:25-31). */
int pcrs_normals(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                 uint32_t k, const float viewpoint[3], float *nx, float *ny, float *nz);

/* SOR + normals of one cloud with a single upload and index build. */
int pcrs_sor_normals(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                     uint32_t k_sor, float std_mul, uint32_t k_normals, const float viewpoint[3],
                     uint8_t *keep, size_t *n_keep, float *thr_out, float *nx, float *ny,
                     float *nz);

/* ------------------------------------------------------------ registration */

/* apply_transform — crates/registration/src/icp.rs:39-47,77-92: p' = R p + t,
 * row-major R, additions left to right, no fused multiply-add. */
int pcrs_apply_transform(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                         const float rotation[9], const float translation[3], float *ox,
                         float *oy, float *oz);
int pcrs_apply_transform_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                             size_t n, const float rotation[9], const float translation[3],
                             float *dox, float *doy, float *doz);

/* RigidTransform::compose — icp.rs:52-73: apply (ra,ta) first, then (rb,tb). */
void pcrs_compose(const float ra[9], const float ta[3], const float rb[9], const float tb[3],
                  float rout[9], float tout[3]);

/* find_correspondences — crates/registration/src/correspondence.rs:16-39,
 * dense form: tgt_idx[n] is the nearest target index or PCRS_NO_INDEX where the
 * source point has no correspondence (dist > max_distance, non-finite point,
 * empty target); dist[n] its Euclidean distance (NaN where none).
 * *n_corr (nullable) counts the correspondences. */
int pcrs_find_correspondences(pcrs_index *target, const float *sx, const float *sy,
                              const float *sz, size_t n, float max_distance, uint32_t *tgt_idx,
                              float *dist, size_t *n_corr);
int pcrs_find_correspondences_dev(pcrs_index *target, const float *dsx, const float *dsy,
                                  const float *dsz, size_t n, float max_distance,
                                  uint32_t *dtgt_idx, float *ddist, size_t *n_corr);

/* ICP state: target index (+ target normals for point-to-plane) and the working
 * copy `current` of the source, all device-resident across iterations
 * (the loop of icp.rs:125-206 / icp_plane.rs:20-97).
 * tnx/tny/tnz may be NULL (point-to-point only); n_normals is the length of
 * the normals columns and must equal m for point-to-plane. */
int pcrs_icp_create(pcrs_ctx *ctx, const float *sx, const float *sy, const float *sz, size_t n,
                    const float *tx, const float *ty, const float *tz, size_t m, const float *tnx,
                    const float *tny, const float *tnz, size_t n_normals, pcrs_icp **out);
int pcrs_icp_create_dev(pcrs_ctx *ctx, const float *dsx, const float *dsy, const float *dsz,
                        size_t n, const float *dtx, const float *dty, const float *dtz, size_t m,
                        const float *dtnx, const float *dtny, const float *dtnz, size_t n_normals,
                        pcrs_icp **out);
void pcrs_icp_destroy(pcrs_icp *icp);
/* current <- source (icp.rs:145) */
int pcrs_icp_reset(pcrs_icp *icp);

/* One correspondence pass over `current` fused with the consumer reduction
 * (correspondence.rs:23-36 + icp.rs:221-244,277-281 / icp_plane.rs:148-180).
 * sums[PCRS_ICP_NSUMS], all f64, host memory:
 *   [0]      correspondence count
 *   [1]      sum of dist*dist (each product rounded to f32 first, icp.rs:279)
 *   mode POINT_TO_PLANE: [2..22] upper triangle of A^T A row by row
 *                        (00 01 02 03 04 05 11 12 ... 55), [23..28] A^T b
 *   mode POINT_TO_POINT: [2..4] sum s, [5..7] sum t, [8..16] sum s t^T row-major
 * Unused entries are zero.  Sums are additive across shards of the source, so
 * a multi-GPU job all-reduces this array between accumulate and update. */
int pcrs_icp_accumulate(pcrs_icp *icp, int mode, float max_distance, double *sums);

/* The incremental transform of one iteration from (all-reduced) sums:
 * solve_point_to_plane's Tikhonov + Cholesky/LU + Rodrigues
 * (icp_plane.rs:182-235) or compute_rigid_transform_svd (icp.rs:246-269). */
int pcrs_icp_solve(int mode, const double *sums, float rotation[9], float translation[3]);

/* current <- R*current + t in place (icp.rs:186, icp_plane.rs:78) */
int pcrs_icp_apply(pcrs_icp *icp, const float rotation[9], const float translation[3]);

/* Downloads the working copy of the source. */
int pcrs_icp_get_current(pcrs_icp *icp, float *x, float *y, float *z);

/* icp_point_to_point / icp_point_to_plane — icp.rs:125-206, icp_plane.rs:20-97.
 * Starts from current = source.  Returns PCRS_ERR_NORMALS_MISMATCH for
 * point-to-plane when the normals length differs from the target length. */
int pcrs_icp_run(pcrs_icp *icp, int mode, uint32_t max_iterations, float tolerance,
                 float max_correspondence_distance, pcrs_icp_result *out);

/* The same loop with the source split over several GPUs (one pcrs_icp per rank over its shard
 * of the source and the whole target).  An iteration of icp.rs:147-188 is cut where the sums of
 * the shards meet; the caller's collective runs between the two halves, and nothing in them
 * waits for the device:
 *
 *   pcrs_icp_shard_begin(icp, n_source_total)
 *   for it in 0 .. max_iterations:                     (queue several, then look at the state)
 *     pcrs_icp_shard_accumulate_dev(icp, mode, max_d, it, dsums, &dd2)
 *         search + reduction of this shard into dsums[PCRS_ICP_NSUMS] (device); *dd2 (nullable
 *         argument) = device address of dist*dist of the shard's points in their input order
 *         (icp.rs:279), owned by the handle and valid until iteration it + 2 is queued
 *     all-gather of dsums (and of dd2 where the reference's own rmse is wanted) over the ranks
 *     pcrs_icp_shard_update_dev(icp, mode, dsums_all, world, tolerance, dd2_all, metrics_only)
 *         adds the world * PCRS_ICP_NSUMS doubles in rank order (identical on every rank), then
 *         metrics, convergence test, solve and compose on the device (icp.rs:163-186); the
 *         increment is applied to the shard by the next accumulate.  dd2_all: dist*dist of ALL
 *         n_source_total points in index order, summed left to right in f32 as icp.rs:277-281
 *         does; NULL takes the rmse from the f64 sums (then pcrs_icp_shard_rmse_dev sets the
 *         reported value afterwards).  metrics_only: the single pass of icp.rs:190-197
 *   pcrs_icp_shard_state(icp, &result, &done, &last_live)   waits for the stream
 *         done: the loop has ended (iterations queued behind it were no-ops); last_live: the
 *         1-based iteration the reported metrics come from (0: none)
 *   pcrs_icp_shard_finish(icp)    applies the last increment to the working copy
 *
 * Every rank passes the same arguments and sees the same state, so all ranks take the same
 * decisions without exchanging them. */
int pcrs_icp_shard_begin(pcrs_icp *icp, size_t n_source_total);
int pcrs_icp_shard_accumulate_dev(pcrs_icp *icp, int mode, float max_distance, uint32_t iteration,
                                  double *dsums, const float **dd2);
int pcrs_icp_shard_update_dev(pcrs_icp *icp, int mode, const double *dsums_all, uint32_t world,
                              float tolerance, const float *dd2_all, int metrics_only);
int pcrs_icp_shard_rmse_dev(pcrs_icp *icp, const float *dd2_all);
int pcrs_icp_shard_state(pcrs_icp *icp, pcrs_icp_result *out, int *done, uint32_t *last_live);
int pcrs_icp_shard_finish(pcrs_icp *icp);

/* Whole registration on host buffers (create + run + destroy). */
int pcrs_icp_register(pcrs_ctx *ctx, int mode, const float *sx, const float *sy, const float *sz, size_t n,
             const float *tx, const float *ty, const float *tz, size_t m, const float *tnx,
             const float *tny, const float *tnz, size_t n_normals, uint32_t max_iterations,
             float tolerance, float max_correspondence_distance, pcrs_icp_result *out);

/* ---------------------------------------------------- several GPUs, from C */

/* One cloud over the GPUs of a node with the collectives issued by the library itself on RCCL
 * (no counterpart in the reference, which is a single-process CPU library): the two sharded
 * drivers of pointclouds_rs/dist.py -- sor_normals_sharded(ownership="slabs") with its fallback
 * to index ranges (dist.py:426-573) and icp_sharded (dist.py:576-641) -- as one call per rank,
 * for hosts without Python.  One process (or thread) per GPU, each with its own context; every
 * rank makes the same call with the same arguments apart from its shard.  RCCL is loaded when
 * the first of these entry points runs (librccl.so.1, the copy already in the process if there is
 * one), so the library itself loads on a machine without it; they fail with PCRS_ERR_HIP when it
 * cannot be found.
 *
 * `nccl_comm` is an ncclComm_t over the participating ranks: one the host already has, or one
 * from pcrs_dist_comm_init -- rank 0 fills id[PCRS_DIST_UNIQUE_ID_BYTES] with
 * pcrs_dist_unique_id and hands the bytes to the other ranks by whatever means the host has (a
 * file, a socket, MPI), then every rank calls pcrs_dist_comm_init with its rank.  The collectives
 * run on the context's stream.
 *
 * The cloud is sharded in index order: rank r holds the contiguous range of n_total points that
 * starts at r * (n_total / world) + min(r, n_total % world), the first n_total % world ranges
 * one point longer (shard_range, dist.py:36-40).
 *
 * pcrs_dist_sor_normals_dev   dx/dy/dz: this rank's n_local points (device).  All ranks gather
 *   the columns, every rank indexes its own slab of space plus a halo and answers the slab's
 *   points (pcrs_slab_index_build_dev, pcrs_sor_normals_selected_dev), the means are routed to
 *   the owners of their index ranges and gathered in index order, and every rank derives the
 *   same threshold from them (statistical_outlier.rs:43-62) and writes the mask of the points it
 *   answered.  Outputs (device, room for `capacity` points each; a slab holds about
 *   n_total / world points, at most n_total): dpoint[i] = index in the whole cloud of the i-th
 *   answered point, dmean / dkeep / dnx / dny / dnz[i] its results -- bit for bit what
 *   pcrs_sor_normals writes at that index on one GPU.  dnx/dny/dnz may be NULL (no normals);
 *   k_sor == 0 skips the outlier test (dmean, dkeep may then be NULL).  The points answered by
 *   all ranks partition the cloud.  When the cloud has no slabs (non-finite points, no extent,
 *   too few points) every rank answers its own index range from an index over the whole cloud
 *   and result->by_slabs is 0.  Fails with PCRS_ERR_INVALID_ARG when `capacity` is smaller than
 *   the number of points this rank has to answer.
 * pcrs_dist_icp_run   the loop of icp.rs:125-206 / icp_plane.rs:20-97 with the source split over
 *   the ranks: `icp` holds this rank's shard of the source (its length must be the rank's index
 *   range of n_source_total) and the whole target.  Per iteration the ranks all-gather their
 *   PCRS_ICP_NSUMS partial sums (and, for tolerance > 0, dist*dist of their points, so that the
 *   loop is steered by the reference's own f32 rmse); every rank ends with the same result, equal
 *   to pcrs_icp_run's on one GPU over the whole source up to the order of the f64 partial sums. */
#define PCRS_DIST_UNIQUE_ID_BYTES 128
typedef struct pcrs_dist_sor_result {
    size_t n_answered;  /* points this rank answered (entries of the outputs) */
    size_t n_keep;      /* kept among them */
    size_t n_escalated; /* answered from the whole-cloud index instead of the slab's */
    float threshold;    /* the same on every rank; NaN when no mean is finite */
    int has_finite;
    int by_slabs;       /* 1: ownership by slabs of space, 0: by index ranges */
} pcrs_dist_sor_result;
int pcrs_dist_unique_id(char id[PCRS_DIST_UNIQUE_ID_BYTES]);
int pcrs_dist_comm_init(pcrs_ctx *ctx, const char id[PCRS_DIST_UNIQUE_ID_BYTES], int rank, int world,
                        void **nccl_comm);
void pcrs_dist_comm_destroy(void *nccl_comm);
int pcrs_dist_sor_normals_dev(pcrs_ctx *ctx, void *nccl_comm, int rank, int world, const float *dx,
                              const float *dy, const float *dz, size_t n_local, size_t n_total,
                              uint32_t k_sor, float std_mul, uint32_t k_normals,
                              const float viewpoint[3], size_t capacity, uint32_t *dpoint,
                              uint8_t *dkeep, float *dmean, float *dnx, float *dny, float *dnz,
                              pcrs_dist_sor_result *result);
int pcrs_dist_icp_run(pcrs_icp *icp, void *nccl_comm, int rank, int world, size_t n_source_total,
                      int mode, uint32_t max_iterations, float tolerance,
                      float max_correspondence_distance, pcrs_icp_result *out);

/* ------------------------------------------------------------ diagnostics */

/* Benchmark instrumentation (no counterpart in the reference): when enabled,
 * every launch of the neighbour-search kernel is bracketed by HIP events on
 * the context's stream.  pcrs_ctx_timing_read synchronises, returns the summed
 * kernel time and the number of bracketed launches since the last read, and
 * resets both. */
int pcrs_ctx_timing(pcrs_ctx *ctx, int enable);
int pcrs_ctx_timing_read(pcrs_ctx *ctx, double *total_ms, uint64_t *launches);

/* Counters of the last query call on this index: [0] queries answered by the
 * grid tile kernel, [1] by the ring-search fallback, [2] by brute force. */
int pcrs_index_last_stats(const pcrs_index *index, uint64_t stats[3]);

/* How many grids besides its own the index holds for clouds of very uneven density (no
 * counterpart in the reference): *n_finer levels over the crowded part, each inside the one
 * before, and *n_coarser grids of wider cells for the sparse part.  Both 0 for an evenly
 * sampled cloud, a brute-force index or an index over a box. */
int pcrs_index_levels(const pcrs_index *index, uint32_t *n_finer, uint32_t *n_coarser);

#ifdef __cplusplus
}
#endif
#endif /* PCRS_HIP_H */
