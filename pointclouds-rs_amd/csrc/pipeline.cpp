// The obstacle pipeline of the reference's KITTI example as ONE call
// (examples/python/kitti_obstacle_detection.py:87-124: voxel_downsample -> statistical_outlier_removal
// -> ransac_plane -> drop the plane's inliers -> euclidean_cluster).  Every stage is the stage's own
// entry point; what this adds is that the cloud goes up once and every intermediate cloud stays
// in device memory: a frame of 57 K points is bound by the host's copies and waits when each stage
// takes host buffers in and out.
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "pcrs_internal.h"

// ransac_plane.rs:141-165 (three distinct indices, up to 100 redraws each) over a SplitMix64
// stream: draw = (next() * n) >> 64.  The reference draws from rand's StdRng, whose stream cannot
// be reproduced without that crate; this is the generator of the Python mirror (sample_triples),
// draw for draw.
extern "C" int pcrs_sample_triples(size_t n, size_t iterations, uint64_t seed, uint32_t *triples,
                                   size_t *n_triples) {
    if (!n_triples || (iterations && !triples)) return PCRS_ERR_INVALID_ARG;
    *n_triples = 0;
    if (n < 3 || n >= 0xFFFFFFFFull) return PCRS_OK;
    uint64_t state = seed;
    auto below = [&](uint64_t m) -> uint64_t {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t v = state;
        v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
        v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
        v ^= v >> 31;
        return (uint64_t)(((unsigned __int128)v * m) >> 64);
    };
    size_t out = 0;
    for (size_t it = 0; it < iterations; ++it) {
        const uint64_t i0 = below(n);
        uint64_t i1 = below(n);
        int attempts = 0;
        while (i1 == i0 && attempts <= 100) {
            i1 = below(n);
            ++attempts;
        }
        if (i1 == i0) continue;
        uint64_t i2 = below(n);
        attempts = 0;
        while ((i2 == i0 || i2 == i1) && attempts <= 100) {
            i2 = below(n);
            ++attempts;
        }
        if (i2 == i0 || i2 == i1) continue;
        triples[3 * out] = (uint32_t)i0;
        triples[3 * out + 1] = (uint32_t)i1;
        triples[3 * out + 2] = (uint32_t)i2;
        ++out;
    }
    *n_triples = out;
    return PCRS_OK;
}

namespace {
int check_hip(pcrs_ctx *ctx, hipError_t e, const char *what) {
    if (e == hipSuccess) return PCRS_OK;
    (void)hipGetLastError();
    return pcrs_set_err(ctx, e == hipErrorOutOfMemory ? PCRS_ERR_OOM : PCRS_ERR_HIP, "%s failed: %s", what,
                        hipGetErrorString(e));
}
}  // namespace

extern "C" int pcrs_obstacle_pipeline(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                                      size_t n, const pcrs_obstacle_params *prm,
                                      pcrs_obstacle_result *res, float *obst_x, float *obst_y,
                                      float *obst_z, uint32_t *cluster_offsets,
                                      uint32_t *cluster_indices) try {
    if (!ctx || !prm || !res) return PCRS_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return pcrs_set_err(ctx, PCRS_ERR_HIP, "hipSetDevice failed");
    *res = pcrs_obstacle_result{};
    res->sor_threshold = std::numeric_limits<float>::quiet_NaN();
    res->plane[2] = 1.0f;  // PlaneModel::default(), ransac_plane.rs:23-30
    if (!(prm->voxel_size > 0.0f) || !std::isfinite(prm->voxel_size))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "voxel_size must be > 0 and finite");
    if (!(prm->sor_std_mul >= 0.0f) || !std::isfinite(prm->sor_std_mul))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "std_mul must be >= 0 and finite");
    if (n >= 0xFFFFFFFFull) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    if (n == 0) return PCRS_OK;
    if (!x || !y || !z) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    hipStream_t st = ctx->stream;
    const size_t nb = n * sizeof(float);
    // the frame goes up once
    DevBuf dx, dy, dz, vx, vy, vz;
    PCRS_TRY(check_hip(ctx, dx.alloc(ctx, nb), "alloc"));
    PCRS_TRY(check_hip(ctx, dy.alloc(ctx, nb), "alloc"));
    PCRS_TRY(check_hip(ctx, dz.alloc(ctx, nb), "alloc"));
    PCRS_TRY(check_hip(ctx, hipMemcpyAsync(dx.p, x, nb, hipMemcpyHostToDevice, st), "upload"));
    PCRS_TRY(check_hip(ctx, hipMemcpyAsync(dy.p, y, nb, hipMemcpyHostToDevice, st), "upload"));
    PCRS_TRY(check_hip(ctx, hipMemcpyAsync(dz.p, z, nb, hipMemcpyHostToDevice, st), "upload"));
    // 1. voxel_downsample (voxel_downsample.rs:12-65)
    PCRS_TRY(check_hip(ctx, vx.alloc(ctx, nb), "alloc"));
    PCRS_TRY(check_hip(ctx, vy.alloc(ctx, nb), "alloc"));
    PCRS_TRY(check_hip(ctx, vz.alloc(ctx, nb), "alloc"));
    size_t nv = 0;
    // (the voxel keys bound the centroids: the index over them needs no pass for its bounding box)
    double vbounds[6] = {NAN, NAN, NAN, NAN, NAN, NAN};
    PCRS_TRY(pcrs_launch_voxel_downsample(ctx, dx.as<float>(), dy.as<float>(), dz.as<float>(), n,
                                          prm->voxel_size, vx.as<float>(), vy.as<float>(),
                                          vz.as<float>(), &nv, vbounds));
    res->n_voxels = nv;
    if (nv == 0) return PCRS_OK;
    // 2. statistical_outlier_removal (statistical_outlier.rs:4-69) -> the kept points, compacted
    DevBuf cx, cy, cz, keep;
    PCRS_TRY(check_hip(ctx, cx.alloc(ctx, nv * sizeof(float)), "alloc"));
    PCRS_TRY(check_hip(ctx, cy.alloc(ctx, nv * sizeof(float)), "alloc"));
    PCRS_TRY(check_hip(ctx, cz.alloc(ctx, nv * sizeof(float)), "alloc"));
    PCRS_TRY(check_hip(ctx, keep.alloc(ctx, nv), "alloc"));
    size_t nk = 0;
    if (prm->sor_k == 0) {  // :5-7: an empty cloud
        nk = 0;
    } else if (nv == 1) {  // :10-12
        nk = 1;
        PCRS_TRY(check_hip(ctx, hipMemsetAsync(keep.p, 1, 1, st), "memset"));
    } else {
        pcrs_index *ix = nullptr;
        PCRS_TRY(pcrs_index_build_hinted_dev(ctx, vx.as<float>(), vy.as<float>(), vz.as<float>(), nv,
                                             prm->sor_k + 1, &ix, vbounds));
        DevBuf mean;
        int rc = check_hip(ctx, mean.alloc(ctx, nv * sizeof(float)), "alloc");
        if (rc == PCRS_OK) rc = pcrs_sor_mean_dist_dev(ix, 0, nv, prm->sor_k, mean.as<float>());
        float thr = 0.0f;
        int ok = 0;
        if (rc == PCRS_OK)
            rc = pcrs_sor_select_dev(ctx, mean.as<float>(), nv, prm->sor_std_mul, mean.as<float>(), nv,
                                     keep.as<uint8_t>(), &thr, &ok, &nk);
        pcrs_index_destroy(ix);
        PCRS_TRY(rc);
        if (ok) res->sor_threshold = thr;
    }
    res->n_kept = nk;
    if (nk == 0) return PCRS_OK;
    PCRS_TRY(pcrs_launch_select(ctx, vx.as<float>(), vy.as<float>(), vz.as<float>(), nv,
                                keep.as<uint8_t>(), 0, cx.as<float>(), cy.as<float>(), cz.as<float>(),
                                nullptr, nullptr));
    // 3. ransac_plane_seeded (ransac_plane.rs:56-138) over the mirror's sample stream
    // (generated straight into pinned memory: the upload is queued, not staged and waited for)
    void *io = nullptr;
    PCRS_TRY(pcrs_ctx_pinned_io(ctx, (3 * (size_t)prm->plane_iterations + 3) * sizeof(uint32_t), &io));
    uint32_t *triples = reinterpret_cast<uint32_t *>(io);
    size_t ns = 0;
    PCRS_TRY(pcrs_sample_triples(nk, prm->plane_iterations, prm->plane_seed, triples, &ns));
    DevBuf dsamples, inl;
    PCRS_TRY(check_hip(ctx, dsamples.alloc(ctx, (3 * ns + 3) * sizeof(uint32_t)), "alloc"));
    PCRS_TRY(check_hip(ctx, inl.alloc(ctx, nk), "alloc"));
    if (ns)
        PCRS_TRY(check_hip(ctx, hipMemcpyAsync(dsamples.p, triples, 3 * ns * sizeof(uint32_t),
                                               hipMemcpyHostToDevice, st), "upload"));
    size_t n_in = 0;
    uint64_t best = 0;
    PCRS_TRY(pcrs_launch_ransac_plane(ctx, cx.as<float>(), cy.as<float>(), cz.as<float>(), nk,
                                      dsamples.as<uint32_t>(), ns, prm->plane_threshold, res->plane,
                                      inl.as<uint8_t>(), &n_in, &best));
    res->n_ground = n_in;
    // 4. the points off the plane (select_inverse, cloud.rs:152-162)
    const size_t no = nk - n_in;
    res->n_obstacles = no;
    if (no == 0) return PCRS_OK;
    DevBuf ox, oy, oz;
    PCRS_TRY(check_hip(ctx, ox.alloc(ctx, nk * sizeof(float)), "alloc"));
    PCRS_TRY(check_hip(ctx, oy.alloc(ctx, nk * sizeof(float)), "alloc"));
    PCRS_TRY(check_hip(ctx, oz.alloc(ctx, nk * sizeof(float)), "alloc"));
    PCRS_TRY(pcrs_launch_select(ctx, cx.as<float>(), cy.as<float>(), cz.as<float>(), nk,
                                inl.as<uint8_t>(), 1, ox.as<float>(), oy.as<float>(), oz.as<float>(),
                                nullptr, nullptr));
    if (obst_x && obst_y && obst_z) {
        PCRS_TRY(check_hip(ctx, hipMemcpyAsync(obst_x, ox.p, no * sizeof(float), hipMemcpyDeviceToHost, st), "download"));
        PCRS_TRY(check_hip(ctx, hipMemcpyAsync(obst_y, oy.p, no * sizeof(float), hipMemcpyDeviceToHost, st), "download"));
        PCRS_TRY(check_hip(ctx, hipMemcpyAsync(obst_z, oz.p, no * sizeof(float), hipMemcpyDeviceToHost, st), "download"));
    }
    // 5. euclidean_cluster (euclidean_cluster.rs:96-187) over the obstacle points
    DevBuf doff, dind;
    PCRS_TRY(check_hip(ctx, doff.alloc(ctx, (no + 1) * sizeof(uint32_t)), "alloc"));
    PCRS_TRY(check_hip(ctx, dind.alloc(ctx, no * sizeof(uint32_t)), "alloc"));
    size_t nc = 0, ni = 0;
    PCRS_TRY(pcrs_launch_euclidean_cluster(ctx, ox.as<float>(), oy.as<float>(), oz.as<float>(), no,
                                           prm->cluster_threshold, prm->cluster_min_size,
                                           prm->cluster_max_size, doff.as<uint32_t>(),
                                           dind.as<uint32_t>(), &nc, &ni));
    res->n_clusters = nc;
    res->n_cluster_points = ni;
    // the cluster lists come back through pinned memory (two queued copies and one wait; a copy
    // into the caller's pageable arrays is staged and waited for, each on its own)
    const size_t noff = cluster_offsets ? nc + 1 : 0, nind = cluster_indices ? ni : 0;
    uint32_t *back = nullptr;
    if (noff + nind) {
        // (the sample triples in this area have been consumed: the plane stage has waited for its result)
        PCRS_TRY(pcrs_ctx_pinned_io(ctx, (noff + nind) * sizeof(uint32_t), &io));
        back = reinterpret_cast<uint32_t *>(io);
        if (noff)
            PCRS_TRY(check_hip(ctx, hipMemcpyAsync(back, doff.p, noff * sizeof(uint32_t), hipMemcpyDeviceToHost, st),
                               "download"));
        if (nind)
            PCRS_TRY(check_hip(ctx, hipMemcpyAsync(back + noff, dind.p, nind * sizeof(uint32_t),
                                                   hipMemcpyDeviceToHost, st), "download"));
    }
    PCRS_TRY(check_hip(ctx, hipStreamSynchronize(st), "stream synchronize"));
    if (noff) std::memcpy(cluster_offsets, back, noff * sizeof(uint32_t));
    if (nind) std::memcpy(cluster_indices, back + noff, nind * sizeof(uint32_t));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}
