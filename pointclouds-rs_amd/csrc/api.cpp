// C-ABI of libpcrs_hip.so: argument checking, device-memory plumbing and the host-side control
// flow of the reference functions (edge cases, ICP outer loop).  All numeric work on point data is
// done by the HIP kernels; there is no CPU fallback — without a GPU every entry point fails.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <exception>
#include <new>

#include "pcrs_internal.h"

// ------------------------------------------------------------------------------------ helpers

int pcrs_set_err(pcrs_ctx *ctx, int code, const char *fmt, ...) {
    if (!ctx) return code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
    return code;
}

// Called from the catch-all of every entry point: nothing may unwind through the C boundary
// (the caller is Rust, C or ctypes).  Turns the exception in flight into a status code.
int pcrs_on_exception(pcrs_ctx *ctx) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    } catch (const std::exception &e) {
        return pcrs_set_err(ctx, PCRS_ERR_HIP, "internal error: %s", e.what());
    } catch (...) {
        return pcrs_set_err(ctx, PCRS_ERR_HIP, "internal error: unknown exception");
    }
}

int pcrs_ctx_scratch(pcrs_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            PCRS_HIP_CHECK(ctx, hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes + bytes / 4 + 4096;
        PCRS_HIP_CHECK(ctx, hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return PCRS_OK;
}

int pcrs_ctx_pinned(pcrs_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->pinned_bytes) {
        if (ctx->pinned) {
            PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            PCRS_HIP_CHECK(ctx, hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
        }
        size_t want = bytes + bytes / 4 + 4096;
        PCRS_HIP_CHECK(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return PCRS_OK;
}

int pcrs_ctx_pinned_io(pcrs_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->pinned_io_bytes) {
        if (ctx->pinned_io) {
            PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            PCRS_HIP_CHECK(ctx, hipHostFree(ctx->pinned_io));
            ctx->pinned_io = nullptr;
            ctx->pinned_io_bytes = 0;
        }
        size_t want = bytes + bytes / 4 + 65536;
        PCRS_HIP_CHECK(ctx, hipHostMalloc(&ctx->pinned_io, want, hipHostMallocDefault));
        ctx->pinned_io_bytes = want;
    }
    *out = ctx->pinned_io;
    return PCRS_OK;
}

hipError_t pcrs_pool_alloc(pcrs_ctx *ctx, size_t bytes, void **p, size_t *cap) {
    // best fit among the free buffers that are not wastefully large
    size_t best = ctx->pool.size();
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
        const size_t c = ctx->pool[i].bytes;
        if (c >= bytes && c <= 2 * bytes + (1u << 20) &&
            (best == ctx->pool.size() || c < ctx->pool[best].bytes))
            best = i;
    }
    if (best != ctx->pool.size()) {
        *p = ctx->pool[best].p;
        *cap = ctx->pool[best].bytes;
        ctx->pool[best] = ctx->pool.back();
        ctx->pool.pop_back();
        return hipSuccess;
    }
    // round up so that slightly different sizes of the next call still fit
    size_t want = bytes < (1u << 20) ? ((bytes + 65535) & ~(size_t)65535)
                                     : ((bytes + bytes / 8 + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1));
    hipError_t e = hipMalloc(p, want);
    ++ctx->pool_mallocs;
    if (e != hipSuccess && !ctx->pool.empty()) {
        // give the pooled memory back and retry once
        (void)hipStreamSynchronize(ctx->stream);
        for (auto &pe : ctx->pool) (void)hipFree(pe.p);
        ctx->pool.clear();
        e = hipMalloc(p, want);
    }
    if (e == hipSuccess) *cap = want;
    else (void)hipGetLastError();  // do not leave the failure for a later check to find
    return e;
}

void pcrs_pool_release(pcrs_ctx *ctx, void *p, size_t cap) {
    // Keep the pool bounded, by dropping the smallest idle buffer: an index over a scene with two
    // dozen crowded regions gives back two hundred buffers at once, and the ones worth keeping
    // are the large ones (a sorted copy of 100 M points is 1.6 GB: allocating and freeing it
    // costs milliseconds, and freeing waits for the stream).
    if (ctx->pool.size() >= 1024) {
        size_t small = 0;
        for (size_t i = 1; i < ctx->pool.size(); ++i)
            if (ctx->pool[i].bytes < ctx->pool[small].bytes) small = i;
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->pool[small].p);
        ++ctx->pool_drops;
        ctx->pool[small] = ctx->pool.back();
        ctx->pool.pop_back();
    }
    ctx->pool.push_back({p, cap});
}

void pcrs_timing_begin(pcrs_ctx *ctx) {
    if (!ctx->timing) return;
    if (ctx->timing_depth++ > 0) return;
    if (ctx->ev_used == ctx->ev_start.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            ctx->timing = false;
            return;
        }
        ctx->ev_start.push_back(a);
        ctx->ev_stop.push_back(b);
    }
    (void)hipEventRecord(ctx->ev_start[ctx->ev_used], ctx->stream);
}

void pcrs_timing_end(pcrs_ctx *ctx) {
    if (!ctx->timing) return;
    if (--ctx->timing_depth > 0) return;
    ctx->timing_depth = 0;
    if (ctx->ev_used >= ctx->ev_stop.size()) return;
    (void)hipEventRecord(ctx->ev_stop[ctx->ev_used], ctx->stream);
    ctx->ev_used++;
}

namespace {

int bind_device(pcrs_ctx *ctx) {
    if (!ctx) return PCRS_ERR_INVALID_ARG;
    PCRS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return PCRS_OK;
}

int upload(pcrs_ctx *ctx, DevBuf &dst, const float *src, size_t n) {
    if (n >= 0xFFFFFFFFull)  // also keeps n * sizeof(float) from wrapping around
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points (got %zu)", n);
    PCRS_HIP_CHECK(ctx, dst.alloc(ctx, n * sizeof(float)));
    if (n) PCRS_HIP_CHECK(ctx, hipMemcpyAsync(dst.p, src, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return PCRS_OK;
}

int download(pcrs_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes) PCRS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return PCRS_OK;
}

int run_query(pcrs_index *ix, const KnnQuery &q, const KnnOut &out, uint32_t kneed) {
    ix->stats[0] = ix->stats[1] = ix->stats[2] = 0;
    ix->stats_dev = nullptr;
    ix->left_list = ix->left_count = nullptr;
    if (ix->algo == PCRS_ALGO_GRID) return pcrs_launch_knn_grid(ix, q, out, kneed);
    ix->stats[2] = q.nq;
    q.xform_fused = false;
    if (q.xform)  // a pending increment of the queries: the brute-force kernel does not apply it
        PCRS_TRY(pcrs_launch_transform_state(ix->ctx, q.wx, q.wy, q.wz, q.nq,
                                             reinterpret_cast<const IcpLoopState *>(q.xform)));
    return pcrs_launch_knn_bf(ix->ctx, ix->x, ix->y, ix->z, ix->m, q, out, kneed, nullptr, nullptr,
                              0);
}

// Below this size the whole cloud fits a handful of LDS tiles and the grid build is not worth it.
constexpr size_t kGridMinPoints = 4096;

int finish_index(pcrs_index *ix, uint32_t k_hint, int algo) {
    if (algo != PCRS_ALGO_AUTO && algo != PCRS_ALGO_BRUTE && algo != PCRS_ALGO_GRID)
        return pcrs_set_err(ix->ctx, PCRS_ERR_INVALID_ARG, "unknown algo %d", algo);
    if (ix->m >= 0xFFFFFFFFull)
        return pcrs_set_err(ix->ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    ix->algo = PCRS_ALGO_BRUTE;
    const bool want_grid =
        algo == PCRS_ALGO_GRID || (algo == PCRS_ALGO_AUTO && ix->m >= kGridMinPoints);
    if (want_grid && ix->m > 0) {
        PCRS_TRY(pcrs_grid_build(ix, k_hint));  // leaves algo == BRUTE when a grid cannot help
        // finer grids over crowded regions, for the searches of the cloud's own points (an index
        // over a caller's box -- a rank's slab -- answers explicit queries and stays single)
        if (ix->box.unbounded()) PCRS_TRY(pcrs_grid_refine(ix, k_hint));
    }
    return PCRS_OK;
}

}  // namespace

int pcrs_index_build_hinted_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                                size_t m, uint32_t k_hint, pcrs_index **out, const double *bounds) {
    *out = nullptr;
    pcrs_index *ix = new (std::nothrow) pcrs_index();
    if (!ix) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    ix->ctx = ctx;
    ix->x = dx;
    ix->y = dy;
    ix->z = dz;
    ix->m = m;
    ix->k_hint = k_hint;
    ix->use_hint = true;
    if (bounds && std::isfinite(bounds[0] + bounds[1] + bounds[2] + bounds[3] + bounds[4] + bounds[5])) {
        ix->have_bounds = true;
        ix->bounds_count_exact = true;
        ix->bounds_n_upper = m;
        for (int a = 0; a < 3; ++a) {
            ix->bounds_lo[a] = bounds[a];
            ix->bounds_hi[a] = bounds[3 + a];
        }
    }
    const int rc = finish_index(ix, k_hint, PCRS_ALGO_AUTO);
    if (rc != PCRS_OK) {
        delete ix;
        return rc;
    }
    *out = ix;
    return PCRS_OK;
}

// ------------------------------------------------------------------------------------ context

extern "C" {

const char *pcrs_version(void) { return "pcrs-hip 0.1 (gfx950)"; }

int pcrs_ctx_create(int device, pcrs_ctx **out) try {
    if (!out) return PCRS_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PCRS_ERR_HIP;
    if (device < 0 || device >= count) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = new (std::nothrow) pcrs_ctx();
    if (!ctx) return PCRS_ERR_OOM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return PCRS_ERR_HIP;
    }
    ctx->own_stream = true;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
        cus > 0)
        ctx->num_cus = cus;
    // the static scratch array of this device, resolved while set-up work is still allowed
    if (pcrs_static_scratch(ctx, &ctx->static_scratch, &ctx->static_scratch_bytes) != PCRS_OK) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return PCRS_ERR_HIP;
    }
    *out = ctx;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(nullptr);
}

int pcrs_ctx_set_stream(pcrs_ctx *ctx, void *hip_stream) try {
    PCRS_TRY(bind_device(ctx));
    // Pooled buffers, the scratch and the pinned read-back area are reused in stream order: work
    // still in flight on the old stream (or on the copy stream) must be through before the
    // first call on the new one can touch them, whoever owns the old stream.
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (ctx->own_stream && ctx->stream) PCRS_HIP_CHECK(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

void pcrs_ctx_destroy(pcrs_ctx *ctx) try {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_io) (void)hipHostFree(ctx->pinned_io);
    if (getenv("PCRS_POOL_DEBUG")) {  // experiments only
        size_t bytes = 0;
        for (auto &pe : ctx->pool) bytes += pe.bytes;
        fprintf(stderr, "pcrs pool: %zu idle buffers, %.1f MB, %zu allocations from the device, %zu dropped\n",
                ctx->pool.size(), (double)bytes / 1e6, ctx->pool_mallocs, ctx->pool_drops);
    }
    for (auto &pe : ctx->pool) (void)hipFree(pe.p);
    for (hipEvent_t e : ctx->ev_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_stop) (void)hipEventDestroy(e);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->copy_event) (void)hipEventDestroy(ctx->copy_event);
    for (hipStream_t st : ctx->side_streams)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    for (hipEvent_t e : ctx->level_events) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
} catch (...) {
}

int pcrs_ctx_timing(pcrs_ctx *ctx, int enable) try {
    PCRS_TRY(bind_device(ctx));
    ctx->timing = enable != 0;
    ctx->ev_used = 0;
    ctx->timing_depth = 0;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_ctx_timing_read(pcrs_ctx *ctx, double *total_ms, uint64_t *launches) try {
    PCRS_TRY(bind_device(ctx));
    if (!total_ms || !launches) return PCRS_ERR_INVALID_ARG;
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0.0f;
        PCRS_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = ctx->ev_used;
    ctx->ev_used = 0;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

const char *pcrs_last_error(const pcrs_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int pcrs_ctx_synchronize(pcrs_ctx *ctx) try {
    PCRS_TRY(bind_device(ctx));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

// -------------------------------------------------------------------------------------- index

int pcrs_index_build_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                         size_t m, uint32_t k_hint, int algo, pcrs_index **out) try {
    if (!out) return PCRS_ERR_INVALID_ARG;
    *out = nullptr;
    PCRS_TRY(bind_device(ctx));
    if (m && (!dx || !dy || !dz))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    pcrs_index *ix = new (std::nothrow) pcrs_index();
    if (!ix) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    ix->ctx = ctx;
    ix->x = dx;
    ix->y = dy;
    ix->z = dz;
    ix->m = m;
    int rc = finish_index(ix, k_hint, algo);
    if (rc != PCRS_OK) {
        delete ix;
        return rc;
    }
    *out = ix;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_index_build(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t m,
                     uint32_t k_hint, int algo, pcrs_index **out) try {
    if (!out) return PCRS_ERR_INVALID_ARG;
    *out = nullptr;
    PCRS_TRY(bind_device(ctx));
    if (m && (!x || !y || !z)) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    pcrs_index *ix = new (std::nothrow) pcrs_index();
    if (!ix) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    ix->ctx = ctx;
    int rc = upload(ctx, ix->own_x, x, m);
    if (rc == PCRS_OK) rc = upload(ctx, ix->own_y, y, m);
    if (rc == PCRS_OK) rc = upload(ctx, ix->own_z, z, m);
    if (rc == PCRS_OK) {
        ix->x = ix->own_x.as<float>();
        ix->y = ix->own_y.as<float>();
        ix->z = ix->own_z.as<float>();
        ix->m = m;
        rc = finish_index(ix, k_hint, algo);
    }
    if (rc == PCRS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = pcrs_set_err(ctx, PCRS_ERR_HIP, "stream synchronize failed after index upload");
    if (rc != PCRS_OK) {
        delete ix;
        return rc;
    }
    *out = ix;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

void pcrs_index_destroy(pcrs_index *index) try {
    if (!index) return;
    // Grid buffers return to the context's pool (stream-ordered reuse, no wait); buffers uploaded
    // from host memory are hipFree'd, which waits for the device by itself.
    if (index->ctx) (void)hipSetDevice(index->ctx->device);
    delete index;
} catch (...) {
}

size_t pcrs_index_len(const pcrs_index *index) { return index ? index->m : 0; }

int pcrs_index_levels(const pcrs_index *index, uint32_t *n_finer, uint32_t *n_coarser) {
    if (!index || !n_finer || !n_coarser) return PCRS_ERR_INVALID_ARG;
    *n_finer = *n_coarser = 0;
    std::vector<const pcrs_index *> todo(index->kids.begin(), index->kids.end());
    while (!todo.empty()) {
        const pcrs_index *l = todo.back();
        todo.pop_back();
        ++*n_finer;
        todo.insert(todo.end(), l->kids.begin(), l->kids.end());
    }
    for (const pcrs_index *l = index->coarse; l; l = l->coarse) ++*n_coarser;
    return PCRS_OK;
}

int pcrs_index_last_stats(const pcrs_index *index, uint64_t stats[3]) try {
    if (!index || !stats) return PCRS_ERR_INVALID_ARG;
    if (index->stats_dev) {
        pcrs_ctx *ctx = index->ctx;
        uint32_t c[4] = {0, 0, 0, 0};
        if (hipSetDevice(ctx->device) != hipSuccess ||
            hipMemcpyAsync(c, index->stats_dev, sizeof(c), hipMemcpyDeviceToHost, ctx->stream) !=
                hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return PCRS_ERR_HIP;
        stats[0] = index->stats_nq - c[0];  // certified by the 27-cell cube
        stats[1] = c[0] - c[3];             // certified by a larger cube (wave-per-query)
        stats[2] = c[3];                    // brute force
        return PCRS_OK;
    }
    for (int i = 0; i < 3; ++i) stats[i] = index->stats[i];
    return PCRS_OK;
} catch (...) {
    return 0;
}

// ---------------------------------------------------------------------------------------- knn

int pcrs_knn_batch_dev(pcrs_index *index, const float *dqx, const float *dqy, const float *dqz,
                       size_t n, uint32_t k, uint32_t *didx, float *ddist, uint32_t *dcount) try {
    if (!index) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    if (n == 0) return PCRS_OK;
    if (!dqx || !dqy || !dqz) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null query pointer");
    if (k == 0) {  // kdtree.rs:65
        if (dcount) PCRS_HIP_CHECK(ctx, hipMemsetAsync(dcount, 0, n * sizeof(uint32_t), ctx->stream));
        return PCRS_OK;
    }
    if (!didx && !ddist && !dcount) return PCRS_OK;
    KnnQuery q;
    q.qx = dqx;
    q.qy = dqy;
    q.qz = dqz;
    q.nq = n;
    KnnOut o;
    o.idx = didx;
    o.dist = ddist;
    o.count = dcount;
    o.k = k;
    return run_query(index, q, o, k);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_knn_batch(pcrs_index *index, const float *qx, const float *qy, const float *qz, size_t n,
                   uint32_t k, uint32_t *idx, float *dist, uint32_t *count) try {
    if (!index) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    if (n == 0) return PCRS_OK;
    if (!qx || !qy || !qz) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null query pointer");
    DevBuf dqx, dqy, dqz, didx, ddist, dcount;
    PCRS_TRY(upload(ctx, dqx, qx, n));
    PCRS_TRY(upload(ctx, dqy, qy, n));
    PCRS_TRY(upload(ctx, dqz, qz, n));
    const size_t nk = n * (size_t)k;
    if (idx) PCRS_HIP_CHECK(ctx, didx.alloc(ctx, nk * sizeof(uint32_t)));
    if (dist) PCRS_HIP_CHECK(ctx, ddist.alloc(ctx, nk * sizeof(float)));
    if (count) PCRS_HIP_CHECK(ctx, dcount.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_TRY(pcrs_knn_batch_dev(index, dqx.as<float>(), dqy.as<float>(), dqz.as<float>(), n, k,
                                idx ? didx.as<uint32_t>() : nullptr,
                                dist ? ddist.as<float>() : nullptr,
                                count ? dcount.as<uint32_t>() : nullptr));
    if (idx) PCRS_TRY(download(ctx, idx, didx.p, nk * sizeof(uint32_t)));
    if (dist) PCRS_TRY(download(ctx, dist, ddist.p, nk * sizeof(float)));
    if (count) PCRS_TRY(download(ctx, count, dcount.p, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_radius_count_dev(pcrs_index *index, const float *dqx, const float *dqy, const float *dqz,
                          size_t n, float radius, uint32_t *dcount) try {
    if (!index || !dcount) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    // kdtree.rs:106-112: radius <= 0 or non-finite -> no results (the brute-force launcher
    // handles those); otherwise the grid counts when the ball fits a small cube of cells
    const bool usable = radius > 0.0f && radius - radius == 0.0f;
    if (n && usable && pcrs_grid_radius_ok(index, radius)) {
        if (!dqx || !dqy || !dqz)
            return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
        KnnQuery q;
        q.qx = dqx;
        q.qy = dqy;
        q.qz = dqz;
        q.nq = n;
        return pcrs_launch_radius_count_grid(index, q, radius, dcount);
    }
    return pcrs_launch_radius_count(ctx, index->x, index->y, index->z, index->m, dqx, dqy, dqz, n,
                                    radius, dcount);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

// radius_outlier.rs:4-18 on device columns: grid with cells of the radius, one counting walk,
// mask.  dcount (nullable) receives the per-point neighbour counts.
int pcrs_radius_outlier_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                            size_t n, float radius, size_t min_neighbors, uint8_t *dkeep,
                            uint32_t *dcount, size_t *n_keep) try {
    PCRS_TRY(bind_device(ctx));
    if (n_keep) *n_keep = 0;
    if (n == 0) return PCRS_OK;
    if (!dx || !dy || !dz || !dkeep) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    if (n >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    DevBuf own_count;
    if (!dcount) {
        PCRS_HIP_CHECK(ctx, own_count.alloc(ctx, n * sizeof(uint32_t)));
        dcount = own_count.as<uint32_t>();
    }
    const bool usable = radius > 0.0f && radius - radius == 0.0f;
    pcrs_index *ix = new (std::nothrow) pcrs_index();
    if (!ix) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    ix->ctx = ctx;
    ix->x = dx;
    ix->y = dy;
    ix->z = dz;
    ix->m = n;
    ix->algo = PCRS_ALGO_BRUTE;
    int rc = PCRS_OK;
    if (usable && n >= kGridMinPoints) rc = pcrs_grid_build(ix, 0, radius);
    if (rc == PCRS_OK) {
        // a grid of a few cells would make every lane scan most of the cloud on its own: the
        // tiled brute-force scan is the better kernel there
        if (usable && ix->ncell > 100 && pcrs_grid_radius_ok(ix, radius)) {
            KnnQuery q;
            q.nq = n;
            q.self = true;
            rc = pcrs_launch_radius_count_grid(ix, q, radius, dcount);
        } else {
            rc = pcrs_launch_radius_count(ctx, dx, dy, dz, n, dx, dy, dz, n, radius, dcount);
        }
    }
    unsigned long long *dtotal = nullptr, *htotal = nullptr;
    if (rc == PCRS_OK && n_keep) {
        void *p = nullptr, *sc = nullptr;
        rc = pcrs_ctx_pinned(ctx, 64, &p);
        if (rc == PCRS_OK) rc = pcrs_ctx_scratch(ctx, 64, &sc);
        htotal = reinterpret_cast<unsigned long long *>(p);
        dtotal = reinterpret_cast<unsigned long long *>(sc);
    }
    if (rc == PCRS_OK) rc = pcrs_launch_count_mask(ctx, dcount, n, min_neighbors, dkeep, dtotal);
    if (rc == PCRS_OK && n_keep) {
        rc = download(ctx, htotal, dtotal, sizeof(unsigned long long));
        if (rc == PCRS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = pcrs_set_err(ctx, PCRS_ERR_HIP, "stream synchronize failed");
        if (rc == PCRS_OK) *n_keep = (size_t)*htotal;
    }
    pcrs_index_destroy(ix);
    return rc;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_radius_outlier(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                        float radius, size_t min_neighbors, uint8_t *keep, uint32_t *count,
                        size_t *n_keep) try {
    PCRS_TRY(bind_device(ctx));
    if (n_keep) *n_keep = 0;
    if (n == 0) return PCRS_OK;
    if (!x || !y || !z || !keep) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    DevBuf dx, dy, dz, dkeep, dcount;
    PCRS_TRY(upload(ctx, dx, x, n));
    PCRS_TRY(upload(ctx, dy, y, n));
    PCRS_TRY(upload(ctx, dz, z, n));
    PCRS_HIP_CHECK(ctx, dkeep.alloc(ctx, n));
    PCRS_HIP_CHECK(ctx, dcount.alloc(ctx, n * sizeof(uint32_t)));
    size_t kept = 0;
    PCRS_TRY(pcrs_radius_outlier_dev(ctx, dx.as<float>(), dy.as<float>(), dz.as<float>(), n, radius,
                                     min_neighbors, dkeep.as<uint8_t>(), dcount.as<uint32_t>(),
                                     &kept));
    PCRS_TRY(download(ctx, keep, dkeep.p, n));
    if (count) PCRS_TRY(download(ctx, count, dcount.p, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_keep) *n_keep = kept;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_radius_count(pcrs_index *index, const float *qx, const float *qy, const float *qz,
                      size_t n, float radius, uint32_t *count) try {
    if (!index || !count) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    if (n == 0) return PCRS_OK;
    DevBuf dqx, dqy, dqz, dcount;
    PCRS_TRY(upload(ctx, dqx, qx, n));
    PCRS_TRY(upload(ctx, dqy, qy, n));
    PCRS_TRY(upload(ctx, dqz, qz, n));
    PCRS_HIP_CHECK(ctx, dcount.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_TRY(pcrs_radius_count_dev(index, dqx.as<float>(), dqy.as<float>(), dqz.as<float>(), n,
                                   radius, dcount.as<uint32_t>()));
    PCRS_TRY(download(ctx, count, dcount.p, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

// KdTree::radius_search for a batch (kdtree.rs:105-135): count, scan, fill, order.
int pcrs_radius_search_dev(pcrs_index *index, const float *dqx, const float *dqy,
                           const float *dqz, size_t n, float radius, uint32_t *doffsets,
                           uint32_t *dindices, size_t capacity, size_t *total) try {
    if (!index || !doffsets || !total) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    *total = 0;
    if (n >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    // pass 1: the counts, scanned in place into offsets[0..n], the total also in 64 bits
    PCRS_TRY(pcrs_radius_count_dev(index, dqx, dqy, dqz, n, radius, doffsets));
    DevBuf sums, tot64;
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, (n / 1024 + 4) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, tot64.alloc(ctx, sizeof(unsigned long long)));
    PCRS_TRY(pcrs_launch_sum_u32(ctx, doffsets, n, tot64.as<unsigned long long>()));
    if (n)
        PCRS_TRY(pcrs_launch_scan_u32(ctx, doffsets, n, sums.as<uint32_t>(), doffsets + n));
    else
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(doffsets, 0, sizeof(uint32_t), ctx->stream));
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
    PCRS_TRY(pcrs_launch_publish(ctx, tot64.p, pin, sizeof(unsigned long long)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long t = *reinterpret_cast<unsigned long long *>(pin);
    *total = (size_t)t;
    if (t >= 0xFFFFFFFFull) {
        // the 32-bit scan has wrapped: the caller gets zeros rather than offsets that mean nothing
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(doffsets, 0, (n + 1) * sizeof(uint32_t), ctx->stream));
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG,
                            "%llu results do not fit 32-bit offsets: search fewer queries at a time", t);
    }
    if (t == 0) return PCRS_OK;
    if (t > capacity || !dindices)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG,
                            "the index list needs room for %llu entries (capacity %zu)", t, capacity);
    // pass 2 (a usable radius and a non-empty cloud, or there would be no results)
    if (pcrs_grid_radius_ok(index, radius)) {
        // the grid meets a query's hits in cell order: keyed by (query, index) and sorted, every
        // list is ascending (kdtree.rs:131-132) and stays where the offsets put it
        int idx_bits = 1, q_bits = 1;
        while (idx_bits < 32 && (index->m >> idx_bits)) ++idx_bits;
        while (q_bits < 32 && (n >> q_bits)) ++q_bits;
        DevBuf keys, keys_alt;  // the key carries the index: a keys-only sort, 16 B per result
        PCRS_HIP_CHECK(ctx, keys.alloc(ctx, t * sizeof(uint64_t)));
        PCRS_HIP_CHECK(ctx, keys_alt.alloc(ctx, t * sizeof(uint64_t)));
        KnnQuery q;
        q.qx = dqx;
        q.qy = dqy;
        q.qz = dqz;
        q.nq = n;
        PCRS_TRY(pcrs_launch_radius_fill_grid(index, q, radius, doffsets, keys.as<uint64_t>(),
                                              idx_bits));
        bool in_alt = false;
        PCRS_TRY(pcrs_sort_pairs_u64(ctx, keys.as<uint64_t>(), nullptr, keys_alt.as<uint64_t>(),
                                     nullptr, (size_t)t, idx_bits + q_bits, &in_alt));
        PCRS_TRY(pcrs_launch_key_low_bits(ctx, in_alt ? keys_alt.as<uint64_t>() : keys.as<uint64_t>(),
                                          (size_t)t, idx_bits, dindices));
        return PCRS_OK;
    }
    return pcrs_launch_radius_fill(ctx, index->x, index->y, index->z, index->m, dqx, dqy, dqz, n,
                                   radius, doffsets, dindices);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_radius_search(pcrs_index *index, const float *qx, const float *qy, const float *qz,
                       size_t n, float radius, uint32_t *offsets, uint32_t *indices,
                       size_t capacity, size_t *total) try {
    if (!index || !offsets || !total) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    *total = 0;
    if (n == 0) {
        offsets[0] = 0;
        return PCRS_OK;
    }
    if (!qx || !qy || !qz) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    DevBuf dqx, dqy, dqz, doff, dind;
    PCRS_TRY(upload(ctx, dqx, qx, n));
    PCRS_TRY(upload(ctx, dqy, qy, n));
    PCRS_TRY(upload(ctx, dqz, qz, n));
    PCRS_HIP_CHECK(ctx, doff.alloc(ctx, (n + 1) * sizeof(uint32_t)));
    if (indices && capacity) PCRS_HIP_CHECK(ctx, dind.alloc(ctx, capacity * sizeof(uint32_t)));
    const int rc = pcrs_radius_search_dev(index, dqx.as<float>(), dqy.as<float>(), dqz.as<float>(),
                                          n, radius, doff.as<uint32_t>(), dind.as<uint32_t>(),
                                          dind.p ? capacity : 0, total);
    // the offsets are valid even when the list did not fit: the caller learns every list length
    PCRS_TRY(download(ctx, offsets, doff.p, (n + 1) * sizeof(uint32_t)));
    if (rc == PCRS_OK && *total) PCRS_TRY(download(ctx, indices, dind.p, *total * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_voxel_downsample_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                              size_t n, float voxel_size, float *dout_x, float *dout_y,
                              float *dout_z, size_t *n_out) try {
    PCRS_TRY(bind_device(ctx));
    if (!n_out) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    *n_out = 0;
    if (!(voxel_size > 0.0f) || !std::isfinite(voxel_size))  // voxel_downsample.rs:13-16
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "voxel_size must be > 0 and finite");
    if (n == 0) return PCRS_OK;
    if (!dx || !dy || !dz || !dout_x || !dout_y || !dout_z)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    return pcrs_launch_voxel_downsample(ctx, dx, dy, dz, n, voxel_size, dout_x, dout_y, dout_z,
                                        n_out);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_voxel_downsample(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                          float voxel_size, float *out_x, float *out_y, float *out_z,
                          size_t *n_out) try {
    PCRS_TRY(bind_device(ctx));
    if (!n_out) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    *n_out = 0;
    if (!(voxel_size > 0.0f) || !std::isfinite(voxel_size))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "voxel_size must be > 0 and finite");
    if (n == 0) return PCRS_OK;
    if (!x || !y || !z || !out_x || !out_y || !out_z)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    DevBuf d[3], o[3];
    const float *src[3] = {x, y, z};
    float *dst[3] = {out_x, out_y, out_z};
    for (int a = 0; a < 3; ++a) {
        PCRS_TRY(upload(ctx, d[a], src[a], n));
        PCRS_HIP_CHECK(ctx, o[a].alloc(ctx, n * sizeof(float)));
    }
    size_t nv = 0;
    PCRS_TRY(pcrs_launch_voxel_downsample(ctx, d[0].as<float>(), d[1].as<float>(), d[2].as<float>(),
                                          n, voxel_size, o[0].as<float>(), o[1].as<float>(),
                                          o[2].as<float>(), &nv));
    for (int a = 0; a < 3; ++a)
        if (nv) PCRS_TRY(download(ctx, dst[a], o[a].p, nv * sizeof(float)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = nv;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_ransac_plane_samples_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                                  size_t n, const uint32_t *dsamples, size_t n_samples,
                                  float distance_threshold, float model[4], uint8_t *dinlier_mask,
                                  size_t *n_inliers, uint64_t *best_count) try {
    PCRS_TRY(bind_device(ctx));
    if (!model || !n_inliers) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    if (n && (!dx || !dy || !dz || !dinlier_mask))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    if (n_samples && !dsamples) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null samples");
    return pcrs_launch_ransac_plane(ctx, dx, dy, dz, n, dsamples, n_samples, distance_threshold,
                                    model, dinlier_mask, n_inliers, best_count);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_ransac_plane_samples(pcrs_ctx *ctx, const float *x, const float *y, const float *z,
                              size_t n, const uint32_t *samples, size_t n_samples,
                              float distance_threshold, float model[4], uint8_t *inlier_mask,
                              size_t *n_inliers, uint64_t *best_count) try {
    PCRS_TRY(bind_device(ctx));
    if (!model || !n_inliers) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    if (n && (!x || !y || !z || !inlier_mask))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    if (n_samples && !samples) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null samples");
    DevBuf d[3], dmask, dsamp;
    const float *src[3] = {x, y, z};
    for (int a = 0; a < 3; ++a) PCRS_TRY(upload(ctx, d[a], src[a], n));
    PCRS_HIP_CHECK(ctx, dmask.alloc(ctx, n));
    PCRS_HIP_CHECK(ctx, dsamp.alloc(ctx, 3 * n_samples * sizeof(uint32_t)));
    if (n_samples)
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(dsamp.p, samples, 3 * n_samples * sizeof(uint32_t),
                                           hipMemcpyHostToDevice, ctx->stream));
    // the mask travels back inside the call's own wait for its results
    PCRS_TRY(pcrs_launch_ransac_plane(ctx, d[0].as<float>(), d[1].as<float>(), d[2].as<float>(), n,
                                      dsamp.as<uint32_t>(), n_samples, distance_threshold, model,
                                      dmask.as<uint8_t>(), n_inliers, best_count,
                                      n ? inlier_mask : nullptr));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_euclidean_cluster_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                               size_t n, float distance_threshold, size_t min_size,
                               size_t max_size, uint32_t *doffsets, uint32_t *dindices,
                               size_t *n_clusters, size_t *n_indices) try {
    PCRS_TRY(bind_device(ctx));
    if (!n_clusters) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    *n_clusters = 0;
    size_t ni = 0;
    if (n_indices) *n_indices = 0;
    if (n == 0) return PCRS_OK;
    if (!dx || !dy || !dz || !doffsets || !dindices)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    PCRS_TRY(pcrs_launch_euclidean_cluster(ctx, dx, dy, dz, n, distance_threshold, min_size,
                                           max_size, doffsets, dindices, n_clusters, &ni));
    if (n_indices) *n_indices = ni;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_euclidean_cluster(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                           float distance_threshold, size_t min_size, size_t max_size,
                           uint32_t *offsets, uint32_t *indices, size_t *n_clusters) try {
    PCRS_TRY(bind_device(ctx));
    if (!n_clusters || !offsets) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    *n_clusters = 0;
    offsets[0] = 0;
    if (n == 0) return PCRS_OK;
    if (!x || !y || !z || !indices)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    DevBuf d[3], doff, didx;
    const float *src[3] = {x, y, z};
    for (int a = 0; a < 3; ++a) PCRS_TRY(upload(ctx, d[a], src[a], n));
    PCRS_HIP_CHECK(ctx, doff.alloc(ctx, (n + 1) * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, didx.alloc(ctx, n * sizeof(uint32_t)));
    size_t nc = 0, ni = 0;
    PCRS_TRY(pcrs_launch_euclidean_cluster(ctx, d[0].as<float>(), d[1].as<float>(), d[2].as<float>(),
                                           n, distance_threshold, min_size, max_size,
                                           doff.as<uint32_t>(), didx.as<uint32_t>(), &nc, &ni));
    if (nc) {
        PCRS_TRY(download(ctx, offsets, doff.p, (nc + 1) * sizeof(uint32_t)));
        if (ni) PCRS_TRY(download(ctx, indices, didx.p, ni * sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_clusters = nc;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

// -------------------------------------------------------------------------- SOR and normals

static int check_range(pcrs_index *index, size_t q_begin, size_t q_count) {
    if (q_begin > index->m || q_count > index->m - q_begin)
        return pcrs_set_err(index->ctx, PCRS_ERR_INVALID_ARG,
                            "query range [%zu, +%zu) exceeds the indexed cloud (%zu points)",
                            q_begin, q_count, index->m);
    return PCRS_OK;
}

int pcrs_sor_normals_dev(pcrs_index *index, size_t q_begin, size_t q_count, uint32_t k_sor,
                         uint32_t k_normals, const float viewpoint[3], float *dmean, float *dnx,
                         float *dny, float *dnz) try {
    if (!index) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    PCRS_TRY(check_range(index, q_begin, q_count));
    const bool want_n = dnx || dny || dnz;
    if (want_n && !(dnx && dny && dnz))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "normals need all three output columns");
    if (dmean && k_sor == 0) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_sor must be >= 1");
    if (want_n && (k_normals == 0 || !viewpoint))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_normals must be >= 1 with a viewpoint");
    if (q_count == 0 || (!dmean && !want_n)) return PCRS_OK;
    KnnQuery q;
    q.qx = index->x + q_begin;
    q.qy = index->y + q_begin;
    q.qz = index->z + q_begin;
    q.nq = q_count;
    q.out_offset = q_begin;
    q.self = true;
    KnnOut o;
    uint32_t kneed = 1;
    if (dmean) {
        o.mean = dmean;
        o.k_sor = k_sor;
        if (k_sor == 0xFFFFFFFFu) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_sor too large");
        kneed = k_sor + 1;
    }
    if (want_n) {
        o.nx = dnx;
        o.ny = dny;
        o.nz = dnz;
        o.k_nrm = k_normals;
        o.vp[0] = viewpoint[0];
        o.vp[1] = viewpoint[1];
        o.vp[2] = viewpoint[2];
        if (k_normals > kneed) kneed = k_normals;
    }
    return run_query(index, q, o, kneed);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_index_slot_bounds(pcrs_index *index, uint32_t parts, size_t *bounds) try {
    if (!index) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    if (!bounds || parts == 0) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "parts must be >= 1");
    if (index->algo != PCRS_ALGO_GRID || index->nvalid != index->m) {
        // no cell order to split by (brute-force index, or non-finite points that have no cell)
        for (uint32_t r = 0; r <= parts; ++r) bounds[r] = 0;
        return PCRS_OK;
    }
    return pcrs_grid_slot_bounds(index, parts, bounds);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_sor_normals_slots_dev(pcrs_index *index, size_t slot_begin, size_t slot_count,
                               uint32_t k_sor, uint32_t k_normals, const float viewpoint[3],
                               uint32_t *dpoint, float *dmean, float *dnx, float *dny, float *dnz) try {
    if (!index) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    if (index->algo != PCRS_ALGO_GRID)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "the index has no cell order");
    if (slot_begin > index->nvalid || slot_count > index->nvalid - slot_begin)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG,
                            "slot range [%zu, +%zu) exceeds the %zu slots of the index", slot_begin,
                            slot_count, index->nvalid);
    const bool want_n = dnx || dny || dnz;
    if (want_n && !(dnx && dny && dnz))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "normals need all three output columns");
    if (dmean && k_sor == 0) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_sor must be >= 1");
    if (want_n && (k_normals == 0 || !viewpoint))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_normals must be >= 1 with a viewpoint");
    if (slot_count == 0) return PCRS_OK;
    if (!dpoint) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null point-index output");
    // the points of the run as explicit queries, already in cell order
    DevBuf qx, qy, qz;
    PCRS_HIP_CHECK(ctx, qx.alloc(ctx, slot_count * sizeof(float)));
    PCRS_HIP_CHECK(ctx, qy.alloc(ctx, slot_count * sizeof(float)));
    PCRS_HIP_CHECK(ctx, qz.alloc(ctx, slot_count * sizeof(float)));
    PCRS_TRY(pcrs_grid_slot_slice(index, slot_begin, slot_count, qx.as<float>(), qy.as<float>(),
                                  qz.as<float>(), dpoint));
    if (!dmean && !want_n) return PCRS_OK;
    KnnQuery q;
    q.qx = qx.as<float>();
    q.qy = qy.as<float>();
    q.qz = qz.as<float>();
    q.nq = slot_count;
    q.coherent = true;
    KnnOut o;
    uint32_t kneed = 1;
    if (dmean) {
        if (k_sor == 0xFFFFFFFFu) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_sor too large");
        o.mean = dmean;
        o.k_sor = k_sor;
        kneed = k_sor + 1;
    }
    if (want_n) {
        o.nx = dnx;
        o.ny = dny;
        o.nz = dnz;
        o.k_nrm = k_normals;
        o.vp[0] = viewpoint[0];
        o.vp[1] = viewpoint[1];
        o.vp[2] = viewpoint[2];
        if (k_normals > kneed) kneed = k_normals;
    }
    return run_query(index, q, o, kneed);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

// ---- one cloud over several GPUs by space: slabs, an index per slab (+ halo), the slab's queries

int pcrs_slab_cuts_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz, size_t m,
                       uint32_t parts, uint32_t k_hint, int *axis, float *cuts, float *halo,
                       size_t *n_finite) try {
    PCRS_TRY(bind_device(ctx));
    if (!axis || !cuts || !halo || !n_finite || parts == 0)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null output or no parts");
    if (m && (!dx || !dy || !dz)) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    return pcrs_slab_cuts(ctx, dx, dy, dz, m, parts, k_hint, axis, cuts, halo, n_finite);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_index_build_box_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                             size_t m, uint32_t k_hint, const float box_lo[3],
                             const float box_hi[3], pcrs_index **out) try {
    if (!out) return PCRS_ERR_INVALID_ARG;
    *out = nullptr;
    PCRS_TRY(bind_device(ctx));
    if (!box_lo || !box_hi) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null box");
    if (m && (!dx || !dy || !dz)) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    for (int a = 0; a < 3; ++a)
        if (!(box_lo[a] <= box_hi[a]))  // NaN included
            return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "box with lo > hi on axis %d", a);
    pcrs_index *ix = new (std::nothrow) pcrs_index();
    if (!ix) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    ix->ctx = ctx;
    ix->x = dx;
    ix->y = dy;
    ix->z = dz;
    ix->m = m;
    for (int a = 0; a < 3; ++a) {
        ix->box.lo[a] = box_lo[a];
        ix->box.hi[a] = box_hi[a];
    }
    ix->k_hint = k_hint;
    int rc = finish_index(ix, k_hint, PCRS_ALGO_GRID);
    if (rc != PCRS_OK) {
        delete ix;
        return rc;
    }
    *out = ix;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

// cut + index over one slab and its halo + selection of the slab's points: the three calls above
// in one, with the cloud measured once (the build takes its bounds from the cut)
int pcrs_slab_index_build_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                              size_t m, uint32_t k_hint, uint32_t parts, uint32_t part,
                              pcrs_index **out, size_t *n_queries, int *axis_out,
                              float slab[2]) try {
    if (!out || !n_queries) return PCRS_ERR_INVALID_ARG;
    *out = nullptr;
    *n_queries = 0;
    PCRS_TRY(bind_device(ctx));
    if (parts == 0 || part >= parts) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "part %u of %u", part, parts);
    if (m && (!dx || !dy || !dz)) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    std::vector<float> cuts((size_t)parts + 1);
    SlabMeasure ms;
    int axis = 0;
    float halo = 0.0f;
    size_t n_finite = 0;
    PCRS_TRY(pcrs_slab_cuts(ctx, dx, dy, dz, m, parts, k_hint, &axis, cuts.data(), &halo, &n_finite, &ms));
    if (axis_out) *axis_out = axis;
    if (slab) {
        slab[0] = cuts[part];
        slab[1] = cuts[part + 1];
    }
    // points without a place in space belong to no slab, and a cloud without extent has no cuts
    if (n_finite != m || ms.hist.empty()) return PCRS_OK;
    pcrs_index *ix = new (std::nothrow) pcrs_index();
    if (!ix) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    ix->ctx = ctx;
    ix->x = dx;
    ix->y = dy;
    ix->z = dz;
    ix->m = m;
    ix->k_hint = k_hint;
    BoxFilter qb = pcrs_no_box();
    qb.lo[axis] = cuts[part];
    qb.hi[axis] = cuts[part + 1];
    if (part > 0) ix->box.lo[axis] = cuts[part] - halo;
    if (part + 1 < parts) ix->box.hi[axis] = cuts[part + 1] + halo;
    // the bounds of what the index takes, and no fewer points than it can take (whole bins of the
    // cut's histogram, one more on either side for the rounding of the bin edges)
    ix->have_bounds = true;
    for (int a = 0; a < 3; ++a) {
        ix->bounds_lo[a] = ms.lo[a];
        ix->bounds_hi[a] = ms.hi[a];
    }
    if ((double)ix->box.lo[axis] > ix->bounds_lo[axis]) ix->bounds_lo[axis] = ix->box.lo[axis];
    if ((double)ix->box.hi[axis] < ix->bounds_hi[axis]) ix->bounds_hi[axis] = ix->box.hi[axis];
    {
        const double w = ms.hist_width;
        const long nbins = (long)ms.hist.size();
        long b0 = (long)std::floor((ix->bounds_lo[axis] - ms.hist_lo) / w) - 1;
        long b1 = (long)std::floor((ix->bounds_hi[axis] - ms.hist_lo) / w) + 1;
        if (b0 < 0) b0 = 0;
        if (b1 > nbins - 1) b1 = nbins - 1;
        size_t upper = 0;
        for (long b = b0; b <= b1; ++b) upper += ms.hist[(size_t)b];
        ix->bounds_n_upper = upper;
    }
    int rc = ix->bounds_n_upper >= 64 ? finish_index(ix, k_hint, PCRS_ALGO_GRID) : PCRS_OK;
    if (rc == PCRS_OK && ix->algo == PCRS_ALGO_GRID) rc = pcrs_grid_select_box(ix, qb, n_queries);
    if (rc != PCRS_OK || ix->algo != PCRS_ALGO_GRID) {  // no grid over this slab: no index either
        delete ix;
        return rc;
    }
    *out = ix;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_index_select_box(pcrs_index *index, const float q_lo[3], const float q_hi[3],
                          size_t *count) try {
    if (!index || !count) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    *count = 0;
    if (!q_lo || !q_hi) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null box");
    if (index->algo != PCRS_ALGO_GRID) {
        // a box with too few points for a grid: nothing is selected by cell order here (the
        // caller falls back to ranges of point indices)
        index->sel_count = 0;
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "the index has no cell order");
    }
    BoxFilter qb;
    for (int a = 0; a < 3; ++a) {
        qb.lo[a] = q_lo[a];
        qb.hi[a] = q_hi[a];
    }
    return pcrs_grid_select_box(index, qb, count);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_sor_normals_selected_dev(pcrs_index *index, uint32_t k_sor, uint32_t k_normals,
                                  const float viewpoint[3], uint32_t *dpoint, float *dmean,
                                  float *dnx, float *dny, float *dnz, size_t *n_escalated) try {
    if (!index) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = index->ctx;
    PCRS_TRY(bind_device(ctx));
    if (n_escalated) *n_escalated = 0;
    if (index->algo != PCRS_ALGO_GRID || !index->sel_list.p)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "no queries selected (pcrs_index_select_box)");
    const bool want_n = dnx || dny || dnz;
    if (want_n && !(dnx && dny && dnz))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "normals need all three output columns");
    if (dmean && k_sor == 0) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_sor must be >= 1");
    if (want_n && (k_normals == 0 || !viewpoint))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_normals must be >= 1 with a viewpoint");
    const size_t n = index->sel_count;
    if (n == 0) return PCRS_OK;
    if (!dpoint) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null point-index output");
    DevBuf qx, qy, qz, cov;
    PCRS_HIP_CHECK(ctx, qx.alloc(ctx, n * sizeof(float)));
    PCRS_HIP_CHECK(ctx, qy.alloc(ctx, n * sizeof(float)));
    PCRS_HIP_CHECK(ctx, qz.alloc(ctx, n * sizeof(float)));
    PCRS_TRY(pcrs_grid_gather_selected(index, qx.as<float>(), qy.as<float>(), qz.as<float>(), dpoint));
    if (!dmean && !want_n) return PCRS_OK;
    KnnQuery q;
    q.qx = qx.as<float>();
    q.qy = qy.as<float>();
    q.qz = qz.as<float>();
    q.nq = n;
    q.coherent = true;
    q.no_brute = true;  // what the box cannot certify is asked of an index over the whole cloud
    KnnOut o;
    uint32_t kneed = 1;
    if (dmean) {
        if (k_sor == 0xFFFFFFFFu) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "k_sor too large");
        o.mean = dmean;
        o.k_sor = k_sor;
        kneed = k_sor + 1;
    }
    if (want_n) {
        o.nx = dnx;
        o.ny = dny;
        o.nz = dnz;
        o.k_nrm = k_normals;
        o.vp[0] = viewpoint[0];
        o.vp[1] = viewpoint[1];
        o.vp[2] = viewpoint[2];
        if (k_normals > kneed) kneed = k_normals;
        // the covariance records are the caller's here: both indices write into them, and they
        // are solved once, at the end
        PCRS_HIP_CHECK(ctx, cov.alloc(ctx, PCRS_REC_FLOATS_HOST * n * sizeof(float)));
        o.cov = cov.as<float>();
    }
    PCRS_TRY(run_query(index, q, o, kneed));
    if (index->left_count) {
        void *pin = nullptr;
        PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
        PCRS_TRY(pcrs_launch_publish(ctx, index->left_count, pin, sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t left = *reinterpret_cast<uint32_t *>(pin);
        if (n_escalated) *n_escalated = left;
        if (left) {
            if (!index->whole) {
                pcrs_index *w = new (std::nothrow) pcrs_index();
                if (!w) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
                w->ctx = ctx;
                w->x = index->x;
                w->y = index->y;
                w->z = index->z;
                w->m = index->m;
                const int rc = finish_index(w, index->k_hint, PCRS_ALGO_AUTO);
                if (rc != PCRS_OK) {
                    delete w;
                    return rc;
                }
                index->whole = w;
            }
            KnnQuery q2 = q;
            q2.no_brute = false;
            q2.only_list = index->left_list;
            q2.only_count = index->left_count;
            PCRS_TRY(run_query(index->whole, q2, o, kneed));
        }
    }
    if (want_n)
        PCRS_TRY(pcrs_launch_normals_from_cov(ctx, n, cov.as<float>(), q.qx, q.qy, q.qz, o));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_scatter_f32_dev(pcrs_ctx *ctx, const float *dvalues, const uint32_t *dpoint, size_t n,
                         float *dout) try {
    PCRS_TRY(bind_device(ctx));
    if (n && (!dvalues || !dpoint || !dout))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    return pcrs_launch_scatter_perm(ctx, dvalues, dpoint, n, dout);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_select_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz, size_t n,
                    const uint8_t *dmask, int invert, float *dox, float *doy, float *doz,
                    uint32_t *dindex, size_t *n_out) try {
    PCRS_TRY(bind_device(ctx));
    if (n_out) *n_out = 0;
    if (n == 0) return PCRS_OK;
    if (!dx || !dy || !dz || !dmask || !dox || !doy || !doz)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    DevBuf total;
    PCRS_HIP_CHECK(ctx, total.alloc(ctx, sizeof(uint32_t)));
    PCRS_TRY(pcrs_launch_select(ctx, dx, dy, dz, n, dmask, invert, dox, doy, doz, dindex,
                                total.as<uint32_t>()));
    if (n_out) {
        void *pin = nullptr;
        PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
        PCRS_TRY(pcrs_launch_publish(ctx, total.p, pin, sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        *n_out = *reinterpret_cast<uint32_t *>(pin);
    }
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_route_pairs_dev(pcrs_ctx *ctx, const uint32_t *dpoint, const float *dvalues, size_t n,
                         size_t n_total, uint32_t world, uint32_t *dsend_point, float *dsend_values,
                         size_t *counts) try {
    PCRS_TRY(bind_device(ctx));
    if (!counts || (n && (!dpoint || !dvalues || !dsend_point || !dsend_values)))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    return pcrs_launch_route_pairs(ctx, dpoint, dvalues, n, n_total, world, dsend_point,
                                   dsend_values, counts);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_scatter_range_f32_dev(pcrs_ctx *ctx, const float *dvalues, const uint32_t *dpoint,
                               size_t n, size_t first, float *dout) try {
    PCRS_TRY(bind_device(ctx));
    if (n && (!dvalues || !dpoint || !dout)) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    if (first >= 0xFFFFFFFFull) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "range start too large");
    return pcrs_launch_scatter_offset(ctx, dvalues, dpoint, n, (uint32_t)first, dout);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_sor_mean_dist_dev(pcrs_index *index, size_t q_begin, size_t q_count, uint32_t k,
                           float *dmean) try {
    if (!dmean) return PCRS_ERR_INVALID_ARG;
    return pcrs_sor_normals_dev(index, q_begin, q_count, k, 0, nullptr, dmean, nullptr, nullptr,
                                nullptr);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_normals_dev(pcrs_index *index, size_t q_begin, size_t q_count, uint32_t k,
                     const float viewpoint[3], float *dnx, float *dny, float *dnz) try {
    if (!dnx || !dny || !dnz) return PCRS_ERR_INVALID_ARG;
    return pcrs_sor_normals_dev(index, q_begin, q_count, 0, k, viewpoint, nullptr, dnx, dny, dnz);
} catch (...) {
    return pcrs_on_exception(index ? index->ctx : nullptr);
}

int pcrs_sor_threshold_dev(pcrs_ctx *ctx, const float *dmean, size_t n, float std_mul, float *thr,
                           int *has_finite) try {
    PCRS_TRY(bind_device(ctx));
    if (!thr || !has_finite) return PCRS_ERR_INVALID_ARG;
    return pcrs_sor_threshold_impl(ctx, dmean, n, std_mul, thr, has_finite);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_sor_mask_dev(pcrs_ctx *ctx, const float *dmean, size_t n, float thr, uint8_t *dkeep,
                      size_t *n_keep) try {
    PCRS_TRY(bind_device(ctx));
    if (n_keep) *n_keep = 0;
    if (n == 0) return PCRS_OK;
    if (!dmean || !dkeep) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    unsigned long long *dcount = nullptr;
    unsigned long long *hcount = nullptr;
    if (n_keep) {
        void *p = nullptr;
        PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &p));
        hcount = reinterpret_cast<unsigned long long *>(p);
        void *s = nullptr;
        PCRS_TRY(pcrs_ctx_scratch(ctx, 64, &s));
        dcount = reinterpret_cast<unsigned long long *>(s);
    }
    PCRS_TRY(pcrs_launch_mask(ctx, dmean, n, thr, dkeep, dcount));
    if (n_keep) {
        PCRS_TRY(download(ctx, hcount, dcount, sizeof(unsigned long long)));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        *n_keep = (size_t)*hcount;
    }
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

// statistical_outlier.rs:43-66 without leaving the device between the statistics and the mask:
// the threshold comes from ALL means (dmean_all), the mask covers the caller's own range (dmean).
// Queues statistics, mask and the publication of {scalars, kept count} into pinned memory;
// select_finish decodes them once the stream has been synchronised.
struct SelectPending {
    void *pin = nullptr;
    size_t sb = 0;
};

static int select_launch(pcrs_ctx *ctx, const float *dmean_all, size_t n_all, float std_mul,
                         const float *dmean, size_t n, uint8_t *dkeep, SelectPending *pend) {
    const size_t sb = (pcrs_sor_stats_scalars_bytes() + 63) & ~(size_t)63;
    const size_t wb = pcrs_sor_stats_workspace(n_all);
    void *scratch = nullptr;
    PCRS_TRY(pcrs_ctx_scratch(ctx, sb + 64 + wb, &scratch));
    char *sc = reinterpret_cast<char *>(scratch);
    unsigned long long *dcount = reinterpret_cast<unsigned long long *>(sc + sb);
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, sb + 64, &pin));
    // the counter of the mask lies right behind the scalars and is cleared with them
    PCRS_TRY(pcrs_launch_sor_stats(ctx, dmean_all, n_all, std_mul, sc, sc + sb + 64, wb,
                                   sb + sizeof(unsigned long long)));
    PCRS_TRY(pcrs_launch_mask_dev_thr(ctx, dmean, n, sc, dkeep, dcount, true));
    PCRS_TRY(pcrs_launch_publish(ctx, sc, pin, sb + 64));
    pend->pin = pin;
    pend->sb = sb;
    return PCRS_OK;
}

static void select_finish(const SelectPending &pend, float *thr, int *has_finite, size_t *n_keep) {
    float t;
    int ok;
    pcrs_sor_stats_decode(pend.pin, nullptr, &t, &ok);
    if (ok) {
        if (thr) *thr = t;
        if (has_finite) *has_finite = 1;
        if (n_keep)
            *n_keep = (size_t)*reinterpret_cast<unsigned long long *>(
                reinterpret_cast<char *>(pend.pin) + pend.sb);
    }
}

int pcrs_sor_select_dev(pcrs_ctx *ctx, const float *dmean_all, size_t n_all, float std_mul,
                        const float *dmean, size_t n, uint8_t *dkeep, float *thr, int *has_finite,
                        size_t *n_keep) try {
    PCRS_TRY(bind_device(ctx));
    if (thr) *thr = std::numeric_limits<float>::quiet_NaN();
    if (has_finite) *has_finite = 0;
    if (n_keep) *n_keep = 0;
    if (n_all == 0) return PCRS_OK;
    if (!dmean_all || (n && (!dmean || !dkeep)))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    SelectPending pend;
    PCRS_TRY(select_launch(ctx, dmean_all, n_all, std_mul, dmean, n, dkeep, &pend));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    select_finish(pend, thr, has_finite, n_keep);
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

// The same three launches with the library's static device array as work space and the results
// left on the device: no allocation, no release, no synchronisation (see pcrs_hip.h).
int pcrs_sor_select_static_dev(pcrs_ctx *ctx, const float *dmean_all, size_t n_all, float std_mul,
                               const float *dmean, size_t n, uint8_t *dkeep,
                               pcrs_sor_select_result *dresult) try {
    PCRS_TRY(bind_device(ctx));
    if (!dresult || (n_all && !dmean_all) || (n && (!dmean || !dkeep)))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    const size_t sb = (pcrs_sor_stats_scalars_bytes() + 63) & ~(size_t)63;
    const size_t wb = pcrs_sor_stats_workspace(n_all);
    if (n_all > PCRS_SOR_SELECT_STATIC_MAX || sb + 64 + wb > ctx->static_scratch_bytes)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG,
                            "pcrs_sor_select_static_dev handles at most %zu means (got %zu): use "
                            "pcrs_sor_select_dev",
                            (size_t)PCRS_SOR_SELECT_STATIC_MAX, n_all);
    char *sc = reinterpret_cast<char *>(ctx->static_scratch);  // the resolved device address
    unsigned long long *dcount = reinterpret_cast<unsigned long long *>(sc + sb);
    // the array keeps the last call's contents: the scalars and the count behind them are
    // cleared by the first kernel of the statistics (or by a memset when there are no means),
    // every other part of the work space is written before it is read
    PCRS_TRY(pcrs_launch_sor_stats(ctx, dmean_all, n_all, std_mul, sc, sc + sb + 64, wb,
                                   sb + sizeof(unsigned long long)));
    if (n_all)  // with no means at all the mask is left alone, as in pcrs_sor_select_dev
        PCRS_TRY(pcrs_launch_mask_dev_thr(ctx, dmean, n, sc, dkeep, dcount, true));
    return pcrs_launch_select_result(ctx, sc, dcount, dresult);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_sor_normals(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                     uint32_t k_sor, float std_mul, uint32_t k_normals, const float viewpoint[3],
                     uint8_t *keep, size_t *n_keep, float *thr_out, float *nx, float *ny,
                     float *nz) try {
    PCRS_TRY(bind_device(ctx));
    const bool want_sor = keep != nullptr;
    const bool want_n = nx || ny || nz;
    if (n_keep) *n_keep = 0;
    if (thr_out) *thr_out = std::numeric_limits<float>::quiet_NaN();
    if (want_n && !(nx && ny && nz && viewpoint))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "normals need three columns and a viewpoint");
    if (n == 0) return PCRS_OK;
    if (!x || !y || !z) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    // statistical_outlier.rs:5-12, estimate.rs:25-31
    bool run_sor = want_sor, run_n = want_n && k_normals > 0;
    if (want_sor) {
        if (k_sor == 0) {
            std::memset(keep, 0, n);
            run_sor = false;
        } else if (n == 1) {
            keep[0] = 1;
            if (n_keep) *n_keep = 1;
            run_sor = false;
        }
    }
    if (!run_sor && !run_n) return PCRS_OK;

    pcrs_index *ix = nullptr;
    const uint32_t hint = run_sor ? k_sor + 1 : k_normals;
    PCRS_TRY(pcrs_index_build(ctx, x, y, z, n, hint, PCRS_ALGO_AUTO, &ix));
    DevBuf dmean, dnx, dny, dnz;
    int rc = PCRS_OK;
    auto fail = [&](int code) {
        // nothing may still read the buffers that are released on the way out
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        pcrs_index_destroy(ix);
        return code;
    };
    if (run_sor && dmean.alloc(ctx, n * sizeof(float)) != hipSuccess)
        return fail(pcrs_set_err(ctx, PCRS_ERR_OOM, "device allocation failed"));
    if (run_n && (dnx.alloc(ctx, n * sizeof(float)) != hipSuccess ||
                  dny.alloc(ctx, n * sizeof(float)) != hipSuccess ||
                  dnz.alloc(ctx, n * sizeof(float)) != hipSuccess))
        return fail(pcrs_set_err(ctx, PCRS_ERR_OOM, "device allocation failed"));
    rc = pcrs_sor_normals_dev(ix, 0, n, run_sor ? k_sor : 0, run_n ? k_normals : 0, viewpoint,
                              run_sor ? dmean.as<float>() : nullptr,
                              run_n ? dnx.as<float>() : nullptr, run_n ? dny.as<float>() : nullptr,
                              run_n ? dnz.as<float>() : nullptr);
    if (rc != PCRS_OK) return fail(rc);
    // statistics and mask stay on the device (statistical_outlier.rs:43-66): only the mask, the
    // normals and three scalars cross PCIe.  The normals are final once the search is through, so
    // they travel on a second stream while the statistics run.
    DevBuf dkeep;
    float thr = std::numeric_limits<float>::quiet_NaN();
    int has_finite = 0;
    size_t kept = 0;
    SelectPending pend;
    const bool overlap = run_n && run_sor;
    if (overlap) {
        if (!ctx->copy_stream &&
            (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
             hipEventCreateWithFlags(&ctx->copy_event, hipEventDisableTiming) != hipSuccess))
            return fail(pcrs_set_err(ctx, PCRS_ERR_HIP, "copy stream creation failed"));
        if (hipEventRecord(ctx->copy_event, ctx->stream) != hipSuccess ||
            hipStreamWaitEvent(ctx->copy_stream, ctx->copy_event, 0) != hipSuccess)
            return fail(pcrs_set_err(ctx, PCRS_ERR_HIP, "event ordering failed"));
    }
    if (run_sor) {
        if (dkeep.alloc(ctx, n) != hipSuccess)
            return fail(pcrs_set_err(ctx, PCRS_ERR_OOM, "device allocation failed"));
        rc = select_launch(ctx, dmean.as<float>(), n, std_mul, dmean.as<float>(), n,
                           dkeep.as<uint8_t>(), &pend);
        if (rc != PCRS_OK) return fail(rc);
    }
    if (run_n) {
        hipStream_t cs = overlap ? ctx->copy_stream : ctx->stream;
        const size_t nb = n * sizeof(float);
        if (hipMemcpyAsync(nx, dnx.p, nb, hipMemcpyDeviceToHost, cs) != hipSuccess ||
            hipMemcpyAsync(ny, dny.p, nb, hipMemcpyDeviceToHost, cs) != hipSuccess ||
            hipMemcpyAsync(nz, dnz.p, nb, hipMemcpyDeviceToHost, cs) != hipSuccess)
            return fail(pcrs_set_err(ctx, PCRS_ERR_HIP, "download of the normals failed"));
    }
    if (run_sor) {
        rc = download(ctx, keep, dkeep.p, n);
        if (rc != PCRS_OK) return fail(rc);
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        (overlap && hipStreamSynchronize(ctx->copy_stream) != hipSuccess))
        return fail(pcrs_set_err(ctx, PCRS_ERR_HIP, "stream synchronize failed"));
    if (run_sor) select_finish(pend, &thr, &has_finite, &kept);
    pcrs_index_destroy(ix);
    if (run_sor) {
        if (has_finite && thr_out) *thr_out = thr;
        if (n_keep) *n_keep = kept;  // no finite mean: every keep[i] is 0 (:49-51)
    }
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_sor(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n, uint32_t k,
             float std_mul, uint8_t *keep, size_t *n_keep, float *thr_out, float *mean_out) try {
    if (!keep && n) return PCRS_ERR_INVALID_ARG;
    if (n == 0) {
        if (n_keep) *n_keep = 0;
        if (thr_out) *thr_out = std::numeric_limits<float>::quiet_NaN();
        return bind_device(ctx);
    }
    if (!mean_out)
        return pcrs_sor_normals(ctx, x, y, z, n, k, std_mul, 0, nullptr, keep, n_keep, thr_out,
                                nullptr, nullptr, nullptr);
    // same flow, keeping the per-point means for the caller
    PCRS_TRY(bind_device(ctx));
    if (n_keep) *n_keep = 0;
    if (thr_out) *thr_out = std::numeric_limits<float>::quiet_NaN();
    if (n == 0) return PCRS_OK;
    if (k == 0) {
        std::memset(keep, 0, n);
        return PCRS_OK;
    }
    if (n == 1) {
        keep[0] = 1;
        if (n_keep) *n_keep = 1;
        return PCRS_OK;
    }
    pcrs_index *ix = nullptr;
    PCRS_TRY(pcrs_index_build(ctx, x, y, z, n, k + 1, PCRS_ALGO_AUTO, &ix));
    DevBuf dmean, dkeep;
    int rc = (dmean.alloc(ctx, n * sizeof(float)) == hipSuccess &&
              dkeep.alloc(ctx, n) == hipSuccess)
                 ? PCRS_OK
                 : pcrs_set_err(ctx, PCRS_ERR_OOM, "device allocation failed");
    if (rc == PCRS_OK) rc = pcrs_sor_mean_dist_dev(ix, 0, n, k, dmean.as<float>());
    float thr = 0.0f;
    int ok = 0;
    size_t kept = 0;
    if (rc == PCRS_OK) rc = download(ctx, mean_out, dmean.p, n * sizeof(float));
    // threshold and mask on the device (the sequential sums included); the means travel back
    // while the statistics run
    if (rc == PCRS_OK)
        rc = pcrs_sor_select_dev(ctx, dmean.as<float>(), n, std_mul, dmean.as<float>(), n,
                                 dkeep.as<uint8_t>(), &thr, &ok, &kept);
    if (rc == PCRS_OK) rc = download(ctx, keep, dkeep.p, n);
    if (rc == PCRS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = pcrs_set_err(ctx, PCRS_ERR_HIP, "stream synchronize failed");
    pcrs_index_destroy(ix);
    if (rc != PCRS_OK) return rc;
    if (ok) {
        if (thr_out) *thr_out = thr;
        if (n_keep) *n_keep = kept;
    }
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_normals(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                 uint32_t k, const float viewpoint[3], float *nx, float *ny, float *nz) try {
    if (!nx || !ny || !nz || !viewpoint) return PCRS_ERR_INVALID_ARG;
    return pcrs_sor_normals(ctx, x, y, z, n, 0, 0.0f, k, viewpoint, nullptr, nullptr, nullptr, nx,
                            ny, nz);
} catch (...) {
    return pcrs_on_exception(ctx);
}

// ------------------------------------------------------------------------------ registration

void pcrs_compose(const float ra[9], const float ta[3], const float rb[9], const float tb[3],
                  float rout[9], float tout[3]) try {
    pcrs_host_compose(ra, ta, rb, tb, rout, tout);
} catch (...) {
}

int pcrs_apply_transform_dev(pcrs_ctx *ctx, const float *dx, const float *dy, const float *dz,
                             size_t n, const float rotation[9], const float translation[3],
                             float *dox, float *doy, float *doz) try {
    PCRS_TRY(bind_device(ctx));
    if (!rotation || !translation) return PCRS_ERR_INVALID_ARG;
    if (n == 0) return PCRS_OK;
    if (!dx || !dy || !dz || !dox || !doy || !doz)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    return pcrs_launch_transform(ctx, dx, dy, dz, n, rotation, translation, dox, doy, doz);
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_apply_transform(pcrs_ctx *ctx, const float *x, const float *y, const float *z, size_t n,
                         const float rotation[9], const float translation[3], float *ox,
                         float *oy, float *oz) try {
    PCRS_TRY(bind_device(ctx));
    if (n == 0) return PCRS_OK;
    if (!x || !y || !z || !ox || !oy || !oz || !rotation || !translation)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null pointer");
    DevBuf dx, dy, dz;
    PCRS_TRY(upload(ctx, dx, x, n));
    PCRS_TRY(upload(ctx, dy, y, n));
    PCRS_TRY(upload(ctx, dz, z, n));
    PCRS_TRY(pcrs_launch_transform(ctx, dx.as<float>(), dy.as<float>(), dz.as<float>(), n,
                                   rotation, translation, dx.as<float>(), dy.as<float>(),
                                   dz.as<float>()));
    PCRS_TRY(download(ctx, ox, dx.p, n * sizeof(float)));
    PCRS_TRY(download(ctx, oy, dy.p, n * sizeof(float)));
    PCRS_TRY(download(ctx, oz, dz.p, n * sizeof(float)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

}  // extern "C"

struct pcrs_icp {
    pcrs_ctx *ctx = nullptr;
    pcrs_index *target = nullptr;  // owned
    const float *tnx = nullptr, *tny = nullptr, *tnz = nullptr;
    size_t n_normals = 0;
    const float *sx = nullptr, *sy = nullptr, *sz = nullptr;
    size_t n = 0, m = 0;
    DevBuf own_s[3], own_t[3], own_n[3];
    DevBuf cur[3];
    DevBuf nn_idx, nn_d2, sums;
    // source in the cell order of a grid over itself (neighbouring lanes then search neighbouring
    // target cells); perm[i] = original index of sorted point i, empty = original order kept
    DevBuf ord_s[3], perm;
    DevBuf state;  // IcpLoopState of pcrs_icp_run
    // dist*dist of the correspondences of the last two iterations at their original source index
    // (two halves of n floats, iteration i in half i & 1), and the work space of the
    // left-to-right f32 sum over one of them (icp.rs:277-281)
    DevBuf d2seq, seq_work;
    size_t seq_scalars_bytes = 0;
    // source split over ranks (pcrs_icp_shard_*): points of all shards together, and the work
    // space of the f32 sum over their gathered dist*dist
    size_t shard_total = 0;
    DevBuf shard_seq;
};

pcrs_ctx *pcrs_icp_context(pcrs_icp *icp) { return icp ? icp->ctx : nullptr; }
void pcrs_icp_lengths(const pcrs_icp *icp, size_t *n_source, size_t *n_target, size_t *n_normals) {
    *n_source = icp->n;
    *n_target = icp->m;
    *n_normals = icp->n_normals;
}

namespace {

// shared by find_correspondences and the ICP accumulate step
int correspond_and_reduce(pcrs_ctx *ctx, pcrs_index *target, int mode, const float *cx,
                          const float *cy, const float *cz, size_t n, const float *tnx,
                          const float *tny, const float *tnz, float max_d, uint32_t *nn_idx,
                          float *nn_d2, double *dsums, uint32_t *tgt_out, float *dist_out,
                          const IcpTail &tail = IcpTail(), bool coherent = false,
                          float *const *writable = nullptr) {
    KnnQuery q;
    q.qx = cx;
    q.qy = cy;
    q.qz = cz;
    q.nq = n;
    q.coherent = coherent;
    if (writable && tail.state) {  // the increment of the previous iteration is still to be applied
        q.xform = tail.state;
        q.wx = writable[0];
        q.wy = writable[1];
        q.wz = writable[2];
    }
    if (tail.state) q.skip_flag = &tail.state->done;
    if (max_d >= 0.0f) q.useful_distance = max_d;  // NaN or negative: no bound
    KnnOut o;
    o.nn_idx = nn_idx;
    o.nn_d2 = nn_d2;
    PCRS_TRY(run_query(target, q, o, 1));
    return pcrs_launch_icp_reduce(ctx, mode, cx, cy, cz, n, target->x, target->y, target->z, tnx,
                                  tny, tnz, nn_idx, nn_d2, max_d, tgt_out, dist_out, dsums, tail);
}

// Reorders the source spatially.  The sums of an ICP step do not depend on the order of the
// correspondences beyond f64 rounding, and pcrs_icp_get_current undoes the permutation.
int icp_order_source(pcrs_icp *icp) {
    pcrs_ctx *ctx = icp->ctx;
    const size_t n = icp->n;
    if (n < 4096) return PCRS_OK;
    pcrs_index *tmp = nullptr;
    PCRS_TRY(pcrs_index_build_dev(ctx, icp->sx, icp->sy, icp->sz, n, 8, PCRS_ALGO_AUTO, &tmp));
    int rc = PCRS_OK;
    if (tmp->algo == PCRS_ALGO_GRID && tmp->nvalid == n) {  // all finite
        for (int a = 0; a < 3 && rc == PCRS_OK; ++a)
            if (icp->ord_s[a].alloc(ctx, n * sizeof(float)) != hipSuccess)
                rc = pcrs_set_err(ctx, PCRS_ERR_OOM, "device allocation failed");
        if (rc == PCRS_OK && icp->perm.alloc(ctx, n * sizeof(uint32_t)) != hipSuccess)
            rc = pcrs_set_err(ctx, PCRS_ERR_OOM, "device allocation failed");
        if (rc == PCRS_OK)
            rc = pcrs_grid_sorted_columns(tmp, icp->ord_s[0].as<float>(), icp->ord_s[1].as<float>(),
                                          icp->ord_s[2].as<float>(), icp->perm.as<uint32_t>());
        if (rc == PCRS_OK) {
            icp->sx = icp->ord_s[0].as<float>();
            icp->sy = icp->ord_s[1].as<float>();
            icp->sz = icp->ord_s[2].as<float>();
        } else {
            icp->perm.release();
        }
    }
    (void)hipStreamSynchronize(ctx->stream);  // tmp's buffers are still in use by the kernels above
    pcrs_index_destroy(tmp);
    return rc;
}

int icp_alloc_work(pcrs_icp *icp) {
    pcrs_ctx *ctx = icp->ctx;
    const size_t n = icp->n;
    for (int a = 0; a < 3; ++a) PCRS_HIP_CHECK(ctx, icp->cur[a].alloc(ctx, n * sizeof(float)));
    PCRS_HIP_CHECK(ctx, icp->nn_idx.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, icp->nn_d2.alloc(ctx, n * sizeof(float)));
    PCRS_HIP_CHECK(ctx, icp->sums.alloc(ctx, PCRS_ICP_NSUMS * sizeof(double)));
    PCRS_HIP_CHECK(ctx, icp->state.alloc(ctx, sizeof(IcpLoopState)));
    PCRS_HIP_CHECK(ctx, icp->d2seq.alloc(ctx, 2 * n * sizeof(float)));
    icp->seq_scalars_bytes = (pcrs_sor_stats_scalars_bytes() + 63) & ~(size_t)63;
    PCRS_HIP_CHECK(ctx, icp->seq_work.alloc(ctx, icp->seq_scalars_bytes + pcrs_sor_stats_workspace(n)));
    return PCRS_OK;
}

// Fresh loop state on the device (identity transform, no previous rmse).  *hs: pinned host copy
// that the read-backs of the loop land in.
int icp_state_init(pcrs_icp *icp, IcpLoopState **hs_out, IcpLoopState **ds_out) {
    pcrs_ctx *ctx = icp->ctx;
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, sizeof(IcpLoopState) + 64, &pin));
    IcpLoopState *hs = reinterpret_cast<IcpLoopState *>(pin);
    std::memset(hs, 0, sizeof(*hs));
    hs->cum_r[0] = hs->cum_r[4] = hs->cum_r[8] = 1.0f;
    hs->inc_r[0] = hs->inc_r[4] = hs->inc_r[8] = 1.0f;
    hs->prev_rmse = hs->last_rmse = std::numeric_limits<float>::infinity();
    IcpLoopState *ds = icp->state.as<IcpLoopState>();
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(ds, hs, sizeof(*hs), hipMemcpyHostToDevice, ctx->stream));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // hs is reused for the read-backs
    *hs_out = hs;
    *ds_out = ds;
    return PCRS_OK;
}

void result_from_state(const IcpLoopState *hs, pcrs_icp_result *out) {
    std::memcpy(out->rotation, hs->cum_r, sizeof(hs->cum_r));
    std::memcpy(out->translation, hs->cum_t, sizeof(hs->cum_t));
    out->fitness = hs->last_fitness;
    out->rmse = hs->last_rmse;
    out->converged = hs->converged ? 1 : 0;
    out->num_iterations = hs->num_iterations;
}

}  // namespace

extern "C" {

int pcrs_find_correspondences_dev(pcrs_index *target, const float *dsx, const float *dsy,
                                  const float *dsz, size_t n, float max_distance,
                                  uint32_t *dtgt_idx, float *ddist, size_t *n_corr) try {
    if (!target) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = target->ctx;
    PCRS_TRY(bind_device(ctx));
    if (n_corr) *n_corr = 0;
    if (n == 0) return PCRS_OK;
    if (!dsx || !dsy || !dsz) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    DevBuf nn_idx, nn_d2, sums;
    PCRS_HIP_CHECK(ctx, nn_idx.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, nn_d2.alloc(ctx, n * sizeof(float)));
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, PCRS_ICP_NSUMS * sizeof(double)));
    PCRS_TRY(correspond_and_reduce(ctx, target, PCRS_ICP_POINT_TO_POINT, dsx, dsy, dsz, n, nullptr,
                                   nullptr, nullptr, max_distance, nn_idx.as<uint32_t>(),
                                   nn_d2.as<float>(), sums.as<double>(), dtgt_idx, ddist));
    double h[2] = {0, 0};
    PCRS_TRY(download(ctx, h, sums.p, sizeof(h)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_corr) *n_corr = (size_t)h[0];
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(target ? target->ctx : nullptr);
}

int pcrs_find_correspondences(pcrs_index *target, const float *sx, const float *sy,
                              const float *sz, size_t n, float max_distance, uint32_t *tgt_idx,
                              float *dist, size_t *n_corr) try {
    if (!target) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = target->ctx;
    PCRS_TRY(bind_device(ctx));
    if (n_corr) *n_corr = 0;
    if (n == 0) return PCRS_OK;
    if (!sx || !sy || !sz) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    DevBuf dx, dy, dz, dti, dd;
    PCRS_TRY(upload(ctx, dx, sx, n));
    PCRS_TRY(upload(ctx, dy, sy, n));
    PCRS_TRY(upload(ctx, dz, sz, n));
    PCRS_HIP_CHECK(ctx, dti.alloc(ctx, n * sizeof(uint32_t)));
    PCRS_HIP_CHECK(ctx, dd.alloc(ctx, n * sizeof(float)));
    PCRS_TRY(pcrs_find_correspondences_dev(target, dx.as<float>(), dy.as<float>(), dz.as<float>(),
                                           n, max_distance, dti.as<uint32_t>(), dd.as<float>(),
                                           n_corr));
    if (tgt_idx) PCRS_TRY(download(ctx, tgt_idx, dti.p, n * sizeof(uint32_t)));
    if (dist) PCRS_TRY(download(ctx, dist, dd.p, n * sizeof(float)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(target ? target->ctx : nullptr);
}

int pcrs_icp_create_dev(pcrs_ctx *ctx, const float *dsx, const float *dsy, const float *dsz,
                        size_t n, const float *dtx, const float *dty, const float *dtz, size_t m,
                        const float *dtnx, const float *dtny, const float *dtnz, size_t n_normals,
                        pcrs_icp **out) try {
    if (!out) return PCRS_ERR_INVALID_ARG;
    *out = nullptr;
    PCRS_TRY(bind_device(ctx));
    if ((n && (!dsx || !dsy || !dsz)) || (m && (!dtx || !dty || !dtz)))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    const bool have_n = dtnx && dtny && dtnz;
    pcrs_icp *icp = new (std::nothrow) pcrs_icp();
    if (!icp) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    icp->ctx = ctx;
    icp->sx = dsx;
    icp->sy = dsy;
    icp->sz = dsz;
    icp->n = n;
    icp->m = m;
    icp->tnx = have_n ? dtnx : nullptr;
    icp->tny = have_n ? dtny : nullptr;
    icp->tnz = have_n ? dtnz : nullptr;
    icp->n_normals = have_n ? n_normals : 0;
    int rc = pcrs_index_build_dev(ctx, dtx, dty, dtz, m, 1, PCRS_ALGO_AUTO, &icp->target);
    if (rc == PCRS_OK) rc = icp_order_source(icp);
    if (rc == PCRS_OK) rc = icp_alloc_work(icp);
    if (rc == PCRS_OK) rc = pcrs_icp_reset(icp);
    if (rc != PCRS_OK) {
        pcrs_icp_destroy(icp);
        return rc;
    }
    *out = icp;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_icp_create(pcrs_ctx *ctx, const float *sx, const float *sy, const float *sz, size_t n,
                    const float *tx, const float *ty, const float *tz, size_t m, const float *tnx,
                    const float *tny, const float *tnz, size_t n_normals, pcrs_icp **out) try {
    if (!out) return PCRS_ERR_INVALID_ARG;
    *out = nullptr;
    PCRS_TRY(bind_device(ctx));
    if ((n && (!sx || !sy || !sz)) || (m && (!tx || !ty || !tz)))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    const bool have_n = tnx && tny && tnz;
    pcrs_icp *icp = new (std::nothrow) pcrs_icp();
    if (!icp) return pcrs_set_err(ctx, PCRS_ERR_OOM, "host allocation failed");
    icp->ctx = ctx;
    icp->n = n;
    icp->m = m;
    const float *s[3] = {sx, sy, sz}, *t[3] = {tx, ty, tz}, *nn[3] = {tnx, tny, tnz};
    int rc = PCRS_OK;
    for (int a = 0; a < 3 && rc == PCRS_OK; ++a) rc = upload(ctx, icp->own_s[a], s[a], n);
    for (int a = 0; a < 3 && rc == PCRS_OK; ++a) rc = upload(ctx, icp->own_t[a], t[a], m);
    if (have_n)
        for (int a = 0; a < 3 && rc == PCRS_OK; ++a) rc = upload(ctx, icp->own_n[a], nn[a], n_normals);
    if (rc == PCRS_OK) {
        icp->sx = icp->own_s[0].as<float>();
        icp->sy = icp->own_s[1].as<float>();
        icp->sz = icp->own_s[2].as<float>();
        if (have_n) {
            icp->tnx = icp->own_n[0].as<float>();
            icp->tny = icp->own_n[1].as<float>();
            icp->tnz = icp->own_n[2].as<float>();
            icp->n_normals = n_normals;
        }
        rc = pcrs_index_build_dev(ctx, icp->own_t[0].as<float>(), icp->own_t[1].as<float>(),
                                  icp->own_t[2].as<float>(), m, 1, PCRS_ALGO_AUTO, &icp->target);
    }
    if (rc == PCRS_OK) rc = icp_order_source(icp);
    if (rc == PCRS_OK) rc = icp_alloc_work(icp);
    if (rc == PCRS_OK) rc = pcrs_icp_reset(icp);
    if (rc == PCRS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = pcrs_set_err(ctx, PCRS_ERR_HIP, "stream synchronize failed");
    if (rc != PCRS_OK) {
        pcrs_icp_destroy(icp);
        return rc;
    }
    *out = icp;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

void pcrs_icp_destroy(pcrs_icp *icp) try {
    if (!icp) return;
    if (icp->ctx) {
        (void)hipSetDevice(icp->ctx->device);
        (void)hipStreamSynchronize(icp->ctx->stream);
    }
    pcrs_index_destroy(icp->target);
    delete icp;
} catch (...) {
}

int pcrs_icp_reset(pcrs_icp *icp) try {
    if (!icp) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    // icp.rs:145: apply_transform(source, identity) — evaluated, not copied, so that the f32
    // result matches the reference for signed zeros and non-finite inputs.
    const float rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    return pcrs_launch_transform(ctx, icp->sx, icp->sy, icp->sz, icp->n, rot, t,
                                 icp->cur[0].as<float>(), icp->cur[1].as<float>(),
                                 icp->cur[2].as<float>());
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_accumulate(pcrs_icp *icp, int mode, float max_distance, double *sums) try {
    if (!icp || !sums) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    if (mode != PCRS_ICP_POINT_TO_POINT && mode != PCRS_ICP_POINT_TO_PLANE)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "unknown ICP mode %d", mode);
    if (mode == PCRS_ICP_POINT_TO_PLANE && (!icp->tnx || icp->n_normals != icp->m))
        return pcrs_set_err(ctx, PCRS_ERR_NORMALS_MISMATCH,
                            "target_normals length (%zu) does not match target cloud length (%zu)",
                            icp->n_normals, icp->m);
    PCRS_TRY(correspond_and_reduce(ctx, icp->target, mode, icp->cur[0].as<float>(),
                                   icp->cur[1].as<float>(), icp->cur[2].as<float>(), icp->n,
                                   icp->tnx, icp->tny, icp->tnz, max_distance,
                                   icp->nn_idx.as<uint32_t>(), icp->nn_d2.as<float>(),
                                   icp->sums.as<double>(), nullptr, nullptr, IcpTail(),
                                   icp->perm.p != nullptr));
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, PCRS_ICP_NSUMS * sizeof(double), &pin));
    PCRS_TRY(download(ctx, pin, icp->sums.p, PCRS_ICP_NSUMS * sizeof(double)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(sums, pin, PCRS_ICP_NSUMS * sizeof(double));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_solve(int mode, const double *sums, float rotation[9], float translation[3]) try {
    if (!sums || !rotation || !translation) return PCRS_ERR_INVALID_ARG;
    if (mode == PCRS_ICP_POINT_TO_PLANE)
        pcrs_host_solve_plane(sums, rotation, translation);
    else if (mode == PCRS_ICP_POINT_TO_POINT)
        pcrs_host_solve_point(sums, rotation, translation);
    else
        return PCRS_ERR_INVALID_ARG;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(nullptr);
}

int pcrs_icp_apply(pcrs_icp *icp, const float rotation[9], const float translation[3]) try {
    if (!icp || !rotation || !translation) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    float *cx = icp->cur[0].as<float>(), *cy = icp->cur[1].as<float>(), *cz = icp->cur[2].as<float>();
    return pcrs_launch_transform(ctx, cx, cy, cz, icp->n, rotation, translation, cx, cy, cz);
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_get_current(pcrs_icp *icp, float *x, float *y, float *z) try {
    if (!icp) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    float *o[3] = {x, y, z};
    DevBuf tmp;
    if (icp->perm.p) PCRS_HIP_CHECK(ctx, tmp.alloc(ctx, icp->n * sizeof(float)));
    for (int a = 0; a < 3; ++a) {
        if (!o[a]) continue;
        const void *src = icp->cur[a].p;
        if (icp->perm.p) {  // back to the caller's point order
            PCRS_TRY(pcrs_launch_scatter_perm(ctx, icp->cur[a].as<float>(),
                                              icp->perm.as<uint32_t>(), icp->n, tmp.as<float>()));
            src = tmp.p;
        }
        PCRS_TRY(download(ctx, o[a], src, icp->n * sizeof(float)));
    }
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

static void result_identity(pcrs_icp_result *r) {
    std::memset(r, 0, sizeof(*r));
    r->rotation[0] = r->rotation[4] = r->rotation[8] = 1.0f;
}

int pcrs_icp_run(pcrs_icp *icp, int mode, uint32_t max_iterations, float tolerance,
                 float max_correspondence_distance, pcrs_icp_result *out) try {
    if (!icp || !out) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    if (mode != PCRS_ICP_POINT_TO_POINT && mode != PCRS_ICP_POINT_TO_PLANE)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "unknown ICP mode %d", mode);
    result_identity(out);
    // icp_plane.rs:27-32 comes before the empty-cloud shortcut
    if (mode == PCRS_ICP_POINT_TO_PLANE && icp->n_normals != icp->m)
        return pcrs_set_err(ctx, PCRS_ERR_NORMALS_MISMATCH,
                            "target_normals length (%zu) does not match target cloud length (%zu)",
                            icp->n_normals, icp->m);
    if (icp->n == 0 || icp->m == 0) {  // icp.rs:131-139
        out->converged = (icp->n == 0 && icp->m == 0) ? 1 : 0;
        return PCRS_OK;
    }
    PCRS_TRY(pcrs_icp_reset(icp));
    // The loop state lives on the device (icp_tail_step carries icp.rs:147-188), so the host only
    // looks at it every few iterations; iterations queued behind the end of the loop are no-ops.
    IcpLoopState *hs = nullptr, *ds = nullptr;
    PCRS_TRY(icp_state_init(icp, &hs, &ds));
    float *cx = icp->cur[0].as<float>(), *cy = icp->cur[1].as<float>(), *cz = icp->cur[2].as<float>();
    // The convergence test |prev_rmse - rmse| < tolerance (icp.rs:173) can only fire for a
    // positive tolerance.  Then every iteration takes its rmse from the reference's own sum
    // (dist*dist in source order, left to right in f32), so that the loop ends in the iteration
    // in which the reference ends.  Otherwise the f64 tree sum steers nothing, and only the rmse
    // that is reported is recomputed that way after the loop.
    IcpTail tail;
    tail.state = ds;
    tail.n_source = icp->n;
    tail.tolerance = tolerance;
    tail.exact = tolerance > 0.0f;
    tail.perm = icp->perm.as<uint32_t>();  // null when the source kept its order
    tail.seq_scalars = icp->seq_work.p;
    tail.seq_work = reinterpret_cast<char *>(icp->seq_work.p) + icp->seq_scalars_bytes;
    tail.seq_work_bytes = icp->seq_work.bytes - icp->seq_scalars_bytes;
    float *const cur_w[3] = {cx, cy, cz};
    // icp.rs:186 (current <- increment(current)) is not a launch of its own: the search of the
    // next iteration applies the pending increment as it loads its queries
    auto iteration = [&](uint32_t it) {
        tail.d2seq = icp->d2seq.as<float>() + (size_t)(it & 1u) * icp->n;
        return correspond_and_reduce(ctx, icp->target, mode, cx, cy, cz, icp->n, icp->tnx, icp->tny,
                                     icp->tnz, max_correspondence_distance,
                                     icp->nn_idx.as<uint32_t>(), icp->nn_d2.as<float>(),
                                     icp->sums.as<double>(), nullptr, nullptr, tail,
                                     icp->perm.p != nullptr, it > 0 ? cur_w : nullptr);
    };
    if (max_iterations == 0) {  // icp.rs:190-197: one metrics pass, identity transform
        tail.metrics_only = true;
        tail.exact = true;
        PCRS_TRY(iteration(0));
    }
    // Iterations are queued in batches and the state is looked at after each: four at a time for
    // the first sixteen, then eight (registrations that converge do so early, and an iteration
    // queued behind the end of the loop still costs its launches -- with a live tolerance also
    // the f32 sum's, which does not look at the flag; long runs pay for a read-back only every
    // eighth iteration).
    for (uint32_t iter = 0; iter < max_iterations;) {
        const uint32_t batch = iter < 16u ? 4u : 8u;
        const uint32_t nb = std::min(batch, max_iterations - iter);
        for (uint32_t j = 0; j < nb; ++j)
            PCRS_TRY(iteration(iter + j));  // [pending increment,] search, reduction, tail
        iter += nb;
        PCRS_TRY(pcrs_launch_publish(ctx, ds, hs, sizeof(*hs)));  // hs is pinned
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (hs->done) break;
    }
    // the increment of the last iteration (a no-op once the loop has ended on its own)
    if (max_iterations > 0) PCRS_TRY(pcrs_launch_transform_state(ctx, cx, cy, cz, icp->n, ds));
    if (max_iterations == 0) {
        PCRS_TRY(pcrs_launch_publish(ctx, ds, hs, sizeof(*hs)));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    } else if (!tail.exact && hs->last_live) {
        // the reported rmse from the reference's sum over the last live iteration's distances
        const float *d2 = icp->d2seq.as<float>() + (size_t)((hs->last_live - 1u) & 1u) * icp->n;
        PCRS_TRY(pcrs_launch_seq_sum(ctx, d2, icp->n, tail.seq_scalars, tail.seq_work,
                                     tail.seq_work_bytes));
        PCRS_TRY(pcrs_launch_icp_exact_rmse(ctx, ds, tail.seq_scalars));
        PCRS_TRY(pcrs_launch_publish(ctx, ds, hs, sizeof(*hs)));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    result_from_state(hs, out);
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

// ---- the loop with the source split over ranks: see the header for the protocol

static int icp_check_mode(pcrs_icp *icp, int mode) {
    pcrs_ctx *ctx = icp->ctx;
    if (mode != PCRS_ICP_POINT_TO_POINT && mode != PCRS_ICP_POINT_TO_PLANE)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "unknown ICP mode %d", mode);
    if (mode == PCRS_ICP_POINT_TO_PLANE && (!icp->tnx || icp->n_normals != icp->m))
        return pcrs_set_err(ctx, PCRS_ERR_NORMALS_MISMATCH,
                            "target_normals length (%zu) does not match target cloud length (%zu)",
                            icp->n_normals, icp->m);
    return PCRS_OK;
}

int pcrs_icp_shard_begin(pcrs_icp *icp, size_t n_source_total) try {
    if (!icp) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    if (n_source_total < icp->n || n_source_total >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG,
                            "n_source_total (%zu) must cover this shard (%zu points)",
                            n_source_total, icp->n);
    if (icp->m == 0) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "empty target");
    PCRS_TRY(pcrs_icp_reset(icp));
    IcpLoopState *hs = nullptr, *ds = nullptr;
    PCRS_TRY(icp_state_init(icp, &hs, &ds));
    icp->shard_total = n_source_total;
    PCRS_HIP_CHECK(ctx, icp->shard_seq.alloc(ctx, icp->seq_scalars_bytes +
                                                      pcrs_sor_stats_workspace(n_source_total)));
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_shard_accumulate_dev(pcrs_icp *icp, int mode, float max_distance, uint32_t iteration,
                                  double *dsums, const float **dd2) try {
    if (!icp) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    if (!dsums) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null sums pointer");
    if (!icp->shard_total && icp->n) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "pcrs_icp_shard_begin has not run");
    PCRS_TRY(icp_check_mode(icp, mode));
    float *d2 = icp->d2seq.as<float>() + (size_t)(iteration & 1u) * icp->n;
    if (dd2) *dd2 = d2;
    if (icp->n == 0) {  // a rank without points adds nothing
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(dsums, 0, PCRS_ICP_NSUMS * sizeof(double), ctx->stream));
        return PCRS_OK;
    }
    float *cx = icp->cur[0].as<float>(), *cy = icp->cur[1].as<float>(), *cz = icp->cur[2].as<float>();
    float *const cur_w[3] = {cx, cy, cz};
    IcpTail tail;
    tail.state = icp->state.as<IcpLoopState>();
    tail.n_source = icp->shard_total;
    tail.sums_only = true;
    tail.d2seq = d2;
    tail.perm = icp->perm.as<uint32_t>();
    return correspond_and_reduce(ctx, icp->target, mode, cx, cy, cz, icp->n, icp->tnx, icp->tny,
                                 icp->tnz, max_distance, icp->nn_idx.as<uint32_t>(),
                                 icp->nn_d2.as<float>(), dsums, nullptr, nullptr, tail,
                                 icp->perm.p != nullptr, iteration > 0 ? cur_w : nullptr);
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_shard_update_dev(pcrs_icp *icp, int mode, const double *dsums_all, uint32_t world,
                              float tolerance, const float *dd2_all, int metrics_only) try {
    if (!icp) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    if (!dsums_all || world == 0) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "no sums to add");
    if (!icp->shard_seq.p) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "pcrs_icp_shard_begin has not run");
    PCRS_TRY(icp_check_mode(icp, mode));
    IcpTail tail;
    tail.state = icp->state.as<IcpLoopState>();
    tail.n_source = icp->shard_total;
    tail.tolerance = tolerance;
    tail.metrics_only = metrics_only != 0;
    tail.exact = dd2_all != nullptr;
    tail.seq_scalars = icp->shard_seq.p;
    tail.seq_work = reinterpret_cast<char *>(icp->shard_seq.p) + icp->seq_scalars_bytes;
    tail.seq_work_bytes = icp->shard_seq.bytes - icp->seq_scalars_bytes;
    if (tail.exact)
        PCRS_TRY(pcrs_launch_seq_sum(ctx, dd2_all, icp->shard_total, tail.seq_scalars, tail.seq_work,
                                     tail.seq_work_bytes, &tail.state->done));
    return pcrs_launch_icp_shard_tail(ctx, mode, dsums_all, world, icp->sums.as<double>(), tail);
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_shard_rmse_dev(pcrs_icp *icp, const float *dd2_all) try {
    if (!icp) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    if (!dd2_all || !icp->shard_seq.p) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "nothing to sum");
    void *scalars = icp->shard_seq.p;
    PCRS_TRY(pcrs_launch_seq_sum(ctx, dd2_all, icp->shard_total, scalars,
                                 reinterpret_cast<char *>(scalars) + icp->seq_scalars_bytes,
                                 icp->shard_seq.bytes - icp->seq_scalars_bytes));
    return pcrs_launch_icp_exact_rmse(ctx, icp->state.as<IcpLoopState>(), scalars);
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_shard_state(pcrs_icp *icp, pcrs_icp_result *out, int *done, uint32_t *last_live) try {
    if (!icp || !out) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, sizeof(IcpLoopState) + 64, &pin));
    IcpLoopState *hs = reinterpret_cast<IcpLoopState *>(pin);
    PCRS_TRY(pcrs_launch_publish(ctx, icp->state.as<IcpLoopState>(), hs, sizeof(*hs)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    result_identity(out);
    result_from_state(hs, out);
    if (done) *done = hs->done ? 1 : 0;
    if (last_live) *last_live = hs->last_live;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_shard_finish(pcrs_icp *icp) try {
    if (!icp) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = icp->ctx;
    PCRS_TRY(bind_device(ctx));
    return pcrs_launch_transform_state(ctx, icp->cur[0].as<float>(), icp->cur[1].as<float>(),
                                       icp->cur[2].as<float>(), icp->n, icp->state.as<IcpLoopState>());
} catch (...) {
    return pcrs_on_exception(icp ? icp->ctx : nullptr);
}

int pcrs_icp_register(pcrs_ctx *ctx, int mode, const float *sx, const float *sy, const float *sz, size_t n,
             const float *tx, const float *ty, const float *tz, size_t m, const float *tnx,
             const float *tny, const float *tnz, size_t n_normals, uint32_t max_iterations,
             float tolerance, float max_correspondence_distance, pcrs_icp_result *out) try {
    if (!out) return PCRS_ERR_INVALID_ARG;
    PCRS_TRY(bind_device(ctx));
    result_identity(out);
    if (mode != PCRS_ICP_POINT_TO_POINT && mode != PCRS_ICP_POINT_TO_PLANE)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "unknown ICP mode %d", mode);
    const bool have_n = tnx && tny && tnz;
    if (mode == PCRS_ICP_POINT_TO_PLANE && ((have_n ? n_normals : 0) != m))
        return pcrs_set_err(ctx, PCRS_ERR_NORMALS_MISMATCH,
                            "target_normals length (%zu) does not match target cloud length (%zu)",
                            have_n ? n_normals : (size_t)0, m);
    if (n == 0 || m == 0) {
        out->converged = (n == 0 && m == 0) ? 1 : 0;
        return PCRS_OK;
    }
    pcrs_icp *icp = nullptr;
    PCRS_TRY(pcrs_icp_create(ctx, sx, sy, sz, n, tx, ty, tz, m, tnx, tny, tnz, n_normals, &icp));
    int rc = pcrs_icp_run(icp, mode, max_iterations, tolerance, max_correspondence_distance, out);
    pcrs_icp_destroy(icp);
    return rc;
} catch (...) {
    return pcrs_on_exception(ctx);
}

}  // extern "C"

// Global SOR statistics: computed on the device with the reference's summation order
// (stats.hip); only the few scalars come back.
int pcrs_sor_threshold_impl(pcrs_ctx *ctx, const float *dmean, size_t n, float std_mul, float *thr,
                            int *has_finite) {
    *thr = std::numeric_limits<float>::quiet_NaN();
    *has_finite = 0;
    if (n == 0) return PCRS_OK;
    const size_t sb = (pcrs_sor_stats_scalars_bytes() + 63) & ~(size_t)63;
    const size_t wb = pcrs_sor_stats_workspace(n);
    void *scratch = nullptr;
    PCRS_TRY(pcrs_ctx_scratch(ctx, sb + wb, &scratch));
    void *pin = nullptr;
    PCRS_TRY(pcrs_ctx_pinned(ctx, sb, &pin));
    PCRS_TRY(pcrs_launch_sor_stats(ctx, dmean, n, std_mul, scratch,
                                   reinterpret_cast<char *>(scratch) + sb, wb));
    PCRS_TRY(download(ctx, pin, scratch, pcrs_sor_stats_scalars_bytes()));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    float t;
    int ok;
    pcrs_sor_stats_decode(pin, nullptr, &t, &ok);
    if (ok) {
        *thr = t;
        *has_finite = 1;
    }
    return PCRS_OK;
}
