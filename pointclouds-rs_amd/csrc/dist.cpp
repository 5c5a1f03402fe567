// One cloud over the GPUs of a node, driven from C: the steps of pointclouds_rs/dist.py (ownership
// by slabs of space for SOR + normals, the source split over the ranks for ICP) with the
// collectives issued on RCCL directly, so that a host without Python -- the reference's Rust
// crates behind a -sys binding -- runs the multi-GPU path with one call per step and rank.
//
// RCCL is loaded on first use (dlopen of librccl.so.1: the copy already in the process when the
// host has one, e.g. PyTorch's), never at library load: a single-GPU host needs no RCCL at all.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

#include "pcrs_internal.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl &rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl &r = g_rccl;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
#define PCRS_SYM(NAME) r.NAME = reinterpret_cast<decltype(r.NAME)>(dlsym(r.lib, "nccl" #NAME))
        PCRS_SYM(GetUniqueId);
        PCRS_SYM(CommInitRank);
        PCRS_SYM(CommDestroy);
        PCRS_SYM(AllGather);
        PCRS_SYM(AllReduce);
        PCRS_SYM(Broadcast);
        PCRS_SYM(Send);
        PCRS_SYM(Recv);
        PCRS_SYM(GroupStart);
        PCRS_SYM(GroupEnd);
        PCRS_SYM(GetErrorString);
#undef PCRS_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce &&
               r.Broadcast && r.Send && r.Recv && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return g_rccl;
}

int need_rccl(pcrs_ctx *ctx) {
    if (rccl().ok) return PCRS_OK;
    return pcrs_set_err(ctx, PCRS_ERR_HIP, "librccl.so.1 could not be loaded: the multi-GPU entry points need RCCL");
}

#define PCRS_NCCL(ctx, expr)                                                                  \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess)                                                                \
            return pcrs_set_err((ctx), PCRS_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(_r)); \
    } while (0)

// [begin, end) of rank r's contiguous index range (the first n % world ranges one longer)
void shard_range(size_t n, int world, int r, size_t *b, size_t *e) {
    const size_t base = n / (size_t)world, rem = n % (size_t)world;
    *b = (size_t)r * base + std::min<size_t>((size_t)r, rem);
    *e = *b + base + ((size_t)r < rem ? 1 : 0);
}

// all-gather of pieces of different lengths: every rank broadcasts its piece into its place
int all_gather_var(pcrs_ctx *ctx, ncclComm_t comm, int rank, int world, const void *mine,
                   void *all, const std::vector<size_t> &counts, size_t elem, ncclDataType_t type) {
    const Rccl &r = rccl();
    bool equal = true;
    for (int p = 1; p < world; ++p) equal = equal && counts[(size_t)p] == counts[0];
    if (equal) {  // the plain collective (in place when the piece already lies in its place)
        if (counts[0]) PCRS_NCCL(ctx, r.AllGather(mine, all, counts[0], type, comm, ctx->stream));
        return PCRS_OK;
    }
    PCRS_NCCL(ctx, r.GroupStart());
    size_t off = 0;
    for (int p = 0; p < world; ++p) {
        char *dst = reinterpret_cast<char *>(all) + off * elem;
        if (counts[(size_t)p])
            PCRS_NCCL(ctx, r.Broadcast(p == rank ? mine : dst, dst, counts[(size_t)p], type, p, comm,
                                       ctx->stream));
        off += counts[(size_t)p];
    }
    PCRS_NCCL(ctx, r.GroupEnd());
    return PCRS_OK;
}

int bind(pcrs_ctx *ctx) {
    if (!ctx) return PCRS_ERR_INVALID_ARG;
    PCRS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return PCRS_OK;
}

}  // namespace

extern "C" {

int pcrs_dist_unique_id(char id[PCRS_DIST_UNIQUE_ID_BYTES]) try {
    static_assert(PCRS_DIST_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id || !rccl().ok) return PCRS_ERR_HIP;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return PCRS_ERR_HIP;
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(nullptr);
}

int pcrs_dist_comm_init(pcrs_ctx *ctx, const char id[PCRS_DIST_UNIQUE_ID_BYTES], int rank, int world,
                        void **comm) try {
    if (!comm || !id) return PCRS_ERR_INVALID_ARG;
    *comm = nullptr;
    PCRS_TRY(bind(ctx));
    PCRS_TRY(need_rccl(ctx));
    if (world < 1 || rank < 0 || rank >= world)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "rank %d of %d", rank, world);
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    PCRS_NCCL(ctx, rccl().CommInitRank(&c, world, u, rank));
    *comm = c;
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

void pcrs_dist_comm_destroy(void *comm) try {
    if (comm && rccl().ok) (void)rccl().CommDestroy(reinterpret_cast<ncclComm_t>(comm));
} catch (...) {
}

int pcrs_dist_sor_normals_dev(pcrs_ctx *ctx, void *nccl_comm, int rank, int world, const float *dx,
                              const float *dy, const float *dz, size_t n_local, size_t n_total,
                              uint32_t k_sor, float std_mul, uint32_t k_normals,
                              const float viewpoint[3], size_t capacity, uint32_t *dpoint,
                              uint8_t *dkeep, float *dmean, float *dnx, float *dny, float *dnz,
                              pcrs_dist_sor_result *result) try {
    PCRS_TRY(bind(ctx));
    PCRS_TRY(need_rccl(ctx));
    if (!nccl_comm || !result || world < 1 || rank < 0 || rank >= world)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null communicator or result, or rank %d of %d", rank, world);
    *result = pcrs_dist_sor_result{};
    result->threshold = std::numeric_limits<float>::quiet_NaN();
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    const Rccl &r = rccl();
    if (n_total >= 0xFFFFFFFFull)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "clouds are limited to 2^32-2 points");
    std::vector<size_t> counts((size_t)world);
    size_t begin = 0, end = 0;
    for (int p = 0; p < world; ++p) {
        size_t b, e;
        shard_range(n_total, world, p, &b, &e);
        counts[(size_t)p] = e - b;
        if (p == rank) {
            begin = b;
            end = e;
        }
    }
    if (n_local != end - begin)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "shard of %zu points, its index range holds %zu",
                            n_local, end - begin);
    const bool want_n = dnx || dny || dnz;
    if (want_n && !(dnx && dny && dnz && viewpoint && k_normals))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "normals need three columns, a viewpoint and k >= 1");
    if (n_total == 0) return PCRS_OK;
    if (n_local && (!dx || !dy || !dz)) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null column pointer");
    if (!dpoint || (k_sor && (!dkeep || !dmean)))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "dpoint is required, and dkeep and dmean with k_sor >= 1");
    if (k_sor == 0 && !want_n) return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "nothing to compute");
    // 1. every rank holds the whole cloud
    DevBuf fx, fy, fz;
    PCRS_HIP_CHECK(ctx, fx.alloc(ctx, n_total * sizeof(float)));
    PCRS_HIP_CHECK(ctx, fy.alloc(ctx, n_total * sizeof(float)));
    PCRS_HIP_CHECK(ctx, fz.alloc(ctx, n_total * sizeof(float)));
    PCRS_TRY(all_gather_var(ctx, comm, rank, world, dx, fx.p, counts, sizeof(float), ncclFloat32));
    PCRS_TRY(all_gather_var(ctx, comm, rank, world, dy, fy.p, counts, sizeof(float), ncclFloat32));
    PCRS_TRY(all_gather_var(ctx, comm, rank, world, dz, fz.p, counts, sizeof(float), ncclFloat32));
    // 2. this rank's slab of it, indexed with a halo; all ranks take the same road
    const uint32_t k_hint = std::max(k_sor ? k_sor + 1u : 0u, want_n ? k_normals : 0u);
    pcrs_index *ix = nullptr;
    size_t nq = 0;
    int axis = 0;
    PCRS_TRY(pcrs_slab_index_build_dev(ctx, fx.as<float>(), fy.as<float>(), fz.as<float>(), n_total,
                                       k_hint, (uint32_t)world, (uint32_t)rank, &ix, &nq, &axis,
                                       nullptr));
    struct Guard {
        pcrs_index *&p;
        ~Guard() { pcrs_index_destroy(p); }
    } guard{ix};
    DevBuf flag;
    PCRS_HIP_CHECK(ctx, flag.alloc(ctx, 2 * sizeof(int)));
    int *pin_flag = nullptr;
    {
        void *pin = nullptr;
        PCRS_TRY(pcrs_ctx_pinned(ctx, 64, &pin));
        pin_flag = reinterpret_cast<int *>(pin);
    }
    const int have = ix ? 1 : 0;
    PCRS_HIP_CHECK(ctx, hipMemcpyAsync(flag.p, &have, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PCRS_NCCL(ctx, r.AllReduce(flag.p, flag.as<int>() + 1, 1, ncclInt32, ncclMin, comm, ctx->stream));
    PCRS_TRY(pcrs_launch_publish(ctx, flag.as<int>() + 1, pin_flag, sizeof(int)));
    PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const bool slabs = *pin_flag != 0;
    result->by_slabs = slabs ? 1 : 0;
    DevBuf mean_full;
    if (k_sor) PCRS_HIP_CHECK(ctx, mean_full.alloc(ctx, n_total * sizeof(float)));
    if (slabs) {
        if (nq > capacity)
            return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "this rank's slab holds %zu points, the outputs %zu", nq, capacity);
        size_t escalated = 0;
        PCRS_TRY(pcrs_sor_normals_selected_dev(ix, k_sor, want_n ? k_normals : 0, viewpoint, dpoint,
                                               k_sor ? dmean : nullptr, dnx, dny, dnz, &escalated));
        result->n_escalated = escalated;
        result->n_answered = nq;
        if (!k_sor) return PCRS_OK;
        // 3. every mean to the owner of its index range, dropped into place there, and the
        //    ranges (in rank order: index order) gathered by everybody
        DevBuf sp, sv, rp, rv, range, dcounts;
        PCRS_HIP_CHECK(ctx, sp.alloc(ctx, (nq + 1) * sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, sv.alloc(ctx, (nq + 1) * sizeof(float)));
        std::vector<size_t> send((size_t)world);
        PCRS_TRY(pcrs_route_pairs_dev(ctx, dpoint, dmean, nq, n_total, (uint32_t)world, sp.as<uint32_t>(),
                                      sv.as<float>(), send.data()));
        // who sends how much to whom: all-gather of every rank's row of counts
        PCRS_HIP_CHECK(ctx, dcounts.alloc(ctx, ((size_t)world + (size_t)world * (size_t)world) * sizeof(unsigned long long)));
        std::vector<unsigned long long> row((size_t)world);
        for (int p = 0; p < world; ++p) row[(size_t)p] = send[(size_t)p];
        unsigned long long *d_row = dcounts.as<unsigned long long>();
        unsigned long long *d_all = d_row + world;
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(d_row, row.data(), (size_t)world * sizeof(unsigned long long),
                                           hipMemcpyHostToDevice, ctx->stream));
        PCRS_NCCL(ctx, r.AllGather(d_row, d_all, (size_t)world, ncclUint64, comm, ctx->stream));
        // (the calls above may have moved the context's pinned block: ask again)
        void *pin = nullptr;
        PCRS_TRY(pcrs_ctx_pinned(ctx, (size_t)world * (size_t)world * sizeof(unsigned long long), &pin));
        unsigned long long *h_all = reinterpret_cast<unsigned long long *>(pin);
        PCRS_HIP_CHECK(ctx, hipMemcpyAsync(h_all, d_all, (size_t)world * (size_t)world * sizeof(unsigned long long),
                                           hipMemcpyDeviceToHost, ctx->stream));
        PCRS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<size_t> recv((size_t)world);
        size_t recv_total = 0;
        for (int p = 0; p < world; ++p) {
            recv[(size_t)p] = (size_t)h_all[(size_t)p * (size_t)world + (size_t)rank];
            recv_total += recv[(size_t)p];
        }
        if (recv_total != n_local)
            return pcrs_set_err(ctx, PCRS_ERR_HIP, "internal: the slabs do not partition the cloud (%zu of %zu)", recv_total, n_local);
        PCRS_HIP_CHECK(ctx, rp.alloc(ctx, (recv_total + 1) * sizeof(uint32_t)));
        PCRS_HIP_CHECK(ctx, rv.alloc(ctx, (recv_total + 1) * sizeof(float)));
        PCRS_NCCL(ctx, r.GroupStart());
        size_t so = 0, ro = 0;
        for (int p = 0; p < world; ++p) {
            if (send[(size_t)p]) {
                PCRS_NCCL(ctx, r.Send(sp.as<uint32_t>() + so, send[(size_t)p], ncclUint32, p, comm, ctx->stream));
                PCRS_NCCL(ctx, r.Send(sv.as<float>() + so, send[(size_t)p], ncclFloat32, p, comm, ctx->stream));
            }
            if (recv[(size_t)p]) {
                PCRS_NCCL(ctx, r.Recv(rp.as<uint32_t>() + ro, recv[(size_t)p], ncclUint32, p, comm, ctx->stream));
                PCRS_NCCL(ctx, r.Recv(rv.as<float>() + ro, recv[(size_t)p], ncclFloat32, p, comm, ctx->stream));
            }
            so += send[(size_t)p];
            ro += recv[(size_t)p];
        }
        PCRS_NCCL(ctx, r.GroupEnd());
        PCRS_HIP_CHECK(ctx, range.alloc(ctx, (n_local + 1) * sizeof(float)));
        PCRS_TRY(pcrs_scatter_range_f32_dev(ctx, rv.as<float>(), rp.as<uint32_t>(), recv_total, begin,
                                            range.as<float>()));
        PCRS_TRY(all_gather_var(ctx, comm, rank, world, range.p, mean_full.p, counts, sizeof(float), ncclFloat32));
        result->n_answered = nq;
    } else {
        // no slabs (points without a place in space, a slab too small for a grid): every rank
        // indexes the whole cloud and answers the points it holds, in their order
        if (n_local > capacity)
            return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "this rank holds %zu points, the outputs %zu", n_local, capacity);
        pcrs_index *whole = nullptr;
        PCRS_TRY(pcrs_index_build_dev(ctx, fx.as<float>(), fy.as<float>(), fz.as<float>(), n_total, k_hint,
                                      PCRS_ALGO_AUTO, &whole));
        int rc = pcrs_sor_normals_dev(whole, begin, n_local, k_sor, want_n ? k_normals : 0, viewpoint,
                                      k_sor ? dmean : nullptr, dnx, dny, dnz);
        pcrs_index_destroy(whole);
        PCRS_TRY(rc);
        PCRS_TRY(pcrs_launch_index_run(ctx, dpoint, n_local, (uint32_t)begin));
        result->n_answered = n_local;
        if (!k_sor) return PCRS_OK;
        PCRS_TRY(all_gather_var(ctx, comm, rank, world, dmean, mean_full.p, counts, sizeof(float), ncclFloat32));
        result->n_answered = n_local;
        nq = n_local;
    }
    // 4. the same threshold on every rank (statistical_outlier.rs:43-62), the mask of its own answers
    float thr = 0.0f;
    int ok = 0;
    size_t kept = 0;
    PCRS_TRY(pcrs_sor_select_dev(ctx, mean_full.as<float>(), n_total, std_mul, dmean, nq, dkeep, &thr, &ok, &kept));
    if (ok) {
        result->threshold = thr;
        result->has_finite = 1;
        result->n_keep = kept;
    }
    return PCRS_OK;
} catch (...) {
    return pcrs_on_exception(ctx);
}

int pcrs_dist_icp_run(pcrs_icp *icp, void *nccl_comm, int rank, int world, size_t n_source_total,
                      int mode, uint32_t max_iterations, float tolerance,
                      float max_correspondence_distance, pcrs_icp_result *out) try {
    if (!icp || !out) return PCRS_ERR_INVALID_ARG;
    pcrs_ctx *ctx = pcrs_icp_context(icp);
    PCRS_TRY(bind(ctx));
    PCRS_TRY(need_rccl(ctx));
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "null communicator, or rank %d of %d", rank, world);
    if (mode != PCRS_ICP_POINT_TO_POINT && mode != PCRS_ICP_POINT_TO_PLANE)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "unknown ICP mode %d", mode);
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    const Rccl &r = rccl();
    size_t n_source_local = 0, n_target = 0, n_normals = 0;
    pcrs_icp_lengths(icp, &n_source_local, &n_target, &n_normals);
    std::memset(out, 0, sizeof(*out));
    out->rotation[0] = out->rotation[4] = out->rotation[8] = 1.0f;
    // icp_plane.rs:27-32 comes before the empty-cloud shortcut
    if (mode == PCRS_ICP_POINT_TO_PLANE && n_normals != n_target)
        return pcrs_set_err(ctx, PCRS_ERR_NORMALS_MISMATCH,
                            "target_normals length (%zu) does not match target cloud length (%zu)",
                            n_normals, n_target);
    std::vector<size_t> counts((size_t)world);
    for (int p = 0; p < world; ++p) {
        size_t b, e;
        shard_range(n_source_total, world, p, &b, &e);
        counts[(size_t)p] = e - b;
    }
    if (counts[(size_t)rank] != n_source_local)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "shard of %zu source points, its index range holds %zu",
                            n_source_local, counts[(size_t)rank]);
    if (n_source_total == 0 || n_target == 0) {  // icp.rs:131-139
        out->converged = (n_source_total == 0 && n_target == 0) ? 1 : 0;
        return PCRS_OK;
    }
    PCRS_TRY(pcrs_icp_shard_begin(icp, n_source_total));
    DevBuf sums, sums_all, d2_all;
    PCRS_HIP_CHECK(ctx, sums.alloc(ctx, PCRS_ICP_NSUMS * sizeof(double)));
    PCRS_HIP_CHECK(ctx, sums_all.alloc(ctx, (size_t)world * PCRS_ICP_NSUMS * sizeof(double)));
    PCRS_HIP_CHECK(ctx, d2_all.alloc(ctx, (n_source_total + 1) * sizeof(float)));
    const bool exact = tolerance > 0.0f;  // the loop is steered by the reference's own f32 rmse
    const float *d2_of[2] = {nullptr, nullptr};
    auto iteration = [&](uint32_t it, bool metrics_only) -> int {
        const float *d2 = nullptr;
        PCRS_TRY(pcrs_icp_shard_accumulate_dev(icp, mode, max_correspondence_distance, it, sums.as<double>(), &d2));
        d2_of[it & 1u] = d2;
        PCRS_NCCL(ctx, r.AllGather(sums.p, sums_all.p, PCRS_ICP_NSUMS, ncclFloat64, comm, ctx->stream));
        const float *all = nullptr;
        if (exact || metrics_only) {
            PCRS_TRY(all_gather_var(ctx, comm, rank, world, d2, d2_all.p, counts, sizeof(float), ncclFloat32));
            all = d2_all.as<float>();
        }
        return pcrs_icp_shard_update_dev(icp, mode, sums_all.as<double>(), (uint32_t)world, tolerance, all,
                                         metrics_only ? 1 : 0);
    };
    if (max_iterations == 0) PCRS_TRY(iteration(0, true));  // icp.rs:190-197
    uint32_t it = 0, last_live = 0;
    int done = 0;
    while (it < max_iterations && !done) {
        const uint32_t nb = std::min<uint32_t>(4u, max_iterations - it);  // look at the state every four
        for (uint32_t j = 0; j < nb; ++j) PCRS_TRY(iteration(it + j, false));
        it += nb;
        PCRS_TRY(pcrs_icp_shard_state(icp, out, &done, &last_live));
    }
    PCRS_TRY(pcrs_icp_shard_finish(icp));
    if (max_iterations > 0 && !exact && last_live) {
        // the reported rmse: the reference's sum over the last live iteration's correspondences
        PCRS_TRY(all_gather_var(ctx, comm, rank, world, d2_of[(last_live - 1u) & 1u], d2_all.p, counts,
                                sizeof(float), ncclFloat32));
        PCRS_TRY(pcrs_icp_shard_rmse_dev(icp, d2_all.as<float>()));
    }
    return pcrs_icp_shard_state(icp, out, &done, &last_live);
} catch (...) {
    return pcrs_on_exception(pcrs_icp_context(icp));
}

}  // extern "C"
