// Device-side building blocks shared by the brute-force and grid neighbour kernels:
// (d2, index) keys, the register top-K list, and the fused per-query consumers
// (SOR mean distance, PCA normal, list output, 1-NN output).
#pragma once

#include "pcrs_internal.h"

#pragma clang fp contract(off)

namespace pcrs {

__device__ __forceinline__ bool finite3(float x, float y, float z) {
    // (v - v) is 0 for finite v and NaN for inf/NaN
    float s = (x - x) + (y - y) + (z - z);
    return s == 0.0f;
}

// kiddo SquaredEuclidean fold for f32 x 3 axes: ((dx*dx + dy*dy) + dz*dz), no fma.
__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float px, float py, float pz) {
    float dx = qx - px;
    float dy = qy - py;
    float dz = qz - pz;
    return (dx * dx + dy * dy) + dz * dz;
}

// The same value with the x and y lanes as one packed pair (v_pk_add_f32 / v_pk_mul_f32 on
// gfx950): six VALU operations instead of eight in the inner loops of the cell walks.
typedef float pcrs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sqdist_pk(pcrs_f2 qxy, float qz, const float4 &p) {
    const pcrs_f2 pxy = {p.x, p.y};
    const pcrs_f2 d = qxy - pxy;
    const pcrs_f2 sq = d * d;
    const float dz = qz - p.z;
    return (sq.x + sq.y) + dz * dz;
}

// Four points at once, as they lie in memory (twelve consecutive floats x0 y0 z0 x1 | y1 z1 x2 y2 |
// z2 x3 y3 z3, loaded as three 16-byte vectors).  Every aligned pair of that stream is a pair of
// coordinates -- (x0 y0) (z0 x1) (y1 z1) (x2 y2) (z2 x3) (y3 z3) -- so six packed subtractions
// against the query's coordinates in the same three pairings and six packed multiplications give
// all twelve squares: 20 VALU operations for four distances instead of 26, and no register moves
// to bring a point's x and y side by side.  Each distance is still ((dx*dx + dy*dy) + dz*dz).
typedef float pcrs_f4u __attribute__((ext_vector_type(4), aligned(4)));
struct pcrs_q3 {
    pcrs_f2 xy, zx, yz;
};
__device__ __forceinline__ pcrs_q3 make_q3(float qx, float qy, float qz) {
    pcrs_q3 q;
    q.xy = pcrs_f2{qx, qy};
    q.zx = pcrs_f2{qz, qx};
    q.yz = pcrs_f2{qy, qz};
    return q;
}
__device__ __forceinline__ void sqdist_pk4(const pcrs_q3 &q, pcrs_f4u a, pcrs_f4u b, pcrs_f4u c,
                                           float (&d2)[4]) {
    pcrs_f2 t0 = q.xy - __builtin_shufflevector(a, a, 0, 1);
    pcrs_f2 t1 = q.zx - __builtin_shufflevector(a, a, 2, 3);
    pcrs_f2 t2 = q.yz - __builtin_shufflevector(b, b, 0, 1);
    pcrs_f2 t3 = q.xy - __builtin_shufflevector(b, b, 2, 3);
    pcrs_f2 t4 = q.zx - __builtin_shufflevector(c, c, 0, 1);
    pcrs_f2 t5 = q.yz - __builtin_shufflevector(c, c, 2, 3);
    t0 *= t0;  // dx0^2 dy0^2
    t1 *= t1;  // dz0^2 dx1^2
    t2 *= t2;  // dy1^2 dz1^2
    t3 *= t3;  // dx2^2 dy2^2
    t4 *= t4;  // dz2^2 dx3^2
    t5 *= t5;  // dy3^2 dz3^2
    d2[0] = (t0.x + t0.y) + t1.x;
    d2[1] = (t1.y + t2.x) + t2.y;
    d2[2] = (t3.x + t3.y) + t4.x;
    d2[3] = (t4.y + t5.x) + t5.y;
}

// Neighbour order: ascending squared distance, then DESCENDING point index.  One unsigned 64-bit
// compare gives exactly that order: d2 >= 0 so its bit pattern is monotone, and the low word
// stores (0xFFFFFFFE - index).  The empty key (d2 = +inf, low word all ones) is above every real
// key because indices stay below 0xFFFFFFFE.
__device__ __forceinline__ uint64_t make_key(float d2, uint32_t idx) {
    return ((uint64_t)__float_as_uint(d2) << 32) | (uint64_t)(0xFFFFFFFEu - idx);
}
__device__ __forceinline__ float key_d2(uint64_t key) { return __uint_as_float((uint32_t)(key >> 32)); }
__device__ __forceinline__ uint32_t key_idx(uint64_t key) { return 0xFFFFFFFEu - (uint32_t)key; }

// Ascending list of the K smallest keys seen, entirely in registers.  With PAY, every key
// carries a 32-bit payload (the grid kernels store the sorted slot of the point there, so that
// the consumers can re-read its coordinates with one coherent float4 load).
template <int K, bool PAY = false>
struct TopK {
    uint64_t key[K];
    uint32_t pay[PAY ? K : 1];
    float worst;  // d2 of key[K-1]; +inf while the list is not full

    __device__ __forceinline__ void init(bool active) {
#pragma unroll
        for (int i = 0; i < K; ++i) key[i] = PCRS_EMPTY_KEY;
        if (PAY) {
#pragma unroll
            for (int i = 0; i < K; ++i) pay[PAY ? i : 0] = 0u;
        }
        // an inactive lane can never pass `d2 <= worst` (d2 >= 0 or NaN)
        worst = active ? __builtin_inff() : -1.0f;
    }

    // Requires nk < key[K-1].
    __device__ __forceinline__ void insert(uint64_t nk, uint32_t np = 0u) {
#pragma unroll
        for (int s = K - 1; s > 0; --s) {
            const uint64_t lower = key[s - 1];
            const uint64_t cur = key[s];
            const bool below = nk < lower, here = nk < cur;
            key[s] = below ? lower : (here ? nk : cur);
            if (PAY) pay[PAY ? s : 0] = below ? pay[PAY ? s - 1 : 0] : (here ? np : pay[PAY ? s : 0]);
        }
        const bool first = nk < key[0];
        key[0] = first ? nk : key[0];
        if (PAY) pay[0] = first ? np : pay[0];
        worst = key_d2(key[K - 1]);  // the empty key decodes to +inf
    }

    __device__ __forceinline__ void offer(float d2, uint32_t idx, uint32_t np = 0u) {
        // NaN d2 (non-finite reference point) fails the compare and is dropped.
        if (d2 <= worst) {
            const uint64_t nk = make_key(d2, idx);
            if (nk < key[K - 1]) insert(nk, np);
        }
    }

    __device__ __forceinline__ uint32_t count() const {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) c += (key[i] != PCRS_EMPTY_KEY) ? 1u : 0u;
        return c;
    }
};

// Adds the per-thread counts of a 256-thread block to *total with ONE atomic per block (atomics
// on a single address serialise: a per-wave add from thousands of waves costs tens of
// microseconds).  Every thread of the block must call it.
__device__ __forceinline__ void block_count_add(unsigned local, unsigned long long *total) {
    __shared__ unsigned pcrs_block_wsum[4];
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0) pcrs_block_wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0 && total) {
        const unsigned long long s = (unsigned long long)pcrs_block_wsum[0] + pcrs_block_wsum[1] +
                                     pcrs_block_wsum[2] + pcrs_block_wsum[3];
        if (s) atomicAdd(total, s);
    }
}

// Neighbour accessors for the consumers: `for_each(c, f)` calls f(j, key_j, payload_j) for j < c in
// order; `fetch(key, payload, x, y, z)` yields the neighbour's coordinates.
template <int K>
struct RegNbrs {  // keys in registers, coordinates gathered from the original columns by index
    const uint64_t (&key)[K];
    const float *rx, *ry, *rz;
    __device__ __forceinline__ RegNbrs(const uint64_t (&k)[K], const float *x, const float *y,
                                       const float *z)
        : key(k), rx(x), ry(y), rz(z) {}
    template <class F>
    __device__ __forceinline__ void for_each(uint32_t c, F f) const {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((uint32_t)j < c) f((uint32_t)j, key[j], 0u);
    }
    __device__ __forceinline__ void fetch(uint64_t k, uint32_t, float &x, float &y, float &z) const {
        const uint32_t id = key_idx(k);
        x = rx[id];
        y = ry[id];
        z = rz[id];
    }
};

template <int K>
struct SlotNbrs {  // keys + sorted-slot payloads in registers, coordinates from the sorted array
    const uint64_t (&key)[K];
    const uint32_t (&pay)[K];
    const float4 *spts;
    __device__ __forceinline__ SlotNbrs(const uint64_t (&k)[K], const uint32_t (&p)[K],
                                        const float4 *s)
        : key(k), pay(p), spts(s) {}
    template <class F>
    __device__ __forceinline__ void for_each(uint32_t c, F f) const {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((uint32_t)j < c) f((uint32_t)j, key[j], pay[j]);
    }
    __device__ __forceinline__ void fetch(uint64_t, uint32_t slot, float &x, float &y,
                                          float &z) const {
        const float4 p = spts[slot];
        x = p.x;
        y = p.y;
        z = p.z;
    }
};

// This lane's neighbours in LDS (column threadIdx.x): lst[] holds their sorted slots in no
// particular order, ord[r] the position in lst[] of the r-th neighbour.  Keys are recomputed
// from the point (one float4 load per neighbour, four in flight).
template <int ROWS, int COLS>
struct LdsNbrs {
    const uint32_t (&lst)[ROWS][COLS];
    const uint8_t (&ord)[ROWS][COLS];
    const float4 *spts;
    float qx, qy, qz;
    __device__ __forceinline__ LdsNbrs(const uint32_t (&l)[ROWS][COLS],
                                       const uint8_t (&o)[ROWS][COLS], const float4 *s, float x,
                                       float y, float z)
        : lst(l), ord(o), spts(s), qx(x), qy(y), qz(z) {}
    __device__ __forceinline__ uint64_t key_of(const float4 &p) const {
        return make_key(sqdist(qx, qy, qz, p.x, p.y, p.z), __float_as_uint(p.w));
    }
    __device__ __forceinline__ uint32_t slot(uint32_t r) const {
        return lst[ord[r][threadIdx.x]][threadIdx.x];
    }
    template <class F>
    __device__ __forceinline__ void for_each(uint32_t c, F f) const {
        uint32_t j = 0;
        for (; j + 3 < c; j += 4) {
            const uint32_t s0 = slot(j), s1 = slot(j + 1), s2 = slot(j + 2), s3 = slot(j + 3);
            const float4 p0 = spts[s0], p1 = spts[s1], p2 = spts[s2], p3 = spts[s3];
            f(j, key_of(p0), s0);
            f(j + 1, key_of(p1), s1);
            f(j + 2, key_of(p2), s2);
            f(j + 3, key_of(p3), s3);
        }
        for (; j < c; ++j) {
            const uint32_t s0 = slot(j);
            f(j, key_of(spts[s0]), s0);
        }
    }
    __device__ __forceinline__ void fetch(uint64_t, uint32_t slot, float &x, float &y,
                                          float &z) const {
        const float4 p = spts[slot];
        x = p.x;
        y = p.y;
        z = p.z;
    }
};

struct ListNbrs {  // ascending keys of this query in memory
    const uint64_t *key;
    const float *rx, *ry, *rz;
    template <class F>
    __device__ __forceinline__ void for_each(uint32_t c, F f) const {
        for (uint32_t j = 0; j < c; ++j) f(j, key[j], 0u);
    }
    __device__ __forceinline__ void fetch(uint64_t k, uint32_t, float &x, float &y, float &z) const {
        const uint32_t id = key_idx(k);
        x = rx[id];
        y = ry[id];
        z = rz[id];
    }
};

struct StagedListNbrs {  // ascending keys of this query and their coordinates, by rank, in LDS
    const uint64_t *key;
    const float *px, *py, *pz;
    template <class F>
    __device__ __forceinline__ void for_each(uint32_t c, F f) const {
        for (uint32_t j = 0; j < c; ++j) f(j, key[j], j);
    }
    __device__ __forceinline__ void fetch(uint64_t, uint32_t rank, float &x, float &y, float &z) const {
        x = px[rank];
        y = py[rank];
        z = pz[rank];
    }
};

// crates/normals/src/estimate.rs:139-238 — smallest-|eigenvalue| eigenvector of a symmetric 3x3,
// f64 inside, written here for the device.
__device__ inline void smallest_eigvec_3x3(float f00, float f01, float f02, float f11, float f12,
                                           float f22, float &ox, float &oy, float &oz) {
    const double a00 = f00, a01 = f01, a02 = f02, a11 = f11, a12 = f12, a22 = f22;
    const double m = (a00 + a11 + a22) / 3.0;
    const double b00 = a00 - m, b11 = a11 - m, b22 = a22 - m;
    const double q = (b00 * (b11 * b22 - a12 * a12) - a01 * (a01 * b22 - a12 * a02) +
                      a02 * (a01 * a12 - b11 * a02)) /
                     2.0;
    const double p =
        (b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12)) / 6.0;
    const double pp = (p > 0.0) ? p : 0.0;
    ox = 0.0f;
    oy = 0.0f;
    oz = 1.0f;
    if (pp < 1e-30) return;
    double ratio = q / (pp * sqrt(pp));
    if (ratio < -1.0) ratio = -1.0;
    if (ratio > 1.0) ratio = 1.0;
    const double phi = acos(ratio) / 3.0;
    const double sp = sqrt(pp);
    const double kFracPi3 = 1.04719755119659774615421446109316763;
    const double eig0 = m + 2.0 * sp * cos(phi + 2.0 * kFracPi3);
    const double eig2 = m + 2.0 * sp * cos(phi);
    const double eig1 = 3.0 * m - eig0 - eig2;
    double lambda;
    if (fabs(eig0) <= fabs(eig1) && fabs(eig0) <= fabs(eig2))
        lambda = eig0;
    else if (fabs(eig1) <= fabs(eig2))
        lambda = eig1;
    else
        lambda = eig2;
    const double r00 = a00 - lambda, r11 = a11 - lambda, r22 = a22 - lambda;
    double ex = a01 * a12 - r11 * a02;
    double ey = a02 * a01 - a12 * r00;
    double ez = r00 * r11 - a01 * a01;
    double len2 = ex * ex + ey * ey + ez * ez;
    if (len2 < 1e-30) {
        ex = a01 * r22 - a12 * a02;
        ey = a02 * a02 - r22 * r00;
        ez = r00 * a12 - a01 * a02;
        len2 = ex * ex + ey * ey + ez * ez;
        if (len2 < 1e-30) {
            ex = r11 * r22 - a12 * a12;
            ey = a12 * a02 - r22 * a01;
            ez = a01 * a12 - r11 * a02;
            len2 = ex * ex + ey * ey + ez * ez;
            if (len2 < 1e-30) return;  // (0, 0, 1)
        }
    }
    const double inv = 1.0 / sqrt(len2);
    ox = (float)(ex * inv);
    oy = (float)(ey * inv);
    oz = (float)(ez * inv);
}

// estimate.rs:83-109 — eigenvector of the covariance, normalised, flipped towards the viewpoint.
__device__ __forceinline__ void normal_from_cov(float c00, float c01, float c02, float c11,
                                                float c12, float c22, float qx, float qy, float qz,
                                                const float *vp, float &nnx, float &nny,
                                                float &nnz) {
    smallest_eigvec_3x3(c00, c01, c02, c11, c12, c22, nnx, nny, nnz);
    const float len = sqrtf((nnx * nnx + nny * nny) + nnz * nnz);
    if (len > 1e-10f) {
        nnx /= len;
        nny /= len;
        nnz /= len;
    }
    const float vx = vp[0] - qx, vy = vp[1] - qy, vz = vp[2] - qz;
    const float dot = (nnx * vx + nny * vy) + nnz * vz;
    if (dot < 0.0f) {
        nnx = -nnx;
        nny = -nny;
        nnz = -nnz;
    }
}

// Per-query consumers.  `cnt` = valid neighbours available (ascending), `qvalid` = finite query,
// `qo` = output slot; neighbour coordinates come from the accessor's fetch().
// COV_ONLY promises that o.cov is set whenever o.nx is, and leaves the solver out of the kernel.
// While the normals are pending, a query's results live in one 32-byte record of o.cov
// ({c00, c01, c02, c11, c12, c22, SOR mean, unused}): the search kernels answer queries in cell
// order, so anything they write per query is scattered over the output, and a whole 32-byte
// sector per query costs a fraction of seven separate 4-byte stores.  normals_from_cov_kernel
// reads the records in index order and writes the caller's columns coalesced.
constexpr size_t PCRS_REC_FLOATS = PCRS_REC_FLOATS_HOST;
// A mean found before the record is written travels in the caller's copy of KnnOut
// (rec_mean) and leaves with the record: two 16-byte stores fill the sector at once.
__device__ __forceinline__ void store_mean(KnnOut &o, size_t qo, float mean) {
    if (o.cov)
        o.rec_mean = mean;
    else
        o.mean[qo] = mean;
}

template <bool COV_ONLY = false, class Nbrs>
__device__ __forceinline__ void consume(const Nbrs &nb, uint32_t cnt, bool qvalid, size_t qo,
                                        float qx, float qy, float qz, const KnnOut &o) {
    if (!qvalid) cnt = 0;
    float rec_mean = o.rec_mean;
    if (o.idx || o.dist || o.count) {
        const uint32_t c = cnt < o.k ? cnt : o.k;
        if (o.count) o.count[qo] = c;
        const size_t base = qo * (size_t)o.k;
        nb.for_each(c, [&](uint32_t j, uint64_t key, uint32_t) {
            if (o.idx) o.idx[base + j] = key_idx(key);
            if (o.dist) o.dist[base + j] = sqrtf(key_d2(key));  // kdtree.rs:76
        });
        for (uint32_t j = c; j < o.k; ++j) {
            if (o.idx) o.idx[base + j] = PCRS_NO_INDEX;
            if (o.dist) o.dist[base + j] = __builtin_nanf("");
        }
    }
    if (o.nn_idx) {
        uint32_t id = PCRS_NO_INDEX;
        float d2 = __builtin_inff();
        nb.for_each(cnt < 1u ? cnt : 1u, [&](uint32_t, uint64_t key, uint32_t) {
            id = key_idx(key);
            d2 = key_d2(key);
        });
        o.nn_idx[qo] = id;
        if (o.nn_d2) o.nn_d2[qo] = d2;
    }
    if (o.mean) {
        // statistical_outlier.rs:22-37
        float mean = __builtin_inff();
        if (qvalid) {
            const uint32_t kk = o.k_sor + 1u;
            const uint32_t c = cnt < kk ? cnt : kk;
            const uint32_t first = c > 1u ? 1u : 0u;
            const uint32_t nn = c - first;
            if (nn > 0u) {
                float sum = 0.0f;
                nb.for_each(c, [&](uint32_t j, uint64_t key, uint32_t) {
                    if (j >= first) sum += sqrtf(key_d2(key));
                });
                mean = sum / (float)nn;
            }
        }
        if (o.cov)
            rec_mean = mean;
        else
            o.mean[qo] = mean;
    }
    if (o.nx) {
        // estimate.rs:45-109
        const uint32_t c = cnt < o.k_nrm ? cnt : o.k_nrm;
        float nnx = 0.0f, nny = 0.0f, nnz = 1.0f;
        if (c >= 1u) {
            const float count = (float)c;
            float cx = 0.0f, cy = 0.0f, cz = 0.0f;
            nb.for_each(c, [&](uint32_t, uint64_t key, uint32_t pay) {
                float px, py, pz;
                nb.fetch(key, pay, px, py, pz);
                cx += px;
                cy += py;
                cz += pz;
            });
            cx /= count;
            cy /= count;
            cz /= count;
            float c00 = 0.0f, c01 = 0.0f, c02 = 0.0f, c11 = 0.0f, c12 = 0.0f, c22 = 0.0f;
            nb.for_each(c, [&](uint32_t, uint64_t key, uint32_t pay) {
                float px, py, pz;
                nb.fetch(key, pay, px, py, pz);
                const float dx = px - cx;
                const float dy = py - cy;
                const float dz = pz - cz;
                c00 += dx * dx;
                c01 += dx * dy;
                c02 += dx * dz;
                c11 += dy * dy;
                c12 += dy * dz;
                c22 += dz * dz;
            });
            if (COV_ONLY || o.cov) {  // the eigenvector is solved by normals_from_cov_kernel
                float4 *rec = reinterpret_cast<float4 *>(o.cov + qo * PCRS_REC_FLOATS);
                rec[0] = make_float4(c00, c01, c02, c11);
                rec[1] = make_float4(c12, c22, rec_mean, 0.0f);
                return;
            }
            if constexpr (!COV_ONLY)
                normal_from_cov(c00, c01, c02, c11, c12, c22, qx, qy, qz, o.vp, nnx, nny, nnz);
        } else if (COV_ONLY || o.cov) {
            // no neighbours (a variance is never negative)
            float4 *rec = reinterpret_cast<float4 *>(o.cov + qo * PCRS_REC_FLOATS);
            rec[0] = make_float4(-1.0f, 0.0f, 0.0f, 0.0f);
            rec[1] = make_float4(0.0f, 0.0f, rec_mean, 0.0f);
            return;
        }
        o.nx[qo] = nnx;
        o.ny[qo] = nny;
        o.nz[qo] = nnz;
    }
}

// Smallest register bucket that holds `kneed` neighbours (kneed <= PCRS_KMAX_REG).
inline int k_bucket(uint32_t kneed) {
    static const int buckets[] = {1, 2, 4, 8, 12, 16, 17, 21, 24, 32, 48, 64};
    for (int b : buckets)
        if ((uint32_t)b >= kneed) return b;
    return 0;
}

// Expands `call(K)` for the bucket chosen at run time.
#define PCRS_DISPATCH_K(kb, CALL) \
    switch (kb) {                 \
        case 1: CALL(1); break;   \
        case 2: CALL(2); break;   \
        case 4: CALL(4); break;   \
        case 8: CALL(8); break;   \
        case 12: CALL(12); break; \
        case 16: CALL(16); break; \
        case 17: CALL(17); break; \
        case 21: CALL(21); break; \
        case 24: CALL(24); break; \
        case 32: CALL(32); break; \
        case 48: CALL(48); break; \
        case 64: CALL(64); break; \
        default: break;           \
    }

}  // namespace pcrs
