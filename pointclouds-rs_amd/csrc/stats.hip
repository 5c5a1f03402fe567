// Global statistics of statistical_outlier_removal on the GPU, bit-identical to the reference's
// strictly left-to-right f32 sums (crates/filters/src/statistical_outlier.rs:43-62):
//     mu = (sum of finite means) / nf ;  var = (sum of (d - mu)^2) / nf ;  thr = mu + mul*sqrt(var)
//
// A left-to-right f32 sum is not associative, but while the running sum s stays inside one
// binade [2^e, 2^(e+1)) its ulp u = 2^(e-23) is fixed and s/u is an integer S in [2^23, 2^24).
// Adding a non-negative x = (X + f) u then gives S + X, rounded up when f > 1/2 and to even when
// f = 1/2 exactly.  So each element is a function S -> S + a[S & 1], and such functions compose
// associatively: (f then g)[p] = f[p] + g[(p + f[p]) & 1].  That turns the sequential sum into a
// scan wherever the binade does not change:
//   1. per 256-element chunk: f64 sum + finite count                      (parallel)
//   2. exclusive scan of the chunk sums -> a close estimate of s at every chunk start
//   3. per chunk: composite function under the binade of that estimate    (parallel)
//   4. one wave walks the chunks in order with the true f32 state: a chunk whose composite
//      applies (same binade going in, still inside it coming out) costs O(1); any other chunk
//      (a binade crossing, the first elements, a wrong estimate) is summed element by element
//      with real f32 adds.  The result is exact by construction; only speed depends on 1-3.
#include <cstddef>
#include <mutex>

#include "knn_common.h"

#pragma clang fp contract(off)

namespace pcrs {

constexpr int ST_CHUNK = 256;           // elements per chunk = one wave x 4
constexpr int ST_PER_LANE = ST_CHUNK / 64;
constexpr uint32_t ST_SAT = 1u << 30;   // saturation of composite increments (>= 2^24 means "left the binade")
constexpr int ST_NO_BINADE = -1000;
// The "binade" of a running sum that is still exactly zero: it stays zero over every element
// that is zero (constant means give a variance pass of nothing but zeros).
constexpr int ST_ZERO = -999;

struct StatsScalars {
    float mu;
    float thr;
    float sigma;
    uint32_t has_finite;
    unsigned long long nf;
    uint32_t max_bits;  // bit pattern of the largest element of the current pass (elements are >= 0)
    float run;          // running f32 sum between two launches of the walker
};

struct Composite {
    uint32_t a0, a1;  // increment of S for even / odd S
    int e;            // binade the composite was built for, ST_NO_BINADE = unusable
    uint32_t pad;
};

// element i of pass P: pass 1 = finite mean (else 0), pass 2 = (mean - mu)^2 of finite means,
// pass 3 = the value itself (a plain left-to-right sum of non-negative numbers; +inf stays +inf)
template <int PASS>
__device__ __forceinline__ float stat_element(const float *mean, size_t n, size_t i, float mu,
                                              bool *finite) {
    *finite = false;
    if (i >= n) return 0.0f;
    const float d = mean[i];
    if (PASS == 3) {
        *finite = true;
        return d;
    }
    if (!(d - d == 0.0f)) return 0.0f;  // inf / NaN are filtered out (:43-47)
    *finite = true;
    if (PASS == 1) return d;
    const float t = d - mu;
    return t * t;  // (d - global_mean).powi(2)
}

template <int PASS>
__global__ __launch_bounds__(256) void chunk_sums_kernel(const float *mean, size_t n,
                                                         const StatsScalars *sc, double *csum,
                                                         uint32_t *ccnt, float *cmax,
                                                         uint32_t nchunks,
                                                         const uint32_t *skip, uint32_t *zero_me,
                                                         uint32_t zero_words) {
    if (skip && *skip) return;  // the caller's loop has ended: nobody will read this sum
    // the first kernel of a run clears the run's scalars (no kernel of this launch reads them,
    // every later one writes them): a memset launch less
    if (zero_me && blockIdx.x == 0)
        for (uint32_t t = threadIdx.x; t < zero_words; t += 256u) zero_me[t] = 0u;
    const uint32_t chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= nchunks) return;
    const int lane = threadIdx.x & 63;
    const float mu = (PASS == 2) ? sc->mu : 0.0f;
    double s = 0.0;
    uint32_t c = 0;
    float mx = 0.0f;
#pragma unroll
    for (int k = 0; k < ST_PER_LANE; ++k) {
        bool fin;
        const float x = stat_element<PASS>(mean, n, (size_t)chunk * ST_CHUNK + lane * ST_PER_LANE + k, mu, &fin);
        s += (double)x;
        c += fin ? 1u : 0u;
        if (x - x == 0.0f) mx = fmaxf(mx, x);  // a squared deviation can overflow to +inf
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        c += __shfl_down(c, off, 64);
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    if (lane == 0) {
        csum[chunk] = s;
        ccnt[chunk] = c;
        cmax[chunk] = mx;
    }
}

// exclusive f64 scan of the chunk sums (in place) and the total finite count
// Exclusive f64 prefix of the chunk sums, the count of finite elements and the largest element,
// in two launches: every block scans one stretch of 4096 chunks in place and leaves the stretch's
// totals; the second launch adds the totals of the stretches before it.  A cloud of one stretch
// is finished by the first launch.
constexpr uint32_t ST_SCAN_SPAN = 4096;

__global__ __launch_bounds__(1024) void chunk_scan_local_kernel(
    double *csum, const uint32_t *ccnt, const float *cmax, uint32_t nchunks, double *ssum,
    unsigned long long *scnt, float *smax, StatsScalars *sc, int write_count,
    const uint32_t *skip) {
    if (skip && *skip) return;
    __shared__ double wsum[16];
    __shared__ unsigned long long wcnt[16];
    __shared__ float wmax[16];
    const uint32_t b = blockIdx.x * ST_SCAN_SPAN + threadIdx.x * 4u;
    double v[4];
    double s = 0.0;
    unsigned long long c = 0;
    float mx = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool in = b + j < nchunks;
        v[j] = in ? csum[b + j] : 0.0;
        s += v[j];
        c += in ? ccnt[b + j] : 0u;
        mx = fmaxf(mx, in ? cmax[b + j] : 0.0f);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_down(c, off, 64);
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) {
        wcnt[wave] = c;
        wmax[wave] = mx;
    }
    __syncthreads();
    double run = incl - s;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (b + j < nchunks) csum[b + j] = run;
        run += v[j];
    }
    if (threadIdx.x == 0) {
        double total_s = 0.0;
        unsigned long long total = 0;
        float m = 0.0f;
        for (int w = 0; w < 16; ++w) {
            total_s += wsum[w];
            total += wcnt[w];
            m = fmaxf(m, wmax[w]);
        }
        if (gridDim.x == 1) {
            sc->max_bits = __float_as_uint(m);
            if (write_count) {
                sc->nf = total;
                sc->has_finite = total ? 1u : 0u;
            }
        } else {
            ssum[blockIdx.x] = total_s;
            scnt[blockIdx.x] = total;
            smax[blockIdx.x] = m;
        }
    }
}

__global__ __launch_bounds__(1024) void chunk_scan_base_kernel(
    double *csum, uint32_t nchunks, const double *ssum, const unsigned long long *scnt,
    const float *smax, StatsScalars *sc, int write_count, const uint32_t *skip) {
    if (skip && *skip) return;
    __shared__ double wsum[16];
    __shared__ unsigned long long wcnt[16];
    __shared__ float wmax[16];
    // stretch totals before this block (block 0: all of them, for the count and the maximum)
    const uint32_t upto = blockIdx.x == 0 ? gridDim.x : blockIdx.x;
    double s = 0.0;
    unsigned long long c = 0;
    float mx = 0.0f;
    for (uint32_t i = threadIdx.x; i < upto; i += 1024u) {
        s += ssum[i];
        c += scnt[i];
        mx = fmaxf(mx, smax[i]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        c += __shfl_down(c, off, 64);
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        wsum[wave] = s;
        wcnt[wave] = c;
        wmax[wave] = mx;
    }
    __syncthreads();
    double base = 0.0;
    unsigned long long total = 0;
    float m = 0.0f;
    for (int w = 0; w < 16; ++w) {
        base += wsum[w];
        total += wcnt[w];
        m = fmaxf(m, wmax[w]);
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            sc->max_bits = __float_as_uint(m);
            if (write_count) {
                sc->nf = total;
                sc->has_finite = total ? 1u : 0u;
            }
        }
        return;
    }
    const uint32_t b = blockIdx.x * ST_SCAN_SPAN + threadIdx.x * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (b + j < nchunks) csum[b + j] += base;
}

__device__ __forceinline__ Composite compose(const Composite &f, const Composite &g) {
    // f first, then g
    Composite h;
    const uint32_t g0 = ((0u + f.a0) & 1u) ? g.a1 : g.a0;
    const uint32_t g1 = ((1u + f.a1) & 1u) ? g.a1 : g.a0;
    h.a0 = min(f.a0 + g0, ST_SAT);
    h.a1 = min(f.a1 + g1, ST_SAT);
    h.e = f.e;
    h.pad = 0;
    return h;
}

// the function S -> S + a[S & 1] of one element x >= 0 in the binade with ulp 2^(e-23):
// x / ulp = m 2^sh from the bits of x; X = its integer part, the remainder decides the rounding
__device__ __forceinline__ Composite element_function(float x, int e) {
    Composite f;
    f.e = e;
    f.pad = 0;
    if (e == ST_ZERO) {  // a sum of zero stays zero only over zeros
        f.a0 = f.a1 = x == 0.0f ? 0u : ST_SAT;
        return f;
    }
    const uint32_t bits = __float_as_uint(x);
    const uint32_t ex = bits >> 23, mant = bits & 0x7FFFFFu;
    const uint32_t m = ex ? (mant | 0x800000u) : mant;  // subnormals have no hidden bit
    const int sh = (int)(ex ? ex : 1u) - 127 - e;
    if (sh >= 0) {  // a whole number of ulps
        const uint32_t X = sh >= 7 ? ST_SAT : (m << sh);
        f.a0 = f.a1 = min(X, ST_SAT);
        return f;
    }
    const int r = -sh;
    if (r >= 25) {  // below half an ulp
        f.a0 = f.a1 = 0;
        return f;
    }
    const uint32_t X = m >> r, rem = m & ((1u << r) - 1u), half = 1u << (r - 1);
    if (rem < half) {
        f.a0 = f.a1 = X;
    } else if (rem > half) {
        f.a0 = f.a1 = X + 1u;
    } else {  // tie: round to even
        f.a0 = X + (X & 1u);          // S even: S + X odd iff X odd
        f.a1 = X + ((X & 1u) ^ 1u);   // S odd:  S + X odd iff X even
    }
    return f;
}

__device__ __forceinline__ int binade_of(float s) {
    const uint32_t b = __float_as_uint(s);
    if (b == 0u) return ST_ZERO;
    const int ex = (int)((b >> 23) & 0xFFu);
    if (ex == 0 || ex == 255 || (b >> 31)) return ST_NO_BINADE;
    const int e = ex - 127;
    if (e < -100 || e > 100) return ST_NO_BINADE;
    return e;
}

// Composites of the chunks [chunk_begin, chunk_end) for W binades each (comp[chunk * W + j]
// records the binade it was built for).  W = 1 (short sums): the binade predicted from
// the f64 prefix.  W = 4 (long sums, walked in sections): a sequential f32 sum of millions of
// elements drifts away from the exact prefix — it runs ahead while the elements round up to a
// whole ulp, falls behind and finally stalls once they are below half an ulp — so the window is
// anchored on what is known for certain: the sum never decreases (at least the binade of the
// true running sum at the start of the section, read from sc->run) and never passes the binade in
// which the largest element is below half an ulp.
template <int PASS, int W>
__global__ __launch_bounds__(256) void chunk_composite_kernel(const float *mean, size_t n,
                                                              const StatsScalars *sc,
                                                              const double *cpre, Composite *comp,
                                                              uint32_t chunk_begin,
                                                              uint32_t chunk_end,
                                                              const uint32_t *skip) {
    if (skip && *skip) return;
    const uint32_t chunk = chunk_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= chunk_end) return;
    const int lane = threadIdx.x & 63;
    const float mu = (PASS == 2) ? sc->mu : 0.0f;
    const int pred = binade_of((float)cpre[chunk]);
    int binades[W];
#pragma unroll
    for (int j = 0; j < W; ++j)
        binades[j] = pred == ST_NO_BINADE ? ST_NO_BINADE
                                          : (pred == ST_ZERO ? (j == 0 ? ST_ZERO : ST_NO_BINADE) : pred - j);
    if (W == 4 && pred != ST_NO_BINADE && pred != ST_ZERO) {
        const int emax = binade_of(__uint_as_float(sc->max_bits));
        const int cap = emax != ST_NO_BINADE ? emax + 25 : 100;
        const int top = min(pred + 1, cap);
        const int floor_e = chunk_begin ? binade_of(sc->run) : ST_NO_BINADE;
        if (floor_e == ST_NO_BINADE) {  // first section: the prefix is still a close estimate
            for (int j = 0; j < W; ++j) binades[j] = top - j;
        } else if (top <= floor_e + 3) {  // the sum never decreases: nothing below floor_e
            for (int j = 0; j < W; ++j) binades[j] = floor_e + 3 - j;
        } else {  // the prefix is far ahead (the f32 sum may have stalled): cover both ends
            binades[0] = top;
            binades[1] = top - 1;
            binades[2] = floor_e + 1;
            binades[3] = floor_e;
        }
    }
    float x[ST_PER_LANE];
#pragma unroll
    for (int k = 0; k < ST_PER_LANE; ++k) {
        bool fin;
        x[k] = stat_element<PASS>(mean, n, (size_t)chunk * ST_CHUNK + lane * ST_PER_LANE + k, mu, &fin);
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
        int ej = binades[j];
        if (ej != ST_NO_BINADE && ej != ST_ZERO && (ej < -100 || ej > 100)) ej = ST_NO_BINADE;
        Composite f;
        f.a0 = f.a1 = 0;
        f.e = ej;
        f.pad = 0;
        if (ej != ST_NO_BINADE) {
#pragma unroll
            for (int k = 0; k < ST_PER_LANE; ++k) f = compose(f, element_function(x[k], ej));
            // ordered tree: lane l absorbs lane l + off (the later elements)
            for (int off = 1; off < 64; off <<= 1) {
                Composite g;
                g.a0 = __shfl_down(f.a0, off, 64);
                g.a1 = __shfl_down(f.a1, off, 64);
                g.e = ej;
                g.pad = 0;
                if ((lane & (2 * off - 1)) == 0) f = compose(f, g);
            }
        }
        if (lane == 0) comp[(size_t)chunk * W + j] = f;
    }
}

// Composite of a whole batch of 256 chunks (the unit of the walker's scan), for the binades of
// the batch's first chunk: where the running sum stays in its binade over all 65536 elements —
// nearly everywhere — the walker applies this one function and never looks at the chunks.
// One wave per batch; lane l folds the chunks 4l .. 4l+3, an ordered tree folds the lanes.
template <int W>
__global__ __launch_bounds__(64) void batch_composite_kernel(const Composite *comp,
                                                             Composite *bcomp,
                                                             uint32_t chunk_begin,
                                                             uint32_t chunk_end,
                                                             const uint32_t *skip) {
    if (skip && *skip) return;
    constexpr int G = 4;
    const int lane = threadIdx.x;
    const uint32_t base = chunk_begin + blockIdx.x * (64u * G);
    if (base >= chunk_end) return;
    Composite mine[G][W];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const uint32_t ci = base + (uint32_t)lane * G + k;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            mine[k][j].a0 = mine[k][j].a1 = 0;
            mine[k][j].e = ST_NO_BINADE;
            mine[k][j].pad = 0;
            if (ci < chunk_end) mine[k][j] = comp[(size_t)ci * W + j];
        }
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int ej = __builtin_amdgcn_readfirstlane(mine[0][j].e);
        Composite f;
        f.a0 = f.a1 = 0;
        f.e = ej;
        f.pad = 0;
        if (ej != ST_NO_BINADE) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const uint32_t ci = base + (uint32_t)lane * G + k;
                Composite u;
                u.e = ej;
                u.pad = 0;
                u.a0 = u.a1 = ci < chunk_end ? ST_SAT : 0u;  // past the end: identity
                if (ci < chunk_end) {
#pragma unroll
                    for (int q = 0; q < W; ++q)
                        if (mine[k][q].e == ej) {
                            u.a0 = mine[k][q].a0;
                            u.a1 = mine[k][q].a1;
                        }
                }
                f = compose(f, u);
            }
            for (int off = 1; off < 64; off <<= 1) {
                Composite g;
                g.a0 = __shfl_down(f.a0, off, 64);
                g.a1 = __shfl_down(f.a1, off, 64);
                g.e = ej;
                g.pad = 0;
                if ((lane & (2 * off - 1)) == 0) f = compose(f, g);
            }
        }
        if (lane == 0) bcomp[(size_t)(base / (64u * G)) * W + j] = f;
    }
}

// One wave walks the chunks in order carrying the exact f32 running sum.  256 chunk composites
// are taken at a time, four consecutive ones per lane: each lane composes its own four, an
// inclusive scan of function composition over the lanes then gives the state after every chunk
// of the batch at once.  The sum only grows (the elements are >= 0), so the first chunk that
// leaves the binade (or was built for another one) is the first whose state fails the test: it
// is found with a ballot, summed element by element with real f32 adds, and the scan restarts
// behind it.
// STAGE (sums of up to ST_STAGE_CHUNKS chunks, one binade per chunk): the walker is one wave, so
// every composite or element it has to fetch in the middle of the walk costs a full, unhidden
// memory round trip -- two per binade crossing, some twenty crossings per sum.  It therefore
// starts by copying all chunk composites into LDS and by fetching the elements of the chunks in
// which the f64 prefix says a crossing falls (and their neighbours), everything in flight
// together; the walk itself then runs out of LDS.  A chunk that was not foreseen is still read
// from memory, so this changes when data arrives, never what is computed.
constexpr uint32_t ST_STAGE_CHUNKS = 4096;
constexpr int ST_PRE = 48;  // chunks whose elements are fetched ahead

constexpr int ST_STAGE_THREADS = 1024;
// The staged walker keeps all of this in LDS at once: it is sized for the 160 KB of a gfx950
// compute unit (the 64 KB of earlier gfx9 parts cannot hold it; a build for one of those must
// lower ST_STAGE_CHUNKS, which only moves more sums to the unstaged walker).
static_assert(ST_STAGE_CHUNKS * sizeof(Composite) + (size_t)ST_PRE * 64 * sizeof(float4) +
                      64 * sizeof(float4) + 64 * sizeof(uint32_t) + 64 <=
                  160 * 1024,
              "staged statistics walker exceeds the LDS of a gfx950 compute unit");

template <int PASS, int W, bool STAGE>
__global__ __launch_bounds__(STAGE ? ST_STAGE_THREADS : 64) void stats_walk_kernel(const float *mean, size_t n,
                                                        StatsScalars *sc, const Composite *comp,
                                                        const Composite *bcomp,
                                                        uint32_t chunk_begin, uint32_t nchunks,
                                                        bool last, float std_mul,
                                                        const double *cpre,
                                                        const uint32_t *skip) {
    if (skip && *skip) return;
    // walks the chunks [chunk_begin, nchunks); the state crosses launches in sc->run
    constexpr int G = 4;
    constexpr uint32_t BATCH = 64u * G;
    static_assert(ST_PER_LANE == 4, "the element-wise path moves one float4 per lane");
    static_assert(!STAGE || W == 1, "staging holds one composite per chunk");
    __shared__ float4 elems[64];
    __shared__ Composite scomp[STAGE ? ST_STAGE_CHUNKS : 1];
    __shared__ float4 pre_elems[STAGE ? ST_PRE : 1][64];
    __shared__ uint32_t pre_chunk[STAGE ? 64 : 1];
    const int lane = threadIdx.x;
    const float mu = (PASS == 2) ? sc->mu : 0.0f;
    float s = chunk_begin ? sc->run : 0.0f;  // uniform across the wave
#ifdef PCRS_STATS_TRACE
    const long long tr_start = wall_clock64();
    long long tr_staged = tr_start, tr_elem = 0, tr_scan = 0;
    unsigned tr_nelem = 0, tr_nscan = 0, tr_nwhole = 0;
#endif
    if constexpr (STAGE) {
        // Staging is done by the whole block (ST_STAGE_THREADS threads: every load of a phase in
        // flight at once); the walk that follows is the first wave's alone.
        __shared__ uint32_t npre_s;
        const uint32_t tid = threadIdx.x, nth = blockDim.x;
        const uint32_t nloc = nchunks - chunk_begin;
        // composites and the binade of every chunk's f64 prefix (kept in the composite's spare word)
        for (uint32_t i = tid; i < nloc; i += nth) {
            Composite c = comp[chunk_begin + i];
            c.pad = (uint32_t)binade_of((float)cpre[chunk_begin + i]);
            scomp[i] = c;
        }
        if (tid < 64u) pre_chunk[tid] = 0xFFFFFFFFu;
        if (tid == 0) npre_s = 0u;
        __syncthreads();
        {
            // every wave of the block looks at its share of the chunks (which of the predicted
            // chunks get a place when there are more than ST_PRE of them does not matter: a chunk
            // that was not fetched ahead is read when the walk reaches it)
            const uint32_t ln = tid & 63u;
            const unsigned long long lt = (1ull << ln) - 1ull;
            for (uint32_t b0 = (tid >> 6) * 64u; b0 < nloc; b0 += nth) {
                const uint32_t c = b0 + ln;
                bool cross = false;
                if (c < nloc) {
                    // a crossing inside chunk c, or inside a neighbour (the f32 sum drifts off the
                    // prefix)
                    const uint32_t lo = c ? c - 1u : 0u, hi = c + 2u < nloc ? c + 2u : nloc - 1u;
                    const int e0 = (int)scomp[lo].pad, e1 = (int)scomp[hi].pad;
                    cross = e0 != e1 || e0 == ST_NO_BINADE;
                }
                const unsigned long long m = __ballot(cross);
                if (m) {
                    uint32_t first = 0;
                    if (ln == 0) first = atomicAdd(&npre_s, (uint32_t)__popcll(m));
                    first = __shfl(first, 0, 64);
                    if (cross) {
                        const uint32_t at = first + (uint32_t)__popcll(m & lt);
                        if (at < (uint32_t)ST_PRE) pre_chunk[at] = chunk_begin + c;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t npre = npre_s < (uint32_t)ST_PRE ? npre_s : (uint32_t)ST_PRE;
        for (uint32_t t = tid; t < npre * 64u; t += nth) {
            const uint32_t p = t >> 6, l = t & 63u;
            bool fin;
            const size_t e0 = (size_t)pre_chunk[p] * ST_CHUNK + l * ST_PER_LANE;
            float4 v;
            v.x = stat_element<PASS>(mean, n, e0 + 0, mu, &fin);
            v.y = stat_element<PASS>(mean, n, e0 + 1, mu, &fin);
            v.z = stat_element<PASS>(mean, n, e0 + 2, mu, &fin);
            v.w = stat_element<PASS>(mean, n, e0 + 3, mu, &fin);
            pre_elems[p][l] = v;
        }
        __syncthreads();
        if (tid >= 64u) return;
    }
#ifdef PCRS_STATS_TRACE
    tr_staged = wall_clock64();
#endif
    // the composites of the next batch are requested while the current one is being scanned
    auto load_batch = [&](uint32_t base, Composite (&dst)[G][W]) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const uint32_t ci = base + (uint32_t)lane * G + k;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                dst[k][j].a0 = dst[k][j].a1 = 0;
                dst[k][j].e = ST_NO_BINADE;
                dst[k][j].pad = 0;
                if (ci < nchunks) {
                    if constexpr (STAGE)
                        dst[k][j] = scomp[ci - chunk_begin];
                    else
                        dst[k][j] = comp[(size_t)ci * W + j];
                }
            }
        }
    };
    auto load_elements = [&](uint32_t chunk) {  // this lane's four elements of a chunk
        bool fin;
        const size_t e0 = (size_t)chunk * ST_CHUNK + lane * ST_PER_LANE;
        float4 v;
        v.x = stat_element<PASS>(mean, n, e0 + 0, mu, &fin);
        v.y = stat_element<PASS>(mean, n, e0 + 1, mu, &fin);
        v.z = stat_element<PASS>(mean, n, e0 + 2, mu, &fin);
        v.w = stat_element<PASS>(mean, n, e0 + 3, mu, &fin);
        return v;
    };
    // The batch composites of the whole section are requested up front, batch 64 t + l in lane l
    // (a section has at most 64 * BPL batches; further ones are walked chunk by chunk); the chunk
    // composites are only loaded where a batch cannot be taken whole.
    constexpr int BPL = 2;
    Composite whole_of[BPL][W];
#pragma unroll
    for (int t = 0; t < BPL; ++t) {
        const uint32_t bbase = chunk_begin + (uint32_t)(t * 64 + lane) * BATCH;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            whole_of[t][j].a0 = whole_of[t][j].a1 = ST_SAT;
            whole_of[t][j].e = ST_NO_BINADE;
            whole_of[t][j].pad = 0;
            if (bbase < nchunks) whole_of[t][j] = bcomp[(size_t)(bbase / BATCH) * W + j];
        }
    }
    for (uint32_t base = chunk_begin; base < nchunks; base += BATCH) {
        const uint32_t bi = (base - chunk_begin) / BATCH;  // uniform
        Composite whole[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            whole[j].a0 = whole[j].a1 = ST_SAT;
            whole[j].e = ST_NO_BINADE;
        }
#pragma unroll
        for (int t = 0; t < BPL; ++t)
            if ((int)(bi >> 6) == t) {
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const int src = (int)(bi & 63u);
                    whole[j].a0 = __builtin_amdgcn_readlane(whole_of[t][j].a0, src);
                    whole[j].a1 = __builtin_amdgcn_readlane(whole_of[t][j].a1, src);
                    whole[j].e = __builtin_amdgcn_readlane(whole_of[t][j].e, src);
                }
            }
        {
            const uint32_t bits = __builtin_amdgcn_readfirstlane(__float_as_uint(s));
            const int e_now = binade_of(__uint_as_float(bits));
            const uint32_t S = (bits & 0x7FFFFFu) | 0x800000u;
            uint32_t inc = ST_SAT;
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (e_now != ST_NO_BINADE && whole[j].e == e_now)
                    inc = (S & 1u) ? whole[j].a1 : whole[j].a0;
            // the sum only grows: in the binade after the batch = in the binade throughout
            if (inc < ST_SAT && S + inc < (1u << 24)) {
                s = __uint_as_float((bits & 0xFF800000u) | ((S + inc) & 0x7FFFFFu));
#ifdef PCRS_STATS_TRACE
                ++tr_nwhole;
#endif
                continue;
            }
        }
        Composite mine[G][W];
        load_batch(base, mine);
        const uint32_t lim = min(BATCH, nchunks - base);
        uint32_t pos = 0;
        while (pos < lim) {
#ifdef PCRS_STATS_TRACE
            const long long tr_a = wall_clock64();
            ++tr_nscan;
#endif
            const uint32_t bits = __builtin_amdgcn_readfirstlane(__float_as_uint(s));
            const int e_now = binade_of(__uint_as_float(bits));
            const uint32_t S = (bits & 0x7FFFFFu) | 0x800000u;
            // composites of the chunks already consumed act as the identity; unusable ones
            // saturate so that everything from them on reads as "left the binade"
            Composite p[G];  // p[k] = this lane's chunks 0..k composed
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const uint32_t ci = (uint32_t)lane * G + k;
                Composite u;
                u.e = e_now;
                u.pad = 0;
                u.a0 = u.a1 = ST_SAT;
                if (ci < pos) {
                    u.a0 = u.a1 = 0;
                } else if (e_now != ST_NO_BINADE && ci < lim) {
#pragma unroll
                    for (int j = 0; j < W; ++j)
                        if (mine[k][j].e == e_now) {
                            u.a0 = mine[k][j].a0;
                            u.a1 = mine[k][j].a1;
                        }
                }
                p[k] = (k == 0) ? u : compose(p[k - 1], u);
            }
            // inclusive scan over the lanes with data-parallel-primitive moves: lanes a move does
            // not reach receive {0, 0}, the identity of the composition
            Composite f = p[G - 1];
#define PCRS_SCAN_STEP(CTRL, ROWS)                                                          \
    {                                                                                       \
        Composite g;                                                                        \
        g.a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.a0, CTRL, ROWS, 0xf, false); \
        g.a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.a1, CTRL, ROWS, 0xf, false); \
        g.e = e_now;                                                                        \
        g.pad = 0;                                                                          \
        f = compose(g, f);                                                                  \
    }
            PCRS_SCAN_STEP(0x111, 0xf)  // row_shr:1
            PCRS_SCAN_STEP(0x112, 0xf)  // row_shr:2
            PCRS_SCAN_STEP(0x114, 0xf)  // row_shr:4
            PCRS_SCAN_STEP(0x118, 0xf)  // row_shr:8
            PCRS_SCAN_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
            PCRS_SCAN_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef PCRS_SCAN_STEP
            Composite ex;  // everything before this lane's chunks (wave_shr:1)
            ex.a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.a0, 0x138, 0xf, 0xf, false);
            ex.a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.a1, 0x138, 0xf, 0xf, false);
            ex.e = e_now;
            ex.pad = 0;
            uint32_t S_after[G];
            int first_k = G;  // first of this lane's chunks whose state left the binade
#pragma unroll
            for (int k = G - 1; k >= 0; --k) {
                const Composite h = compose(ex, p[k]);
                S_after[k] = S + ((S & 1u) ? h.a1 : h.a0);
                if (!(S_after[k] < (1u << 24))) first_k = k;
            }
            const unsigned long long bad = __ballot(first_k < G);
            uint32_t first_bad = lim;
            if (bad) {
                const int bl = __builtin_ctzll(bad);
                const uint32_t fb =
                    (uint32_t)bl * G + (uint32_t)__builtin_amdgcn_readlane(first_k, bl);
                if (fb < first_bad) first_bad = fb;
            }
            if (first_bad > pos) {
                const uint32_t j = first_bad - 1, kk = j % G;
                const uint32_t sel = kk == 0 ? S_after[0] : kk == 1 ? S_after[1]
                                   : kk == 2 ? S_after[2] : S_after[3];
                const uint32_t Sj = __builtin_amdgcn_readlane(sel, j / G);
                s = __uint_as_float((bits & 0xFF800000u) | (Sj & 0x7FFFFFu));
            }
#ifdef PCRS_STATS_TRACE
            const long long tr_b = wall_clock64();
            tr_scan += tr_b - tr_a;
#endif
            if (first_bad < lim) {
#ifdef PCRS_STATS_TRACE
                ++tr_nelem;
#endif
                // element-by-element f32 adds over this chunk (lane-major element order)
                const uint32_t bad_chunk = base + first_bad;
                float4 mine4;
                bool staged = false;
                if constexpr (STAGE) {
                    const unsigned long long hit = __ballot(pre_chunk[lane] == bad_chunk);
                    if (hit) {
                        mine4 = pre_elems[__builtin_ctzll(hit)][lane];
                        staged = true;
                    }
                }
                if (!staged) mine4 = load_elements(bad_chunk);
                // through shared memory: every lane reads all 256 values back (broadcast reads)
                // and runs the same chain of adds
                elems[lane] = mine4;
                __builtin_amdgcn_wave_barrier();  // one wave: its LDS accesses complete in order
#pragma unroll 16
                for (int l = 0; l < 64; ++l) {
                    const float4 x = elems[l];
                    s = s + x.x;
                    s = s + x.y;
                    s = s + x.z;
                    s = s + x.w;
                }
                __builtin_amdgcn_wave_barrier();
            }
#ifdef PCRS_STATS_TRACE
            tr_elem += wall_clock64() - tr_b;
#endif
            pos = first_bad + 1;
        }
    }
#ifdef PCRS_STATS_TRACE
    if (lane == 0)
        printf("walk pass %d: staging %lld, walk %lld ticks (100 MHz); %u whole batches, %u scans (%lld ticks), "
               "%u element-wise chunks (%lld ticks)\n", PASS, tr_staged - tr_start, wall_clock64() - tr_staged,
               tr_nwhole, tr_nscan, tr_scan, tr_nelem, tr_elem);
#endif
    if (lane == 0) sc->run = s;
    if (lane == 0 && last) {
        const float nn = (float)sc->nf;  // finite_dists.len() as f32
        if (PASS == 1) {
            sc->mu = s / nn;
        } else if (PASS == 2) {
            const float var = s / nn;
            const float sd = sqrtf(var);
            sc->sigma = sd;
            sc->thr = sc->mu + std_mul * sd;
        }
    }
}

// keep[i] = mean[i] <= thr with the threshold read from device memory; four means per thread
// where the pointers allow it, one atomic per block for the count
__global__ __launch_bounds__(256) void mask_dev_thr_kernel(const float *mean, size_t n,
                                                           const StatsScalars *sc, uint8_t *keep,
                                                           unsigned long long *count) {
    const bool any = sc->has_finite != 0;
    const float thr = any ? sc->thr : -__builtin_inff();
    unsigned local = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t done = 0;
    if (((reinterpret_cast<uintptr_t>(mean) & 15u) == 0) &&
        ((reinterpret_cast<uintptr_t>(keep) & 3u) == 0)) {
        const float4 *mean4 = reinterpret_cast<const float4 *>(mean);
        uint32_t *keep4 = reinterpret_cast<uint32_t *>(keep);
        const size_t n4 = n / 4;
        for (size_t i = tid; i < n4; i += stride) {
            const float4 m = mean4[i];
            const uint32_t k0 = any && m.x <= thr, k1 = any && m.y <= thr;
            const uint32_t k2 = any && m.z <= thr, k3 = any && m.w <= thr;
            keep4[i] = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24);
            local += k0 + k1 + k2 + k3;
        }
        done = n4 * 4;
    }
    for (size_t i = done + tid; i < n; i += stride) {
        const bool k = any && (mean[i] <= thr);
        keep[i] = k ? 1 : 0;
        local += k ? 1u : 0u;
    }
    if (!count) return;
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    __shared__ unsigned wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long s = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (s) atomicAdd(count, s);
    }
}

}  // namespace pcrs

using namespace pcrs;

static int stats_window(size_t n);

// Composites + walk of one pass.  Short sums: one launch each.  Long sums: sections of 32768
// chunks (8 M elements), the composites of a section built once the true running sum at its start
// is known (everything is queued on the stream; nothing comes back to the host).
template <int PASS>
static void run_pass(pcrs_ctx *ctx, int W, const float *dmean, size_t n, StatsScalars *sc,
                     const double *csum, Composite *comp, Composite *bcomp, uint32_t nchunks,
                     float std_mul, const uint32_t *skip) {
    constexpr uint32_t kBatch = 256;  // chunks per batch composite = per scan of the walker
    if (W == 1) {
        hipLaunchKernelGGL((chunk_composite_kernel<PASS, 1>), dim3((nchunks + 3) / 4), dim3(256), 0,
                           ctx->stream, dmean, n, sc, csum, comp, 0u, nchunks, skip);
        hipLaunchKernelGGL((batch_composite_kernel<1>), dim3((nchunks + kBatch - 1) / kBatch),
                           dim3(64), 0, ctx->stream, comp, bcomp, 0u, nchunks, skip);
        if (nchunks <= ST_STAGE_CHUNKS)
            hipLaunchKernelGGL((stats_walk_kernel<PASS, 1, true>), dim3(1), dim3(ST_STAGE_THREADS), 0, ctx->stream,
                               dmean, n, sc, comp, bcomp, 0u, nchunks, true, std_mul, csum, skip);
        else
            hipLaunchKernelGGL((stats_walk_kernel<PASS, 1, false>), dim3(1), dim3(64), 0, ctx->stream,
                               dmean, n, sc, comp, bcomp, 0u, nchunks, true, std_mul, csum, skip);
        return;
    }
    // Before element i the f32 sum s and the exact prefix p obey p / (1 + i 2^-24) <= s <=
    // p / (1 - i 2^-24) (every add is off by at most half an ulp of s), so over the first 2^23
    // elements s stays within one binade of p and needs no anchor: one long first section.
    // Later sections are as long as the first (the most batch composites one walk launch holds in
    // registers).  A section's window reaches three binades above the sum at its start; a sum of
    // like-sized elements that already holds 2^23 of them does not outrun that within 2^23 more,
    // and where one does (sorted input), the chunks outside the window are added element by
    // element, as with any section length.  Every section costs three dependent launches: 2.9
    // instead of 3.8 ms per 100 M elements against sections of 8192 chunks.
    constexpr uint32_t kFirst = (1u << 23) / ST_CHUNK, kSection = kFirst;
    static_assert(kSection % kBatch == 0 && kFirst % kBatch == 0, "sections start on a batch");
    for (uint32_t c0 = 0, c1; c0 < nchunks; c0 = c1) {
        const uint32_t len = c0 == 0 ? kFirst : kSection;
        c1 = c0 + len < nchunks ? c0 + len : nchunks;
        hipLaunchKernelGGL((chunk_composite_kernel<PASS, 4>), dim3((c1 - c0 + 3) / 4), dim3(256), 0,
                           ctx->stream, dmean, n, sc, csum, comp, c0, c1, skip);
        hipLaunchKernelGGL((batch_composite_kernel<4>), dim3((c1 - c0 + kBatch - 1) / kBatch),
                           dim3(64), 0, ctx->stream, comp, bcomp, c0, c1, skip);
        hipLaunchKernelGGL((stats_walk_kernel<PASS, 4, false>), dim3(1), dim3(64), 0, ctx->stream,
                           dmean, n, sc, comp, bcomp, c0, c1, c1 == nchunks, std_mul, csum, skip);
    }
}

// The work space of one statistics run, carved out of one allocation.
struct StatsWork {
    uint32_t nchunks, nspans;
    int W;
    double *csum;
    Composite *comp, *bcomp;
    uint32_t *ccnt;
    float *cmax;
    double *ssum;
    unsigned long long *scnt;
    float *smax;
};

static StatsWork carve_work(size_t n, void *d_work) {
    StatsWork k;
    k.nchunks = (uint32_t)((n + ST_CHUNK - 1) / ST_CHUNK);
    k.W = stats_window(n);
    char *w = reinterpret_cast<char *>(d_work);
    k.csum = reinterpret_cast<double *>(w);
    k.comp = reinterpret_cast<Composite *>(w + (size_t)k.nchunks * sizeof(double));
    char *after = w + (size_t)k.nchunks * (sizeof(double) + (size_t)k.W * sizeof(Composite));
    k.ccnt = reinterpret_cast<uint32_t *>(after);
    k.cmax = reinterpret_cast<float *>(after + (size_t)k.nchunks * sizeof(uint32_t));
    k.nspans = (k.nchunks + ST_SCAN_SPAN - 1) / ST_SCAN_SPAN;
    uintptr_t sp = reinterpret_cast<uintptr_t>(k.cmax + k.nchunks);
    sp = (sp + 7u) & ~(uintptr_t)7u;
    k.ssum = reinterpret_cast<double *>(sp);
    k.scnt = reinterpret_cast<unsigned long long *>(k.ssum + k.nspans);
    k.smax = reinterpret_cast<float *>(k.scnt + k.nspans);
    uintptr_t bp = reinterpret_cast<uintptr_t>(k.smax + k.nspans);
    bp = (bp + 15u) & ~(uintptr_t)15u;
    k.bcomp = reinterpret_cast<Composite *>(bp);
    return k;
}

// chunk sums -> f64 prefix -> composites -> walk, for one pass
template <int PASS>
static void run_whole_pass(pcrs_ctx *ctx, const StatsWork &k, const float *v, size_t n,
                           StatsScalars *sc, int write_count, float std_mul,
                           const uint32_t *skip = nullptr, uint32_t zero_words = 0) {
    hipLaunchKernelGGL(chunk_sums_kernel<PASS>, dim3((k.nchunks + 3) / 4), dim3(256), 0, ctx->stream,
                       v, n, sc, k.csum, k.ccnt, k.cmax, k.nchunks, skip,
                       zero_words ? reinterpret_cast<uint32_t *>(sc) : nullptr, zero_words);
    hipLaunchKernelGGL(chunk_scan_local_kernel, dim3(k.nspans), dim3(1024), 0, ctx->stream, k.csum,
                       k.ccnt, k.cmax, k.nchunks, k.ssum, k.scnt, k.smax, sc, write_count, skip);
    if (k.nspans > 1)
        hipLaunchKernelGGL(chunk_scan_base_kernel, dim3(k.nspans), dim3(1024), 0, ctx->stream,
                           k.csum, k.nchunks, k.ssum, k.scnt, k.smax, sc, write_count, skip);
    run_pass<PASS>(ctx, k.W, v, n, sc, k.csum, k.comp, k.bcomp, k.nchunks, std_mul, skip);
}

// Runs both passes; leaves {mu, thr, sigma, has_finite, nf} in *d_scalars (device).  No host sync.
int pcrs_launch_sor_stats(pcrs_ctx *ctx, const float *dmean, size_t n, float std_mul,
                          void *d_scalars, void *d_work, size_t work_bytes, size_t clear_bytes) {
    if (work_bytes < pcrs_sor_stats_workspace(n))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "stats workspace too small");
    StatsScalars *sc = reinterpret_cast<StatsScalars *>(d_scalars);
    // clear_bytes from d_scalars on start at zero (at least the scalars themselves): cleared by
    // the first kernel of the run, or by a memset when there is no kernel to run
    if (clear_bytes < sizeof(StatsScalars)) clear_bytes = sizeof(StatsScalars);
    if (n == 0) {
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(sc, 0, clear_bytes, ctx->stream));
        return PCRS_OK;
    }
    const StatsWork k = carve_work(n, d_work);
    run_whole_pass<1>(ctx, k, dmean, n, sc, 1, std_mul, nullptr, (uint32_t)((clear_bytes + 3) / 4));  // mean
    run_whole_pass<2>(ctx, k, dmean, n, sc, 0, std_mul);  // variance
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// The left-to-right f32 sum of n non-negative values (what `iter().sum::<f32>()` gives), left in
// the `run` field of *d_scalars; pcrs_seq_sum_result_offset() tells where that float lies.
int pcrs_launch_seq_sum(pcrs_ctx *ctx, const float *values, size_t n, void *d_scalars, void *d_work,
                        size_t work_bytes, const uint32_t *skip) {
    if (work_bytes < pcrs_sor_stats_workspace(n))
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "stats workspace too small");
    StatsScalars *sc = reinterpret_cast<StatsScalars *>(d_scalars);
    if (n == 0) {
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(sc, 0, sizeof(StatsScalars), ctx->stream));
        return PCRS_OK;
    }
    const StatsWork k = carve_work(n, d_work);
    run_whole_pass<3>(ctx, k, values, n, sc, 1, 0.0f, skip, (uint32_t)(sizeof(StatsScalars) / 4));
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

size_t pcrs_seq_sum_result_offset() { return offsetof(StatsScalars, run); }

// Binades per chunk composite.  After i adds the f32 sum is within a factor 1 -+ i 2^-24 of the
// exact prefix; elements far below half an ulp (squared deviations) all round the same way, so
// the bound is reached.  Up to 2^20 elements that is 6 %: one binade, predicted from the prefix,
// with a few element-wise chunks around every crossing.  Longer sums take a window of four
// (PCRS_STATS_WINDOW=1|4 overrides the choice, for experiments).
static int stats_window(size_t n) {
    static const char *force = getenv("PCRS_STATS_WINDOW");
    if (force && (force[0] == '1' || force[0] == '4')) return force[0] - '0';
    return n > (size_t)1 << 20 ? 4 : 1;
}

size_t pcrs_sor_stats_workspace(size_t n) {
    const size_t nchunks = (n + ST_CHUNK - 1) / ST_CHUNK + 1;
    return nchunks * (sizeof(double) + (size_t)stats_window(n) * sizeof(Composite) +
                      sizeof(uint32_t) + sizeof(float)) +
           (nchunks / ST_SCAN_SPAN + 2) * (sizeof(double) + sizeof(unsigned long long) + sizeof(float)) +
           (nchunks / 256 + 2) * (size_t)stats_window(n) * sizeof(Composite) + 160;
}

size_t pcrs_sor_stats_scalars_bytes() { return sizeof(StatsScalars); }

int pcrs_launch_mask_dev_thr(pcrs_ctx *ctx, const float *mean, size_t n, const void *d_scalars,
                             uint8_t *keep, unsigned long long *d_count, bool count_is_clear) {
    if (d_count && !count_is_clear)
        PCRS_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, sizeof(*d_count), ctx->stream));
    if (n == 0) return PCRS_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > (size_t)ctx->num_cus) blocks = (size_t)ctx->num_cus;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(mask_dev_thr_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, mean,
                       n, reinterpret_cast<const StatsScalars *>(d_scalars), keep, d_count);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// ---------------------------------------------------------------------------------------------
// Work space for the entry point that must not allocate (pcrs_sor_select_static_dev): a static
// device array instead of the context's grown-on-demand scratch.  Every GPU the library runs on
// holds its own copy at its own address; the name below is only a host-side handle, so the
// address is resolved once per device and host code uses that pointer from then on.
static __device__ __align__(256) unsigned char scratch[PCRS_SOR_SELECT_STATIC_BYTES];

namespace {
constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices];  // per device, nullptr = not resolved yet
std::mutex g_scratch_mutex;
}  // namespace

int pcrs_static_scratch(pcrs_ctx *ctx, void **ptr, size_t *bytes) {
    if (ctx->device < 0 || ctx->device >= kMaxDevices)
        return pcrs_set_err(ctx, PCRS_ERR_INVALID_ARG, "device %d has no static scratch slot",
                            ctx->device);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[ctx->device]) {
        void *p = nullptr;
        PCRS_HIP_CHECK(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratch_ptr[ctx->device] = p;
    }
    *ptr = g_scratch_ptr[ctx->device];
    *bytes = sizeof(scratch);
    return PCRS_OK;
}

namespace pcrs {
__global__ void select_result_kernel(const StatsScalars *sc, const unsigned long long *count,
                                     pcrs_sor_select_result *out) {
    if (threadIdx.x != 0) return;
    const bool any = sc->has_finite != 0;
    out->thr = any ? sc->thr : __builtin_nanf("");
    out->has_finite = any ? 1 : 0;
    out->n_keep = any ? *count : 0ull;
}
}  // namespace pcrs

int pcrs_launch_select_result(pcrs_ctx *ctx, const void *d_scalars,
                              const unsigned long long *d_count, pcrs_sor_select_result *d_result) {
    hipLaunchKernelGGL(pcrs::select_result_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       reinterpret_cast<const StatsScalars *>(d_scalars), d_count, d_result);
    PCRS_HIP_CHECK(ctx, hipGetLastError());
    return PCRS_OK;
}

// host view of the device scalars
void pcrs_sor_stats_decode(const void *host_copy, float *mu, float *thr, int *has_finite) {
    const StatsScalars *sc = reinterpret_cast<const StatsScalars *>(host_copy);
    if (mu) *mu = sc->mu;
    if (thr) *thr = sc->thr;
    if (has_finite) *has_finite = sc->has_finite ? 1 : 0;
}
