/* Runs every exported function of the CPU oracle on small awkward inputs under
 * AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle san`).  The oracle is test
 * infrastructure; this program checks the checker's own memory safety on the host.  It includes
 * the oracle's source so that the static helpers are instrumented too. */
#include "pcrs_oracle.c"

#include <stdio.h>

static uint64_t sm_state;
static float rnd(void) { /* SplitMix64 -> [0, 1) */
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((double)(z >> 40) / 16777216.0);
}

static void fill(float *x, float *y, float *z, size_t n, int kind) {
    for (size_t i = 0; i < n; ++i) {
        x[i] = rnd() * 10.0f;
        y[i] = rnd() * 10.0f;
        z[i] = kind == 1 ? 2.5f : rnd() * 10.0f; /* kind 1: co-planar */
        if (kind == 2) { /* lattice with duplicates */
            x[i] = (float)(i % 4);
            y[i] = (float)((i / 4) % 4);
            z[i] = (float)((i / 16) % 2);
        }
    }
    if (kind == 3 && n > 8) { /* non-finite rows */
        x[1] = NAN;
        y[4] = INFINITY;
        z[7] = -INFINITY;
    }
}

int main(void) {
    const size_t sizes[] = {0, 1, 2, 3, 17, 64, 300, 5000};
    size_t checks = 0;
    for (int kind = 0; kind < 4; ++kind)
        for (size_t si = 0; si < sizeof(sizes) / sizeof(sizes[0]); ++si) {
            const size_t n = sizes[si], cap = n ? n : 1;
            sm_state = 1234 + 17 * kind + si;
            float *x = malloc(cap * 4), *y = malloc(cap * 4), *z = malloc(cap * 4);
            fill(x, y, z, n, kind);
            for (int algo = 0; algo < 2; ++algo)
                for (uint32_t k = 0; k <= 33; k += (k < 2 ? 1 : 15)) {
                    const size_t nk = cap * (k ? k : 1);
                    uint32_t *idx = malloc(nk * 4), *cnt = malloc(cap * 4);
                    float *dist = malloc(nk * 4), *d2 = malloc(nk * 4), *mean = malloc(cap * 4);
                    unsigned char *keep = malloc(cap);
                    float *nx = malloc(cap * 4), *ny = malloc(cap * 4), *nz = malloc(cap * 4);
                    float thr, vp[3] = {0, 0, 20};
                    orc_knn_batch(x, y, z, n, x, y, z, n, k, algo, 2, idx, dist, d2, cnt);
                    orc_sor(x, y, z, n, k, 1.0f, algo, 2, keep, &thr, mean);
                    orc_normals(x, y, z, n, k, vp, algo, 2, nx, ny, nz);
                    free(idx), free(cnt), free(dist), free(d2), free(mean), free(keep);
                    free(nx), free(ny), free(nz);
                    checks += 3;
                }
            {
                unsigned char *keep = malloc(cap);
                uint32_t *cnt = malloc(cap * 4), *off = malloc((cap + 1) * 4), *ind = malloc(cap * 4);
                float *ox = malloc(cap * 4), *oy = malloc(cap * 4), *oz = malloc(cap * 4);
                orc_radius_outlier(x, y, z, n, 1.5f, 3, 2, keep, cnt);
                orc_radius_outlier(x, y, z, n, -1.0f, 0, 2, keep, cnt);
                orc_voxel_downsample(x, y, z, n, 0.7f, ox, oy, oz);
                orc_voxel_downsample(x, y, z, n, 1e-30f, ox, oy, oz);
                orc_euclidean_cluster(x, y, z, n, 0.8f, 1, cap, off, ind);
                if (n >= 3) {
                    uint32_t smp[30];
                    for (int s = 0; s < 30; ++s) smp[s] = (uint32_t)(rnd() * (float)n) % (uint32_t)n;
                    float model[4];
                    uint64_t best;
                    orc_ransac_plane_samples(x, y, z, n, smp, 10, 0.2f, model, keep, &best);
                }
                free(keep), free(cnt), free(off), free(ind), free(ox), free(oy), free(oz);
                checks += 6;
            }
            if (n >= 2) { /* registration: target = the cloud, source = the cloud moved a little */
                float *sx = malloc(cap * 4), *sy = malloc(cap * 4), *sz = malloc(cap * 4);
                float *nx = malloc(cap * 4), *ny = malloc(cap * 4), *nz = malloc(cap * 4);
                float vp[3] = {0, 0, 20};
                const float rot[9] = {0.9995f, -0.0316f, 0, 0.0316f, 0.9995f, 0, 0, 0, 1};
                const float t[3] = {0.05f, -0.02f, 0.01f};
                orc_apply_transform(x, y, z, n, rot, t, sx, sy, sz);
                orc_normals(x, y, z, n, 8, vp, 1, 2, nx, ny, nz);
                for (int mode = 0; mode < 3; ++mode) {
                    orc_icp_result res;
                    float trace[2 * 12];
                    orc_icp(mode, sx, sy, sz, n, x, y, z, nx, ny, nz, n, n, 12, 1e-6f, 2.0f, mode & 1, 2,
                            &res, trace);
                    ++checks;
                }
                double plane[42], point[15], sd2;
                float rmse;
                orc_icp_step(sx, sy, sz, n, x, y, z, nx, ny, nz, n, INFINITY, 1, 2, plane, point, &rmse,
                             &sd2, NULL, NULL);
                free(sx), free(sy), free(sz), free(nx), free(ny), free(nz);
                ++checks;
            }
            free(x), free(y), free(z);
        }
    printf("oracle sanitizer check: %zu calls clean\n", checks);
    return 0;
}
