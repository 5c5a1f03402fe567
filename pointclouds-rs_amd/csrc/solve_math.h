// Small dense solves of the ICP outer loop, usable from host and device code: a 6x6
// normal-equation solve + Rodrigues rotation for point-to-plane
// (crates/registration/src/icp_plane.rs:182-235) and a 3x3 SVD for point-to-point
// (crates/registration/src/icp.rs:246-269).
#pragma once

#include <cmath>

#include <hip/hip_runtime.h>

// always inlined: a solver array whose address crosses a call stays in (scratch) memory
#define PCRS_HD __host__ __device__ __attribute__((always_inline))

namespace pcrs_math {


PCRS_HD inline void identity(float rot[9], float t[3]) {
    for (int i = 0; i < 9; ++i) rot[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    t[0] = t[1] = t[2] = 0.0f;
}

// Every array the solves index with a run-time subscript.  On the device such arrays would live
// in scratch memory, and a kernel that needs scratch pays for it at every launch; there the
// caller puts one Work into LDS.  The host keeps it on the stack.
struct Work {
    double A[6][6], L[6][6], M[6][7], b[6], x[6], y[6];                       // 6x6 system
    double S[3][3], H[3][3], U[3][3], V[3][3], Vs[3][3], R[3][3], sv[3], nrm[3], u[3];  // 3x3 SVD
    int ord[3];
    bool have[3];
};

// Cholesky-Banachiewicz (row by row) factorisation and the two triangular solves.
PCRS_HD inline bool cholesky6(Work &w, const double A[6][6], const double b[6], double x[6]) {
    double (&L)[6][6] = w.L;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = A[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i][i] = sqrt(s);
            } else {
                L[i][j] = s / L[j][j];
            }
        }
    }
    double (&y)[6] = w.y;
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
    return true;
}

// Gaussian elimination with partial pivoting (the reference's LU fallback).
PCRS_HD inline bool lu6(Work &w, const double A[6][6], const double b[6], double x[6]) {
    double (&M)[6][7] = w.M;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) M[i][j] = A[i][j];
        M[i][6] = b[i];
    }
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r)
            if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (M[piv][c] == 0.0 || !(M[piv][c] - M[piv][c] == 0.0)) return false;
        if (piv != c)
            for (int j = 0; j < 7; ++j) {
                const double tmp = M[c][j];
                M[c][j] = M[piv][j];
                M[piv][j] = tmp;
            }
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int j = c; j < 7; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 5; i >= 0; --i) {
        double s = M[i][6];
        for (int k = i + 1; k < 6; ++k) s -= M[i][k] * x[k];
        x[i] = s / M[i][i];
    }
    return true;
}

// One-sided (Hestenes) Jacobi SVD of a 3x3: A = U diag(s) V^T, s descending.
PCRS_HD inline void svd3(Work &w, const double Ain[3][3], double U[3][3], double s[3],
                         double V[3][3]) {
    double (&A)[3][3] = w.S;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = Ain[i][j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 3; ++k) {
                    alpha += A[k][p] * A[k][p];
                    beta += A[k][q] * A[k][q];
                    gamma += A[k][p] * A[k][q];
                }
                if (gamma == 0.0) continue;
                off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t =
                    (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double ap = A[k][p], aq = A[k][q];
                    A[k][p] = c * ap - sn * aq;
                    A[k][q] = sn * ap + c * aq;
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - sn * vq;
                    V[k][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double (&nrm)[3] = w.nrm;
    for (int j = 0; j < 3; ++j)
        nrm[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    int (&ord)[3] = w.ord;
    ord[0] = 0;
    ord[1] = 1;
    ord[2] = 2;
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (nrm[ord[j]] > nrm[ord[i]]) {
                const int tmp = ord[i];
                ord[i] = ord[j];
                ord[j] = tmp;
            }
    double (&Vs)[3][3] = w.Vs;
    bool (&have)[3] = w.have;
    for (int c = 0; c < 3; ++c) {
        const int o = ord[c];
        s[c] = nrm[o];
        for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][o];
        have[c] = nrm[o] > 1e-13 * (nrm[ord[0]] > 0 ? nrm[ord[0]] : 1.0) && nrm[o] > 0;
        if (have[c])
            for (int r = 0; r < 3; ++r) U[r][c] = A[r][o] / nrm[o];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = Vs[i][j];
    // complete U to an orthonormal basis where singular values vanish
    for (int c = 0; c < 3; ++c) {
        if (have[c]) continue;
        const int a = (c + 1) % 3, b = (c + 2) % 3;
        double (&u)[3] = w.u;
        if (have[a] && have[b]) {
            u[0] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
            u[1] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
            u[2] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
        } else {
            int best = 0;
            double bestv = 1e300;
            for (int e = 0; e < 3; ++e) {
                double mx = 0;
                for (int p = 0; p < 3; ++p)
                    if (have[p]) mx = fmax(mx, fabs(U[e][p]));
                if (mx < bestv) {
                    bestv = mx;
                    best = e;
                }
            }
            u[0] = u[1] = u[2] = 0.0;
            u[best] = 1.0;
            for (int p = 0; p < 3; ++p)
                if (have[p]) {
                    const double d = u[0] * U[0][p] + u[1] * U[1][p] + u[2] * U[2][p];
                    for (int r = 0; r < 3; ++r) u[r] -= d * U[r][p];
                }
        }
        const double l = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int r = 0; r < 3; ++r) U[r][c] = u[r] / l;
        have[c] = true;
    }
}

PCRS_HD inline double det3(const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) -
           M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}


// icp.rs:52-73 — result applies (ra,ta) first, then (rb,tb).
PCRS_HD inline void compose(const float ra[9], const float ta[3], const float rb[9], const float tb[3],
                       float ro[9], float to[3]) {
    float r[9], t[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = rb[i * 3 + 0] * ra[0 * 3 + j];
            s += rb[i * 3 + 1] * ra[1 * 3 + j];
            s += rb[i * 3 + 2] * ra[2 * 3 + j];
            r[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i) {
        float s = rb[i * 3 + 0] * ta[0];
        s += rb[i * 3 + 1] * ta[1];
        s += rb[i * 3 + 2] * ta[2];
        t[i] = s + tb[i];
    }
    for (int i = 0; i < 9; ++i) ro[i] = r[i];
    for (int i = 0; i < 3; ++i) to[i] = t[i];
}

// icp_plane.rs:137-139,182-235 from the accumulated upper triangle of A^T A and A^T b.
PCRS_HD inline void solve_plane(Work &w, const double *sums, float rot[9], float t[3]) {
    identity(rot, t);
    if (!(sums[0] > 0.0)) return;
    double (&A)[6][6] = w.A;
    double (&b)[6] = w.b;
    int s = 2;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
            A[r][c] = sums[s];
            A[c][r] = sums[s];
            ++s;
        }
    for (int r = 0; r < 6; ++r) b[r] = sums[23 + r];
    double diag_max = 0.0;
    for (int i = 0; i < 6; ++i) diag_max = fmax(diag_max, fabs(A[i][i]));
    const double lambda = 1e-6 * fmax(diag_max, 1e-12);
    for (int i = 0; i < 6; ++i) A[i][i] += lambda;
    double (&x)[6] = w.x;
    if (!cholesky6(w, A, b, x) && !lu6(w, A, b, x)) return;
    const float alpha = (float)x[0], beta = (float)x[1], gamma = (float)x[2];
    t[0] = (float)x[3];
    t[1] = (float)x[4];
    t[2] = (float)x[5];
    const float angle = sqrt((alpha * alpha + beta * beta) + gamma * gamma);
    if (angle < 1e-10f) {
        const float r[9] = {1.0f, -gamma, beta, gamma, 1.0f, -alpha, -beta, alpha, 1.0f};
        for (int i = 0; i < 9; ++i) rot[i] = r[i];
        return;
    }
    const float ax = alpha / angle, ay = beta / angle, az = gamma / angle;
    // evaluated in f64 and rounded once: the device's own f32 cos / sin are a unit in the last
    // place off a correctly rounded libm now and then, and one such bit moves the whole
    // trajectory of the loop away from the host's
    const float c = (float)cos((double)angle), sn = (float)sin((double)angle), tt = 1.0f - c;
    rot[0] = tt * ax * ax + c;
    rot[1] = tt * ax * ay - sn * az;
    rot[2] = tt * ax * az + sn * ay;
    rot[3] = tt * ax * ay + sn * az;
    rot[4] = tt * ay * ay + c;
    rot[5] = tt * ay * az - sn * ax;
    rot[6] = tt * ax * az - sn * ay;
    rot[7] = tt * ay * az + sn * ax;
    rot[8] = tt * az * az + c;
}

// icp.rs:210-270 from f64 sums: centroids, H = sum s t^T - n sbar tbar^T, Kabsch rotation.
PCRS_HD inline void solve_point(Work &w, const double *sums, float rot[9], float t[3]) {
    identity(rot, t);
    const double n = sums[0];
    if (!(n > 0.0)) return;
    double sbar[3], tbar[3];
    double (&H)[3][3] = w.H;
    for (int a = 0; a < 3; ++a) {
        sbar[a] = sums[2 + a] / n;
        tbar[a] = sums[5 + a] / n;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r][c] = sums[8 + r * 3 + c] - n * sbar[r] * tbar[c];
    double (&U)[3][3] = w.U;
    double (&sv)[3] = w.sv;
    double (&V)[3][3] = w.V;
    double (&R)[3][3] = w.R;
    svd3(w, H, U, sv, V);
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                R[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + V[i][2] * U[j][2];
        if (pass == 1 || det3(R) >= 0.0) break;
        for (int r = 0; r < 3; ++r) V[r][2] = -V[r][2];  // icp.rs:256-261
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rot[i * 3 + j] = (float)R[i][j];
    const float sc[3] = {(float)sbar[0], (float)sbar[1], (float)sbar[2]};
    const float tc[3] = {(float)tbar[0], (float)tbar[1], (float)tbar[2]};
    for (int r = 0; r < 3; ++r) {
        float s = rot[r * 3 + 0] * sc[0];
        s += rot[r * 3 + 1] * sc[1];
        s += rot[r * 3 + 2] * sc[2];
        t[r] = tc[r] - s;
    }
}

}  // namespace pcrs_math
