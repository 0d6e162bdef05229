/*
 * sqpnp.c — CPU oracle for chalkydri_sqpnp and the AprilTags::process glue.  TEST INFRASTRUCTURE ONLY (see ck_oracle.h).
 *
 * Restates crates/chalkydri_sqpnp/src/lib.rs line by line (citations at each function).  The linear-algebra
 * primitives the reference takes from nalgebra 0.34.1 [EXT: crates/chalkydri_sqpnp/Cargo.toml:7, source not under
 * /root/reference] are restated from their published definitions:
 *     Matrix3::svd            -> one-sided Jacobi SVD (singular values sorted descending, like nalgebra)
 *     Matrix9::symmetric_eigen-> cyclic Jacobi eigen-decomposition
 *     Matrix15::lu().solve    -> LU with partial (row) pivoting; None when a pivot is exactly zero
 *     Matrix3::try_inverse    -> adjugate / determinant; None when the determinant is zero
 *     Rotation3::from_matrix  -> nearest rotation (polar factor), the fixed point of nalgebra's iteration
 *     euler_angles            -> Slabaugh's formulas as nalgebra documents them
 * PARITY UNPINNED: the reference has no test or golden vector for this crate; results are pinned by an independent
 * numpy restatement and by algebraic invariants on synthetic scenes (tests/test_sqpnp_oracle.py).  Tolerance, not
 * bit-exactness, is the contract for this floating-point path: |dR|, |dt| <= 1e-9 between implementations.
 */
#include "ck_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* lib.rs:29-39 */
#define XY_STD_DEV_SCALAR 5.0
#define THETA_STD_DEV_SCALAR 2.0
#define MAX_TRUSTABLE_RMS 0.1
#define MAX_GYRO_DELTA 30.0
#define TAG_SIZE 0.1651
#define CORNER_DISTANCE (TAG_SIZE / 2.0)
#define CK_PI 3.14159265358979323846

/* ---- small dense helpers (column-major where the reference is: r = vec(R) by columns, lib.rs:43,271) ---- */
static void quat_to_mat(const double q[4], double R[9]) { /* row-major R */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double n = sqrt(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static void mat_to_quat(const double R[9], double q[4]) { /* row-major R -> (w,x,y,z), w >= 0 branchwise */
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
    } else {
        double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
    }
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
    for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
static double mat3_det(const double m[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static int mat3_try_inverse(const double m[9], double o[9]) {
    double det = mat3_det(m);
    if (det == 0.0) return 0;
    o[0] = (m[4] * m[8] - m[5] * m[7]) / det; o[1] = (m[2] * m[7] - m[1] * m[8]) / det; o[2] = (m[1] * m[5] - m[2] * m[4]) / det;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / det; o[4] = (m[0] * m[8] - m[2] * m[6]) / det; o[5] = (m[2] * m[3] - m[0] * m[5]) / det;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / det; o[7] = (m[1] * m[6] - m[0] * m[7]) / det; o[8] = (m[0] * m[4] - m[1] * m[3]) / det;
    return 1;
}

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major); V columns = eigenvectors */
static void jacobi_eigen(double *A, int n, double *V, double *w) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    /* sweeps stop once the off-diagonal part is below 1e-16 of the matrix in the Frobenius norm (rotations preserve that
     * norm, so it is taken once): further sweeps would only push already negligible entries towards underflow */
    double tot = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) tot += A[i * n + j] * A[i * n + j];
    const double stop = 1e-32 * tot;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off <= stop) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* 3x3 SVD M = U diag(s) V^T via the eigen-decomposition of M^T M (singular values sorted descending) */
static void svd3(const double M[9], double U[9], double s[3], double V[9]) {
    double MtM[9], Vt[9], w[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) MtM[i * 3 + j] = M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
    jacobi_eigen(MtM, 3, Vt, w);
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (w[idx[j]] > w[idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    for (int c = 0; c < 3; c++) {
        s[c] = sqrt(w[idx[c]] > 0 ? w[idx[c]] : 0);
        for (int r = 0; r < 3; r++) V[r * 3 + c] = Vt[r * 3 + idx[c]];
    }
    /* U columns = M v / s.  Columns whose singular value vanishes are completed deterministically (a rank-deficient
     * guess is normal here: tags on one axis-aligned wall give exact null eigenvectors e0..e2 of Omega). */
    for (int c = 0; c < 3; c++) {
        double v[3] = {V[c], V[3 + c], V[6 + c]}, u[3];
        mat3_vec(M, v, u);
        double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (n > 1e-12 * (s[0] > 0 ? s[0] : 1.0)) { for (int r = 0; r < 3; r++) U[r * 3 + c] = u[r] / n; }
        else if (c == 2) { /* complete a right-handed frame */
            double ua[3] = {U[0], U[3], U[6]}, ub[3] = {U[1], U[4], U[7]};
            double cr[3] = {ua[1] * ub[2] - ua[2] * ub[1], ua[2] * ub[0] - ua[0] * ub[2], ua[0] * ub[1] - ua[1] * ub[0]};
            double cn = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
            for (int r = 0; r < 3; r++) U[r * 3 + 2] = cn > 0 ? cr[r] / cn : (r == 2);
        } else if (c == 1) { /* rank 1: the coordinate axis least aligned with u0 (first on ties), made orthogonal to u0 */
            double u0[3] = {U[0], U[3], U[6]};
            int k = 0;
            for (int r = 1; r < 3; r++)
                if (fabs(u0[r]) < fabs(u0[k])) k = r;
            double e[3] = {0, 0, 0};
            e[k] = 1.0;
            double d = u0[k], g[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
            double gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            for (int r = 0; r < 3; r++) U[r * 3 + 1] = g[r] / gn;
        } else { /* zero matrix: U = I */
            for (int r = 0; r < 3; r++) U[r * 3 + 0] = (r == 0);
        }
    }
}

/* lib.rs:42-59: U V^T, third column of U flipped when det < 0.  r_vec is column-major. */
static int nearest_so3(const double r_vec[9], double out[9]) {
    double M[9];
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) M[r * 3 + c] = r_vec[c * 3 + r];
    double U[9], s[3], V[9], Vt[9], rot[9];
    svd3(M, U, s, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Vt[i * 3 + j] = V[j * 3 + i];
    mat3_mul(U, Vt, rot);
    if (mat3_det(rot) < 0.0) {
        for (int r = 0; r < 3; r++) U[r * 3 + 2] = -U[r * 3 + 2];
        mat3_mul(U, Vt, rot);
    }
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) out[c * 3 + r] = rot[r * 3 + c];
    return 1;
}

/* lib.rs:62-95 */
static void constraints_and_jacobian(const double r[9], double h[6], double J[6 * 9]) {
    const double *c1 = r, *c2 = r + 3, *c3 = r + 6;
    h[0] = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2] - 1.0;
    h[1] = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2] - 1.0;
    h[2] = c3[0] * c3[0] + c3[1] * c3[1] + c3[2] * c3[2] - 1.0;
    h[3] = c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
    h[4] = c1[0] * c3[0] + c1[1] * c3[1] + c1[2] * c3[2];
    h[5] = c2[0] * c3[0] + c2[1] * c3[1] + c2[2] * c3[2];
    memset(J, 0, sizeof(double) * 54);
    for (int k = 0; k < 3; k++) {
        J[0 * 9 + 0 + k] = 2.0 * c1[k]; J[1 * 9 + 3 + k] = 2.0 * c2[k]; J[2 * 9 + 6 + k] = 2.0 * c3[k];
        J[3 * 9 + 0 + k] = c2[k]; J[3 * 9 + 3 + k] = c1[k];
        J[4 * 9 + 0 + k] = c3[k]; J[4 * 9 + 6 + k] = c1[k];
        J[5 * 9 + 3 + k] = c3[k]; J[5 * 9 + 6 + k] = c2[k];
    }
}

/* 15x15 LU with partial pivoting; returns 0 when a pivot is exactly zero (nalgebra's solve -> None) */
static int lu_solve15(double *A, double *b) {
    const int n = 15;
    for (int col = 0; col < n; col++) {
        int piv = col;
        double best = fabs(A[col * n + col]);
        for (int r = col + 1; r < n; r++)
            if (fabs(A[r * n + col]) > best) { best = fabs(A[r * n + col]); piv = r; }
        if (best == 0.0) return 0;
        if (piv != col) {
            for (int k = 0; k < n; k++) { double t = A[col * n + k]; A[col * n + k] = A[piv * n + k]; A[piv * n + k] = t; }
            double t = b[col]; b[col] = b[piv]; b[piv] = t;
        }
        for (int r = col + 1; r < n; r++) {
            double f = A[r * n + col] / A[col * n + col];
            if (f == 0.0) continue;
            for (int k = col; k < n; k++) A[r * n + k] -= f * A[col * n + k];
            b[r] -= f * b[col];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[r * n + k] * b[k];
        b[r] = s / A[r * n + r];
    }
    return 1;
}

/* lib.rs:98-115: KKT step [[Omega, J^T],[J, 0]] [delta; lambda] = [-Omega r; -h] */
static int solve_newton(const double r[9], const double omega[81], const double h[6], const double J[54], double delta[9]) {
    double lhs[225], rhs[15];
    memset(lhs, 0, sizeof lhs);
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) lhs[i * 15 + j] = omega[i * 9 + j];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 9; j++) { lhs[j * 15 + 9 + i] = J[i * 9 + j]; lhs[(9 + i) * 15 + j] = J[i * 9 + j]; }
    for (int i = 0; i < 9; i++) {
        double s = 0;
        for (int j = 0; j < 9; j++) s += omega[i * 9 + j] * r[j];
        rhs[i] = -s;
    }
    for (int i = 0; i < 6; i++) rhs[9 + i] = -h[i];
    if (!lu_solve15(lhs, rhs)) return 0;
    memcpy(delta, rhs, sizeof(double) * 9);
    return 1;
}

/* lib.rs:463-480 */
static double optimization(int max_iter, double tol_sq, double r[9], const double omega[81]) {
    for (int it = 0; it < max_iter; it++) {
        double h[6], J[54], d[9];
        constraints_and_jacobian(r, h, J);
        if (!solve_newton(r, omega, h, J, d)) break;
        double n2 = 0;
        for (int k = 0; k < 9; k++) { r[k] += d[k]; n2 += d[k] * d[k]; }
        if (n2 < tol_sq) break;
    }
    double e = 0;
    for (int i = 0; i < 9; i++) {
        double s = 0;
        for (int j = 0; j < 9; j++) s += omega[i * 9 + j] * r[j];
        e += r[i] * s;
    }
    return e;
}

typedef struct { double omega[81]; double q_tt_inv[9]; double q_rt[27]; /* 9x3 row-major */ } linsys_t;

/* lib.rs:124-180 */
static void build_linear_system(const double *p3, const double *p2, int n, linsys_t *sys) {
    double q_rr[81], q_rt[27], q_tt[9];
    memset(q_rr, 0, sizeof q_rr); memset(q_rt, 0, sizeof q_rt); memset(q_tt, 0, sizeof q_tt);
    for (int k = 0; k < n; k++) {
        const double *v = p2 + 3 * k, *X = p3 + 3 * k;
        double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        double inv = 1.0 / sq;
        double P[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) P[i * 3 + j] = (i == j ? 1.0 : 0.0) - (v[i] * v[j]) * inv;
        for (int i = 0; i < 9; i++) q_tt[i] += P[i];
        for (int a = 0; a < 3; a++) {
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) q_rt[(3 * a + i) * 3 + j] += P[i * 3 + j] * X[a];
            for (int b = 0; b < 3; b++)
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) q_rr[(3 * a + i) * 9 + 3 * b + j] += (P[i * 3 + j] * X[a]) * X[b];
        }
    }
    if (!mat3_try_inverse(q_tt, sys->q_tt_inv)) memset(sys->q_tt_inv, 0, sizeof sys->q_tt_inv); /* unwrap_or_default, lib.rs:171 */
    double temp[27];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 3; j++)
            temp[i * 3 + j] = q_rt[i * 3] * sys->q_tt_inv[j] + q_rt[i * 3 + 1] * sys->q_tt_inv[3 + j] + q_rt[i * 3 + 2] * sys->q_tt_inv[6 + j];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++)
            sys->omega[i * 9 + j] = q_rr[i * 9 + j] - (temp[i * 3] * q_rt[j * 3] + temp[i * 3 + 1] * q_rt[j * 3 + 1] + temp[i * 3 + 2] * q_rt[j * 3 + 2]);
    memcpy(sys->q_rt, q_rt, sizeof q_rt);
}

typedef struct { double r[9]; double energy; } cand_t;

/* lib.rs:396-428 */
static int solve_rotation_candidates(const ck_sqpnp_params_t *prm, const double omega[81], const double fwd_in_cam[3], double gyro_cos,
                                     double gyro_sin, double sign_change_error, cand_t cands[6]) {
    double A[81], V[81], w[9];
    memcpy(A, omega, sizeof A);
    jacobi_eigen(A, 9, V, w);
    int idx[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 1; i < 9; i++) { /* stable insertion sort by eigenvalue, like sort_by(total_cmp) */
        int v = idx[i], j = i - 1;
        while (j >= 0 && w[idx[j]] > w[v]) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = v;
    }
    int n = 0;
    for (int t = 0; t < 3; t++) {
        for (int sg = 0; sg < 2; sg++) {
            double sign = sg == 0 ? -1.0 : 1.0, guess[9], r[9];
            for (int k = 0; k < 9; k++) guess[k] = V[k * 9 + idx[t]] * sign;
            if (!nearest_so3(guess, r)) continue;
            double energy = optimization(prm->max_iter, prm->tol_sq, r, omega);
            const double *d = fwd_in_cam;
            double fx = r[0] * d[0] + r[1] * d[1] + r[2] * d[2];
            double fy = r[3] * d[0] + r[4] * d[1] + r[5] * d[2];
            double dot = fx * gyro_cos + fy * gyro_sin;
            double angle_error = 1.0 - dot;
            if (angle_error < 0.0) angle_error = 0.0;
            energy += sign_change_error * angle_error;
            memcpy(cands[n].r, r, sizeof r);
            cands[n].energy = energy;
            n++;
        }
    }
    for (int i = 1; i < n; i++) { /* stable sort by penalised energy (lib.rs:427) */
        cand_t v = cands[i];
        int j = i - 1;
        while (j >= 0 && cands[j].energy > v.energy) { cands[j + 1] = cands[j]; j--; }
        cands[j + 1] = v;
    }
    return n;
}

/* nearest rotation to a (near-orthonormal) matrix: what Rotation3::from_matrix converges to (lib.rs:289) */
static void rot_from_matrix(const double Rm[9], double out[9]) {
    double U[9], s[3], V[9], Vt[9];
    svd3(Rm, U, s, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Vt[i * 3 + j] = V[j * 3 + i];
    mat3_mul(U, Vt, out);
    if (mat3_det(out) < 0.0) {
        for (int r = 0; r < 3; r++) U[r * 3 + 2] = -U[r * 3 + 2];
        mat3_mul(U, Vt, out);
    }
}

/* lib.rs:224-246 */
static void compute_std_devs(double energy, double distance, int n_tags, double out[3]) {
    double n_points = (double)(n_tags * 4);
    double rms = sqrt(energy / n_points);
    if (rms > MAX_TRUSTABLE_RMS) { out[0] = out[1] = out[2] = DBL_MAX; return; }
    double mult = 1.0 + (distance / TAG_SIZE);
    double xy = ((rms * mult) / sqrt((double)n_tags)) * XY_STD_DEV_SCALAR;
    xy = xy < 0.01 ? 0.01 : (xy > 10.0 ? 10.0 : xy);
    double th = (((rms / TAG_SIZE) * mult) / sqrt((double)n_tags)) * THETA_STD_DEV_SCALAR;
    th = th < 0.05 ? 0.05 : (th > CK_PI ? CK_PI : th);
    out[0] = xy; out[1] = xy; out[2] = th;
}

static double euler_yaw(const double R[9]) { /* nalgebra Rotation3::euler_angles().2 */
    if (fabs(R[6]) < 1.0) {
        double pitch = -asin(R[6]);
        double tc = cos(pitch);
        return atan2(R[3] / tc, R[0] / tc);
    }
    return 0.0;
}

/* lib.rs:297-377 with solve (248-295) and corner_points_from_center (379-394) inlined */
int ora_sqpnp_solve_robot_pose(const ck_sqpnp_params_t *prm, const ck_iso3_t *tags, int n_tags, const double *bearings, int n_bearings,
                               const ck_iso3_t *robot_to_cam, double gyro, double sign_change_error, ck_sqpnp_result_t *out) {
    memset(out, 0, sizeof *out);
    double gyro_cos = cos(gyro), gyro_sin = sin(gyro);
    double Rrc[9];
    quat_to_mat(robot_to_cam->q, Rrc);
    double fwd_in_cam[3] = {Rrc[0], Rrc[3], Rrc[6]}; /* column 0 (lib.rs:313-318) */
    int n = 4 * n_tags;
    if (n < 3 || n != n_bearings) return 0; /* lib.rs:255 */
    double *buf = (double *)malloc(sizeof(double) * 3 * (size_t)n), *loc = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    static const double cp[4][3] = {{0, -CORNER_DISTANCE, -CORNER_DISTANCE}, {0, CORNER_DISTANCE, -CORNER_DISTANCE},
                                    {0, CORNER_DISTANCE, CORNER_DISTANCE}, {0, -CORNER_DISTANCE, CORNER_DISTANCE}};
    for (int t = 0; t < n_tags; t++) {
        double R[9];
        quat_to_mat(tags[t].q, R);
        for (int c = 0; c < 4; c++) {
            double p[3];
            mat3_vec(R, cp[c], p);
            for (int k = 0; k < 3; k++) buf[(4 * t + c) * 3 + k] = p[k] + tags[t].t[k];
        }
    }
    double centroid[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) centroid[k] += buf[i * 3 + k];
    for (int k = 0; k < 3; k++) centroid[k] /= (double)n;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) loc[i * 3 + k] = buf[i * 3 + k] - centroid[k];
    linsys_t sys;
    build_linear_system(loc, bearings, n, &sys);
    cand_t cands[6];
    int nc = solve_rotation_candidates(prm, sys.omega, fwd_in_cam, gyro_cos, gyro_sin, sign_change_error, cands);
    int found = 0;
    double best_score = DBL_MAX, bestR[9], bestT[3], best_energy = 0;
    for (int ci = 0; ci < nc; ci++) {
        const double *r = cands[ci].r;
        double Rm[9];
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 3; rr++) Rm[rr * 3 + c] = r[c * 3 + rr];
        double qtr[3], tl[3], Rc[3], t[3];
        for (int j = 0; j < 3; j++) { /* q_rt^T r */
            double s = 0;
            for (int i = 0; i < 9; i++) s += sys.q_rt[i * 3 + j] * r[i];
            qtr[j] = s;
        }
        mat3_vec(sys.q_tt_inv, qtr, tl);
        mat3_vec(Rm, centroid, Rc);
        for (int k = 0; k < 3; k++) t[k] = -tl[k] - Rc[k];
        int all_in_front = 1;
        for (int i = 0; i < n && all_in_front; i++) {
            double pc[3];
            mat3_vec(Rm, buf + 3 * i, pc);
            if (!(pc[2] + t[2] > 0.0)) all_in_front = 0;
        }
        if (!all_in_front) continue;
        if (cands[ci].energy < best_score) {
            best_score = cands[ci].energy;
            double e = 0;
            for (int i = 0; i < 9; i++) {
                double s = 0;
                for (int j = 0; j < 9; j++) s += sys.omega[i * 9 + j] * r[j];
                e += r[i] * s;
            }
            best_energy = e;
            rot_from_matrix(Rm, bestR);
            memcpy(bestT, t, sizeof t);
            found = 1;
        }
    }
    free(buf); free(loc);
    if (!found) return 0;
    double distance = sqrt(bestT[0] * bestT[0] + bestT[1] * bestT[1] + bestT[2] * bestT[2]);
    compute_std_devs(best_energy, distance, n_tags, out->std_devs);
    /* t_world_robot = world_to_cam^-1 * robot_to_cam (lib.rs:328-337) */
    double Rt[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[i * 3 + j] = bestR[j * 3 + i];
    double d[3] = {robot_to_cam->t[0] - bestT[0], robot_to_cam->t[1] - bestT[1], robot_to_cam->t[2] - bestT[2]};
    double robot_pos[3], robot_rot[9];
    mat3_vec(Rt, d, robot_pos);
    mat3_mul(Rt, Rrc, robot_rot);
    double tag_centroid[3] = {0, 0, 0};
    for (int t = 0; t < n_tags; t++)
        for (int k = 0; k < 3; k++) tag_centroid[k] += tags[t].t[k];
    for (int k = 0; k < 3; k++) tag_centroid[k] /= (double)n_tags;
    double vision_yaw = atan2(robot_rot[3], robot_rot[0]);
    double delta_yaw = gyro - vision_yaw;
    delta_yaw = fmod(delta_yaw + CK_PI, 2.0 * CK_PI);
    if (delta_yaw < 0) delta_yaw += 2.0 * CK_PI; /* rem_euclid */
    delta_yaw -= CK_PI;
    double delta_deg = fabs(delta_yaw) * (180.0 / CK_PI);
    double weight = delta_deg / MAX_GYRO_DELTA;
    weight = weight < 0 ? 0 : (weight > 1 ? 1 : weight);
    weight = weight * weight * (3.0 - 2.0 * weight);
    double applied = delta_yaw * weight;
    double cz = cos(applied), sz = sin(applied);
    double rotz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double rel[3] = {robot_pos[0] - tag_centroid[0], robot_pos[1] - tag_centroid[1], robot_pos[2] - tag_centroid[2]}, piv[3];
    mat3_vec(rotz, rel, piv);
    for (int k = 0; k < 3; k++) out->pos[k] = tag_centroid[k] + piv[k];
    mat3_mul(rotz, robot_rot, out->rot);
    out->yaw = euler_yaw(out->rot);
    out->energy = best_energy;
    out->valid = 1;
    return 1;
}

/* lib.rs:430-461 */
void ora_sqpnp_create_solver_camera_transform(double fwd_m, double left_m, double up_m, double roll_deg, double pitch_deg, double yaw_deg,
                                              ck_iso3_t *out) {
    double r = roll_deg * (CK_PI / 180.0), p = pitch_deg * (CK_PI / 180.0), y = yaw_deg * (CK_PI / 180.0);
    double cr = cos(r * 0.5), sr = sin(r * 0.5), cpp = cos(p * 0.5), sp = sin(p * 0.5), cy = cos(y * 0.5), sy = sin(y * 0.5);
    double q[4] = {cr * cpp * cy + sr * sp * sy, sr * cpp * cy - cr * sp * sy, cr * sp * cy + sr * cpp * sy, cr * cpp * sy - sr * sp * cy};
    double Rn[9];
    quat_to_mat(q, Rn);
    static const double nwu_to_cv[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
    double Rc[9], Rinv[9];
    mat3_mul(Rn, nwu_to_cv, Rc); /* rotation of (robot_pose_of_cam_nwu * nwu_to_cv) */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rinv[i * 3 + j] = Rc[j * 3 + i];
    double T[3] = {fwd_m, left_m, up_m}, ti[3];
    mat3_vec(Rinv, T, ti);
    for (int k = 0; k < 3; k++) out->t[k] = -ti[k];
    mat_to_quat(Rinv, out->q);
}

/* OpenCVModel5 unprojection [EXT: camera-intrinsic-model, git branch main, crates/apriltags/Cargo.toml:23; call site
 * crates/apriltags/src/lib.rs:316-322].  Restated from the published OpenCV model: normalise, then undo
 * (k1,k2,p1,p2,k3) by fixed-point iteration; the bearing is (x,y,1)/|(x,y,1)|. */
int ora_unproject_opencv5(const ck_opencv5_t *c, const double *px, int n, double *bearings, uint8_t *ok) {
    for (int i = 0; i < n; i++) {
        double xd = (px[2 * i] - c->cx) / c->fx, yd = (px[2 * i + 1] - c->cy) / c->fy;
        double x = xd, y = yd;
        int conv = 0;
        for (int it = 0; it < 50; it++) {
            double r2 = x * x + y * y;
            double radial = 1.0 + r2 * (c->k1 + r2 * (c->k2 + r2 * c->k3));
            double dx = 2.0 * c->p1 * x * y + c->p2 * (r2 + 2.0 * x * x);
            double dy = c->p1 * (r2 + 2.0 * y * y) + 2.0 * c->p2 * x * y;
            double nx = (xd - dx) / radial, ny = (yd - dy) / radial;
            double ex = nx - x, ey = ny - y;
            x = nx; y = ny;
            if (ex * ex + ey * ey < 1e-24) { conv = 1; break; }
        }
        double nrm = sqrt(x * x + y * y + 1.0);
        bearings[3 * i] = x / nrm; bearings[3 * i + 1] = y / nrm; bearings[3 * i + 2] = 1.0 / nrm;
        if (ok) ok[i] = (uint8_t)(conv && isfinite(x) && isfinite(y));
    }
    return 0;
}

/* AprilTags::process for one frame (crates/apriltags/src/lib.rs:293-379) */
int ora_process_frame(const uint8_t *img, int w, int h, int stride, const ck_config_t *cfg, const ck_process_params_t *pp, double gyro,
                      int has_gyro, ck_vision_measurement_t *out, int *valid) {
    memset(out, 0, sizeof *out);
    out->camera_id = pp->camera_id;
    *valid = 0;
    enum { CAP = 256 };
    ck_detection_t *dets = (ck_detection_t *)malloc(sizeof(ck_detection_t) * CAP);
    int nd = 0;
    uint32_t st = 0;
    ora_detect(img, w, h, stride, cfg, dets, CAP, &nd, &st);
    if (nd > 0 && has_gyro) {
        ck_iso3_t *world = (ck_iso3_t *)malloc(sizeof(ck_iso3_t) * (size_t)nd);
        double *cam = (double *)malloc(sizeof(double) * 12 * (size_t)nd);
        int nt = 0;
        for (int i = 0; i < nd; i++) {
            const ck_field_tag_t *tag = NULL;
            for (int k = 0; k < pp->n_field; k++)
                if (pp->field[k].id == dets[i].id) { tag = &pp->field[k]; break; }
            if (!tag) continue; /* unknown tag (lib.rs:306-308) */
            if (!pp->allow_unverified_ids && (uint32_t)dets[i].id >= cfg->families[dets[i].family]->n_upstream) continue; /* not an upstream id */
            double px[8], b[12];
            uint8_t ok[4];
            for (int c = 0; c < 4; c++) { px[2 * c] = dets[i].p[c][0]; px[2 * c + 1] = dets[i].p[c][1]; }
            ora_unproject_opencv5(&pp->cam, px, 4, b, ok);
            if (!(ok[0] && ok[1] && ok[2] && ok[3])) continue; /* lib.rs:324 */
            world[nt] = tag->pose;
            memcpy(cam + 12 * nt, b, sizeof b);
            nt++;
        }
        ck_sqpnp_result_t res;
        if (ora_sqpnp_solve_robot_pose(&pp->sqpnp, world, nt, cam, 4 * nt, &pp->robot_to_cam, gyro, pp->sign_change_error, &res)) {
            out->pose_x = res.pos[0]; out->pose_y = res.pos[1]; out->pose_rot = res.yaw;
            out->std_x = res.std_devs[0]; out->std_y = res.std_devs[1]; out->std_rot = res.std_devs[2];
            out->tag_count = (uint8_t)(nd > 255 ? 255 : nd); /* ALL detections (lib.rs:354) */
            *valid = 1;
        }
        free(world); free(cam);
    }
    free(dets);
    return 0;
}
