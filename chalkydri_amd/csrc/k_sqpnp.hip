// k_sqpnp.hip — batched SQPnP robot-pose solve, OpenCV-5 unprojection and the AprilTags::process glue.
//
// Replaces chalkydri_sqpnp::SqPnP::solve_robot_pose (crates/chalkydri_sqpnp/src/lib.rs:297-377, with solve :248-295,
// build_linear_system :124-180, solve_rotation_candidates :396-428, optimization :463-480, nearest_so3 :42-59) and
// the per-frame glue of crates/apriltags/src/lib.rs:293-379.  All arithmetic is f64.
//
// Two waves per problem.  The work is ~0.3 MFLOP and latency-bound, so the mapping favours determinism over peak rate:
// every entry of Q_rr/Q_rt/Q_tt is accumulated by ONE lane over the points in index order (same sums as the scalar code,
// no atomics, no reduction tree); the 9x9 Jacobi eigen-decomposition runs its rotations with 9 lanes updating one
// row/column element each; the six SQP refinements run as six groups of 16 lanes, each lane holding one row of the 15x15
// KKT system in registers (LU with partial pivoting: pivot rows travel by shuffle, rows are never moved); the cheirality
// test of the winner selection is spread over a wave.  MFMA is deliberately not used: the only contraction,
// (9 x 3N)(3N x 9) with N <= 120, is 0.2 MFLOP — see DESIGN.md.
#include <math.h>
#include <string.h>

#include <vector>

#include "ck_internal.h"

namespace {

constexpr double XY_STD_DEV_SCALAR = 5.0, THETA_STD_DEV_SCALAR = 2.0, MAX_TRUSTABLE_RMS = 0.1, MAX_GYRO_DELTA = 30.0;
constexpr double TAG_SIZE = 0.1651, CORNER_DISTANCE = TAG_SIZE / 2.0, PI_D = 3.14159265358979323846;
constexpr double DBLMAX = 1.7976931348623157e308;

__device__ void quat_to_mat(const double q[4], double R[9]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double n = sqrt(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
__device__ void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ void mat3_vec(const double A[9], const double v[3], double o[3]) {
    for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
__device__ double mat3_det(const double m[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ int mat3_try_inverse(const double m[9], double o[9]) {
    double det = mat3_det(m);
    if (det == 0.0) return 0;
    o[0] = (m[4] * m[8] - m[5] * m[7]) / det; o[1] = (m[2] * m[7] - m[1] * m[8]) / det; o[2] = (m[1] * m[5] - m[2] * m[4]) / det;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / det; o[4] = (m[0] * m[8] - m[2] * m[6]) / det; o[5] = (m[2] * m[3] - m[0] * m[5]) / det;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / det; o[7] = (m[1] * m[6] - m[0] * m[7]) / det; o[8] = (m[0] * m[4] - m[1] * m[3]) / det;
    return 1;
}
// serial cyclic Jacobi for small symmetric matrices (used for the 3x3 cases)
__device__ void jacobi3(double A[9], double V[9], double w[3]) {
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0);
    double tot = 0; // same stop rule and summation order as the oracle's jacobi_eigen
    for (int i = 0; i < 9; i++) tot += A[i] * A[i];
    const double stop = 1e-32 * tot;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        off += A[1] * A[1]; off += A[2] * A[2]; off += A[5] * A[5];
        if (off <= stop) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p * 3 + q];
                if (fabs(apq) < 1e-300) continue;
                double app = A[p * 3 + p], aqq = A[q * 3 + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { double akp = A[k * 3 + p], akq = A[k * 3 + q]; A[k * 3 + p] = c * akp - s * akq; A[k * 3 + q] = s * akp + c * akq; }
                for (int k = 0; k < 3; k++) { double apk = A[p * 3 + k], aqk = A[q * 3 + k]; A[p * 3 + k] = c * apk - s * aqk; A[q * 3 + k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; k++) { double vkp = V[k * 3 + p], vkq = V[k * 3 + q]; V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq; }
            }
    }
    for (int i = 0; i < 3; i++) w[i] = A[i * 3 + i];
}
__device__ void svd3(const double M[9], double U[9], double s[3], double V[9]) {
    double MtM[9], Vt[9], w[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) MtM[i * 3 + j] = M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
    jacobi3(MtM, Vt, w);
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (w[idx[j]] > w[idx[i]]) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    for (int c = 0; c < 3; c++) {
        s[c] = sqrt(w[idx[c]] > 0 ? w[idx[c]] : 0);
        for (int r = 0; r < 3; r++) V[r * 3 + c] = Vt[r * 3 + idx[c]];
    }
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
// nearest rotation of a row-major 3x3 (U V^T with the chirality fix)
__device__ void polar_rotation(const double M[9], double out[9]) {
    double U[9], s[3], V[9], Vt[9];
    svd3(M, U, s, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Vt[i * 3 + j] = V[j * 3 + i];
    mat3_mul(U, Vt, out);
    if (mat3_det(out) < 0.0) {
        for (int r = 0; r < 3; r++) U[r * 3 + 2] = -U[r * 3 + 2];
        mat3_mul(U, Vt, out);
    }
}
__device__ void nearest_so3(const double r_vec[9], double out[9]) { // column-major in and out (lib.rs:42-59)
    double M[9], rot[9];
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) M[r * 3 + c] = r_vec[c * 3 + r];
    polar_rotation(M, rot);
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) out[c * 3 + r] = rot[r * 3 + c];
}

// One SQP refinement by a group of 16 lanes (lib.rs:98-115, 463-480): lane `gl` of the group owns one row of the
// 15x15 KKT system [[Omega, J^T], [J, 0]] in registers (lane 15 idles); r and the solution are replicated in every lane.
// LU with partial pivoting without moving rows: a lane remembers which logical row it holds (`lrow`), the pivot of a
// column is the unpivoted lane with the largest |entry| (smallest logical row on ties, like the sequential scan), its row
// is broadcast by shuffles and every other unpivoted lane eliminates in registers.  Each entry sees exactly the operations
// of the sequential code, in the same order, so the result is bit-identical to it.
// one step of an all-reduce over a DPP row (16 lanes): combine with the lane N places round the row (row_ror:N)
template <int N>
__device__ __forceinline__ void row_max_step(double &best, int &meta) {
    const long long bits = __double_as_longlong(best);
    const int lo = __builtin_amdgcn_update_dpp((int)bits, (int)bits, 0x120 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(bits >> 32), (int)(bits >> 32), 0x120 + N, 0xF, 0xF, false);
    const int om = __builtin_amdgcn_update_dpp(meta, meta, 0x120 + N, 0xF, 0xF, false);
    const double ob = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    if (ob > best || (ob == best && om < meta)) { best = ob; meta = om; }
}
__device__ double optimization16(int max_iter, double tol_sq, double r[9], const double *omega, int gl) {
    const int row = gl; // 0..14 own a row; 15 computes along on a zero row and is never a pivot
    for (int it = 0; it < max_iter; it++) {
        const double *c1 = r, *c2 = r + 3, *c3 = r + 6;
        double h[6];
        h[0] = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2] - 1.0;
        h[1] = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2] - 1.0;
        h[2] = c3[0] * c3[0] + c3[1] * c3[1] + c3[2] * c3[2] - 1.0;
        h[3] = c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
        h[4] = c1[0] * c3[0] + c1[1] * c3[1] + c1[2] * c3[2];
        h[5] = c2[0] * c3[0] + c2[1] * c3[1] + c2[2] * c3[2];
        // J (6x9), rows: 0:(2c1,0,0) 1:(0,2c2,0) 2:(0,0,2c3) 3:(c2,c1,0) 4:(c3,0,c1) 5:(0,c3,c2)
        double J[6][9];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) J[i][j] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            J[0][k] = 2.0 * c1[k]; J[1][3 + k] = 2.0 * c2[k]; J[2][6 + k] = 2.0 * c3[k];
            J[3][k] = c2[k]; J[3][3 + k] = c1[k];
            J[4][k] = c3[k]; J[4][6 + k] = c1[k];
            J[5][3 + k] = c3[k]; J[5][6 + k] = c2[k];
        }
        double A[15], b = 0.0;
#pragma unroll
        for (int j = 0; j < 15; j++) A[j] = 0.0;
        if (row < 9) {
            double sacc = 0;
#pragma unroll
            for (int j = 0; j < 9; j++) { const double o = omega[row * 9 + j]; A[j] = o; sacc += o * r[j]; }
            b = -sacc;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < 9; j++) v = (j == row) ? J[i][j] : v;
                A[9 + i] = v;
            }
        } else if (row < 15) {
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (i == row - 9) {
#pragma unroll
                    for (int j = 0; j < 9; j++) A[j] = J[i][j];
                    b = -h[i];
                }
        }
        int lrow = row;          // logical row currently held by this lane
        bool pivoted = row >= 15; // lane 15 never takes part
        bool singular = false;
#pragma unroll
        for (int col = 0; col < 15; col++) {
            // pivot: largest |A[.][col]| among unpivoted lanes, smallest logical row on ties (the sequential scan keeps the
            // first maximum because it only replaces on a strictly larger value)
            // The 16 lanes of a group are one DPP row: four rotations (by 8, 4, 2, 1) with this combiner leave the same winner
            // in every lane — the order (value descending, logical row ascending) is total over the unpivoted lanes, so the
            // reduction order does not matter — and cost register moves instead of sixteen trips through the LDS crossbar.
            double best = pivoted ? -1.0 : fabs(A[col]);
            int meta = ((pivoted ? 99 : lrow) << 8) | gl; // logical row, then the lane that holds it
            row_max_step<8>(best, meta); row_max_step<4>(best, meta); row_max_step<2>(best, meta); row_max_step<1>(best, meta);
            const int bl = meta >> 8, bs = meta & 0xFF;
            if (best == 0.0) { singular = true; break; }
            // the lane that held logical row `col` takes over the pivot lane's logical row (a swap, without moving data)
            if (!pivoted && lrow == col && gl != bs) lrow = bl;
            const bool is_piv = gl == bs;
            if (is_piv) { lrow = col; }
            double P[15];
#pragma unroll
            for (int k = col; k < 15; k++) P[k] = __shfl(A[k], bs, 16);
            const double pb = __shfl(b, bs, 16);
            if (is_piv) pivoted = true;
            else if (!pivoted) {
                const double f = A[col] / P[col];
                if (f != 0.0) {
#pragma unroll
                    for (int k = col; k < 15; k++) A[k] -= f * P[k];
                    b -= f * pb;
                }
            }
        }
        if (singular) break;
        // back substitution over logical rows 14..0; the owner of a row finishes it and broadcasts the unknown
        double x[15];
#pragma unroll
        for (int rr = 14; rr >= 0; rr--) {
            double sv = b;
#pragma unroll
            for (int k = rr + 1; k < 15; k++) sv -= A[k] * x[k];
            sv = sv / A[rr];
            const unsigned long long own = __ballot(lrow == rr && gl < 15);
            const int src = (int)(__builtin_ctzll((own >> (threadIdx.x & 48)) & 0xFFFFull)); // owner inside this group of 16
            x[rr] = __shfl(sv, src, 16);
        }
        double n2 = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) { r[k] += x[k]; n2 += x[k] * x[k]; }
        if (n2 < tol_sq) break;
    }
    double e = 0;
    for (int i = 0; i < 9; i++) {
        double sacc = 0;
        for (int j = 0; j < 9; j++) sacc += omega[i * 9 + j] * r[j];
        e += r[i] * sacc;
    }
    return e;
}

struct SolveArgs {
    ck_sqpnp_params_t prm;
    const ck_sqpnp_problem_t *problems;
    const ck_iso3_t *tags;
    const double *bearings;
    ck_sqpnp_result_t *out;
    int n;
    int max_points; // capacity of the per-problem world-point scratch
    double *world;  // [n][max_points][3]
    int stop_after; // diagnostics (CK_SQ_STOP_AFTER): 1 after Omega, 2 after the eigen-decomposition, 3 after the refinements
};
static int sq_stop_after() { static const int v = CK_KNOB("CK_SQ_STOP_AFTER", 99); return v; }

constexpr int SQ_NT = 128; // two waves: six candidate groups of 16 lanes in the refinement, 128-wide loops elsewhere
__global__ __launch_bounds__(SQ_NT) void k_sqpnp(SolveArgs a) {
    __shared__ double sQrr[81], sQrt[27], sQtt[9], sQttInv[9], sOmega[81], sA[81], sV[81], sW[9];
    __shared__ double sCandR[6][9], sCandE[6];
    __shared__ double sCentroid[3], sRot[2];
    __shared__ int sIdx[9];
    const int lane = threadIdx.x, pi = blockIdx.x;
    if (pi >= a.n) return;
    const ck_sqpnp_problem_t pr = a.problems[pi];
    ck_sqpnp_result_t *res = &a.out[pi];
    const int n_tags = pr.n_tags, n = 4 * pr.n_tags;
    if (lane == 0) { res->valid = 0; res->pad = 0; }
    if (n < 3 || n != pr.n_bearings || n > a.max_points) return; // lib.rs:255
    const ck_iso3_t *tags = a.tags + pr.tag_offset;
    const double *p2 = a.bearings + (size_t)3 * pr.bearing_offset;
    double *world = a.world + (size_t)pi * a.max_points * 3;
    const double cp[4][3] = {{0, -CORNER_DISTANCE, -CORNER_DISTANCE}, {0, CORNER_DISTANCE, -CORNER_DISTANCE},
                             {0, CORNER_DISTANCE, CORNER_DISTANCE}, {0, -CORNER_DISTANCE, CORNER_DISTANCE}};
    for (int i = lane; i < n; i += SQ_NT) { // corner_points_from_center (lib.rs:379-394)
        int t = i >> 2, c = i & 3;
        double R[9], p[3];
        quat_to_mat(tags[t].q, R);
        mat3_vec(R, cp[c], p);
        for (int k = 0; k < 3; k++) world[i * 3 + k] = p[k] + tags[t].t[k];
    }
    __syncthreads();
    if (lane < 3) { // centroid: index order, like the fold in lib.rs:259-260
        double s = 0;
        for (int i = 0; i < n; i++) s += world[i * 3 + lane];
        sCentroid[lane] = s / (double)n;
    }
    __syncthreads();
    // build_linear_system (lib.rs:124-180): entry e of [Q_rr(81) | Q_rt(27) | Q_tt(9)] belongs to one lane
    for (int e = lane; e < 117; e += SQ_NT) {
        double acc = 0;
        for (int k = 0; k < n; k++) {
            const double *v = p2 + 3 * k;
            double X[3] = {world[k * 3] - sCentroid[0], world[k * 3 + 1] - sCentroid[1], world[k * 3 + 2] - sCentroid[2]};
            double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            double inv = 1.0 / sq;
            if (e < 81) {
                int row = e / 9, col = e - row * 9;
                int ai = row / 3, i = row - ai * 3, bi = col / 3, j = col - bi * 3;
                double P = (i == j ? 1.0 : 0.0) - (v[i] * v[j]) * inv;
                acc += (P * X[ai]) * X[bi];
            } else if (e < 108) {
                int q = e - 81, row = q / 3, j = q - row * 3;
                int ai = row / 3, i = row - ai * 3;
                double P = (i == j ? 1.0 : 0.0) - (v[i] * v[j]) * inv;
                acc += P * X[ai];
            } else {
                int q = e - 108, i = q / 3, j = q - i * 3;
                acc += (i == j ? 1.0 : 0.0) - (v[i] * v[j]) * inv;
            }
        }
        if (e < 81) sQrr[e] = acc; else if (e < 108) sQrt[e - 81] = acc; else sQtt[e - 108] = acc;
    }
    __syncthreads();
    if (lane == 0) {
        double inv[9];
        if (!mat3_try_inverse(sQtt, inv)) for (int i = 0; i < 9; i++) inv[i] = 0.0; // unwrap_or_default (lib.rs:171)
        for (int i = 0; i < 9; i++) sQttInv[i] = inv[i];
    }
    __syncthreads();
    for (int e = lane; e < 81; e += SQ_NT) {
        int i = e / 9, j = e - i * 9;
        double t0 = sQrt[i * 3] * sQttInv[0] + sQrt[i * 3 + 1] * sQttInv[3] + sQrt[i * 3 + 2] * sQttInv[6];
        double t1 = sQrt[i * 3] * sQttInv[1] + sQrt[i * 3 + 1] * sQttInv[4] + sQrt[i * 3 + 2] * sQttInv[7];
        double t2 = sQrt[i * 3] * sQttInv[2] + sQrt[i * 3 + 1] * sQttInv[5] + sQrt[i * 3 + 2] * sQttInv[8];
        double om = sQrr[e] - (t0 * sQrt[j * 3] + t1 * sQrt[j * 3 + 1] + t2 * sQrt[j * 3 + 2]);
        sOmega[e] = om; sA[e] = om; sV[e] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (a.stop_after == 1) return;
    // symmetric eigen-decomposition of Omega: cyclic Jacobi, 9 lanes update one element of the rotated rows/columns.  The
    // first wave does it alone: a wave's LDS accesses execute in program order, so the three hand-overs of a rotation need
    // no workgroup barrier (36 rotations x ~10 sweeps x 3 barriers were a third of this kernel's dependency chain).
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    double jtot = 0; // Frobenius norm^2 of Omega: the sweeps stop at 1e-32 of it (oracle: jacobi_eigen)
    for (int e = 0; e < 81; e++) jtot += sA[e] * sA[e];
    const double jstop = 1e-32 * jtot;
    if (lane < 64)
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int i = 0; i < 9; i++)
            for (int j = i + 1; j < 9; j++) off += sA[i * 9 + j] * sA[i * 9 + j];
        if (off <= jstop) break;
        for (int p = 0; p < 9; p++)
            for (int q = p + 1; q < 9; q++) {
                double apq = sA[p * 9 + q];
                if (fabs(apq) < 1e-300) continue; // uniform: every lane reads the same LDS value
                double app = sA[p * 9 + p], aqq = sA[q * 9 + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                wave_sync();
                if (lane < 9) { int k = lane; double akp = sA[k * 9 + p], akq = sA[k * 9 + q]; sA[k * 9 + p] = c * akp - s * akq; sA[k * 9 + q] = s * akp + c * akq; }
                wave_sync();
                if (lane < 9) { int k = lane; double apk = sA[p * 9 + k], aqk = sA[q * 9 + k]; sA[p * 9 + k] = c * apk - s * aqk; sA[q * 9 + k] = s * apk + c * aqk; }
                if (lane < 9) { int k = lane; double vkp = sV[k * 9 + p], vkq = sV[k * 9 + q]; sV[k * 9 + p] = c * vkp - s * vkq; sV[k * 9 + q] = s * vkp + c * vkq; }
                wave_sync();
            }
    }
    __syncthreads();
    if (lane == 0) {
        for (int i = 0; i < 9; i++) { sW[i] = sA[i * 9 + i]; sIdx[i] = i; }
        for (int i = 1; i < 9; i++) { // stable ascending order of eigenvalues (lib.rs:400-401)
            int v = sIdx[i], j = i - 1;
            while (j >= 0 && sW[sIdx[j]] > sW[v]) { sIdx[j + 1] = sIdx[j]; j--; }
            sIdx[j + 1] = v;
        }
        sRot[0] = cos(pr.gyro); sRot[1] = sin(pr.gyro);
    }
    __syncthreads();
    if (a.stop_after == 2) return;
    double Rrc[9];
    quat_to_mat(pr.robot_to_cam.q, Rrc);
    const double fwd[3] = {Rrc[0], Rrc[3], Rrc[6]}; // column 0 (lib.rs:313-318)
    {   // solve_rotation_candidates (lib.rs:403-425): group g of 16 lanes = candidate 2*t + sign index
        const int g = lane >> 4, gl = lane & 15;
        if (g < 6) {
            int t = g >> 1;
            double sign = (g & 1) ? 1.0 : -1.0, guess[9], r[9];
            for (int k = 0; k < 9; k++) guess[k] = sV[k * 9 + sIdx[t]] * sign;
            nearest_so3(guess, r);
            double energy = optimization16(a.prm.max_iter, a.prm.tol_sq, r, sOmega, gl);
            double fx = r[0] * fwd[0] + r[1] * fwd[1] + r[2] * fwd[2];
            double fy = r[3] * fwd[0] + r[4] * fwd[1] + r[5] * fwd[2];
            double dot = fx * sRot[0] + fy * sRot[1];
            double ae = 1.0 - dot;
            if (ae < 0.0) ae = 0.0;
            energy += pr.sign_change_error * ae;
            if (gl == 0) {
                for (int k = 0; k < 9; k++) sCandR[g][k] = r[k];
                sCandE[g] = energy;
            }
        }
    }
    __syncthreads();
    if (a.stop_after == 3) return;
    if (lane >= 64) return; // the first wave picks the winner: every lane replays the (cheap) selection, the cheirality test
                            // over the points is spread over the lanes
    int order[6] = {0, 1, 2, 3, 4, 5};
    for (int i = 1; i < 6; i++) { // stable sort by penalised energy (lib.rs:427)
        int v = order[i], j = i - 1;
        while (j >= 0 && sCandE[order[j]] > sCandE[v]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    bool found = false;
    double best_score = DBLMAX, bestR[9], bestT[3], best_energy = 0;
    for (int oi = 0; oi < 6; oi++) {
        const double *r = sCandR[order[oi]];
        double Rm[9];
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 3; rr++) Rm[rr * 3 + c] = r[c * 3 + rr];
        double qtr[3], tl[3], Rc[3], t[3];
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int i = 0; i < 9; i++) s += sQrt[i * 3 + j] * r[i];
            qtr[j] = s;
        }
        mat3_vec(sQttInv, qtr, tl);
        mat3_vec(Rm, sCentroid, Rc);
        for (int k = 0; k < 3; k++) t[k] = -tl[k] - Rc[k];
        bool behind = false;
        for (int i = lane; i < n; i += 64) {
            double pc[3];
            mat3_vec(Rm, world + 3 * i, pc);
            if (!(pc[2] + t[2] > 0.0)) behind = true;
        }
        if (__ballot(behind)) continue; // uniform: all points must be in front (lib.rs:276-283)
        if (sCandE[order[oi]] < best_score) {
            best_score = sCandE[order[oi]];
            double e = 0;
            for (int i = 0; i < 9; i++) {
                double s = 0;
                for (int j = 0; j < 9; j++) s += sOmega[i * 9 + j] * r[j];
                e += r[i] * s;
            }
            best_energy = e;
            polar_rotation(Rm, bestR); // Rot3::from_matrix (lib.rs:289)
            for (int k = 0; k < 3; k++) bestT[k] = t[k];
            found = true;
        }
    }
    if (!found || lane != 0) return;
    // compute_std_devs (lib.rs:224-246)
    double distance = sqrt(bestT[0] * bestT[0] + bestT[1] * bestT[1] + bestT[2] * bestT[2]);
    {
        double n_points = (double)(n_tags * 4);
        double rms = sqrt(best_energy / n_points);
        if (rms > MAX_TRUSTABLE_RMS) { res->std_devs[0] = res->std_devs[1] = res->std_devs[2] = DBLMAX; }
        else {
            double mult = 1.0 + (distance / TAG_SIZE);
            double xy = ((rms * mult) / sqrt((double)n_tags)) * XY_STD_DEV_SCALAR;
            xy = xy < 0.01 ? 0.01 : (xy > 10.0 ? 10.0 : xy);
            double th = (((rms / TAG_SIZE) * mult) / sqrt((double)n_tags)) * THETA_STD_DEV_SCALAR;
            th = th < 0.05 ? 0.05 : (th > PI_D ? PI_D : th);
            res->std_devs[0] = xy; res->std_devs[1] = xy; res->std_devs[2] = th;
        }
    }
    // world_to_cam^-1 * robot_to_cam, then the yaw pivot about the tag centroid (lib.rs:328-376)
    double Rt[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[i * 3 + j] = bestR[j * 3 + i];
    double d[3] = {pr.robot_to_cam.t[0] - bestT[0], pr.robot_to_cam.t[1] - bestT[1], pr.robot_to_cam.t[2] - bestT[2]};
    double robot_pos[3], robot_rot[9];
    mat3_vec(Rt, d, robot_pos);
    mat3_mul(Rt, Rrc, robot_rot);
    double tc[3] = {0, 0, 0};
    for (int t = 0; t < n_tags; t++)
        for (int k = 0; k < 3; k++) tc[k] += tags[t].t[k];
    for (int k = 0; k < 3; k++) tc[k] /= (double)n_tags;
    double vision_yaw = atan2(robot_rot[3], robot_rot[0]);
    double delta_yaw = pr.gyro - vision_yaw;
    delta_yaw = fmod(delta_yaw + PI_D, 2.0 * PI_D);
    if (delta_yaw < 0) delta_yaw += 2.0 * PI_D;
    delta_yaw -= PI_D;
    double delta_deg = fabs(delta_yaw) * (180.0 / PI_D);
    double weight = delta_deg / MAX_GYRO_DELTA;
    weight = weight < 0 ? 0 : (weight > 1 ? 1 : weight);
    weight = weight * weight * (3.0 - 2.0 * weight);
    double applied = delta_yaw * weight;
    double cz = cos(applied), sz = sin(applied);
    double rotz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double rel[3] = {robot_pos[0] - tc[0], robot_pos[1] - tc[1], robot_pos[2] - tc[2]}, piv[3], R2[9];
    mat3_vec(rotz, rel, piv);
    mat3_mul(rotz, robot_rot, R2);
    for (int k = 0; k < 3; k++) res->pos[k] = tc[k] + piv[k];
    for (int k = 0; k < 9; k++) res->rot[k] = R2[k];
    double yaw = 0.0;
    if (fabs(R2[6]) < 1.0) { double pitch = -asin(R2[6]); double tcs = cos(pitch); yaw = atan2(R2[3] / tcs, R2[0] / tcs); }
    res->yaw = yaw;
    res->energy = best_energy;
    res->valid = 1;
}

__device__ __forceinline__ bool unproject_one(const ck_opencv5_t &c, double u, double v, double b[3]) {
    double xd = (u - c.cx) / c.fx, yd = (v - c.cy) / c.fy;
    double x = xd, y = yd;
    bool conv = false;
    for (int it = 0; it < 50; it++) {
        double r2 = x * x + y * y;
        double radial = 1.0 + r2 * (c.k1 + r2 * (c.k2 + r2 * c.k3));
        double dx = 2.0 * c.p1 * x * y + c.p2 * (r2 + 2.0 * x * x);
        double dy = c.p1 * (r2 + 2.0 * y * y) + 2.0 * c.p2 * x * y;
        double nx = (xd - dx) / radial, ny = (yd - dy) / radial;
        double ex = nx - x, ey = ny - y;
        x = nx; y = ny;
        if (ex * ex + ey * ey < 1e-24) { conv = true; break; }
    }
    double nrm = sqrt(x * x + y * y + 1.0);
    b[0] = x / nrm; b[1] = y / nrm; b[2] = 1.0 / nrm;
    return conv && isfinite(x) && isfinite(y);
}
__global__ void k_unproject(ck_opencv5_t cam, const double *px, int n, double *bearings, uint8_t *ok) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double b[3];
    bool good = unproject_one(cam, px[2 * i], px[2 * i + 1], b);
    bearings[3 * i] = b[0]; bearings[3 * i + 1] = b[1]; bearings[3 * i + 2] = b[2];
    ok[i] = good ? 1 : 0;
}

// AprilTags::process glue, one thread per frame: known-tag filter + unprojection -> one SQPnP problem per frame
struct GlueArgs {
    ck_stage_ws ws;
    ck_opencv5_t cam;
    ck_iso3_t robot_to_cam;
    const ck_field_tag_t *field; int n_field;
    const double *gyro; const uint8_t *has_gyro;
    double sign_change_error;
    const ck_dev_family *fams; int allow_unverified; // ids past a family's verified prefix are not upstream ids (ck_family_t.n_upstream)
    ck_sqpnp_problem_t *problems; ck_iso3_t *tags; double *bearings; // per frame: det_cap tags, 4*det_cap bearings
    int n;
};
__global__ void k_glue(GlueArgs a) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.n) return;
    const ck_stage_ws &ws = a.ws;
    uint32_t nd = ws.d_counters[(size_t)f * CK_CNT_STRIDE + CK_CNT_DETS];
    const ck_detection_t *dets = ws.d_dets + (size_t)f * ws.det_cap;
    ck_iso3_t *tags = a.tags + (size_t)f * ws.det_cap;
    double *bear = a.bearings + (size_t)f * ws.det_cap * 12;
    int nt = 0;
    if (nd > 0 && a.has_gyro[f]) {
        for (uint32_t i = 0; i < nd; i++) {
            int k = -1;
            for (int j = 0; j < a.n_field; j++)
                if (a.field[j].id == dets[i].id) { k = j; break; }
            if (k < 0) continue; // unknown tag (apriltags/src/lib.rs:306-308)
            if (!a.allow_unverified && (uint32_t)dets[i].id >= a.fams[dets[i].family].n_upstream) continue; // not an upstream id
            double b[12];
            bool ok = true;
            for (int c = 0; c < 4; c++) ok = unproject_one(a.cam, dets[i].p[c][0], dets[i].p[c][1], b + 3 * c) && ok;
            if (!ok) continue; // lib.rs:324
            tags[nt] = a.field[k].pose;
            for (int q = 0; q < 12; q++) bear[nt * 12 + q] = b[q];
            nt++;
        }
    }
    ck_sqpnp_problem_t pr;
    pr.n_tags = nt; pr.n_bearings = 4 * nt;
    pr.tag_offset = f * ws.det_cap; pr.bearing_offset = f * ws.det_cap * 4;
    pr.robot_to_cam = a.robot_to_cam; pr.gyro = a.gyro[f]; pr.sign_change_error = a.sign_change_error;
    a.problems[f] = pr;
}
__global__ void k_measure(ck_stage_ws ws, const ck_sqpnp_result_t *res, uint8_t camera_id, ck_vision_measurement_t *out, int32_t *valid, int n) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    ck_vision_measurement_t m;
    memset(&m, 0, sizeof m);
    m.camera_id = camera_id;
    int ok = res[f].valid;
    if (ok) {
        uint32_t nd = ws.d_counters[(size_t)f * CK_CNT_STRIDE + CK_CNT_DETS];
        m.pose_x = res[f].pos[0]; m.pose_y = res[f].pos[1]; m.pose_rot = res[f].yaw;
        m.std_x = res[f].std_devs[0]; m.std_y = res[f].std_devs[1]; m.std_rot = res[f].std_devs[2];
        m.tag_count = (uint8_t)(nd > 255 ? 255 : nd); // ALL detections (lib.rs:354)
    }
    out[f] = m;
    valid[f] = ok;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, sizeof(T) * (n ? n : 1)) == hipSuccess ? CK_OK : CK_ENOMEM; }
};

} // namespace

extern "C" int ck_sqpnp_solve_batch(ck_handle_t *h, const ck_sqpnp_params_t *params, const ck_sqpnp_problem_t *problems, int32_t n,
                                    const ck_iso3_t *tags, int32_t n_tags_total, const double *bearings, int32_t n_bearings_total,
                                    ck_sqpnp_result_t *out) {
    if (!h || !params || !problems || !out || n < 0 || n_tags_total < 0 || n_bearings_total < 0 || (n_tags_total > 0 && !tags) ||
        (n_bearings_total > 0 && !bearings)) return CK_EINVAL;
    if (n == 0) return CK_OK;
    int max_pts = 4;
    for (int i = 0; i < n; i++) {
        const ck_sqpnp_problem_t &p = problems[i];
        if (p.n_tags < 0 || p.n_bearings < 0 || p.tag_offset < 0 || p.bearing_offset < 0 || (int64_t)p.tag_offset + p.n_tags > n_tags_total ||
            (int64_t)p.bearing_offset + p.n_bearings > n_bearings_total) return CK_EINVAL;
        if (4 * p.n_tags > max_pts) max_pts = 4 * p.n_tags;
    }
    CK_HIP(hipSetDevice(h->device));
    DevBuf<ck_sqpnp_problem_t> dp; DevBuf<ck_iso3_t> dt; DevBuf<double> db, dw; DevBuf<ck_sqpnp_result_t> dr;
    if (dp.alloc((size_t)n) || dt.alloc((size_t)n_tags_total) || db.alloc((size_t)3 * n_bearings_total) || dw.alloc((size_t)n * max_pts * 3) || dr.alloc((size_t)n)) return CK_ENOMEM;
    CK_HIP(hipMemcpyAsync(dp.p, problems, sizeof(ck_sqpnp_problem_t) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    if (n_tags_total) CK_HIP(hipMemcpyAsync(dt.p, tags, sizeof(ck_iso3_t) * (size_t)n_tags_total, hipMemcpyHostToDevice, h->stream));
    if (n_bearings_total) CK_HIP(hipMemcpyAsync(db.p, bearings, sizeof(double) * 3 * (size_t)n_bearings_total, hipMemcpyHostToDevice, h->stream));
    SolveArgs a;
    a.prm = *params; a.problems = dp.p; a.tags = dt.p; a.bearings = db.p; a.out = dr.p; a.n = n; a.max_points = max_pts; a.world = dw.p; a.stop_after = sq_stop_after();
    hipLaunchKernelGGL(k_sqpnp, dim3((unsigned)n), dim3(SQ_NT), 0, h->stream, a);
    CK_HIP(hipGetLastError());
    CK_HIP(hipMemcpyAsync(out, dr.p, sizeof(ck_sqpnp_result_t) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    return CK_OK;
}

extern "C" void ck_sqpnp_create_solver_camera_transform(double fwd_m, double left_m, double up_m, double roll_deg, double pitch_deg,
                                                        double yaw_deg, ck_iso3_t *out) {
    // lib.rs:430-461 — construction-time host arithmetic (the reference calls it once in AprilTags::new)
    const double D2R = PI_D / 180.0;
    double r = roll_deg * D2R, p = pitch_deg * D2R, y = yaw_deg * D2R;
    double cr = cos(r * 0.5), sr = sin(r * 0.5), cp = cos(p * 0.5), sp = sin(p * 0.5), cy = cos(y * 0.5), sy = sin(y * 0.5);
    double qw = cr * cp * cy + sr * sp * sy, qx = sr * cp * cy - cr * sp * sy, qy = cr * sp * cy + sr * cp * sy, qz = cr * cp * sy - sr * sp * cy;
    double Rn[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                    2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                    2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
    const double C[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0}; // nwu -> cv
    double Rc[9], Ri[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rc[i * 3 + j] = Rn[i * 3] * C[j] + Rn[i * 3 + 1] * C[3 + j] + Rn[i * 3 + 2] * C[6 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ri[i * 3 + j] = Rc[j * 3 + i];
    double T[3] = {fwd_m, left_m, up_m};
    for (int i = 0; i < 3; i++) out->t[i] = -(Ri[i * 3] * T[0] + Ri[i * 3 + 1] * T[1] + Ri[i * 3 + 2] * T[2]);
    double tr = Ri[0] + Ri[4] + Ri[8], q[4];
    if (tr > 0) { double s = sqrt(tr + 1.0) * 2; q[0] = 0.25 * s; q[1] = (Ri[7] - Ri[5]) / s; q[2] = (Ri[2] - Ri[6]) / s; q[3] = (Ri[3] - Ri[1]) / s; }
    else if (Ri[0] > Ri[4] && Ri[0] > Ri[8]) { double s = sqrt(1.0 + Ri[0] - Ri[4] - Ri[8]) * 2; q[0] = (Ri[7] - Ri[5]) / s; q[1] = 0.25 * s; q[2] = (Ri[1] + Ri[3]) / s; q[3] = (Ri[2] + Ri[6]) / s; }
    else if (Ri[4] > Ri[8]) { double s = sqrt(1.0 + Ri[4] - Ri[0] - Ri[8]) * 2; q[0] = (Ri[2] - Ri[6]) / s; q[1] = (Ri[1] + Ri[3]) / s; q[2] = 0.25 * s; q[3] = (Ri[5] + Ri[7]) / s; }
    else { double s = sqrt(1.0 + Ri[8] - Ri[0] - Ri[4]) * 2; q[0] = (Ri[3] - Ri[1]) / s; q[1] = (Ri[2] + Ri[6]) / s; q[2] = (Ri[5] + Ri[7]) / s; q[3] = 0.25 * s; }
    for (int i = 0; i < 4; i++) out->q[i] = q[i];
}

extern "C" int ck_unproject_opencv5(const ck_opencv5_t *cam, const double *px, int32_t n, double *bearings, uint8_t *ok) {
    if (!cam || !px || !bearings || !ok || n < 0) return CK_EINVAL;
    if (n == 0) return CK_OK;
    if (ck_device_count() <= 0) return CK_ENODEVICE;
    DevBuf<double> dpx, db; DevBuf<uint8_t> dok;
    if (dpx.alloc((size_t)2 * n) || db.alloc((size_t)3 * n) || dok.alloc((size_t)n)) return CK_ENOMEM;
    CK_HIP(hipMemcpy(dpx.p, px, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_unproject, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, 0, *cam, dpx.p, n, db.p, dok.p);
    CK_HIP(hipGetLastError());
    CK_HIP(hipMemcpy(bearings, db.p, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    CK_HIP(hipMemcpy(ok, dok.p, (size_t)n, hipMemcpyDeviceToHost));
    return CK_OK;
}

// detect (already run) -> glue -> solve -> measurement, all on the device; only 64-byte records come back
int ck_run_pose(ck_handle *h, int n, const ck_process_params_t *pp, const double *gyro, const uint8_t *has_gyro,
                ck_vision_measurement_t *out, int32_t *valid, bool upload_field, bool sync) {
    ck_stage_ws &ws = h->ws;
    if (pp->n_field > ws.field_cap) return CK_ECAPACITY;
    // gyro / has_gyro / out / valid may be host or device pointers (hipMemcpyDefault resolves them)
    if (pp->n_field && upload_field) CK_HIP(hipMemcpyAsync(ws.d_field, pp->field, sizeof(ck_field_tag_t) * (size_t)pp->n_field, hipMemcpyDefault, h->stream));
    CK_HIP(hipMemcpyAsync(ws.d_gyro, gyro, sizeof(double) * (size_t)n, hipMemcpyDefault, h->stream));
    CK_HIP(hipMemcpyAsync(ws.d_has_gyro, has_gyro, (size_t)n, hipMemcpyDefault, h->stream));
    GlueArgs g;
    g.ws = ws; g.cam = pp->cam; g.robot_to_cam = pp->robot_to_cam; g.field = ws.d_field; g.n_field = pp->n_field; g.gyro = ws.d_gyro;
    g.fams = h->d_fams; g.allow_unverified = pp->allow_unverified_ids;
    g.has_gyro = ws.d_has_gyro; g.sign_change_error = pp->sign_change_error; g.problems = ws.d_problems; g.tags = ws.d_pose_tags;
    g.bearings = ws.d_bearings; g.n = n;
    hipLaunchKernelGGL(k_glue, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, g);
    SolveArgs a;
    a.prm = pp->sqpnp; a.problems = ws.d_problems; a.tags = ws.d_pose_tags; a.bearings = ws.d_bearings; a.out = ws.d_results; a.n = n;
    a.max_points = ws.det_cap * 4; a.world = ws.d_world; a.stop_after = sq_stop_after();
    hipLaunchKernelGGL(k_sqpnp, dim3((unsigned)n), dim3(SQ_NT), 0, h->stream, a);
    hipLaunchKernelGGL(k_measure, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, ws, ws.d_results, pp->camera_id, ws.d_meas, ws.d_valid, n);
    CK_HIP(hipGetLastError());
    CK_HIP(hipMemcpyAsync(out, ws.d_meas, sizeof(ck_vision_measurement_t) * (size_t)n, hipMemcpyDefault, h->stream));
    CK_HIP(hipMemcpyAsync(valid, ws.d_valid, sizeof(int32_t) * (size_t)n, hipMemcpyDefault, h->stream));
    if (sync) CK_HIP(hipStreamSynchronize(h->stream));
    return CK_OK;
}
