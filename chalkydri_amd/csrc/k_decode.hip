// k_decode.hip — homography, gray-model bit sampling, sharpening, codebook match, de-duplication.
//
// Replaces the decode stage of the external AprilTag-3 detector behind `detector.detect(&image)`
// (crates/apriltags/src/lib.rs:301); the outputs are exactly what the reference consumes: id() (:306),
// corners() (:310-314) and the number of detections (:302,354).  Bit-exact with oracle/detector.c
// (homography_compute, quad_decode, ora_decode_quads): every accumulation that the oracle does sequentially is
// done by one lane in the same order; order-free work (border and bit sampling, sharpening cells, codebook search)
// is spread over the wave.
#include "ck_internal.h"

namespace {

struct DecodeArgs {
    const uint8_t *im; int w, h, stride; size_t pitch;
    double sharpening; int max_hamming, n_families;
    const ck_dev_family *fams;
    ck_stage_ws ws;
    ck_detection_t *cands; int cand_cap; uint32_t *cand_count; // [n][cand_cap], [n]
};

// A: 72 doubles of workspace.  The elimination indexes it with run-time rows, so a private array would live in scratch
// memory; the caller passes LDS.
__device__ int homography_compute(const double corr[4][4], double *H, double *A) {
    for (int i = 0; i < 4; i++) {
        double x = corr[i][0], y = corr[i][1], u = corr[i][2], v = corr[i][3];
        double *r0 = &A[(2 * i) * 9], *r1 = &A[(2 * i + 1) * 9];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = v;
    }
    for (int col = 0; col < 8; col++) {
        double max_val = 0; int max_idx = -1;
        for (int row = col; row < 8; row++) {
            double val = fabs(A[row * 9 + col]);
            if (val > max_val) { max_val = val; max_idx = row; }
        }
        if (max_val < 1e-10) return 0;
        if (max_idx != col)
            for (int i = col; i < 9; i++) { double t = A[col * 9 + i]; A[col * 9 + i] = A[max_idx * 9 + i]; A[max_idx * 9 + i] = t; }
        for (int i = col + 1; i < 8; i++) {
            double f = A[i * 9 + col] / A[col * 9 + col];
            A[i * 9 + col] = 0;
            for (int j = col + 1; j < 9; j++) A[i * 9 + j] -= f * A[col * 9 + j];
        }
    }
    for (int col = 7; col >= 0; col--) {
        double sum = 0;
        for (int i = col + 1; i < 8; i++) sum += A[col * 9 + i] * A[i * 9 + 8];
        A[col * 9 + 8] = (A[col * 9 + 8] - sum) / A[col * 9 + col];
    }
    for (int i = 0; i < 8; i++) H[i] = A[i * 9 + 8];
    H[8] = 1.0;
    return 1;
}
// (Round 4: the same elimination spread over the lanes of the wave — every element update of a step by the lane that owns the element,
// pivot by ballot, the back substitution's sums in index order from the lanes' products — returned the same bits and measured SLOWER,
// 0.293 against 0.280 ms per 256 frames: eight steps of barriers and broadcasts cost more than one lane's walk through LDS.)
__device__ __forceinline__ void hproject(const double *H, double x, double y, double *ox, double *oy) {
    double xx = H[0] * x + H[1] * y + H[2];
    double yy = H[3] * x + H[4] * y + H[5];
    double zz = H[6] * x + H[7] * y + H[8];
    *ox = xx / zz; *oy = yy / zz;
}
__device__ __forceinline__ double value_for_pixel(const uint8_t *im, int w, int h, int stride, double px, double py) {
    double fx = px - 0.5, fy = py - 0.5;
    int x1 = (int)floor(fx), x2 = (int)ceil(fx);
    double x = fx - (double)x1;
    int y1 = (int)floor(fy), y2 = (int)ceil(fy);
    double y = fy - (double)y1;
    if (x1 < 0 || x2 >= w || y1 < 0 || y2 >= h) return -1.0;
    return (double)im[(size_t)y1 * stride + x1] * (1.0 - x) * (1.0 - y) + (double)im[(size_t)y1 * stride + x2] * x * (1.0 - y) +
           (double)im[(size_t)y2 * stride + x1] * (1.0 - x) * y + (double)im[(size_t)y2 * stride + x2] * x * y;
}
__device__ __forceinline__ unsigned long long code_rotate90(unsigned long long w, int nbits) {
    int p = nbits; unsigned long long l = 0;
    if (nbits % 4 == 1) { p = nbits - 1; l = 1; }
    w = ((w >> l) << (p / 4 + l)) | (w >> (3 * p / 4 + l) << l) | (w & l);
    w &= ((1ull << nbits) - 1);
    return w;
}

// one wave per (frame, quad); loops over families.  (Round 4: about 115 of a frame's 1024 quad slots are in use on the bench batch, and
// the kernel's time is the live quads' chains of dependent steps, some 50 us each at four waves per SIMD — not the dispatch of the
// idle workgroups: a grid of 128 / 256 / 512 workgroups per frame looping over the quads measured 0.39-0.40 against 0.33 ms, five
// waves per SIMD at 96 registers 0.32 against 0.29.)
__global__ __launch_bounds__(64) void k_decode(DecodeArgs a) {
    __shared__ double sH[9], sA[72], sSamp[128][3];
    __shared__ double sC[2][3];      // gray models: [0] white, [1] black
    __shared__ double sVal[256], sSharp[256];
    __shared__ int sOk;
    const int lane = threadIdx.x, frame = blockIdx.y, qi = blockIdx.x;
    const ck_stage_ws &ws = a.ws;
    const uint32_t nq = min(ws.d_counters[(size_t)frame * CK_CNT_STRIDE + CK_CNT_QUADS], (uint32_t)ws.quad_cap);
    if ((uint32_t)qi >= nq) return;
    const ck_quad_t q = ws.d_quads[(size_t)frame * ws.quad_cap + qi];
    const uint8_t *im = a.im + (size_t)frame * a.pitch;
    const int w = a.w, h = a.h, stride = a.stride;
    if (lane == 0) {
        double corr[4][4];
        for (int i = 0; i < 4; i++) {
            corr[i][0] = (i == 0 || i == 3) ? -1 : 1;
            corr[i][1] = (i == 0 || i == 1) ? -1 : 1;
            corr[i][2] = q.p[i][0]; corr[i][3] = q.p[i][1];
        }
        double H[9];
        int ok = homography_compute(corr, H, sA);
        for (int i = 0; i < 9; i++) sH[i] = H[i];
        sOk = ok;
    }
    __syncthreads();
    if (!sOk) return;
    for (int f = 0; f < a.n_families; f++) {
        const ck_dev_family &fam = a.fams[f];
        if ((fam.reversed_border != 0) != (q.reversed_border != 0)) continue;
        const double wb = (double)fam.width_at_border;
        const int tw = fam.total_width;
        __syncthreads();
        // gray models.  The 8 * width_at_border border samples are independent (one image read each): every lane takes
        // samples, then lane 0 adds up the white ones and lane 1 the black ones from LDS in the oracle's order.
        {
            const double patterns[8][5] = {
                {-0.5, 0.5, 0, 1, 1}, {0.5, 0.5, 0, 1, 0}, {wb + 0.5, 0.5, 0, 1, 1}, {wb - 0.5, 0.5, 0, 1, 0},
                {0.5, -0.5, 1, 0, 1}, {0.5, 0.5, 1, 0, 0}, {0.5, wb + 0.5, 1, 0, 1}, {0.5, wb - 0.5, 1, 0, 0}};
            const int wab = fam.width_at_border, nsamp = 8 * wab; // <= 128: total_width <= 16
            for (int j = lane; j < nsamp; j += 64) {
                const int pi = j / wab, i = j - pi * wab;
                double tagx01 = (patterns[pi][0] + (double)i * patterns[pi][2]) / wb;
                double tagy01 = (patterns[pi][1] + (double)i * patterns[pi][3]) / wb;
                double tagx = 2.0 * (tagx01 - 0.5), tagy = 2.0 * (tagy01 - 0.5);
                double px, py;
                hproject(sH, tagx, tagy, &px, &py);
                int ix = (int)px, iy = (int)py;
                double gray = -1.0; // outside the image: the sample is skipped
                if (!(px < 0 || py < 0 || ix < 0 || iy < 0 || ix >= w || iy >= h)) gray = (double)im[(size_t)iy * stride + ix];
                sSamp[j][0] = tagx; sSamp[j][1] = tagy; sSamp[j][2] = gray;
            }
            __syncthreads();
            if (lane < 2) {
                const int want_white = (lane == 0);
                double A00 = 0, A01 = 0, A02 = 0, A11 = 0, A12 = 0, A22 = 0, B0 = 0, B1 = 0, B2 = 0;
                for (int pi = 0; pi < 8; pi++) {
                    int is_white = patterns[pi][4] != 0;
                    if (is_white != want_white) continue;
                    for (int i = 0; i < wab; i++) {
                        const double tagx = sSamp[pi * wab + i][0], tagy = sSamp[pi * wab + i][1], gray = sSamp[pi * wab + i][2];
                        if (gray < 0) continue;
                        A00 += tagx * tagx; A01 += tagx * tagy; A02 += tagx;
                        A11 += tagy * tagy; A12 += tagy; A22 += 1;
                        B0 += tagx * gray; B1 += tagy * gray; B2 += gray;
                    }
                }
                double l00 = sqrt(A00);
                double l10 = A01 / l00, l20 = A02 / l00;
                double l11 = sqrt(A11 - l10 * l10);
                double l21 = (A12 - l10 * l20) / l11;
                double l22 = sqrt(A22 - l20 * l20 - l21 * l21);
                double y0 = B0 / l00;
                double y1 = (B1 - l10 * y0) / l11;
                double y2 = (B2 - l20 * y0 - l21 * y1) / l22;
                double c2 = y2 / l22;
                double c1 = (y1 - l21 * c2) / l11;
                double c0 = (y0 - l10 * c1 - l20 * c2) / l00;
                sC[1 - want_white][0] = c0; sC[1 - want_white][1] = c1; sC[1 - want_white][2] = c2;
            }
        }
        for (int i = lane; i < tw * tw; i += 64) sVal[i] = 0.0;
        __syncthreads();
        // (white(0,0) - black(0,0) < 0) != reversed_border -> reject
        {
            double wv = sC[0][0] * 0.0 + sC[0][1] * 0.0 + sC[0][2];
            double bv = sC[1][0] * 0.0 + sC[1][1] * 0.0 + sC[1][2];
            if (((wv - bv) < 0) != (fam.reversed_border != 0)) continue;
        }
        const int min_coord = (fam.width_at_border - tw) / 2;
        for (uint32_t i = lane; i < fam.nbits; i += 64) {
            int bx = (int)fam.bit_x[i], by = (int)fam.bit_y[i];
            double tagx = 2.0 * (((double)bx + 0.5) / wb - 0.5), tagy = 2.0 * (((double)by + 0.5) / wb - 0.5);
            double px, py;
            hproject(sH, tagx, tagy, &px, &py);
            double v = value_for_pixel(im, w, h, stride, px, py);
            if (v == -1.0) continue;
            double bm = sC[1][0] * tagx + sC[1][1] * tagy + sC[1][2];
            double wm = sC[0][0] * tagx + sC[0][1] * tagy + sC[0][2];
            double thr = (bm + wm) / 2.0;
            sVal[tw * (by - min_coord) + bx - min_coord] = v - thr;
        }
        __syncthreads();
        for (int i = lane; i < tw * tw; i += 64) {
            int y = i / tw, x = i - y * tw;
            double s = 0.0;
            if (y > 0) s += -sVal[(y - 1) * tw + x];
            if (x > 0) s += -sVal[y * tw + x - 1];
            s += 4.0 * sVal[y * tw + x];
            if (x < tw - 1) s += -sVal[y * tw + x + 1];
            if (y < tw - 1) s += -sVal[(y + 1) * tw + x];
            sSharp[i] = s;
        }
        __syncthreads();
        for (int i = lane; i < tw * tw; i += 64) sVal[i] = sVal[i] + a.sharpening * sSharp[i];
        __syncthreads();
        // The code word and the two scores are sums in bit order (the oracle's): lane i brings bit i's value (its cell of the
        // sharpened grid; nbits <= 64), every lane adds them up in that order from the lanes' registers — as one lane's loop over the
        // family's bit tables in global memory and the grid in LDS this was a chain of 36 memory round trips.
        unsigned long long rcode = 0;
        double margin;
        {
            double myv = 0.0;
            if ((uint32_t)lane < fam.nbits) myv = sVal[((int)fam.bit_y[lane] - min_coord) * tw + (int)fam.bit_x[lane] - min_coord];
            double black_score = 0, white_score = 0, black_cnt = 1, white_cnt = 1;
            for (uint32_t i = 0; i < fam.nbits; i++) {
                const double v = __shfl(myv, (int)i, 64);
                rcode <<= 1;
                if (v > 0) { white_score += v; white_cnt += 1; rcode |= 1; }
                else { black_score -= v; black_cnt += 1; }
            }
            const double ma = white_score / white_cnt, mb = black_score / black_cnt;
            margin = ma < mb ? ma : mb;
        }
        // codebook search: minimum (hamming, rotation, id).  A lane's codes of a pass are asked for together and meet all four
        // rotations of the code word in registers (one memory round trip per pass of 64 x DEC_CPL codes, not one per code and rotation).
        uint32_t bestkey = 0xFFFFFFFFu;
        {
            unsigned long long rc4[4];
            rc4[0] = rcode;
#pragma unroll
            for (int rot = 1; rot < 4; rot++) rc4[rot] = code_rotate90(rc4[rot - 1], (int)fam.nbits);
            constexpr int DEC_CPL = 12;
            for (uint32_t k0 = 0; k0 < fam.ncodes; k0 += 64u * DEC_CPL) {
                unsigned long long cw[DEC_CPL];
#pragma unroll
                for (int j = 0; j < DEC_CPL; j++) { const uint32_t k = k0 + (uint32_t)lane + 64u * j; cw[j] = k < fam.ncodes ? fam.codes[k] : 0ull; }
#pragma unroll
                for (int rot = 0; rot < 4; rot++)
#pragma unroll
                    for (int j = 0; j < DEC_CPL; j++) {
                        const uint32_t k = k0 + (uint32_t)lane + 64u * j;
                        const uint32_t hd = (uint32_t)__popcll(rc4[rot] ^ cw[j]);
                        const uint32_t key = (hd << 24) | ((uint32_t)rot << 20) | k;
                        bestkey = min(bestkey, k < fam.ncodes ? key : 0xFFFFFFFFu);
                    }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) bestkey = min(bestkey, (uint32_t)__shfl_xor((int)bestkey, d, 64));
        }
        const int hd = (int)(bestkey >> 24), rot = (int)((bestkey >> 20) & 3), id = (int)(bestkey & 0xFFFFFu);
        if (hd > a.max_hamming) continue;
        if (!(margin >= 0)) continue;
        if (lane == 0) {
            ck_detection_t d;
            d.id = id; d.hamming = hd; d.family = f; d.decision_margin = (float)margin;
            const double cs[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
            double c = cs[rot][0], s = cs[rot][1];
            double Hr[9];
            for (int r = 0; r < 3; r++) {
                Hr[r * 3 + 0] = c * sH[r * 3 + 0] + s * sH[r * 3 + 1];
                Hr[r * 3 + 1] = -s * sH[r * 3 + 0] + c * sH[r * 3 + 1];
                Hr[r * 3 + 2] = sH[r * 3 + 2];
            }
            hproject(Hr, 0, 0, &d.c[0], &d.c[1]);
            const double tc[4][2] = {{-1, 1}, {1, 1}, {1, -1}, {-1, -1}};
            for (int i = 0; i < 4; i++) hproject(Hr, tc[i][0], tc[i][1], &d.p[i][0], &d.p[i][1]);
            uint32_t pos = atomicAdd(&a.cand_count[frame], 1u);
            if (pos < (uint32_t)a.cand_cap) a.cands[(size_t)frame * a.cand_cap + pos] = d;
        }
    }
}

__device__ __forceinline__ int det_cmp(const ck_detection_t &x, const ck_detection_t &y) {
    if (x.family != y.family) return x.family < y.family ? -1 : 1;
    if (x.id != y.id) return x.id < y.id ? -1 : 1;
    if (x.hamming != y.hamming) return x.hamming < y.hamming ? -1 : 1;
    if (x.decision_margin != y.decision_margin) return x.decision_margin > y.decision_margin ? -1 : 1;
    if (x.c[0] != y.c[0]) return x.c[0] < y.c[0] ? -1 : 1;
    if (x.c[1] != y.c[1]) return x.c[1] < y.c[1] ? -1 : 1;
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 2; k++)
            if (x.p[i][k] != y.p[i][k]) return x.p[i][k] < y.p[i][k] ? -1 : 1;
    return 0;
}
__device__ __forceinline__ int point_in_quad(const double q[4][2], double x, double y) {
    int pos = 0, neg = 0;
    for (int i = 0; i < 4; i++) {
        int j = (i + 1) & 3;
        double cr = (q[j][0] - q[i][0]) * (y - q[i][1]) - (q[j][1] - q[i][1]) * (x - q[i][0]);
        if (cr > 0) pos++; else if (cr < 0) neg++;
    }
    return pos == 0 || neg == 0;
}

// one workgroup per frame: total order on the candidates, then sequential duplicate removal in that order
constexpr int FNT = 256;
__global__ __launch_bounds__(FNT) void k_finalize(DecodeArgs a) {
    extern __shared__ int sOrder[]; // [cand_cap]
    const int frame = blockIdx.x, tid = threadIdx.x;
    const ck_stage_ws &ws = a.ws;
    uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
    uint32_t n = a.cand_count[frame];
    if (n > (uint32_t)a.cand_cap) { n = (uint32_t)a.cand_cap; if (tid == 0) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_DETS_OVERFLOW); }
    const ck_detection_t *cand = a.cands + (size_t)frame * a.cand_cap;
    ck_detection_t *out = ws.d_dets + (size_t)frame * ws.det_cap;
    for (uint32_t i = tid; i < n; i += FNT) {
        int rank = 0;
        for (uint32_t j = 0; j < n; j++) {
            if (j == i) continue;
            int c = det_cmp(cand[j], cand[i]);
            if (c < 0 || (c == 0 && j < i)) rank++;
        }
        sOrder[rank] = (int)i;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t kept = 0;
        bool overflow = false, unverified = false;
        for (uint32_t r = 0; r < n; r++) {
            const ck_detection_t &d = cand[sOrder[r]];
            bool dup = false;
            for (uint32_t j = 0; j < kept && !dup; j++) {
                const ck_detection_t &k = out[j];
                if (k.family != d.family || k.id != d.id) continue;
                if (point_in_quad(k.p, d.c[0], d.c[1]) || point_in_quad(d.p, k.c[0], k.c[1])) dup = true;
            }
            if (dup) continue;
            if (kept >= (uint32_t)ws.det_cap) { overflow = true; break; }
            out[kept++] = d;
            unverified = unverified || (uint32_t)d.id >= a.fams[d.family].n_upstream;
        }
        counters[CK_CNT_DETS] = kept;
        if (unverified) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_UNVERIFIED_ID);
        if (overflow) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_DETS_OVERFLOW);
    }
}

} // namespace

int ck_launch_decode(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n) {
    ck_stage_ws &ws = h->ws;
    DecodeArgs a;
    a.im = frames; a.w = h->w; a.h = h->h; a.stride = stride; a.pitch = pitch;
    a.sharpening = h->cfg.decode_sharpening; a.max_hamming = h->cfg.max_hamming; a.n_families = h->cfg.n_families;
    a.fams = h->d_fams; a.ws = ws;
    // candidate buffer lives behind the work lists in the fit scratch; the per-frame counts were zeroed with the cluster tables
    const ck_fit_layout fl = ck_fit_scratch_layout(ws, h->cfg.max_batch);
    a.cand_cap = ws.quad_cap * h->cfg.n_families;
    a.cand_count = fl.cand_count;
    a.cands = fl.cands;
    hipLaunchKernelGGL(k_decode, dim3((unsigned)ws.quad_cap, (unsigned)n), dim3(64), 0, h->stream, a);
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)n), dim3(FNT), sizeof(int) * (size_t)a.cand_cap, h->stream, a);
    CK_HIP(hipGetLastError());
    return CK_OK;
}
