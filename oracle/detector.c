/*
 * detector.c — CPU oracle for the AprilTag detect/decode stages.  TEST INFRASTRUCTURE ONLY (see ck_oracle.h).
 *
 * PARITY UNPINNED: the production detector of the reference is the external AprilTag-3 C library [EXT]
 * (apriltag-sys, git branch master, unpinned) called at crates/apriltags/src/lib.rs:301; it is not under
 * /root/reference and the reference has no golden vectors for it.  This file restates AprilTag-3's published
 * stage structure (threshold → union-find segmentation → gradient clusters → quad fit → edge refinement →
 * homography decode → de-duplication) with the defaults the reference inherits.
 * The only in-tree reference code for a stage is CAT's connected_components
 * (crates/chalkydri-apriltags/src/lib.rs:501-549), whose connectivity rule ora_segment() follows exactly.
 *
 * Deliberate integer-exact choices (so a GPU implementation can be bit-identical):
 *   - canonical component label = smallest pixel index of the component;
 *   - points of a cluster are ordered by an exact 60-bit angular key (octant + 30-bit ratio + x + y), not by
 *     a float slope;
 *   - line-fit moments are int64 prefix sums over half-pixel coordinates with integer weights
 *     W = isqrt(gx^2+gy^2)+1 (AprilTag-3 uses double sums with W = sqrt(..)+1); the line fit divides once (reciprocal) and multiplies;
 *   - the edge-refinement line normal uses the closed-form eigenvector (AprilTag-3: atan2f/cosf/sinf);
 *   - codebook lookup is a brute-force minimum-Hamming search over ids x 4 rotations (AprilTag-3: hash table);
 *   - duplicates are detections of one (family,id) whose centre lies inside the other's quad.
 * Floating point: only + - * / sqrt on doubles in a fixed order; compile with -ffp-contract=off.
 */
#include "ck_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CK_INVALID_LABEL 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------------- */
void ora_decimate(const uint8_t *img, int w, int h, int stride, int f, uint8_t *out) {
    int ow = w / f, oh = h / f;
    for (int y = 0; y < oh; y++)
        for (int x = 0; x < ow; x++) out[(size_t)y * ow + x] = img[(size_t)(y * f) * stride + x * f];
}

/* Adaptive threshold: 4x4 tile min/max, 3x3 tile dilation, per-pixel tri-state. */
void ora_threshold(const uint8_t *img, int w, int h, int stride, int min_wb_diff, uint8_t *out) {
    const int ts = 4;
    int tw = w / ts, th = h / ts;
    uint8_t *tmin = (uint8_t *)malloc((size_t)tw * th), *tmax = (uint8_t *)malloc((size_t)tw * th);
    uint8_t *dmin = (uint8_t *)malloc((size_t)tw * th), *dmax = (uint8_t *)malloc((size_t)tw * th);
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++) {
            uint8_t mn = 255, mx = 0;
            for (int dy = 0; dy < ts; dy++)
                for (int dx = 0; dx < ts; dx++) {
                    uint8_t v = img[(size_t)(ty * ts + dy) * stride + tx * ts + dx];
                    if (v < mn) mn = v;
                    if (v > mx) mx = v;
                }
            tmin[ty * tw + tx] = mn; tmax[ty * tw + tx] = mx;
        }
    for (int ty = 0; ty < th; ty++)
        for (int tx = 0; tx < tw; tx++) {
            uint8_t mn = 255, mx = 0;
            for (int dy = -1; dy <= 1; dy++) {
                if (ty + dy < 0 || ty + dy >= th) continue;
                for (int dx = -1; dx <= 1; dx++) {
                    if (tx + dx < 0 || tx + dx >= tw) continue;
                    uint8_t a = tmin[(ty + dy) * tw + tx + dx], b = tmax[(ty + dy) * tw + tx + dx];
                    if (a < mn) mn = a;
                    if (b > mx) mx = b;
                }
            }
            dmin[ty * tw + tx] = mn; dmax[ty * tw + tx] = mx;
        }
    for (int y = 0; y < h; y++) {
        int ty = y / ts; if (ty > th - 1) ty = th - 1;
        for (int x = 0; x < w; x++) {
            int tx = x / ts; if (tx > tw - 1) tx = tw - 1;   /* ragged right/bottom edge uses the nearest tile */
            int mn = dmin[ty * tw + tx], mx = dmax[ty * tw + tx];
            uint8_t o;
            if (mx - mn < min_wb_diff) o = 127;
            else {
                int thr = mn + (mx - mn) / 2;
                o = img[(size_t)y * stride + x] > thr ? 255 : 0;
            }
            out[(size_t)y * w + x] = o;
        }
    }
    free(tmin); free(tmax); free(dmin); free(dmax);
}

/* Union-find segmentation.  Connectivity exactly as CAT's connected_components
 * (crates/chalkydri-apriltags/src/lib.rs:506-545): origin pixels x in [1, w-2]; same-value left and up
 * neighbours are joined; white (255) pixels are also joined to up-left and up-right. */
static uint32_t uf_find(uint32_t *parent, uint32_t i) {
    while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
    return i;
}
static void uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) parent[b] = a; else parent[a] = b;   /* root = smallest index */
}
void ora_segment(const uint8_t *t, int w, int h, uint32_t *labels, uint32_t *sizes) {
    size_t n = (size_t)w * h;
    uint32_t *parent = (uint32_t *)malloc(n * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) parent[i] = (uint32_t)i;
    for (int y = 0; y < h; y++)
        for (int x = 1; x < w - 1; x++) {
            uint32_t i = (uint32_t)(y * w + x);
            uint8_t v = t[i];
            if (v == 127) continue;
            if (t[i - 1] == v) uf_union(parent, i, i - 1);
            if (y > 0) {
                if (t[i - w] == v) uf_union(parent, i, i - w);
                if (v == 255) {
                    if (t[i - w - 1] == v) uf_union(parent, i, i - w - 1);
                    if (t[i - w + 1] == v) uf_union(parent, i, i - w + 1);
                }
            }
        }
    uint32_t *cnt = sizes ? (uint32_t *)calloc(n, sizeof(uint32_t)) : NULL;
    for (size_t i = 0; i < n; i++) {
        if (t[i] == 127) { labels[i] = CK_INVALID_LABEL; continue; }
        labels[i] = uf_find(parent, (uint32_t)i);
        if (cnt) cnt[labels[i]]++;
    }
    if (sizes)
        for (size_t i = 0; i < n; i++) sizes[i] = (labels[i] == CK_INVALID_LABEL) ? 0 : cnt[labels[i]];
    free(cnt); free(parent);
}

void ora_threshold_segment(const uint8_t *img, int w, int h, int stride, int min_wb_diff, uint8_t *thresh,
                           uint32_t *labels, uint32_t *sizes) {
    ora_threshold(img, w, h, stride, min_wb_diff, thresh);
    ora_segment(thresh, w, h, labels, sizes);
}

/* ------------------------------------------------------------------------------------------------- */
/* Gradient clusters: one point per 8-neighbour pair of opposite colour whose components both have at
 * least min_component_px pixels; key = (min label, max label). */
typedef struct { uint64_t key; uint32_t emit; ck_cluster_point_t p; } rawpt_t;
static int rawpt_cmp(const void *a, const void *b) {
    const rawpt_t *x = (const rawpt_t *)a, *y = (const rawpt_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->emit != y->emit) return x->emit < y->emit ? -1 : 1;
    return 0;
}
int ora_clusters(const uint8_t *t, const uint32_t *labels, const uint32_t *sizes, int w, int h, int min_comp,
                 ck_cluster_t *clusters, int cluster_cap, int *n_clusters, ck_cluster_point_t *points,
                 int point_cap, int *n_points) {
    static const int off[4][2] = {{1, 0}, {0, 1}, {-1, 1}, {1, 1}};
    size_t cap = 1 << 16, n = 0;
    rawpt_t *raw = (rawpt_t *)malloc(cap * sizeof(rawpt_t));
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            int i = y * w + x;
            int v0 = t[i];
            if (v0 == 127) continue;
            if ((int)sizes[i] < min_comp) continue;
            uint32_t rep0 = labels[i];
            for (int k = 0; k < 4; k++) {
                int dx = off[k][0], dy = off[k][1];
                int j = (y + dy) * w + x + dx;
                int v1 = t[j];
                if (v0 + v1 != 255) continue;
                if ((int)sizes[j] < min_comp) continue;
                uint32_t rep1 = labels[j];
                uint32_t a = rep0 < rep1 ? rep0 : rep1, b = rep0 < rep1 ? rep1 : rep0;
                if (n == cap) { cap *= 2; raw = (rawpt_t *)realloc(raw, cap * sizeof(rawpt_t)); }
                rawpt_t *r = &raw[n++];
                r->key = ((uint64_t)a << 32) | b;
                r->emit = ((uint32_t)i << 2) | (uint32_t)k;
                r->p.x = (uint16_t)(2 * x + dx); r->p.y = (uint16_t)(2 * y + dy);
                int s = v1 - v0 > 0 ? 1 : -1;
                r->p.gx = (int8_t)(dx * s); r->p.gy = (int8_t)(dy * s); r->p.pad = 0;
            }
        }
    qsort(raw, n, sizeof(rawpt_t), rawpt_cmp);
    int nc = 0, overflow = 0;
    size_t np = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && raw[j].key == raw[i].key) j++;
        if (nc >= cluster_cap || np + (j - i) > (size_t)point_cap) { overflow = 1; break; }
        clusters[nc].rep0 = (uint32_t)(raw[i].key >> 32); clusters[nc].rep1 = (uint32_t)raw[i].key;
        clusters[nc].start = (uint32_t)np; clusters[nc].count = (uint32_t)(j - i);
        for (size_t k = i; k < j; k++) points[np++] = raw[k].p;
        nc++; i = j;
    }
    *n_clusters = nc; *n_points = (int)np;
    free(raw);
    return overflow;
}

/* ------------------------------------------------------------------------------------------------- */
/* Quad fitting */
typedef struct { int64_t Mx, My, Mxx, Mxy, Myy, W; } lfps_t;

/* exact angular key of a point about the (offset) bounding-box centre; see header comment */
static uint64_t angle_key(int x, int y, int xmin, int xmax, int ymin, int ymax) {
    int64_t dx = 4 * (int64_t)x - 2 * ((int64_t)xmin + xmax) - 1;   /* odd, never 0 */
    int64_t dy = 4 * (int64_t)y - 2 * ((int64_t)ymin + ymax) + 1;   /* odd, never 0 */
    int64_t ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    int oct, inv;
    int64_t num, den;
    if (dy < 0) {
        if (dx < 0) { if (ay <= ax) { oct = 0; inv = 0; num = ay; den = ax; } else { oct = 1; inv = 1; num = ax; den = ay; } }
        else        { if (ay > ax)  { oct = 2; inv = 0; num = ax; den = ay; } else { oct = 3; inv = 1; num = ay; den = ax; } }
    } else {
        if (dx > 0) { if (ay <= ax) { oct = 4; inv = 0; num = ay; den = ax; } else { oct = 5; inv = 1; num = ax; den = ay; } }
        else        { if (ay > ax)  { oct = 6; inv = 0; num = ax; den = ay; } else { oct = 7; inv = 1; num = ay; den = ax; } }
    }
    uint64_t frac = ((uint64_t)num << 30) / (uint64_t)den;          /* in [0, 2^30] */
    if (inv) frac = ((uint64_t)1 << 30) - frac;
    return ((uint64_t)oct << 57) | (frac << 26) | ((uint64_t)x << 13) | (uint64_t)y;
}
static int u64_cmp(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y ? 1 : 0;
}
static uint32_t isqrt_u32(uint32_t v) {
    uint32_t r = (uint32_t)sqrt((double)v);
    while ((uint64_t)r * r > v) r--;
    while ((uint64_t)(r + 1) * (r + 1) <= v) r++;
    return r;
}

static void fit_line(const lfps_t *lf, int sz, int i0, int i1, double *lineparm, double *err, double *mse) {
    int64_t Mx, My, Mxx, Mxy, Myy, W;
    int N;
    if (i0 < i1) {
        N = i1 - i0 + 1;
        Mx = lf[i1].Mx; My = lf[i1].My; Mxx = lf[i1].Mxx; Mxy = lf[i1].Mxy; Myy = lf[i1].Myy; W = lf[i1].W;
        if (i0 > 0) {
            Mx -= lf[i0 - 1].Mx; My -= lf[i0 - 1].My; Mxx -= lf[i0 - 1].Mxx; Mxy -= lf[i0 - 1].Mxy;
            Myy -= lf[i0 - 1].Myy; W -= lf[i0 - 1].W;
        }
    } else { /* wraps around the end of the array; i0 >= 1 here */
        Mx = lf[sz - 1].Mx - lf[i0 - 1].Mx + lf[i1].Mx; My = lf[sz - 1].My - lf[i0 - 1].My + lf[i1].My;
        Mxx = lf[sz - 1].Mxx - lf[i0 - 1].Mxx + lf[i1].Mxx; Mxy = lf[sz - 1].Mxy - lf[i0 - 1].Mxy + lf[i1].Mxy;
        Myy = lf[sz - 1].Myy - lf[i0 - 1].Myy + lf[i1].Myy; W = lf[sz - 1].W - lf[i0 - 1].W + lf[i1].W;
        N = sz - i0 + i1 + 1;
    }
    /* moments are in half-pixel units: x_px = X/2 */
    /* one reciprocal, five products (a design choice shared by the HIP kernel; AprilTag-3 divides five times) */
    double inv = 1.0 / (double)W;
    double Ex = (0.5 * (double)Mx) * inv;
    double Ey = (0.5 * (double)My) * inv;
    double Cxx = (0.25 * (double)Mxx) * inv - Ex * Ex;
    double Cxy = (0.25 * (double)Mxy) * inv - Ex * Ey;
    double Cyy = (0.25 * (double)Myy) * inv - Ey * Ey;
    double d = Cxx - Cyy;
    double q = 4.0 * Cxy;
    double disc = sqrt(d * d + q * Cxy);
    double tr = Cxx + Cyy;
    double eig_small = 0.5 * (tr - disc);
    if (lineparm) {
        lineparm[0] = Ex; lineparm[1] = Ey;
        double eig = 0.5 * (tr + disc);
        double nx1 = Cxx - eig, ny1 = Cxy;
        double M1 = nx1 * nx1 + ny1 * ny1;
        double nx2 = Cxy, ny2 = Cyy - eig;
        double M2 = nx2 * nx2 + ny2 * ny2;
        double nx, ny, M;
        if (M1 > M2) { nx = nx1; ny = ny1; M = M1; } else { nx = nx2; ny = ny2; M = M2; }
        double length = sqrt(M);
        if (length < 1e-12) { lineparm[2] = 0; lineparm[3] = 0; }
        else { lineparm[2] = nx / length; lineparm[3] = ny / length; }
    }
    if (err) *err = (double)N * eig_small;
    if (mse) *mse = eig_small;
}

typedef struct {
    const uint8_t *qim; int qw, qh, qstride;   /* image the clusters were extracted from */
    const ck_config_t *cfg;
    int min_tag_width; int normal_ok, reversed_ok;
    int max_cluster_points;
} fitctx_t;

static const double k_smooth[7] = {0.011108996538242306, 0.1353352832366127, 0.6065306597126334, 1.0,
                                   0.6065306597126334, 0.1353352832366127, 0.011108996538242306};

static int dbl_desc(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return x > y ? -1 : x < y ? 1 : 0;
}

/* debug statistics: where clusters leave fit_quad (read through ora_fit_stats) */
static long g_fit_stats[16];
void ora_fit_stats(long *out, int reset) { for (int i = 0; i < 16; i++) { out[i] = g_fit_stats[i]; if (reset) g_fit_stats[i] = 0; } }
#define FIT_STAT(k) (g_fit_stats[k]++)

static int fit_quad(const fitctx_t *c, const ck_cluster_point_t *pts, int sz0, ck_quad_t *quad) {
    FIT_STAT(0);
    if (sz0 < c->cfg->min_cluster_pixels || sz0 < 24) { FIT_STAT(1); return 0; }
    if (sz0 > c->max_cluster_points) { FIT_STAT(2); return 0; }
    int xmin = pts[0].x, xmax = pts[0].x, ymin = pts[0].y, ymax = pts[0].y;
    for (int i = 1; i < sz0; i++) {
        if (pts[i].x < xmin) xmin = pts[i].x; if (pts[i].x > xmax) xmax = pts[i].x;
        if (pts[i].y < ymin) ymin = pts[i].y; if (pts[i].y > ymax) ymax = pts[i].y;
    }
    if ((xmax - xmin) * (ymax - ymin) < c->min_tag_width) { FIT_STAT(3); return 0; }
    /* border direction: sum over points of (p - centre) . gradient, exact in integers */
    int64_t dot = 0;
    for (int i = 0; i < sz0; i++) {
        int64_t dx = 4 * (int64_t)pts[i].x - 2 * ((int64_t)xmin + xmax) - 1;
        int64_t dy = 4 * (int64_t)pts[i].y - 2 * ((int64_t)ymin + ymax) + 1;
        dot += dx * pts[i].gx + dy * pts[i].gy;
    }
    int reversed = dot < 0;
    if (reversed && !c->reversed_ok) { FIT_STAT(4); return 0; }
    if (!reversed && !c->normal_ok) { FIT_STAT(4); return 0; }
    /* sort by angle, drop duplicate coordinates */
    uint64_t *keys = (uint64_t *)malloc((size_t)sz0 * sizeof(uint64_t));
    for (int i = 0; i < sz0; i++) keys[i] = angle_key(pts[i].x, pts[i].y, xmin, xmax, ymin, ymax);
    qsort(keys, (size_t)sz0, sizeof(uint64_t), u64_cmp);
    int sz = 0;
    for (int i = 0; i < sz0; i++)
        if (i == 0 || keys[i] != keys[i - 1]) keys[sz++] = keys[i];
    int ok = 0;
    lfps_t *lf = NULL; double *errs = NULL, *sm = NULL; int *maxima = NULL; double *maxima_errs = NULL;
    if (sz < 24) { FIT_STAT(5); goto done; }
    /* line-fit prefix sums */
    lf = (lfps_t *)malloc((size_t)sz * sizeof(lfps_t));
    {
        lfps_t acc = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < sz; i++) {
            int x = (int)((keys[i] >> 13) & 0x1FFF), y = (int)(keys[i] & 0x1FFF);
            int ix = (x + 1) >> 1, iy = (y + 1) >> 1;
            int64_t W = 1;
            if (ix > 0 && ix + 1 < c->qw && iy > 0 && iy + 1 < c->qh) {
                int gx = (int)c->qim[(size_t)iy * c->qstride + ix + 1] - (int)c->qim[(size_t)iy * c->qstride + ix - 1];
                int gy = (int)c->qim[(size_t)(iy + 1) * c->qstride + ix] - (int)c->qim[(size_t)(iy - 1) * c->qstride + ix];
                W = (int64_t)isqrt_u32((uint32_t)(gx * gx + gy * gy)) + 1;
            }
            int64_t X = x + 1, Y = y + 1;   /* half-pixel units incl. the +0.5 px pixel-centre offset */
            acc.Mx += W * X; acc.My += W * Y; acc.Mxx += W * X * X; acc.Mxy += W * X * Y; acc.Myy += W * Y * Y;
            acc.W += W;
            lf[i] = acc;
        }
    }
    /* corner candidates: local maxima of the smoothed line-fit error */
    int ksz = sz / 12 < 20 ? sz / 12 : 20;
    if (ksz < 2) { FIT_STAT(5); goto done; }
    errs = (double *)malloc((size_t)sz * sizeof(double));
    sm = (double *)malloc((size_t)sz * sizeof(double));
    for (int i = 0; i < sz; i++) fit_line(lf, sz, (i + sz - ksz) % sz, (i + ksz) % sz, NULL, &errs[i], NULL);
    for (int i = 0; i < sz; i++) {
        double acc = 0.0;
        for (int j = 0; j < 7; j++) acc += errs[(i + j - 3 + sz) % sz] * k_smooth[j];
        sm[i] = acc;
    }
    maxima = (int *)malloc((size_t)sz * sizeof(int));
    maxima_errs = (double *)malloc((size_t)sz * sizeof(double));
    int nmax = 0;
    for (int i = 0; i < sz; i++)
        if (sm[i] > sm[(i + 1) % sz] && sm[i] > sm[(i + sz - 1) % sz]) { maxima[nmax] = i; maxima_errs[nmax] = sm[i]; nmax++; }
    if (nmax < 4) { FIT_STAT(6); goto done; }
    int max_nmaxima = c->cfg->max_nmaxima;
    if (nmax > max_nmaxima) {
        double *cp = (double *)malloc((size_t)nmax * sizeof(double));
        memcpy(cp, maxima_errs, (size_t)nmax * sizeof(double));
        qsort(cp, (size_t)nmax, sizeof(double), dbl_desc);
        double thr = cp[max_nmaxima];
        free(cp);
        int out = 0;
        for (int in = 0; in < nmax; in++) {
            if (maxima_errs[in] <= thr) continue;
            maxima[out++] = maxima[in];
        }
        nmax = out;
    }
    /* exhaustive search over 4-subsets of the maxima */
    int best[4] = {0, 0, 0, 0};
    double best_error = HUGE_VAL;
    double max_mse = c->cfg->max_line_fit_mse, max_dot = c->cfg->cos_critical_rad;
    for (int m0 = 0; m0 < nmax - 3; m0++) {
        int i0 = maxima[m0];
        for (int m1 = m0 + 1; m1 < nmax - 2; m1++) {
            int i1 = maxima[m1];
            double p01[4], e01, s01;
            fit_line(lf, sz, i0, i1, p01, &e01, &s01);
            if (s01 > max_mse) continue;
            for (int m2 = m1 + 1; m2 < nmax - 1; m2++) {
                int i2 = maxima[m2];
                double p12[4], e12, s12;
                fit_line(lf, sz, i1, i2, p12, &e12, &s12);
                if (s12 > max_mse) continue;
                double dp = p01[2] * p12[2] + p01[3] * p12[3];
                if (fabs(dp) > max_dot) continue;
                for (int m3 = m2 + 1; m3 < nmax; m3++) {
                    int i3 = maxima[m3];
                    double e23, s23, e30, s30;
                    fit_line(lf, sz, i2, i3, NULL, &e23, &s23);
                    if (s23 > max_mse) continue;
                    fit_line(lf, sz, i3, i0, NULL, &e30, &s30);
                    if (s30 > max_mse) continue;
                    double e = e01 + e12 + e23 + e30;
                    if (e < best_error) { best_error = e; best[0] = i0; best[1] = i1; best[2] = i2; best[3] = i3; }
                }
            }
        }
    }
    if (best_error == HUGE_VAL) { FIT_STAT(7); goto done; }
    if (best_error / (double)sz >= max_mse) { FIT_STAT(8); goto done; }
    {
        double lines[4][4];
        for (int i = 0; i < 4; i++) {
            double mse;
            fit_line(lf, sz, best[i], best[(i + 1) & 3], lines[i], NULL, &mse);
            if (mse > max_mse) { FIT_STAT(9); goto done; }
        }
        for (int i = 0; i < 4; i++) {
            int j = (i + 1) & 3;
            double A00 = lines[i][3], A01 = -lines[j][3], A10 = -lines[i][2], A11 = lines[j][2];
            double B0 = -lines[i][0] + lines[j][0], B1 = -lines[i][1] + lines[j][1];
            double det = A00 * A11 - A10 * A01;
            if (fabs(det) < 0.001) { FIT_STAT(10); goto done; }
            double W00 = A11 / det, W01 = -A01 / det;
            double L0 = W00 * B0 + W01 * B1;
            quad->p[i][0] = lines[i][0] + L0 * A00;
            quad->p[i][1] = lines[i][1] + L0 * A10;
        }
    }
    /* area: two triangles, Heron */
    {
        double area = 0.0;
        static const int tri[2][3] = {{0, 1, 2}, {2, 3, 0}};
        for (int t = 0; t < 2; t++) {
            double len[3];
            for (int i = 0; i < 3; i++) {
                int a = tri[t][i], b = tri[t][(i + 1) % 3];
                double ddx = quad->p[b][0] - quad->p[a][0], ddy = quad->p[b][1] - quad->p[a][1];
                len[i] = sqrt(ddx * ddx + ddy * ddy);
            }
            double p = (len[0] + len[1] + len[2]) / 2.0;
            area += sqrt(p * (p - len[0]) * (p - len[1]) * (p - len[2]));
        }
        double tw = (double)c->min_tag_width;
        if (area < 0.95 * tw * tw) { FIT_STAT(11); goto done; }
    }
    /* corner angles and winding */
    for (int i = 0; i < 4; i++) {
        int i0 = i, i1 = (i + 1) & 3, i2 = (i + 2) & 3;
        double dx1 = quad->p[i1][0] - quad->p[i0][0], dy1 = quad->p[i1][1] - quad->p[i0][1];
        double dx2 = quad->p[i2][0] - quad->p[i1][0], dy2 = quad->p[i2][1] - quad->p[i1][1];
        double cs = (dx1 * dx2 + dy1 * dy2) / sqrt((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2));
        if (cs > c->cfg->cos_critical_rad || cs < -c->cfg->cos_critical_rad) { FIT_STAT(12); goto done; }
        if (dx1 * dy2 < dy1 * dx2) { FIT_STAT(13); goto done; }
    }
    quad->reversed_border = reversed;
    FIT_STAT(14);
    ok = 1;
done:
    free(keys); free(lf); free(errs); free(sm); free(maxima); free(maxima_errs);
    return ok;
}

/* Snap each quad edge to the strongest nearby gradient in the full-resolution image. */
static void refine_edges(const uint8_t *im, int w, int h, int stride, int decimate, ck_quad_t *quad) {
    double lines[4][4];
    for (int edge = 0; edge < 4; edge++) {
        int a = edge, b = (edge + 1) & 3;
        double nx = quad->p[b][1] - quad->p[a][1];
        double ny = -quad->p[b][0] + quad->p[a][0];
        double mag = sqrt(nx * nx + ny * ny);
        nx = nx / mag; ny = ny / mag;
        if (quad->reversed_border) { nx = -nx; ny = -ny; }
        int nsamples = (int)(mag / 8.0);
        if (nsamples < 16) nsamples = 16;
        double Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, N = 0;
        for (int s = 0; s < nsamples; s++) {
            double alpha = (1.0 + (double)s) / ((double)nsamples + 1.0);
            double x0 = alpha * quad->p[a][0] + (1.0 - alpha) * quad->p[b][0];
            double y0 = alpha * quad->p[a][1] + (1.0 - alpha) * quad->p[b][1];
            double Mn = 0, Mcount = 0;
            int range = decimate + 1;
            for (int n = -range; n <= range; n++) {
                double grange = 1.0;
                int x1 = (int)(x0 + ((double)n + grange) * nx), y1 = (int)(y0 + ((double)n + grange) * ny);
                if (x1 < 0 || x1 >= w || y1 < 0 || y1 >= h) continue;
                int x2 = (int)(x0 + ((double)n - grange) * nx), y2 = (int)(y0 + ((double)n - grange) * ny);
                if (x2 < 0 || x2 >= w || y2 < 0 || y2 >= h) continue;
                int g1 = im[(size_t)y1 * stride + x1], g2 = im[(size_t)y2 * stride + x2];
                if (g1 < g2) continue;
                double weight = (double)((g2 - g1) * (g2 - g1));
                Mn += weight * (double)n;
                Mcount += weight;
            }
            if (Mcount == 0) continue;
            double n0 = Mn / Mcount;
            double bx = x0 + n0 * nx, by = y0 + n0 * ny;
            Mx += bx; My += by; Mxx += bx * bx; Mxy += bx * by; Myy += by * by; N += 1.0;
        }
        if (N < 2.0) { /* nothing to fit: keep the current edge */
            lines[edge][0] = 0.5 * (quad->p[a][0] + quad->p[b][0]); lines[edge][1] = 0.5 * (quad->p[a][1] + quad->p[b][1]);
            lines[edge][2] = nx; lines[edge][3] = ny;
            continue;
        }
        double Ex = Mx / N, Ey = My / N;
        double Cxx = Mxx / N - Ex * Ex, Cxy = Mxy / N - Ex * Ey, Cyy = Myy / N - Ey * Ey;
        double d = Cxx - Cyy, q = 4.0 * Cxy;
        double disc = sqrt(d * d + q * Cxy);
        double eig = 0.5 * (Cxx + Cyy + disc);
        double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
        double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
        double fx, fy, M;
        if (M1 > M2) { fx = nx1; fy = ny1; M = M1; } else { fx = nx2; fy = ny2; M = M2; }
        double len = sqrt(M);
        lines[edge][0] = Ex; lines[edge][1] = Ey;
        if (len < 1e-12) { lines[edge][2] = nx; lines[edge][3] = ny; }
        else { lines[edge][2] = fx / len; lines[edge][3] = fy / len; }
    }
    for (int i = 0; i < 4; i++) {
        int j = (i + 1) & 3;
        double A00 = lines[i][3], A01 = -lines[j][3], A10 = -lines[i][2], A11 = lines[j][2];
        double B0 = -lines[i][0] + lines[j][0], B1 = -lines[i][1] + lines[j][1];
        double det = A00 * A11 - A10 * A01;
        if (fabs(det) > 0.001) {
            double W00 = A11 / det, W01 = -A01 / det;
            double L0 = W00 * B0 + W01 * B1;
            quad->p[j][0] = lines[i][0] + L0 * A00;
            quad->p[j][1] = lines[i][1] + L0 * A10;
        }
    }
}

static int quad_cmp(const void *a, const void *b) {
    const ck_quad_t *x = (const ck_quad_t *)a, *y = (const ck_quad_t *)b;
    if (x->rep0 != y->rep0) return x->rep0 < y->rep0 ? -1 : 1;
    if (x->rep1 != y->rep1) return x->rep1 < y->rep1 ? -1 : 1;
    return 0;
}

int ora_fit_quads(const uint8_t *qim, int qw, int qh, int qstride, const uint8_t *orig, int w, int h, int stride,
                  const ck_config_t *cfg, const ck_cluster_t *clusters, int n_clusters,
                  const ck_cluster_point_t *points, ck_quad_t *quads, int quad_cap, int *n_quads) {
    fitctx_t c;
    c.qim = qim; c.qw = qw; c.qh = qh; c.qstride = qstride; c.cfg = cfg;
    c.normal_ok = 0; c.reversed_ok = 0; c.min_tag_width = 1 << 30;
    for (int f = 0; f < cfg->n_families; f++) {
        if (cfg->families[f]->width_at_border < c.min_tag_width) c.min_tag_width = cfg->families[f]->width_at_border;
        if (cfg->families[f]->reversed_border) c.reversed_ok = 1; else c.normal_ok = 1;
    }
    int dec = cfg->quad_decimate < 1 ? 1 : cfg->quad_decimate;
    c.min_tag_width /= dec;
    if (c.min_tag_width < 3) c.min_tag_width = 3;
    c.max_cluster_points = 3 * (2 * qw + 2 * qh);
    int nq = 0, overflow = 0;
    for (int k = 0; k < n_clusters; k++) {
        ck_quad_t q;
        memset(&q, 0, sizeof q);
        if (!fit_quad(&c, points + clusters[k].start, (int)clusters[k].count, &q)) continue;
        q.rep0 = clusters[k].rep0; q.rep1 = clusters[k].rep1;
        if (dec > 1)
            for (int i = 0; i < 4; i++) {
                q.p[i][0] = (q.p[i][0] - 0.5) * (double)dec + 0.5;
                q.p[i][1] = (q.p[i][1] - 0.5) * (double)dec + 0.5;
            }
        if (cfg->refine_edges) refine_edges(orig, w, h, stride, dec, &q);
        if (nq >= quad_cap) { overflow = 1; break; }
        quads[nq++] = q;
    }
    qsort(quads, (size_t)nq, sizeof(ck_quad_t), quad_cmp);
    *n_quads = nq;
    return overflow;
}

/* ------------------------------------------------------------------------------------------------- */
/* Decode */
static int homography_compute(const double corr[4][4], double *H) {
    double A[8 * 9];
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
static inline void hproject(const double *H, double x, double y, double *ox, double *oy) {
    double xx = H[0] * x + H[1] * y + H[2];
    double yy = H[3] * x + H[4] * y + H[5];
    double zz = H[6] * x + H[7] * y + H[8];
    *ox = xx / zz; *oy = yy / zz;
}
typedef struct { double A[3][3]; double B[3]; double C[3]; } graymodel_t;
static void gm_add(graymodel_t *g, double x, double y, double gray) {
    g->A[0][0] += x * x; g->A[0][1] += x * y; g->A[0][2] += x;
    g->A[1][1] += y * y; g->A[1][2] += y; g->A[2][2] += 1;
    g->B[0] += x * gray; g->B[1] += y * gray; g->B[2] += gray;
}
static void gm_solve(graymodel_t *g) {
    /* symmetric 3x3 solve by Cholesky A = L L^T */
    double a00 = g->A[0][0], a01 = g->A[0][1], a02 = g->A[0][2], a11 = g->A[1][1], a12 = g->A[1][2], a22 = g->A[2][2];
    double l00 = sqrt(a00);
    double l10 = a01 / l00, l20 = a02 / l00;
    double l11 = sqrt(a11 - l10 * l10);
    double l21 = (a12 - l10 * l20) / l11;
    double l22 = sqrt(a22 - l20 * l20 - l21 * l21);
    /* forward: L y = B */
    double y0 = g->B[0] / l00;
    double y1 = (g->B[1] - l10 * y0) / l11;
    double y2 = (g->B[2] - l20 * y0 - l21 * y1) / l22;
    /* backward: L^T c = y */
    double c2 = y2 / l22;
    double c1 = (y1 - l21 * c2) / l11;
    double c0 = (y0 - l10 * c1 - l20 * c2) / l00;
    g->C[0] = c0; g->C[1] = c1; g->C[2] = c2;
}
static inline double gm_interp(const graymodel_t *g, double x, double y) { return g->C[0] * x + g->C[1] * y + g->C[2]; }

static double value_for_pixel(const uint8_t *im, int w, int h, int stride, double px, double py) {
    double fx = px - 0.5, fy = py - 0.5;
    int x1 = (int)floor(fx), x2 = (int)ceil(fx);
    double x = fx - (double)x1;
    int y1 = (int)floor(fy), y2 = (int)ceil(fy);
    double y = fy - (double)y1;
    if (x1 < 0 || x2 >= w || y1 < 0 || y2 >= h) return -1.0;
    return (double)im[(size_t)y1 * stride + x1] * (1.0 - x) * (1.0 - y) + (double)im[(size_t)y1 * stride + x2] * x * (1.0 - y) +
           (double)im[(size_t)y2 * stride + x1] * (1.0 - x) * y + (double)im[(size_t)y2 * stride + x2] * x * y;
}
static uint64_t code_rotate90(uint64_t w, int nbits) {
    int p = nbits; uint64_t l = 0;
    if (nbits % 4 == 1) { p = nbits - 1; l = 1; }
    w = ((w >> l) << (p / 4 + l)) | (w >> (3 * p / 4 + l) << l) | (w & l);
    w &= (((uint64_t)1 << nbits) - 1);
    return w;
}

/* returns decision margin (negative = reject); fills id/hamming/rotation */
static double quad_decode(const uint8_t *im, int w, int h, int stride, const ck_family_t *fam, const double *H,
                          double sharpening, int max_hamming, int *id, int *hamming, int *rotation) {
    double wb = (double)fam->width_at_border;
    double patterns[8][5] = {
        {-0.5, 0.5, 0, 1, 1}, {0.5, 0.5, 0, 1, 0}, {wb + 0.5, 0.5, 0, 1, 1}, {wb - 0.5, 0.5, 0, 1, 0},
        {0.5, -0.5, 1, 0, 1}, {0.5, 0.5, 1, 0, 0}, {0.5, wb + 0.5, 1, 0, 1}, {0.5, wb - 0.5, 1, 0, 0}};
    graymodel_t wm, bm;
    memset(&wm, 0, sizeof wm); memset(&bm, 0, sizeof bm);
    for (int pi = 0; pi < 8; pi++) {
        int is_white = patterns[pi][4] != 0;
        for (int i = 0; i < fam->width_at_border; i++) {
            double tagx01 = (patterns[pi][0] + (double)i * patterns[pi][2]) / wb;
            double tagy01 = (patterns[pi][1] + (double)i * patterns[pi][3]) / wb;
            double tagx = 2.0 * (tagx01 - 0.5), tagy = 2.0 * (tagy01 - 0.5);
            double px, py;
            hproject(H, tagx, tagy, &px, &py);
            int ix = (int)px, iy = (int)py;
            if (px < 0 || py < 0 || ix < 0 || iy < 0 || ix >= w || iy >= h) continue;
            int v = im[(size_t)iy * stride + ix];
            if (is_white) gm_add(&wm, tagx, tagy, (double)v); else gm_add(&bm, tagx, tagy, (double)v);
        }
    }
    gm_solve(&wm); gm_solve(&bm);
    if (((gm_interp(&wm, 0, 0) - gm_interp(&bm, 0, 0)) < 0) != (fam->reversed_border != 0)) return -1.0;
    int tw = fam->total_width;
    double values[16 * 16];
    for (int i = 0; i < tw * tw; i++) values[i] = 0.0;
    int min_coord = (fam->width_at_border - tw) / 2;
    for (uint32_t i = 0; i < fam->nbits; i++) {
        int bx = (int)fam->bit_x[i], by = (int)fam->bit_y[i];
        double tagx = 2.0 * (((double)bx + 0.5) / wb - 0.5), tagy = 2.0 * (((double)by + 0.5) / wb - 0.5);
        double px, py;
        hproject(H, tagx, tagy, &px, &py);
        double v = value_for_pixel(im, w, h, stride, px, py);
        if (v == -1.0) continue;
        double thr = (gm_interp(&bm, tagx, tagy) + gm_interp(&wm, tagx, tagy)) / 2.0;
        values[tw * (by - min_coord) + bx - min_coord] = v - thr;
    }
    /* Laplacian sharpening over the total_width^2 grid */
    double sharp[16 * 16];
    for (int y = 0; y < tw; y++)
        for (int x = 0; x < tw; x++) {
            double s = 0.0;
            if (y > 0) s += -values[(y - 1) * tw + x];
            if (x > 0) s += -values[y * tw + x - 1];
            s += 4.0 * values[y * tw + x];
            if (x < tw - 1) s += -values[y * tw + x + 1];
            if (y < tw - 1) s += -values[(y + 1) * tw + x];
            sharp[y * tw + x] = s;
        }
    for (int i = 0; i < tw * tw; i++) values[i] = values[i] + sharpening * sharp[i];
    uint64_t rcode = 0;
    double black_score = 0, white_score = 0, black_cnt = 1, white_cnt = 1;
    for (uint32_t i = 0; i < fam->nbits; i++) {
        int bx = (int)fam->bit_x[i], by = (int)fam->bit_y[i];
        rcode <<= 1;
        double v = values[(by - min_coord) * tw + bx - min_coord];
        if (v > 0) { white_score += v; white_cnt += 1; rcode |= 1; }
        else { black_score -= v; black_cnt += 1; }
    }
    /* minimum-Hamming match over rotations and ids; ties: fewer rotations first, then smaller id */
    int best_h = 1 << 30, best_id = -1, best_rot = 0;
    uint64_t rc = rcode;
    for (int rot = 0; rot < 4; rot++) {
        for (uint32_t k = 0; k < fam->ncodes; k++) {
            int hd = __builtin_popcountll(rc ^ fam->codes[k]);
            if (hd < best_h) { best_h = hd; best_id = (int)k; best_rot = rot; }
        }
        rc = code_rotate90(rc, (int)fam->nbits);
    }
    if (best_h > max_hamming) return -1.0;
    *id = best_id; *hamming = best_h; *rotation = best_rot;
    double a = white_score / white_cnt, b = black_score / black_cnt;
    return a < b ? a : b;
}

static int point_in_quad(const double q[4][2], double x, double y) {
    int pos = 0, neg = 0;
    for (int i = 0; i < 4; i++) {
        int j = (i + 1) & 3;
        double cr = (q[j][0] - q[i][0]) * (y - q[i][1]) - (q[j][1] - q[i][1]) * (x - q[i][0]);
        if (cr > 0) pos++; else if (cr < 0) neg++;
    }
    return pos == 0 || neg == 0;
}
static int det_cmp(const void *a, const void *b) {
    const ck_detection_t *x = (const ck_detection_t *)a, *y = (const ck_detection_t *)b;
    if (x->family != y->family) return x->family < y->family ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    if (x->hamming != y->hamming) return x->hamming < y->hamming ? -1 : 1;
    if (x->decision_margin != y->decision_margin) return x->decision_margin > y->decision_margin ? -1 : 1;
    if (x->c[0] != y->c[0]) return x->c[0] < y->c[0] ? -1 : 1;
    if (x->c[1] != y->c[1]) return x->c[1] < y->c[1] ? -1 : 1;
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 2; k++)
            if (x->p[i][k] != y->p[i][k]) return x->p[i][k] < y->p[i][k] ? -1 : 1;
    return 0;
}

int ora_decode_quads(const uint8_t *im, int w, int h, int stride, const ck_config_t *cfg, const ck_quad_t *quads,
                     int n_quads, ck_detection_t *dets, int det_cap, int *n_dets) {
    int cap = n_quads * (cfg->n_families > 0 ? cfg->n_families : 1) + 1;
    ck_detection_t *all = (ck_detection_t *)malloc((size_t)cap * sizeof(ck_detection_t));
    int n = 0;
    for (int qi = 0; qi < n_quads; qi++) {
        const ck_quad_t *q = &quads[qi];
        double corr[4][4];
        for (int i = 0; i < 4; i++) {
            corr[i][0] = (i == 0 || i == 3) ? -1 : 1;
            corr[i][1] = (i == 0 || i == 1) ? -1 : 1;
            corr[i][2] = q->p[i][0]; corr[i][3] = q->p[i][1];
        }
        double H[9];
        if (!homography_compute(corr, H)) continue;
        for (int f = 0; f < cfg->n_families; f++) {
            const ck_family_t *fam = cfg->families[f];
            if ((fam->reversed_border != 0) != (q->reversed_border != 0)) continue;
            int id = -1, hd = 0, rot = 0;
            double margin = quad_decode(im, w, h, stride, fam, H, cfg->decode_sharpening, cfg->max_hamming, &id, &hd, &rot);
            if (!(margin >= 0) || id < 0) continue;
            ck_detection_t d;
            memset(&d, 0, sizeof d);
            d.id = id; d.hamming = hd; d.family = f; d.decision_margin = (float)margin;
            /* H' = H * Rz(rot * 90 deg), exact */
            static const double cs[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
            double c = cs[rot][0], s = cs[rot][1];
            double Hr[9];
            for (int r = 0; r < 3; r++) {
                Hr[r * 3 + 0] = c * H[r * 3 + 0] + s * H[r * 3 + 1];
                Hr[r * 3 + 1] = -s * H[r * 3 + 0] + c * H[r * 3 + 1];
                Hr[r * 3 + 2] = H[r * 3 + 2];
            }
            hproject(Hr, 0, 0, &d.c[0], &d.c[1]);
            static const double tc[4][2] = {{-1, 1}, {1, 1}, {1, -1}, {-1, -1}};
            for (int i = 0; i < 4; i++) hproject(Hr, tc[i][0], tc[i][1], &d.p[i][0], &d.p[i][1]);
            all[n++] = d;
        }
    }
    qsort(all, (size_t)n, sizeof(ck_detection_t), det_cmp);
    int out = 0, overflow = 0;
    for (int i = 0; i < n; i++) {
        int dup = 0;
        for (int j = 0; j < out && !dup; j++) {
            const ck_detection_t *k = &dets[j];
            if (k->family != all[i].family || k->id != all[i].id) continue;
            if (point_in_quad(k->p, all[i].c[0], all[i].c[1]) || point_in_quad(all[i].p, k->c[0], k->c[1])) dup = 1;
        }
        if (dup) continue;
        if (out >= det_cap) { overflow = 1; break; }
        dets[out++] = all[i];
    }
    free(all);
    *n_dets = out;
    return overflow;
}

int ora_detect(const uint8_t *img, int w, int h, int stride, const ck_config_t *cfg, ck_detection_t *dets, int det_cap,
               int *n_dets, uint32_t *status) {
    int dec = cfg->quad_decimate < 1 ? 1 : cfg->quad_decimate;
    int qw = w / dec, qh = h / dec;
    uint8_t *qim = NULL;
    const uint8_t *q = img; int qstride = stride;
    if (dec > 1) { qim = (uint8_t *)malloc((size_t)qw * qh); ora_decimate(img, w, h, stride, dec, qim); q = qim; qstride = qw; }
    size_t n = (size_t)qw * qh;
    uint8_t *th = (uint8_t *)malloc(n);
    uint32_t *labels = (uint32_t *)malloc(n * 4), *sizes = (uint32_t *)malloc(n * 4);
    ora_threshold(q, qw, qh, qstride, cfg->min_white_black_diff, th);
    ora_segment(th, qw, qh, labels, sizes);
    int pcap = cfg->max_points_per_frame > 0 ? cfg->max_points_per_frame : (int)(4 * n);
    int ccap = cfg->max_clusters_per_frame > 0 ? cfg->max_clusters_per_frame : (n / 32 < 1024 ? 1024 : (int)(n / 32)); /* the library's defaults */
    int qcap = cfg->max_quads_per_frame > 0 ? cfg->max_quads_per_frame : 1024;
    ck_cluster_t *cl = (ck_cluster_t *)malloc((size_t)ccap * sizeof *cl);
    ck_cluster_point_t *pts = (ck_cluster_point_t *)malloc((size_t)pcap * sizeof *pts);
    ck_quad_t *quads = (ck_quad_t *)malloc((size_t)qcap * sizeof *quads);
    int nc = 0, np = 0, nq = 0;
    uint32_t st = 0;
    if (ora_clusters(th, labels, sizes, qw, qh, cfg->min_component_px, cl, ccap, &nc, pts, pcap, &np)) st |= CK_FRAME_POINTS_OVERFLOW;
    if (ora_fit_quads(q, qw, qh, qstride, img, w, h, stride, cfg, cl, nc, pts, quads, qcap, &nq)) st |= CK_FRAME_QUADS_OVERFLOW;
    if (ora_decode_quads(img, w, h, stride, cfg, quads, nq, dets, det_cap, n_dets)) st |= CK_FRAME_DETS_OVERFLOW;
    for (int i = 0; i < *n_dets; i++) /* an id past the family's verified prefix is not an upstream id (ck_family_t.n_upstream) */
        if ((uint32_t)dets[i].id >= cfg->families[dets[i].family]->n_upstream) st |= CK_FRAME_UNVERIFIED_ID;
    if (status) *status = st;
    free(qim); free(th); free(labels); free(sizes); free(cl); free(pts); free(quads);
    return CK_OK;
}
