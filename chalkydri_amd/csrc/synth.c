// Deterministic synthetic frame renderer (SURVEY.md §8d "Synthetic inputs").
//
// Frames are rendered from known tag IDs and homographies so that tests have exact ground truth and the
// benchmark has inputs of the shape BASELINE.json names.  Only integer ops and IEEE + - * / are used (no
// libm), and the file is compiled with -ffp-contract=off, so every machine regenerates identical bytes.
//
// Pixel convention: pixel (x,y) covers [x,x+1) x [y,y+1); coordinates of ground-truth corners are in that
// continuous frame (the same one the detector reports corners in).
#include "synth.h"

#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r) { // xorshift64*
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static inline int rng_range(rng_t *r, int lo, int hi) { // inclusive
    return lo + (int)((rng_next(r) >> 33) % (uint64_t)(hi - lo + 1));
}

void ck_synth_params_default(ck_synth_params_t *p, int32_t width, int32_t height, int32_t n_tags) {
    memset(p, 0, sizeof *p);
    p->width = width; p->height = height; p->n_tags = n_tags;
    p->min_side = 32; p->max_side = 220; p->max_tilt_t64 = 33; // tan(27.5 deg)*64 -> tilt <= 55 deg
    p->noise_amp = 3; p->ramp_amp = 24; p->black = 24; p->white = 232; p->bg = 128;
    p->family_mode = 0; p->max_id = -1;
}

// cell colour map of a tag: 1 white, 0 black, over total_width^2 cells
static void cell_map(const ck_family_t *f, int id, uint8_t *cells) {
    int tw = f->total_width, wb = f->width_at_border, b = (tw - wb) / 2;
    uint64_t code = f->codes[id];
    for (int cy = 0; cy < tw; cy++)
        for (int cx = 0; cx < tw; cx++) {
            int bx = cx - b, by = cy - b;
            uint8_t v;
            if (bx < 0 || by < 0 || bx >= wb || by >= wb) v = f->reversed_border ? 0 : 1;
            else v = f->reversed_border ? 1 : 0; // border ring and anything not covered by a data bit
            cells[cy * tw + cx] = v;
        }
    for (uint32_t i = 0; i < f->nbits; i++) {
        int bit = (int)((code >> (f->nbits - 1 - i)) & 1);
        cells[(f->bit_y[i] + b) * tw + f->bit_x[i] + b] = (uint8_t)bit;
    }
}

static void mat33_inv(const double *m, double *o) {
    double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    double A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
    double D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
    double G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
    double det = a * A + b * D + c * G;
    o[0] = A / det; o[1] = B / det; o[2] = C / det;
    o[3] = D / det; o[4] = E / det; o[5] = F / det;
    o[6] = G / det; o[7] = H / det; o[8] = I / det;
}
static inline void project(const double *H, double x, double y, double *ox, double *oy) {
    double xx = H[0] * x + H[1] * y + H[2];
    double yy = H[3] * x + H[4] * y + H[5];
    double zz = H[6] * x + H[7] * y + H[8];
    *ox = xx / zz; *oy = yy / zz;
}

void ck_synth_fill_truth(ck_synth_tag_t *t) {
    // detection corner order: (-1,1), (1,1), (1,-1), (-1,-1)   (apriltag convention)
    static const double tc[4][2] = {{-1, 1}, {1, 1}, {1, -1}, {-1, -1}};
    for (int i = 0; i < 4; i++) project(t->H, tc[i][0], tc[i][1], &t->corners[i][0], &t->corners[i][1]);
    project(t->H, 0, 0, &t->center[0], &t->center[1]);
}

void ck_synth_background(uint64_t seed, const ck_synth_params_t *p, uint8_t *out, int32_t stride) {
    rng_t r = { splitmix64(seed ^ 0xB5AD4ECEDA1CE2A9ULL) | 1 };
    int ax = p->ramp_amp ? rng_range(&r, -p->ramp_amp, p->ramp_amp) : 0;
    int ay = p->ramp_amp ? rng_range(&r, -p->ramp_amp, p->ramp_amp) : 0;
    int w = p->width, h = p->height, A = p->noise_amp;
    uint64_t nseed = splitmix64(seed * 0x9E3779B97F4A7C15ULL + 77);
    for (int y = 0; y < h; y++) {
        int ry = ay * (2 * y - h) / h;
        for (int x = 0; x < w; x++) {
            int rx = ax * (2 * x - w) / w;
            int v = p->bg + (rx + ry) / 2;
            if (A > 0) {
                uint64_t hsh = splitmix64(nseed ^ ((uint64_t)y * (uint64_t)w + (uint64_t)x));
                v += (int)((hsh >> 32) % (uint64_t)(2 * A + 1)) - A;
            }
            out[(size_t)y * stride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

int ck_synth_draw_tag(const ck_synth_params_t *p, const ck_family_t *fam, const ck_synth_tag_t *t,
                      uint8_t *out, int32_t stride) {
    if (t->id < 0 || (uint32_t)t->id >= fam->ncodes || fam->total_width > 16) return -1;
    uint8_t cells[16 * 16];
    cell_map(fam, t->id, cells);
    int tw = fam->total_width;
    double q = (double)tw / (double)fam->width_at_border; // quiet-zone half extent in tag units
    double Hi[9];
    mat33_inv(t->H, Hi);
    // bounding box of the quiet-zone quad
    double bx0 = 1e30, by0 = 1e30, bx1 = -1e30, by1 = -1e30;
    static const double sg[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    for (int i = 0; i < 4; i++) {
        double x, y;
        project(t->H, sg[i][0] * q, sg[i][1] * q, &x, &y);
        if (x < bx0) bx0 = x; if (x > bx1) bx1 = x;
        if (y < by0) by0 = y; if (y > by1) by1 = y;
    }
    int x0 = (int)bx0 - 1, x1 = (int)bx1 + 2, y0 = (int)by0 - 1, y1 = (int)by1 + 2;
    if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0;
    if (x1 > p->width) x1 = p->width; if (y1 > p->height) y1 = p->height;
    double scale = (double)fam->width_at_border * 0.5; // tag units -> cells
    int b = (tw - fam->width_at_border) / 2;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            int n_in = 0, sum = 0;
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) {
                    double sx = (double)x + ((double)i + 0.5) * 0.25;
                    double sy = (double)y + ((double)j + 0.5) * 0.25;
                    double tx, ty;
                    project(Hi, sx, sy, &tx, &ty);
                    if (tx < -q || tx >= q || ty < -q || ty >= q) continue;
                    int cx = (int)((tx + 1.0) * scale + (double)b + 16.0) - 16; // floor for values > -16
                    int cy = (int)((ty + 1.0) * scale + (double)b + 16.0) - 16;
                    if (cx < 0) cx = 0; if (cx >= tw) cx = tw - 1;
                    if (cy < 0) cy = 0; if (cy >= tw) cy = tw - 1;
                    sum += cells[cy * tw + cx] ? p->white : p->black;
                    n_in++;
                }
            if (n_in == 0) continue;
            int bgv = out[(size_t)y * stride + x];
            int v = (sum + (16 - n_in) * bgv + 8) / 16;
            out[(size_t)y * stride + x] = (uint8_t)v;
        }
    return 0;
}

// H = K [r1 r2 T] for a tag of half-size 1 rotated in-plane by 2*atan(tz) (+ optional 180 deg) and tilted by
// 2*atan(tx) about x and 2*atan(ty) about y, centred at pixel (pcx,pcy) with apparent side `side` px.
static void pose_homography(double f, double cx0, double cy0, double pcx, double pcy, double side, double tz,
                            int flip, double tx, double ty, double *H) {
    double cz = (1 - tz * tz) / (1 + tz * tz), sz = 2 * tz / (1 + tz * tz);
    if (flip) { cz = -cz; sz = -sz; }
    double cxr = (1 - tx * tx) / (1 + tx * tx), sxr = 2 * tx / (1 + tx * tx);
    double cyr = (1 - ty * ty) / (1 + ty * ty), syr = 2 * ty / (1 + ty * ty);
    // R = Rx * Ry * Rz ; columns r1 = R e1, r2 = R e2
    // Rz e1 = (cz, sz, 0), Rz e2 = (-sz, cz, 0)
    double v1[3] = {cz, sz, 0}, v2[3] = {-sz, cz, 0};
    double r[2][3];
    double *v[2] = {v1, v2};
    for (int k = 0; k < 2; k++) {
        // Ry
        double a0 = cyr * v[k][0] + syr * v[k][2], a1 = v[k][1], a2 = -syr * v[k][0] + cyr * v[k][2];
        // Rx
        r[k][0] = a0; r[k][1] = cxr * a1 - sxr * a2; r[k][2] = sxr * a1 + cxr * a2;
    }
    double Z = 2.0 * f / side;
    double T[3] = {(pcx - cx0) * Z / f, (pcy - cy0) * Z / f, Z};
    double M[9] = {r[0][0], r[1][0], T[0], r[0][1], r[1][1], T[1], r[0][2], r[1][2], T[2]};
    for (int j = 0; j < 3; j++) {
        H[0 + j] = f * M[0 + j] + cx0 * M[6 + j];
        H[3 + j] = f * M[3 + j] + cy0 * M[6 + j];
        H[6 + j] = M[6 + j];
    }
}

int ck_synth_render(uint64_t seed, const ck_synth_params_t *p, const ck_family_t *const *fams, int32_t n_fams,
                    uint8_t *out, int32_t stride, ck_synth_tag_t *truth, int32_t truth_cap, int32_t *n_truth) {
    if (!p || !out || !fams || n_fams < 1 || stride < p->width) return -1;
    ck_synth_background(seed, p, out, stride);
    rng_t r = { splitmix64(seed) | 1 };
    int w = p->width, h = p->height, placed = 0;
    double f = (double)w, cx0 = (double)w * 0.5, cy0 = (double)h * 0.5;
    int max_side = p->max_side;
    if (max_side > h / 2) max_side = h / 2;
    if (max_side < p->min_side) max_side = p->min_side; // a frame too low for min_side: the placement loop finds no room and draws no tag
    ck_synth_tag_t *tags = (ck_synth_tag_t *)calloc((size_t)(p->n_tags > 0 ? p->n_tags : 1), sizeof *tags);
    double *rad = (double *)calloc((size_t)(p->n_tags > 0 ? p->n_tags : 1), sizeof *rad);
    if (!tags || !rad) { free(tags); free(rad); return -2; }
    for (int k = 0; k < p->n_tags; k++) {
        int fi = p->family_mode == 1 ? (k % n_fams) : 0;
        const ck_family_t *fam = fams[fi];
        int max_id = (p->max_id >= 0 && (uint32_t)p->max_id < fam->ncodes) ? p->max_id : (int)fam->ncodes - 1;
        for (int attempt = 0; attempt < 200; attempt++) {
            int side = rng_range(&r, p->min_side, max_side);
            if (attempt > 100) side = p->min_side + (side - p->min_side) / 4;
            double qz = (double)fam->total_width / (double)fam->width_at_border;
            double R = 0.75 * (double)side * qz + 3.0; // bounding radius incl. quiet zone
            int lo_x = (int)R + 2, hi_x = w - (int)R - 2, lo_y = (int)R + 2, hi_y = h - (int)R - 2;
            if (hi_x <= lo_x || hi_y <= lo_y) continue;
            int pcx = rng_range(&r, lo_x, hi_x), pcy = rng_range(&r, lo_y, hi_y);
            int tz = rng_range(&r, -64, 64), flip = rng_range(&r, 0, 1);
            int tx = rng_range(&r, -p->max_tilt_t64, p->max_tilt_t64);
            int ty = rng_range(&r, -p->max_tilt_t64, p->max_tilt_t64);
            int id = rng_range(&r, 0, max_id);
            int ok = 1;
            for (int j = 0; j < placed; j++) {
                double dx = tags[j].center[0] - (double)pcx, dy = tags[j].center[1] - (double)pcy;
                double rr = rad[j] + R;
                if (dx * dx + dy * dy < rr * rr) { ok = 0; break; }
                if (tags[j].family == fi && tags[j].id == id) { ok = 0; break; } // unique ids per frame
            }
            if (!ok) continue;
            ck_synth_tag_t *t = &tags[placed];
            t->family = fi; t->id = id;
            pose_homography(f, cx0, cy0, (double)pcx, (double)pcy, (double)side, (double)tz / 64.0, flip,
                            (double)tx / 64.0, (double)ty / 64.0, t->H);
            ck_synth_fill_truth(t);
            rad[placed] = R;
            placed++;
            break;
        }
    }
    for (int k = 0; k < placed; k++) ck_synth_draw_tag(p, fams[tags[k].family], &tags[k], out, stride);
    if (truth) for (int k = 0; k < placed && k < truth_cap; k++) truth[k] = tags[k];
    if (n_truth) *n_truth = placed;
    free(tags); free(rad);
    return 0;
}
