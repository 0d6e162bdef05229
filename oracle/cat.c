/*
 * cat.c — CPU oracle for "CAT", the experimental detector front-end in crates/chalkydri-apriltags.
 * TEST INFRASTRUCTURE ONLY (see ck_oracle.h).
 *
 * Follows crates/chalkydri-apriltags/src/lib.rs and src/utils.rs line by line (citations at each function).
 * The order statistics come from statrs 0.18.0 [EXT: crates/chalkydri-apriltags/Cargo.toml:16, source not under
 * /root/reference]; its published OrderStatistics semantics are restated in data_median()/data_quantile():
 *     median     : n odd -> x[(n)/2]; n even -> (x[n/2-1] + x[n/2]) / 2
 *     quantile(t): h = (n + 1/3) t + 1/3, hf = trunc(h); hf <= 0 -> min; hf >= n -> max;
 *                  else x[hf-1] + (h - hf) (x[hf] - x[hf-1])           (R-8 estimator)
 * PARITY UNPINNED: the reference has no test, fixture or golden vector for this crate (its bench fixture test.png is
 * absent), so "statrs-assumed" applies to calc_otsu until it can be checked against a real build.
 *
 * Where the reference reads outside its buffer (undefined behaviour) the restatement defines the result and says so:
 *   - process_pixel: a pixel whose radius-3 samples fall outside the class map is not a corner
 *     (the reference loops x,y up to width-3 / height-3 inclusive, lib.rs:293-294, and indexes x+3 / y+3, :372-378);
 *   - check_edge: a sample whose flat index y*width+x (+-5, wrapping as the reference's unchecked usize arithmetic
 *     does, :433-434,456-461) falls outside [0, width*height) makes the edge fail.
 * Inside the buffer the flat-index semantics of utils.rs:27-29 are kept exactly, including reads that wrap into the
 * neighbouring row.
 */
#include "ck_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { BLACK = 0, WHITE = 1, OTHER = 2 }; /* utils.rs:2-6 */

/* utils.rs:33-46 */
uint8_t ora_cat_grayscale(uint8_t r, uint8_t g, uint8_t b) {
    float v = fmaf((float)r, 0.33f, fmaf((float)g, 0.33f, (float)b * 0.33f));
    if (!(v > 0.0f)) return 0; /* `as u8`: saturating, NaN -> 0 */
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

static double data_median(const double *x, int n) {
    int k = n / 2;
    if (n % 2 != 0) return x[k];
    return (x[k > 0 ? k - 1 : 0] + x[k]) / 2.0;
}
static double data_quantile(const double *x, int n, double tau) {
    double h = ((double)n + 1.0 / 3.0) * tau + 1.0 / 3.0;
    long long hf = (long long)h;
    if (hf <= 0 || tau == 0.0) return x[0];
    if (hf >= (long long)n) return x[n - 1];
    double a = x[hf - 1], b = x[hf];
    return a + (h - (double)hf) * (b - a);
}
static uint8_t f64_as_u8(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

/* lib.rs:191-259 */
void ora_cat_calc_otsu(const uint8_t *rgb, int w, int h, uint8_t *classes) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double px[25];
            int n = 0;
            int x_min = x - 2 < 0 ? 0 : x - 2, x_max = x + 2 > w - 1 ? w - 1 : x + 2;
            int y_min = y - 2 < 0 ? 0 : y - 2, y_max = y + 2 > h - 1 ? h - 1 : y + 2;
            for (int xx = x_min; xx <= x_max; xx++)
                for (int yy = y_min; yy <= y_max; yy++) {
                    size_t i = (size_t)yy * w + xx;
                    px[n++] = (double)ora_cat_grayscale(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
                }
            for (int i = 1; i < n; i++) { /* sort ascending */
                double v = px[i];
                int j = i - 1;
                while (j >= 0 && px[j] > v) { px[j + 1] = px[j]; j--; }
                px[j + 1] = v;
            }
            size_t i = (size_t)y * w + x;
            uint8_t p = ora_cat_grayscale(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
            uint8_t c;
            if ((y > 0 && x > 0) && (px[n - 1] - px[0]) < 5.0) {
                double gray = data_median(px, n);
                c = gray < 60.0 ? BLACK : (gray > 160.0 ? WHITE : OTHER);
            } else {
                if (p >= f64_as_u8(data_quantile(px, n, 0.75))) c = WHITE;
                else if (p <= f64_as_u8(data_quantile(px, n, 0.25))) c = BLACK;
                else c = OTHER;
            }
            classes[i] = c;
        }
}

/* lib.rs:319-334 */
void ora_cat_thresh(const uint8_t *rgb, int w, int h, uint8_t *classes) {
    for (size_t i = 0; i < (size_t)w * h; i++) {
        uint8_t gray = ora_cat_grayscale(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
        classes[i] = gray < 60 ? BLACK : (gray > 160 ? WHITE : OTHER);
    }
}

/* class at flat index, 3 = outside the buffer (the reference would read out of bounds) */
static int cls_at(const uint8_t *c, long long idx, long long n) { return (idx < 0 || idx >= n) ? 3 : c[idx]; }

/* lib.rs:291-309 + 345-400; points in the reference's order: x outer, y inner */
int ora_cat_detect_corners(const uint8_t *c, int w, int h, uint32_t *pts, int cap) {
    long long n = (long long)w * h;
    int np = 0;
    for (int x = 3; x <= w - 3; x++)
        for (int y = 3; y <= h - 3; y++) {
            long long i = (long long)y * w + x;
            if (c[i] != BLACK) continue;
            int ul = c[i - w - 1] == BLACK, ur = c[i - w + 1] == BLACK, dl = c[i + w - 1] == BLACK, dr = c[i + w + 1] == BLACK;
            if (!(ul ^ ur ^ dl ^ dr)) continue;
            int p3 = cls_at(c, i - 3LL * w + 3, n), p7 = cls_at(c, i + 3LL * w + 3, n);
            int p11 = cls_at(c, i + 3LL * w - 3, n), p15 = cls_at(c, i - 3LL * w - 3, n);
            if (p3 == 3 || p7 == 3 || p11 == 3 || p15 == 3) continue; /* reference: out-of-bounds read */
            if (p3 == OTHER || p7 == OTHER || p11 == OTHER || p15 == OTHER) continue;
            if (!((p3 == BLACK) ^ (p7 == BLACK) ^ (p11 == BLACK) ^ (p15 == BLACK))) continue;
            if (np < cap) { pts[2 * np] = (uint32_t)x; pts[2 * np + 1] = (uint32_t)y; }
            np++;
        }
    return np;
}

/* lib.rs:409-476: returns the number of lines pushed for this ordered pair (0, 1 or 2) */
static int check_edge(const uint8_t *c, int w, long long n, long long x1, long long y1, long long x2, long long y2) {
    const long long OFF = 5;
    long long mx = (x1 + x2) / 2, my = (y1 + y2) / 2;
    long long xdiff = (x1 > x2 ? x1 - x2 : x2 - x1), ydiff = (y1 > y2 ? y1 - y2 : y2 - y1);
    int is_v = (x1 == x2) || xdiff < ydiff, is_h = (y1 == y2) || ydiff < xdiff;
    long long m1x = (mx + x1) / 2, m1y = (my + y1) / 2, m2x = (mx + x2) / 2, m2y = (my + y2) / 2;
    int pushed = 0;
    if (is_v) {
        int r1 = cls_at(c, m1y * w + m1x + OFF, n), r2 = cls_at(c, m2y * w + m2x + OFF, n);
        int l1 = cls_at(c, m1y * w + m1x - OFF, n), l2 = cls_at(c, m2y * w + m2x - OFF, n);
        if (l1 < 2 && l2 < 2 && r1 < 2 && r2 < 2)
            if (((l1 == BLACK) ^ (r2 == BLACK)) && ((l2 == BLACK) ^ (r1 == BLACK)) && l1 == l2) pushed++;
    }
    if (is_h) {
        int t1 = cls_at(c, (m1y - OFF) * w + m1x, n), t2 = cls_at(c, (m2y - OFF) * w + m2x, n);
        int b1 = cls_at(c, (m1y + OFF) * w + m1x, n), b2 = cls_at(c, (m2y + OFF) * w + m2x, n);
        if (t1 < 2 && t2 < 2 && b1 < 2 && b2 < 2)
            if (((t1 == BLACK) ^ (b2 == BLACK)) && ((t2 == BLACK) ^ (b1 == BLACK)) && t1 == t2) pushed++;
    }
    return pushed;
}

/* lib.rs:480-499: every point against every point in reverse order */
int ora_cat_check_edges(const uint8_t *c, int w, int h, const uint32_t *pts, int np, uint32_t *lines, int cap) {
    long long n = (long long)w * h;
    int nl = 0;
    for (int i = 0; i < np; i++)
        for (int j = np - 1; j >= 0; j--) {
            int k = check_edge(c, w, n, pts[2 * i], pts[2 * i + 1], pts[2 * j], pts[2 * j + 1]);
            for (int q = 0; q < k; q++) {
                if (nl < cap) { lines[4 * nl] = pts[2 * i]; lines[4 * nl + 1] = pts[2 * i + 1]; lines[4 * nl + 2] = pts[2 * j]; lines[4 * nl + 3] = pts[2 * j + 1]; }
                nl++;
            }
        }
    return nl;
}

/* lib.rs:42-113: recursive find with path compression, union by size (ties keep root1) */
static uint64_t cat_find(uint64_t *parent, uint64_t id) {
    uint64_t root = id;
    while (parent[root] != root) root = parent[root];
    while (parent[id] != root) { uint64_t nx = parent[id]; parent[id] = root; id = nx; } /* same end state as the recursion */
    return root;
}
static void cat_union(uint64_t *parent, uint64_t *sizes, uint64_t a, uint64_t b) {
    uint64_t r1 = cat_find(parent, a), r2 = cat_find(parent, b);
    if (r1 == r2) return;
    if (sizes[r1] < sizes[r2]) { parent[r1] = r2; sizes[r2] += sizes[r1]; }
    else { parent[r2] = r1; sizes[r1] += sizes[r2]; }
}
/* lib.rs:501-549 */
void ora_cat_connected_components(const uint8_t *c, int w, int h, uint64_t *parent, uint64_t *sizes) {
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++) { parent[i] = i; sizes[i] = 1; }
    for (int y = 0; y < h; y++)
        for (int x = 1; x < w - 1; x++) {
            size_t i = (size_t)y * w + x;
            uint8_t p = c[i];
            if (p == OTHER) continue;
            if (c[i - 1] == p) cat_union(parent, sizes, i, i - 1);
            if (y > 0) {
                if (c[i - w] == p) cat_union(parent, sizes, i, i - w);
                if (p == WHITE) {
                    if (c[i - w - 1] == p) cat_union(parent, sizes, i, i - w - 1);
                    if (c[i - w + 1] == p) cat_union(parent, sizes, i, i - w + 1);
                }
            }
        }
}
void ora_cat_connected_components_canonical(const uint8_t *c, int w, int h, uint32_t *roots, uint32_t *sizes_out) {
    size_t n = (size_t)w * h;
    uint64_t *parent = (uint64_t *)malloc(n * 8), *sizes = (uint64_t *)malloc(n * 8);
    uint32_t *minidx = (uint32_t *)malloc(n * 4);
    ora_cat_connected_components(c, w, h, parent, sizes);
    for (size_t i = 0; i < n; i++) minidx[i] = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        uint64_t r = cat_find(parent, i);
        if ((uint32_t)i < minidx[r]) minidx[r] = (uint32_t)i;
    }
    for (size_t i = 0; i < n; i++) {
        uint64_t r = cat_find(parent, i);
        roots[i] = minidx[r];
        sizes_out[i] = (uint32_t)sizes[r];
    }
    free(parent); free(sizes); free(minidx);
}
