// k_cat.hip — "CAT", the experimental detector front-end of crates/chalkydri-apriltags, as HIP kernels.
//
// Entry points mirror the crate's Detector methods (src/lib.rs): calc_otsu :191-259, thresh :319-334,
// detect_corners :291-309 (+ process_pixel :345-400), check_edges :480-499 (+ check_edge :409-476),
// connected_components :501-549, process_frame :265-287.  Outputs are bit-identical to oracle/cat.c, including the
// order the reference produces points (x-major) and lines (i forward, j reversed, vertical test before horizontal).
//
//   k_otsu      one thread per pixel: 5x5 clamped window of f32-FMA gray values, insertion sort in registers,
//               statrs' median / R-8 quartiles in f64, tri-state class.
//   k_corner_flag + scan + k_corner_scatter   ordered stream compaction of the corner test in x-major order.
//   k_edge_count / k_edge_write               one workgroup per first point i: the P ordered pairs (i, j) are
//               tested by the lanes, per-pair push counts are scanned in the reference's reversed-j order.
//   connected_components reuses the union-find tile kernels of k_ccl.hip on the class map (Black->0, White->255,
//               Other->127): CAT's connectivity rule is exactly the one those kernels implement.
#include <vector>

#include "ck_internal.h"

namespace {

enum { BLACK = 0, WHITE = 1, OTHER = 2 };
constexpr int NT = 256;

__device__ __forceinline__ uint8_t grayscale(uint8_t r, uint8_t g, uint8_t b) { // utils.rs:33-46
    float v = __fmaf_rn((float)r, 0.33f, __fmaf_rn((float)g, 0.33f, (float)b * 0.33f));
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
__device__ __forceinline__ uint8_t f64_as_u8(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}
__device__ __forceinline__ double data_quantile(const double *x, int n, double tau) { // statrs 0.18 OrderStatistics::quantile
    double h = ((double)n + 1.0 / 3.0) * tau + 1.0 / 3.0;
    long long hf = (long long)h;
    if (hf <= 0 || tau == 0.0) return x[0];
    if (hf >= (long long)n) return x[n - 1];
    double a = x[hf - 1], b = x[hf];
    return a + (h - (double)hf) * (b - a);
}

__global__ __launch_bounds__(NT) void k_otsu(const uint8_t *__restrict__ rgb, int w, int h, uint8_t *__restrict__ classes) {
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= w * h) return;
    int y = i / w, x = i - y * w;
    double px[25];
    int n = 0;
    int x_min = x - 2 < 0 ? 0 : x - 2, x_max = x + 2 > w - 1 ? w - 1 : x + 2;
    int y_min = y - 2 < 0 ? 0 : y - 2, y_max = y + 2 > h - 1 ? h - 1 : y + 2;
    for (int xx = x_min; xx <= x_max; xx++)
        for (int yy = y_min; yy <= y_max; yy++) {
            size_t k = (size_t)yy * w + xx;
            double v = (double)grayscale(rgb[k * 3], rgb[k * 3 + 1], rgb[k * 3 + 2]);
            int j = n - 1; // insertion keeps px sorted ascending
            while (j >= 0 && px[j] > v) { px[j + 1] = px[j]; j--; }
            px[j + 1] = v;
            n++;
        }
    uint8_t p = grayscale(rgb[(size_t)i * 3], rgb[(size_t)i * 3 + 1], rgb[(size_t)i * 3 + 2]);
    uint8_t c;
    if ((y > 0 && x > 0) && (px[n - 1] - px[0]) < 5.0) {
        int k = n / 2;
        double gray = (n % 2 != 0) ? px[k] : (px[k > 0 ? k - 1 : 0] + px[k]) / 2.0;
        c = gray < 60.0 ? BLACK : (gray > 160.0 ? WHITE : OTHER);
    } else {
        if (p >= f64_as_u8(data_quantile(px, n, 0.75))) c = WHITE;
        else if (p <= f64_as_u8(data_quantile(px, n, 0.25))) c = BLACK;
        else c = OTHER;
    }
    classes[i] = c;
}

__global__ __launch_bounds__(NT) void k_fixed_thresh(const uint8_t *__restrict__ rgb, int n, uint8_t *__restrict__ classes) {
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    uint8_t gray = grayscale(rgb[(size_t)i * 3], rgb[(size_t)i * 3 + 1], rgb[(size_t)i * 3 + 2]);
    classes[i] = gray < 60 ? BLACK : (gray > 160 ? WHITE : OTHER);
}

__device__ __forceinline__ int cls_at(const uint8_t *c, long long idx, long long n) { return (idx < 0 || idx >= n) ? 3 : c[idx]; }

// flag index t enumerates pixels in the reference's loop order: x outer (3..=w-3), y inner (3..=h-3)
__global__ __launch_bounds__(NT) void k_corner_flag(const uint8_t *__restrict__ c, int w, int h, uint32_t *__restrict__ flags, int total) {
    int t = blockIdx.x * NT + threadIdx.x;
    if (t >= total) return;
    int ny = h - 5; // number of y values
    int x = 3 + t / ny, y = 3 + t % ny;
    long long n = (long long)w * h, i = (long long)y * w + x;
    uint32_t f = 0;
    if (c[i] == BLACK) {
        int ul = c[i - w - 1] == BLACK, ur = c[i - w + 1] == BLACK, dl = c[i + w - 1] == BLACK, dr = c[i + w + 1] == BLACK;
        if (ul ^ ur ^ dl ^ dr) {
            int p3 = cls_at(c, i - 3LL * w + 3, n), p7 = cls_at(c, i + 3LL * w + 3, n);
            int p11 = cls_at(c, i + 3LL * w - 3, n), p15 = cls_at(c, i - 3LL * w - 3, n);
            if (p3 < 2 && p7 < 2 && p11 < 2 && p15 < 2 && ((p3 == BLACK) ^ (p7 == BLACK) ^ (p11 == BLACK) ^ (p15 == BLACK))) f = 1;
        }
    }
    flags[t] = f;
}

// ---- device-wide exclusive scan of u32 (three small kernels) ---------------------------------------------------
constexpr int SB = 1024; // elements per scan block
__global__ __launch_bounds__(NT) void k_scan_blocks(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *__restrict__ sums, int n) {
    __shared__ uint32_t s[SB];
    int base = blockIdx.x * SB;
    for (int k = threadIdx.x; k < SB; k += NT) s[k] = (base + k < n) ? in[base + k] : 0u;
    __syncthreads();
    for (int d = 1; d < SB; d <<= 1) {
        uint32_t v[SB / NT];
        for (int q = 0; q < SB / NT; q++) { int k = threadIdx.x + q * NT; v[q] = (k >= d) ? s[k - d] : 0u; }
        __syncthreads();
        for (int q = 0; q < SB / NT; q++) { int k = threadIdx.x + q * NT; s[k] += v[q]; }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < SB; k += NT)
        if (base + k < n) out[base + k] = s[k] - in[base + k]; // exclusive within the block
    if (threadIdx.x == 0) sums[blockIdx.x] = s[SB - 1];
}
__global__ void k_scan_sums(uint32_t *sums, int nb, uint32_t *total) { // one thread: nb is a few thousand at most
    uint32_t acc = 0;
    for (int i = 0; i < nb; i++) { uint32_t v = sums[i]; sums[i] = acc; acc += v; }
    *total = acc;
}
__global__ __launch_bounds__(NT) void k_corner_scatter(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ excl,
                                                       const uint32_t *__restrict__ sums, int h, int total, uint32_t *__restrict__ pts, int cap) {
    int t = blockIdx.x * NT + threadIdx.x;
    if (t >= total || !flags[t]) return;
    uint32_t pos = excl[t] + sums[t / SB];
    if (pos >= (uint32_t)cap) return;
    int ny = h - 5;
    pts[2 * pos] = (uint32_t)(3 + t / ny);
    pts[2 * pos + 1] = (uint32_t)(3 + t % ny);
}

// lib.rs:409-476: number of lines this ordered pair pushes (0, 1 or 2)
__device__ __forceinline__ int check_edge(const uint8_t *c, int w, long long n, long long x1, long long y1, long long x2, long long y2) {
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
// one workgroup per first point i; rowcount[i] = lines pushed by (i, *)
__global__ __launch_bounds__(NT) void k_edge_count(const uint8_t *__restrict__ c, int w, int h, const uint32_t *__restrict__ pts, int np,
                                                   uint32_t *__restrict__ rowcount) {
    __shared__ uint32_t s[NT];
    int i = blockIdx.x;
    long long n = (long long)w * h;
    uint32_t acc = 0;
    for (int j = threadIdx.x; j < np; j += NT) acc += (uint32_t)check_edge(c, w, n, pts[2 * i], pts[2 * i + 1], pts[2 * j], pts[2 * j + 1]);
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = NT / 2; d > 0; d >>= 1) { if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) rowcount[i] = s[0];
}
// writes row i's lines at rowoff[i]...; inside the row the order is j = np-1 .. 0 (points.iter().rev(), lib.rs:493)
__global__ __launch_bounds__(NT) void k_edge_write(const uint8_t *__restrict__ c, int w, int h, const uint32_t *__restrict__ pts, int np,
                                                   const uint32_t *__restrict__ rowexcl, const uint32_t *__restrict__ rowsums,
                                                   uint32_t *__restrict__ lines, int cap) {
    __shared__ uint32_t s[NT];
    __shared__ uint32_t carry;
    int i = blockIdx.x;
    long long n = (long long)w * h;
    if (threadIdx.x == 0) carry = rowexcl[i] + rowsums[i / SB];
    __syncthreads();
    for (int base = 0; base < np; base += NT) {
        int r = base + threadIdx.x;         // position in the reversed order
        int j = np - 1 - r;
        uint32_t k = 0;
        if (r < np) k = (uint32_t)check_edge(c, w, n, pts[2 * i], pts[2 * i + 1], pts[2 * j], pts[2 * j + 1]);
        s[threadIdx.x] = k;
        __syncthreads();
        for (int d = 1; d < NT; d <<= 1) {
            uint32_t v = (threadIdx.x >= d) ? s[threadIdx.x - d] : 0u;
            __syncthreads();
            s[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t pos = carry + s[threadIdx.x] - k;
        for (uint32_t q = 0; q < k; q++)
            if (pos + q < (uint32_t)cap) {
                uint32_t *l = lines + 4 * (size_t)(pos + q);
                l[0] = pts[2 * i]; l[1] = pts[2 * i + 1]; l[2] = pts[2 * j]; l[3] = pts[2 * j + 1];
            }
        __syncthreads();
        if (threadIdx.x == NT - 1) carry += s[NT - 1];
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void k_class_to_tri(const uint8_t *__restrict__ classes, int w, int h, int stride, uint8_t *__restrict__ out) {
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= w * h) return;
    int y = i / w, x = i - y * w;
    uint8_t c = classes[i];
    out[(size_t)y * stride + x] = c == BLACK ? 0 : (c == WHITE ? 255 : 127);
}
// canonical labels -> CAT roots/sizes: pixels without a component are singleton sets (UnionFind::new, lib.rs:57-60)
__global__ __launch_bounds__(NT) void k_cat_roots(uint32_t *__restrict__ roots, uint32_t *__restrict__ sizes, int n) {
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    if (roots[i] == CK_LBL_INVALID) { roots[i] = (uint32_t)i; sizes[i] = 1; }
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, sizeof(T) * (n ? n : 1)) == hipSuccess ? CK_OK : CK_ENOMEM; }
};

int exclusive_scan(hipStream_t st, const uint32_t *d_in, uint32_t *d_excl, uint32_t *d_sums, uint32_t *d_total, int n) {
    int nb = (n + SB - 1) / SB;
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nb), dim3(NT), 0, st, d_in, d_excl, d_sums, n);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1), 0, st, d_sums, nb, d_total);
    return CK_OK;
}

int corners_device(ck_handle *h, const uint8_t *d_classes, int w, int ht, uint32_t *d_pts, int cap, int *n_points) {
    if (w < 7 || ht < 7) { *n_points = 0; return CK_OK; }
    int total = (w - 5) * (ht - 5);
    DevBuf<uint32_t> flags, excl, sums, tot;
    if (flags.alloc((size_t)total) || excl.alloc((size_t)total) || sums.alloc((size_t)(total + SB - 1) / SB) || tot.alloc(1)) return CK_ENOMEM;
    hipLaunchKernelGGL(k_corner_flag, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, h->stream, d_classes, w, ht, flags.p, total);
    exclusive_scan(h->stream, flags.p, excl.p, sums.p, tot.p, total);
    hipLaunchKernelGGL(k_corner_scatter, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, h->stream, flags.p, excl.p, sums.p, ht, total, d_pts, cap);
    uint32_t t = 0;
    CK_HIP(hipMemcpyAsync(&t, tot.p, sizeof t, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    *n_points = (int)t;
    return CK_OK;
}

int edges_device(ck_handle *h, const uint8_t *d_classes, int w, int ht, const uint32_t *d_pts, int np, uint32_t *d_lines, int cap, int *n_lines) {
    *n_lines = 0;
    if (np <= 0) return CK_OK;
    DevBuf<uint32_t> rowcount, excl, sums, tot;
    if (rowcount.alloc((size_t)np) || excl.alloc((size_t)np) || sums.alloc((size_t)(np + SB - 1) / SB) || tot.alloc(1)) return CK_ENOMEM;
    hipLaunchKernelGGL(k_edge_count, dim3((unsigned)np), dim3(NT), 0, h->stream, d_classes, w, ht, d_pts, np, rowcount.p);
    exclusive_scan(h->stream, rowcount.p, excl.p, sums.p, tot.p, np);
    hipLaunchKernelGGL(k_edge_write, dim3((unsigned)np), dim3(NT), 0, h->stream, d_classes, w, ht, d_pts, np, excl.p, sums.p, d_lines, cap);
    uint32_t t = 0;
    CK_HIP(hipMemcpyAsync(&t, tot.p, sizeof t, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    *n_lines = (int)t;
    return CK_OK;
}

} // namespace

extern "C" int ck_cat_calc_otsu(ck_handle_t *h, const uint8_t *rgb, int32_t w, int32_t ht, uint8_t *classes_out) {
    if (!h || !rgb || !classes_out || w < 1 || ht < 1) return CK_EINVAL;
    CK_HIP(hipSetDevice(h->device));
    size_t n = (size_t)w * ht;
    DevBuf<uint8_t> drgb, dcls;
    if (drgb.alloc(n * 3) || dcls.alloc(n)) return CK_ENOMEM;
    CK_HIP(hipMemcpyAsync(drgb.p, rgb, n * 3, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_otsu, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, h->stream, drgb.p, w, ht, dcls.p);
    CK_HIP(hipGetLastError());
    CK_HIP(hipMemcpyAsync(classes_out, dcls.p, n, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    return CK_OK;
}

extern "C" int ck_cat_thresh(ck_handle_t *h, const uint8_t *rgb, int32_t w, int32_t ht, uint8_t *classes_out) {
    if (!h || !rgb || !classes_out || w < 1 || ht < 1) return CK_EINVAL;
    CK_HIP(hipSetDevice(h->device));
    size_t n = (size_t)w * ht;
    DevBuf<uint8_t> drgb, dcls;
    if (drgb.alloc(n * 3) || dcls.alloc(n)) return CK_ENOMEM;
    CK_HIP(hipMemcpyAsync(drgb.p, rgb, n * 3, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_fixed_thresh, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, h->stream, drgb.p, (int)n, dcls.p);
    CK_HIP(hipGetLastError());
    CK_HIP(hipMemcpyAsync(classes_out, dcls.p, n, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    return CK_OK;
}

extern "C" int ck_cat_detect_corners(ck_handle_t *h, const uint8_t *classes, int32_t w, int32_t ht, uint32_t *points_xy, int32_t cap, int32_t *n_points) {
    if (!h || !classes || !points_xy || !n_points || cap < 0 || w < 1 || ht < 1) return CK_EINVAL;
    CK_HIP(hipSetDevice(h->device));
    size_t n = (size_t)w * ht;
    DevBuf<uint8_t> dcls; DevBuf<uint32_t> dpts;
    if (dcls.alloc(n) || dpts.alloc((size_t)2 * cap)) return CK_ENOMEM;
    CK_HIP(hipMemcpyAsync(dcls.p, classes, n, hipMemcpyHostToDevice, h->stream));
    int np = 0;
    int rc = corners_device(h, dcls.p, w, ht, dpts.p, cap, &np);
    if (rc != CK_OK) return rc;
    CK_HIP(hipGetLastError());
    int ncopy = np < cap ? np : cap;
    CK_HIP(hipMemcpy(points_xy, dpts.p, sizeof(uint32_t) * 2 * (size_t)ncopy, hipMemcpyDeviceToHost));
    *n_points = np;
    return CK_OK;
}

extern "C" int ck_cat_check_edges(ck_handle_t *h, const uint8_t *classes, int32_t w, int32_t ht, const uint32_t *points_xy, int32_t n_points,
                                  uint32_t *lines_xyxy, int32_t cap, int32_t *n_lines) {
    if (!h || !classes || !lines_xyxy || !n_lines || n_points < 0 || (n_points > 0 && !points_xy) || cap < 0 || w < 1 || ht < 1) return CK_EINVAL;
    CK_HIP(hipSetDevice(h->device));
    size_t n = (size_t)w * ht;
    DevBuf<uint8_t> dcls; DevBuf<uint32_t> dpts, dlines;
    if (dcls.alloc(n) || dpts.alloc((size_t)2 * n_points) || dlines.alloc((size_t)4 * cap)) return CK_ENOMEM;
    CK_HIP(hipMemcpyAsync(dcls.p, classes, n, hipMemcpyHostToDevice, h->stream));
    if (n_points) CK_HIP(hipMemcpyAsync(dpts.p, points_xy, sizeof(uint32_t) * 2 * (size_t)n_points, hipMemcpyHostToDevice, h->stream));
    int nl = 0;
    int rc = edges_device(h, dcls.p, w, ht, dpts.p, n_points, dlines.p, cap, &nl);
    if (rc != CK_OK) return rc;
    CK_HIP(hipGetLastError());
    int ncopy = nl < cap ? nl : cap;
    CK_HIP(hipMemcpy(lines_xyxy, dlines.p, sizeof(uint32_t) * 4 * (size_t)ncopy, hipMemcpyDeviceToHost));
    *n_lines = nl;
    return CK_OK;
}

extern "C" int ck_cat_connected_components(ck_handle_t *h, const uint8_t *classes, int32_t w, int32_t ht, uint32_t *roots_out, uint32_t *sizes_out) {
    if (!h || !classes || !roots_out || !sizes_out) return CK_EINVAL;
    if (w != h->qw || ht != h->qh) return CK_EINVAL; // the detector is created for one geometry (Detector::new, lib.rs:158)
    CK_HIP(hipSetDevice(h->device));
    size_t n = (size_t)w * ht;
    const int stride = (w + 15) / 16 * 16;
    DevBuf<uint8_t> dcls, dtri; DevBuf<uint32_t> droots, dsizes;
    if (dcls.alloc(n) || dtri.alloc((size_t)stride * ht) || droots.alloc(n) || dsizes.alloc(n)) return CK_ENOMEM;
    CK_HIP(hipMemcpyAsync(dcls.p, classes, n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_class_to_tri, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, h->stream, dcls.p, w, ht, stride, dtri.p);
    int rc = ck_launch_threshold_segment(h, dtri.p, stride, (size_t)stride * ht, 1, true);
    if (rc != CK_OK) return rc;
    rc = ck_launch_canonical_labels(h, 1, droots.p, dsizes.p);
    if (rc != CK_OK) return rc;
    hipLaunchKernelGGL(k_cat_roots, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, h->stream, droots.p, dsizes.p, (int)n);
    CK_HIP(hipGetLastError());
    CK_HIP(hipMemcpyAsync(roots_out, droots.p, n * 4, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipMemcpyAsync(sizes_out, dsizes.p, n * 4, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    return CK_OK;
}

extern "C" int ck_cat_process_frame(ck_handle_t *h, const uint8_t *rgb, size_t rgb_len, int32_t w, int32_t ht, uint8_t *classes_out,
                                    uint32_t *points_xy, int32_t point_cap, int32_t *n_points, uint32_t *lines_xyxy, int32_t line_cap,
                                    int32_t *n_lines) {
    if (!h || !rgb || !classes_out || !points_xy || !n_points || !lines_xyxy || !n_lines || w < 1 || ht < 1 || point_cap < 0 || line_cap < 0) return CK_EINVAL;
    if (rgb_len != (size_t)w * ht * 3) return CK_EINVAL; // the reference asserts (lib.rs:267)
    CK_HIP(hipSetDevice(h->device));
    size_t n = (size_t)w * ht;
    DevBuf<uint8_t> drgb, dcls; DevBuf<uint32_t> dpts, dlines;
    if (drgb.alloc(n * 3) || dcls.alloc(n) || dpts.alloc((size_t)2 * point_cap) || dlines.alloc((size_t)4 * line_cap)) return CK_ENOMEM;
    CK_HIP(hipMemcpyAsync(drgb.p, rgb, n * 3, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_otsu, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, h->stream, drgb.p, w, ht, dcls.p);
    int np = 0, nl = 0;
    int rc = corners_device(h, dcls.p, w, ht, dpts.p, point_cap, &np);
    if (rc != CK_OK) return rc;
    int npu = np < point_cap ? np : point_cap;
    rc = edges_device(h, dcls.p, w, ht, dpts.p, npu, dlines.p, line_cap, &nl);
    if (rc != CK_OK) return rc;
    CK_HIP(hipGetLastError());
    CK_HIP(hipMemcpy(classes_out, dcls.p, n, hipMemcpyDeviceToHost));
    CK_HIP(hipMemcpy(points_xy, dpts.p, sizeof(uint32_t) * 2 * (size_t)npu, hipMemcpyDeviceToHost));
    int nlu = nl < line_cap ? nl : line_cap;
    CK_HIP(hipMemcpy(lines_xyxy, dlines.p, sizeof(uint32_t) * 4 * (size_t)nlu, hipMemcpyDeviceToHost));
    *n_points = np; *n_lines = nl;
    return CK_OK;
}
