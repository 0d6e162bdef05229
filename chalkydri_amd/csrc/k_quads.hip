// k_quads.hip — quad fitting.  Two forms of the same function:
//   * k_fit: one workgroup per gradient cluster through all six phases below (calls of a few frames: the size classes side by
//     side on three streams; and batches too small for the split form to pay);
//   * the SPLIT fit of a batch, three kernels cut where the fit's parallelism changes: k_seq (per cluster: phases 1-2, the cluster's
//     points leave as an "extended sequence", ck_internal.h) -> k_chunk (per POSITION of a frame's sequences, no notion of clusters:
//     phases 3-4, results filed by position) -> k_tail (one wave per cluster: phases 5-6).  ck_launch_fit_quads decides.
//
// Replaces the quad-fit and edge-refinement stages of the external AprilTag-3 detector reached at
// crates/apriltags/src/lib.rs:301.  Bit-exact with oracle/detector.c (fit_quad, refine_edges): all decisions
// are made on int64 moments or on doubles evaluated in the oracle's order with -ffp-contract=off.
//
// Per cluster (template: NTH threads, up to CAP points, chunks of CH points):
//   1. points arrive packed (4 bytes, ck_internal.h); bounding box and border direction are integer reductions (DPP);
//   2. exact 60-bit angular keys -> bucketed rank sort (counting pass over angle buckets + ranks inside a bucket; LDS
//      for the classes whose keys fit in registers, a per-workgroup L2-resident scratch for the large one; the bitonic
//      network remains as the fallback for clusters whose angles pile up) -> duplicate coordinates dropped (ballot scan);
//   3. gradient weights: one u16 gather per point from the weight image k_weight_image wrote;
//   4. chunk loop: moment prefix sums over the chunk + halo (DPP wave scans; first-order sums in 32 bit), windowed
//      line-fit error, 7-tap smoothing fused with the maxima test, maxima appended to a per-cluster list;
//   5. the max_nmaxima strongest maxima (rank-based selection), moment sums at them from a 32/128-point table, one line
//      fit per ordered pair, every 4-subset evaluated in parallel (argmin with the oracle's lexicographic tie-break);
//   6. lines, corners, area/angle/winding checks on four lanes, wave-parallel edge refinement.
// Clusters are dispatched through per-size-class work lists built by k_classify (<= 256, <= 512, <= 1024, <= 2048, <= 4096,
// <= 8192, <= 16384 points, and the rest); each variant is a persistent grid whose workgroups take a first chunk of their list by index and the
// rest from a dequeue counter.  The phases can be cut short for measurements with CK_FIT_STOP_AFTER (tools/ablate_fit.sh).
#include <stdlib.h>

#include "ck_internal.h"

namespace {

struct FitArgs {
    const uint8_t *qim; int qw, qh, qstride; size_t qpitch;     // image the clusters came from
    const uint16_t *wimg;                                       // its gradient-magnitude weights, [n][qh][qw]
    const uint8_t *im; int w, h, stride; size_t pitch;          // full resolution image
    int decimate, refine, max_nmaxima, min_tag_width, normal_ok, reversed_ok;
    double cos_critical, max_mse;
    ck_stage_ws ws;
    int stop_after;         // diagnostics (CK_FIT_STOP_AFTER): end every cluster after phase k; 99 = run everything
    int guided;             // chunks handed out by the work counter shrink towards the end of the list
    int list_cap;           // capacity of one class list
    const uint32_t *list;   // work list of this size class: frame << 20 | cluster index
    const uint32_t *list_count;
    uint32_t *head;         // dequeue counter
};

struct M6 { long long Mx, My, Mxx, Mxy, Myy, W; };
__device__ __forceinline__ M6 m6_add(M6 a, M6 b) { a.Mx += b.Mx; a.My += b.My; a.Mxx += b.Mxx; a.Mxy += b.Mxy; a.Myy += b.Myy; a.W += b.W; return a; }
__device__ __forceinline__ M6 m6_sub(M6 a, M6 b) { a.Mx -= b.Mx; a.My -= b.My; a.Mxx -= b.Mxx; a.Mxy -= b.Mxy; a.Myy -= b.Myy; a.W -= b.W; return a; }
__device__ __forceinline__ M6 m6_zero() { M6 z = {0, 0, 0, 0, 0, 0}; return z; }

__device__ __forceinline__ unsigned long long angle_key(int x, int y, int xmin, int xmax, int ymin, int ymax) {
    // everything fits 32 bits (|dx|, |dy| < 2^15) and the octant is picked without branches:
    //   quadrant q: 0 (dy<0,dx<0)  1 (dy<0,dx>=0)  2 (dy>=0,dx>0)  3 (dy>=0,dx<=0);  inside a quadrant the second octant
    //   (fraction counted downwards, inv) starts where |dy| exceeds |dx| for even q, where it stops exceeding it for odd q;
    //   the fraction is always min/max of the two magnitudes
    const int dx = 4 * x - 2 * (xmin + xmax) - 1;
    const int dy = 4 * y - 2 * (ymin + ymax) + 1;
    const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    const int q = dy < 0 ? (dx < 0 ? 0 : 1) : (dx > 0 ? 2 : 3);
    const int steep = ay > ax ? 1 : 0;
    const int inv = (q & 1) ? 1 - steep : steep;
    const int oct = 2 * q + inv;
    const int num = ax < ay ? ax : ay, den = ax < ay ? ay : ax;
    // floor(num * 2^30 / den), num <= den < 2^15 (the oracle does two 15-bit long-division steps).  Here: one IEEE f64
    // division.  num * 2^30 < 2^45 and den are exact doubles; a quotient that is not an integer lies at least 1/den > 2^-15
    // away from one while the rounding error is below 2^-22, so truncating the correctly rounded quotient gives the exact
    // floor — and two integer divisions (~100 instructions on this hardware) become ~20.  den >= 1: dx is odd.
    const double qd = ((double)(uint32_t)num * 1073741824.0) / (double)(uint32_t)den;
    unsigned long long frac = (unsigned long long)(uint32_t)qd;
    if (inv) frac = (1ull << 30) - frac;
    return ((unsigned long long)oct << 57) | (frac << 26) | ((unsigned long long)x << 13) | (unsigned long long)y;
}
__device__ __forceinline__ uint32_t isqrt_u32(uint32_t v) {
    uint32_t r = (uint32_t)__fsqrt_rn((float)v); // v <= 2*255^2: exact after the +-1 correction
    if (r * r > v) r--;
    if ((r + 1) * (r + 1) <= v) r++;
    return r;
}

// an integer below 2^52 as a double: its bits under the exponent of 2^52, less 2^52 (exact; the generic 64-bit conversion is two
// conversions and a scaling).  The moment sums of a window or a cluster side are such integers (<= 49 140 points x 362 x 8192^2 < 2^51).
__device__ __forceinline__ double f64_of_u52(unsigned long long x) {
    return __longlong_as_double((long long)(0x4330000000000000ull | x)) - 4503599627370496.0;
}

// line fit from a moment sum over N points (half-pixel units) — mirrors fit_line() of the oracle exactly
__device__ __forceinline__ void fit_line_m(const M6 &m, int N, double *lineparm, double *err, double *mse) {
    // (the sums of a cluster side are non-negative integers below 2^52: f64_of_u52 gives the same doubles as the casts)
    double inv = 1.0 / f64_of_u52((unsigned long long)m.W); // one reciprocal, five products — same operations as the oracle
    double Ex = (0.5 * f64_of_u52((unsigned long long)m.Mx)) * inv;
    double Ey = (0.5 * f64_of_u52((unsigned long long)m.My)) * inv;
    double Cxx = (0.25 * f64_of_u52((unsigned long long)m.Mxx)) * inv - Ex * Ex;
    double Cxy = (0.25 * f64_of_u52((unsigned long long)m.Mxy)) * inv - Ex * Ey;
    double Cyy = (0.25 * f64_of_u52((unsigned long long)m.Myy)) * inv - Ey * Ey;
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

__device__ const double k_smooth[7] = {0.011108996538242306, 0.1353352832366127, 0.6065306597126334, 1.0,
                                       0.6065306597126334, 0.1353352832366127, 0.011108996538242306};

// s_waitcnt vmcnt(0) as an instruction the compiler's own wait insertion knows about (gfx9 encoding: vmcnt in bits 3:0 and 15:14,
// expcnt 6:4 and lgkmcnt 11:8 left at "do not wait"): placed BEFORE a prefetch is issued, so that what the iteration is about to use
// is known to be there and the compiler's later waits (which, inside loops, are for everything outstanding) find nothing to sit on
constexpr int WAIT_VMCNT0 = 0x0F70;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "WAIT_VMCNT0 is the gfx9-family encoding of s_waitcnt (this library is written for gfx950): another target needs its own"
#endif
// Barriers that order LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which the compiler turns into a wait for
// EVERY outstanding memory operation — a prefetch issued for the next cluster would be waited for at the first barrier behind it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// a workgroup of one wave: its LDS operations are carried out in order, only the compiler has to keep them in place
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

template <int NTH>
struct Block {
    static constexpr int NW = NTH / 64;
    __device__ __forceinline__ static void sync() { if constexpr (NTH == 64) wave_sync(); else __syncthreads(); }
    // inclusive sum over the workgroup of one 64-bit value per thread; scratch: NW+1 values of LDS
    __device__ static long long scan_incl(long long v, long long *scratch, long long *total) {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        long long x = (long long)wave_scan_u64((unsigned long long)v);
        if (NW > 1) {
            if (lane == 63) scratch[wv] = x;
            __syncthreads();
            long long base = 0, tot = 0;
            for (int k = 0; k < NW; k++) { long long t = scratch[k]; if (k < wv) base += t; tot += t; }
            x += base;
            if (total) *total = tot;
            __syncthreads();
        } else if (total) *total = __shfl(x, 63, 64);
        return x;
    }
    __device__ static int reduce_min(int v, int *scratch) {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
        if (NW > 1) {
            if (lane == 0) scratch[wv] = v;
            __syncthreads();
            int r = scratch[0];
            for (int k = 1; k < NW; k++) r = min(r, scratch[k]);
            __syncthreads();
            return r;
        }
        return v;
    }
    __device__ static long long reduce_add(long long v, long long *scratch) {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        {
            const unsigned long long t = wave_scan_u64((unsigned long long)v); // lane 63 holds the wave's sum
            v = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(t >> 32), 63) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, 63));
        }
        if (NW > 1) {
            if (lane == 0) scratch[wv] = v;
            __syncthreads();
            long long r = 0;
            for (int k = 0; k < NW; k++) r += scratch[k];
            __syncthreads();
            return r;
        }
        return v;
    }
    // inclusive scan of three 64-bit and three 32-bit values per thread, one barrier pair; both arrays are replaced.
    // scratch: 5 * NW long long (the 32-bit values travel as a second array inside it)
    __device__ static void scan_incl_3x64_3x32(unsigned long long a[3], uint32_t b[3], long long *scratch) {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int q = 0; q < 3; q++) { a[q] = wave_scan_u64(a[q]); b[q] = wave_scan_u32(b[q]); }
        if (NW > 1) {
            unsigned long long *s64 = reinterpret_cast<unsigned long long *>(scratch);
            uint32_t *s32 = reinterpret_cast<uint32_t *>(scratch + 3 * NW);
            if (lane == 63)
#pragma unroll
                for (int q = 0; q < 3; q++) { s64[q * NW + wv] = a[q]; s32[q * NW + wv] = b[q]; }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 3; q++) {
                unsigned long long ba = 0;
                uint32_t bb = 0;
                for (int k = 0; k < wv; k++) { ba += s64[q * NW + k]; bb += s32[q * NW + k]; }
                a[q] += ba; b[q] += bb;
            }
            __syncthreads();
        }
    }
    // minima of four values per thread over the workgroup (every thread gets them), one barrier pair
    __device__ static void reduce_min4(int v[4], int *scratch /* [4*NW] */) {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = wave_min_i32(v[q]);
        if (NW > 1) {
            if (lane == 0)
#pragma unroll
                for (int q = 0; q < 4; q++) scratch[q * NW + wv] = v[q];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int r = scratch[q * NW];
                for (int k = 1; k < NW; k++) r = min(r, scratch[q * NW + k]);
                v[q] = r;
            }
            __syncthreads();
        }
    }
    // inclusive count of a 0/1 flag over the workgroup: ballot + popcount, one barrier pair
    __device__ static int scan_flag(int flag, int *scratch /* [NW] */, int *total) {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const unsigned long long bal = __ballot(flag != 0);
        int incl = __popcll(bal & ((2ull << lane) - 1ull));
        int tot = __popcll(bal);
        if (NW > 1) {
            if (lane == 0) scratch[wv] = tot;
            __syncthreads();
            int base = 0;
            tot = 0;
            for (int k = 0; k < NW; k++) { int t = scratch[k]; if (k < wv) base += t; tot += t; }
            incl += base;
            __syncthreads();
        }
        *total = tot;
        return incl;
    }
};

// Diagnostic build only (-DCK_FIT_PROFILE): per-phase cycle totals of k_fit, written to a buffer of their own.
#ifdef CK_FIT_PROFILE
__device__ unsigned long long g_fit_prof[3][16];
#define PROF_DECL unsigned long long prof_t0 = __builtin_readcyclecounter(); const int prof_cls = (CAP == 512) ? 0 : (CAP <= 4096 ? 1 : 2)
#define PROF(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_fit_prof[prof_cls][k], t_ - prof_t0); prof_t0 = t_; } while (0)
#else
#define PROF_DECL
#define PROF(k)
#endif

constexpr int HALO = 24;   // 20 (window) + 3 (smoothing) + 1 (maxima neighbour)
constexpr int MAXSEL = 12; // largest max_nmaxima supported

struct PairFit { double err, mse, nx, ny; };

// all 4-subsets of {0..MAXSEL-1}, packed m0<<12|m1<<8|m2<<4|m3
// all pairs a < b of {0..MAXSEL-1}, ordered by b then a and packed a << 4 | b: for nsel selected maxima the first
// nsel*(nsel-1)/2 entries are exactly the pairs below nsel
struct PairTable {
    uint8_t v[MAXSEL * (MAXSEL - 1) / 2];
    constexpr PairTable() : v() {
        int k = 0;
        for (int b = 1; b < MAXSEL; b++)
            for (int a = 0; a < b; a++) v[k++] = (uint8_t)((a << 4) | b);
    }
};
__device__ const PairTable g_pair_table{};
struct ComboTable {
    uint16_t v[495];
    constexpr ComboTable() : v() {
        int k = 0;
        for (int m3 = 3; m3 < MAXSEL; m3++) // ordered by the largest member: the subsets of {0..nsel-1} are the first C(nsel, 4) entries
            for (int m2 = 2; m2 < m3; m2++)
                for (int m1 = 1; m1 < m2; m1++)
                    for (int m0 = 0; m0 < m1; m0++) v[k++] = (uint16_t)((m0 << 12) | (m1 << 8) | (m2 << 4) | m3);
    }
};
__device__ const ComboTable g_combo_table{}; // built at compile time: nothing to upload, valid on every device of the process

// Reciprocals of the window weights of k_chunk: a window holds at most 2 * 20 + 1 points of weight <= 362, so its weight sum is an
// integer below INV_N and 1.0 / W a table entry — the IEEE quotient, formed by the compiler: the same bits as the device's division
// (a dozen instructions around v_rcp_f64, 70 clocks of a SIMD per position by tools/probes/valu_rate_probe.hip) for one 8-byte load
// from a 116 KB table that lives in L2.  A position nobody wrote can hold any weight: beyond the table it divides.
constexpr int INV_N = 41 * 362 + 1;
struct InvTable {
    double v[INV_N];
    constexpr InvTable() : v() {
        v[0] = __builtin_huge_val(); // 1.0 / 0.0
        for (int n = 1; n < INV_N; n++) v[n] = 1.0 / (double)n;
    }
};
__device__ const InvTable g_inv_table{};

__device__ __forceinline__ int wrap_index(int i, int sz) { // i in [-HALO, sz + CH + HALO): bring into [0, sz)
    if (i < 0) i += sz;   // one step each way is enough: i >= -HALO >= -sz, and a span ends before sz + HALO <= 2 * sz
    if (i >= sz) i -= sz;
    return i;
}
__device__ __forceinline__ M6 moments_of(uint32_t xy, uint32_t Wt) {
    // W <= 362, X, Y <= 8192: the first-order products fit 32 bits, the second-order ones are one 32x32->64 multiply each
    const uint32_t X = (xy >> 13) + 1, Y = (xy & 0x1FFF) + 1;
    const uint32_t wx = Wt * X, wy = Wt * Y;
    M6 m;
    m.Mx = (long long)wx; m.My = (long long)wy;
    m.Mxx = (long long)((unsigned long long)wx * X); m.Mxy = (long long)((unsigned long long)wx * Y); m.Myy = (long long)((unsigned long long)wy * Y);
    m.W = (long long)Wt;
    return m;
}


// Bitonic sort of CAP = NTH*EPL keys, thread t holding elements t*EPL .. t*EPL+EPL-1 in registers.  Strides below EPL are
// compare-exchanges inside a thread, strides below 64*EPL are lane exchanges (ds_bpermute, no memory, no barrier); only
// strides that cross a wave go through the LDS buffer.  Always sorts all CAP slots (unused ones hold ~0).
template <int NTH, int EPL, int K, int J>
__device__ __forceinline__ void sort_stage(unsigned long long (&k)[EPL], unsigned long long *buf) {
    const int t = threadIdx.x;
    if constexpr (J < EPL) {
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            if ((e & J) == 0) {
                const int i = t * EPL + e;
                const bool up = (i & K) == 0;
                unsigned long long a = k[e], b = k[e | J];
                const bool sw = (a > b) == up;
                k[e] = sw ? b : a; k[e | J] = sw ? a : b;
            }
        }
    } else {
        constexpr int TJ = J / EPL;       // partner thread = t ^ TJ
        const bool lower = (t & TJ) == 0; // this thread holds the smaller index of every pair
        if constexpr (TJ < 64) {
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int i = t * EPL + e;
                const bool up = (i & K) == 0;
                unsigned long long mine = k[e];
                unsigned long long other = __shfl_xor(mine, TJ, 64);
                const bool take_min = (lower == up);
                k[e] = take_min ? (mine < other ? mine : other) : (mine > other ? mine : other);
            }
        } else {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPL; e++) buf[t * EPL + e] = k[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int i = t * EPL + e;
                const bool up = (i & K) == 0;
                unsigned long long mine = k[e];
                unsigned long long other = buf[(t ^ TJ) * EPL + e];
                const bool take_min = (lower == up);
                k[e] = take_min ? (mine < other ? mine : other) : (mine > other ? mine : other);
            }
        }
    }
}
template <int NTH, int EPL, int K, int J>
struct SortJ {
    __device__ __forceinline__ static void run(unsigned long long (&k)[EPL], unsigned long long *buf) {
        sort_stage<NTH, EPL, K, J>(k, buf);
        if constexpr (J > 1) SortJ<NTH, EPL, K, J / 2>::run(k, buf);
    }
};
template <int NTH, int EPL, int K>
struct SortK {
    __device__ __forceinline__ static void run(unsigned long long (&k)[EPL], unsigned long long *buf) {
        SortJ<NTH, EPL, K, K / 2>::run(k, buf);
        if constexpr (K < NTH * EPL) SortK<NTH, EPL, K * 2>::run(k, buf);
    }
};
template <int NTH, int EPL>
__device__ __forceinline__ void sort_registers(unsigned long long (&k)[EPL], unsigned long long *buf) {
    SortK<NTH, EPL, 2>::run(k, buf);
}

// keys of one cluster -> registers (thread t owns slots t*EPL .. t*EPL+EPL-1 of an NTH*EPL array), border-direction sum,
// sort, write back.  Returns dot (valid on every thread).  `raw` = staged 8-byte points in sKeys[0..sz0).
template <int NTH, int EPL>
__device__ __forceinline__ long long keys_sort(unsigned long long *sKeys, long long *sScratch, int sz0, int xmin, int xmax, int ymin, int ymax,
                                               int normal_ok, int reversed_ok, bool do_sort = true) {
    using B = Block<NTH>;
    const int tid = threadIdx.x;
    unsigned long long kreg[EPL];
    long long dot = 0;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        int i = tid * EPL + e;
        unsigned long long key = ~0ull;
        if (i < sz0) {
            unsigned long long raw = sKeys[i];
            int px = (int)(raw & 0xFFFF), py = (int)((raw >> 16) & 0xFFFF);
            int pgx = (int)(signed char)((raw >> 32) & 0xFF), pgy = (int)(signed char)((raw >> 40) & 0xFF);
            const int dx = 4 * px - 2 * (xmin + xmax) - 1, dy = 4 * py - 2 * (ymin + ymax) + 1; // |.| < 2^15
            dot += (long long)(dx * pgx + dy * pgy);
            key = angle_key(px, py, xmin, xmax, ymin, ymax);
        }
        kreg[e] = key;
    }
    dot = B::reduce_add(dot, sScratch);
    if (do_sort && ((dot < 0) ? reversed_ok : normal_ok)) { // uniform: skip the sort when the border direction is rejected anyway
        sort_registers<NTH, EPL>(kreg, sKeys);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPL; e++) sKeys[tid * EPL + e] = kreg[e];
    }
    return dot;
}

// Bucketed rank sort for the LDS-resident classes: a counting pass over nb angle buckets (octant + leading bits of the
// in-octant fraction, so bucket order = key order), then every key ranks itself inside its bucket.  About one LDS atomic,
// two scatters and a handful of compares per key, against log^2(n)/2 compare-exchanges for the bitonic network.  A
// cluster whose angles pile up in one bucket (more than BUCKET_LIMIT keys) is left to the bitonic path (*done = false;
// the staged points are still intact then).  Same result either way: equal keys are adjacent, their order is irrelevant.
constexpr int BUCKET_LIMIT = 32;
template <int NTH, int EPLS>
__device__ __forceinline__ long long keys_bucket_sort(unsigned long long *sKeys, uint32_t *hist /* [nb + 1] */, long long *sScratch, int sz0,
                                                      int xmin, int xmax, int ymin, int ymax, int normal_ok, int reversed_ok, bool do_sort,
                                                      bool *done) {
    using B = Block<NTH>;
    const int tid = threadIdx.x;
    int n2 = NTH;
    while (n2 < sz0) n2 <<= 1;
    int nb = n2 > 2048 ? 2048 : n2; // about one key per bucket; n2 >= NTH
    const int lgq = (31 - __clz(nb)) - 3; // fraction bits per octant
    for (int i = tid; i <= nb; i += NTH) hist[i] = 0;
    unsigned long long kreg[EPLS];
    uint32_t meta[EPLS]; // bucket << 16 | arrival index inside the bucket
    long long dot = 0;
#pragma unroll
    for (int e = 0; e < EPLS; e++) {
        const int i = tid + e * NTH;
        kreg[e] = 0; meta[e] = 0;
        if (i < sz0) {
            unsigned long long raw = sKeys[i];
            int px = (int)(raw & 0xFFFF), py = (int)((raw >> 16) & 0xFFFF);
            int pgx = (int)(signed char)((raw >> 32) & 0xFF), pgy = (int)(signed char)((raw >> 40) & 0xFF);
            const int dx = 4 * px - 2 * (xmin + xmax) - 1, dy = 4 * py - 2 * (ymin + ymax) + 1; // |.| < 2^15
            dot += (long long)(dx * pgx + dy * pgy);
            kreg[e] = angle_key(px, py, xmin, xmax, ymin, ymax);
        }
    }
    dot = B::reduce_add(dot, sScratch);
    *done = true;
    if (!do_sort || !((dot < 0) ? reversed_ok : normal_ok)) return dot; // uniform: rejected by border direction anyway
    B::sync(); // histogram zeroed, every staged point read
#pragma unroll
    for (int e = 0; e < EPLS; e++) {
        const int i = tid + e * NTH;
        if (i < sz0) {
            const uint32_t oct = (uint32_t)(kreg[e] >> 57), frac = (uint32_t)(kreg[e] >> 26) & 0x7FFFFFFFu;
            uint32_t fb = frac >> (30 - lgq);
            fb = fb > (1u << lgq) - 1 ? (1u << lgq) - 1 : fb; // frac == 2^30 (end of an inverted octant) stays in its octant
            const uint32_t b = (oct << lgq) | fb;
            meta[e] = (b << 16) | atomicAdd(&hist[b], 1u);
        }
    }
    B::sync();
    {   // exclusive scan of the counts in place (hist[nb] = sz0) and the largest count
        const int per = nb / NTH;
        uint32_t sum = 0, mx = 0;
        for (int q = 0; q < per; q++) { uint32_t c = hist[tid * per + q]; sum += c; mx = mx > c ? mx : c; }
        long long total;
        const long long incl = B::scan_incl((long long)sum, sScratch, &total);
        const int worst = -B::reduce_min(-(int)mx, reinterpret_cast<int *>(sScratch));
        if (worst > BUCKET_LIMIT) { *done = false; B::sync(); return dot; }
        uint32_t run = (uint32_t)(incl - sum);
        for (int q = 0; q < per; q++) { uint32_t c = hist[tid * per + q]; hist[tid * per + q] = run; run += c; }
        if (tid == NTH - 1) hist[nb] = run;
    }
    B::sync();
#pragma unroll
    for (int e = 0; e < EPLS; e++) {
        const int i = tid + e * NTH;
        if (i < sz0) sKeys[hist[meta[e] >> 16] + (meta[e] & 0xFFFFu)] = kreg[e];
    }
    B::sync();
    uint32_t pos[EPLS];
#pragma unroll
    for (int e = 0; e < EPLS; e++) {
        const int i = tid + e * NTH;
        pos[e] = 0;
        if (i < sz0) {
            const uint32_t b = meta[e] >> 16, s0 = hist[b], s1 = hist[b + 1], mine = s0 + (meta[e] & 0xFFFFu);
            const unsigned long long key = kreg[e];
            uint32_t rank = 0;
            for (uint32_t j = s0; j < s1; j++) {
                const unsigned long long o = sKeys[j];
                rank += (o < key || (o == key && j < mine)) ? 1u : 0u;
            }
            pos[e] = s0 + rank;
        }
    }
    B::sync();
#pragma unroll
    for (int e = 0; e < EPLS; e++) {
        const int i = tid + e * NTH;
        if (i < sz0) sKeys[pos[e]] = kreg[e];
    }
    return dot;
}

// The same sort for the large class, whose keys do not fit in registers: the counting pass only counts, the scatter goes
// to the cluster's slice of an 8-byte-per-point global scratch (ck_stage_ws::d_lscratch), and the rank
// pass walks that slice and writes the sorted keys to LDS.  The staged points stay intact until the bucket sizes are known,
// so the bitonic fallback can still start from them.
constexpr int BUCKET_LIMIT_L = 96;
// NOLIMIT: the largest class has no other sort to fall back on; its rank pass takes whatever the buckets hold
template <int NTH, bool NOLIMIT>
__device__ __forceinline__ long long keys_bucket_sort_global(unsigned long long *sKeys, uint32_t *hist /* [2 * (1024 + 1)] */, long long *sScratch,
                                                             unsigned long long *gtmp, int sz0, int xmin, int xmax, int ymin, int ymax,
                                                             int normal_ok, int reversed_ok, bool do_sort, bool *done) {
    using B = Block<NTH>;
    const int tid = threadIdx.x;
    constexpr int nb = 1024, lgq = 7;
    uint32_t *cursor = hist + nb + 1;
    auto bucket_of = [&](unsigned long long key) -> uint32_t {
        const uint32_t oct = (uint32_t)(key >> 57), frac = (uint32_t)(key >> 26) & 0x7FFFFFFFu;
        uint32_t fb = frac >> (30 - lgq);
        fb = fb > (1u << lgq) - 1 ? (1u << lgq) - 1 : fb;
        return (oct << lgq) | fb;
    };
    for (int i = tid; i <= nb; i += NTH) hist[i] = 0;
    __syncthreads();
    long long dot = 0;
    for (int i = tid; i < sz0; i += NTH) {
        unsigned long long raw = sKeys[i];
        int px = (int)(raw & 0xFFFF), py = (int)((raw >> 16) & 0xFFFF);
        int pgx = (int)(signed char)((raw >> 32) & 0xFF), pgy = (int)(signed char)((raw >> 40) & 0xFF);
        const int dx = 4 * px - 2 * (xmin + xmax) - 1, dy = 4 * py - 2 * (ymin + ymax) + 1; // |.| < 2^15
        dot += (long long)(dx * pgx + dy * pgy);
        atomicAdd(&hist[bucket_of(angle_key(px, py, xmin, xmax, ymin, ymax))], 1u);
    }
    dot = B::reduce_add(dot, sScratch);
    *done = true;
    if (!do_sort || !((dot < 0) ? reversed_ok : normal_ok)) return dot;
    __syncthreads();
    {
        constexpr int per = nb / NTH;
        uint32_t sum = 0, mx = 0;
        for (int q = 0; q < per; q++) { uint32_t c = hist[tid * per + q]; sum += c; mx = mx > c ? mx : c; }
        long long total;
        const long long incl = B::scan_incl((long long)sum, sScratch, &total);
        const int worst = -B::reduce_min(-(int)mx, reinterpret_cast<int *>(sScratch));
        if (!NOLIMIT && worst > BUCKET_LIMIT_L) { *done = false; __syncthreads(); return dot; }
        uint32_t run = (uint32_t)(incl - sum);
        for (int q = 0; q < per; q++) { uint32_t c = hist[tid * per + q]; hist[tid * per + q] = run; cursor[tid * per + q] = run; run += c; }
        if (tid == NTH - 1) hist[nb] = run;
    }
    __syncthreads();
    for (int i = tid; i < sz0; i += NTH) {
        unsigned long long raw = sKeys[i];
        const unsigned long long key = angle_key((int)(raw & 0xFFFF), (int)((raw >> 16) & 0xFFFF), xmin, xmax, ymin, ymax);
        gtmp[atomicAdd(&cursor[bucket_of(key)], 1u)] = key;
    }
    __syncthreads(); // workgroup-scope release/acquire: the slice written above is read by other waves below
    for (int j = tid; j < sz0; j += NTH) {
        const unsigned long long key = gtmp[j];
        const uint32_t b = bucket_of(key), s0 = hist[b], s1 = hist[b + 1];
        uint32_t rank = 0;
        for (uint32_t k = s0; k < s1; k++) {
            const unsigned long long o = gtmp[k];
            rank += (o < key || (o == key && k < (uint32_t)j)) ? 1u : 0u;
        }
        sKeys[s0 + rank] = key;
    }
    return dot;
}

// NTH threads per cluster, up to CAP points, chunks of CH points; MLDS: the maxima list fits in LDS; GK: the per-point arrays
// (keys, then coordinates and weights) live in a per-workgroup slice of global memory instead of LDS (the largest class)
template <int NTH, int CAP, int CH, bool MLDS, int WPS, bool GK = false>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(WPS, 8))) void k_fit(FitArgs a) {
    using B = Block<NTH>;
    constexpr int SL = CH + 2 * HALO;
    // granularity of the cumulative-moment table: fine for the LDS-resident classes so that a selected maximum needs only
    // a few points beyond its table entry (LPM lanes x 4 points), one entry per chunk for the large class
    constexpr int G = (CAP <= 4096) ? 32 : 128;
    constexpr int LPM = (CAP > 4096) ? 32 : (NTH >= MAXSEL * 8) ? 8 : 4; // lanes per selected maximum, G / LPM points each
    static_assert(MAXSEL * LPM <= NTH && CH % G == 0, "one lane group per selected maximum");
    constexpr int NG = CAP / G;
    constexpr int MAXM = MLDS ? (CAP / 2 > 2 * MAXSEL * 6 ? CAP / 2 : 2 * MAXSEL * 6) : 1;
    // sKeys holds the 64-bit sort keys; after duplicate removal its first half is reused for the packed
    // coordinates (x<<13|y, u32) and the third quarter for the u16 weights.
    __shared__ unsigned long long sKeysL[GK ? 1 : CAP];
    // (the largest class's slice of global memory is laid out for the frame's own bound on a cluster's points, not for the template's)
    const int capr = GK ? a.ws.hcap : CAP;
    unsigned long long *const sKeys = GK ? a.ws.d_hscratch + (size_t)blockIdx.x * 2 * capr : sKeysL;
    // inclusive moment prefix sums over the current span of SL points: Mxx, Mxy, Myy as 64-bit and Mx, My, W as 32-bit
    // (a span's sums stay below 304 * 362 * 8192 < 2^32); the same bytes are the sort's histogram before and the pair-fit
    // table after the chunk loop
    constexpr int SPB = 36 * SL > (int)sizeof(PairFit) * MAXSEL * MAXSEL ? 36 * SL : (int)sizeof(PairFit) * MAXSEL * MAXSEL;
    __shared__ __attribute__((aligned(16))) unsigned char sPraw[SPB];
    __shared__ double sErr[SL];
    __shared__ long long sTot[NG + 1][6];    // cumulative moments of the first k*G sorted points
    __shared__ long long sScratch[6 * (NTH / 64) + 2];
    __shared__ double sMaxVal[MAXM];
    __shared__ int sSelIdx[MAXSEL];
    // moment prefix sums at the selected maxima: they are born after the maxima list has been consumed, so the
    // LDS-resident classes keep them in its bytes
    // ... and the large class keeps them in the error array, which is dead once the chunk loop has produced the maxima
    static_assert(MLDS || sizeof(double) * SL >= sizeof(long long) * 2 * MAXSEL * 6, "selected-maxima sums must fit in sErr");
    static_assert(!MLDS || sizeof(double) * MAXM >= sizeof(long long) * 2 * MAXSEL * 6, "selected-maxima sums must fit in sMaxVal");
    long long (*sSelI)[6] = reinterpret_cast<long long (*)[6]>(MLDS ? reinterpret_cast<long long *>(sMaxVal) : reinterpret_cast<long long *>(sErr));
    long long (*sSelE)[6] = sSelI + MAXSEL;
    __shared__ double sRed[NTH / 64 + 1];
    __shared__ int sRedI[NTH / 64 + 1];
    __shared__ uint32_t sWork;
    __shared__ int sNmax, sFlag;
    __shared__ double sLines[4][4];
    __shared__ double sQuad[4][2];
    // per edge, per sample of the current round: refined point (x,y); x = NaN: no point.  Lives in the pair-fit table's bytes
    double (*sRefine)[16][2] = reinterpret_cast<double (*)[16][2]>(sPraw);
    static_assert(SPB >= ((CAP < 2048 ? CAP : 2048) + 1) * 4 || CAP <= 512, "sort histogram must fit in sPraw");
    long long (*sP64)[SL] = reinterpret_cast<long long (*)[SL]>(sPraw);            // [3][SL]: Mxx, Mxy, Myy
    uint32_t (*sP32)[SL] = reinterpret_cast<uint32_t (*)[SL]>(sPraw + 24 * SL);     // [3][SL]: Mx, My, W
    const int tid = threadIdx.x;
    const ck_stage_ws &ws = a.ws;
    uint32_t *sXY = reinterpret_cast<uint32_t *>(sKeys);
    uint16_t *sW = reinterpret_cast<uint16_t *>(sKeys) + 2 * capr; // bytes [4*CAP, 6*CAP)
    uint16_t *sMaxIdx = reinterpret_cast<uint16_t *>(sKeys) + 3 * capr; // bytes [6*CAP, 8*CAP): free once the keys are packed

    // chunked dequeue from the class work list: DQ consecutive clusters per atomic.  One atomic per cluster is too much for
    // the small class (520 k clusters per batch through one counter cost more than any phase of the fit), static striding
    // leaves the slowest workgroup ~25 % behind the average (the sum of ~100 clusters' work still varies that much); a
    // chunk keeps the counter traffic at a few thousand adds per launch and the tail at one chunk.
    const uint32_t n_work = min(*a.list_count, (uint32_t)a.list_cap);
    // A short list (one frame per call) is handed out one cluster at a time so that it still spreads over the whole grid.
    constexpr uint32_t DQMAX = CAP <= 512 ? 16u : (CAP <= 2048 ? 4u : 1u);
    const uint32_t per_wg4 = n_work / (gridDim.x * 4u);
    const uint32_t DQ = per_wg4 < 1u ? 1u : (per_wg4 > DQMAX ? DQMAX : per_wg4);
    // The first chunk of every workgroup is its own (no atomic: a launch with little or no work — one frame per call, an empty
    // class — must not queue thousands of adds on one address); the counter hands out what lies beyond those.
    const uint32_t static_total = gridDim.x * DQ;
    uint32_t chunk_base = blockIdx.x * DQ, chunk_left = DQ, chunk_len = DQ, dq_next = DQ;
    for (;;) {
        __syncthreads();
        if (chunk_left == 0) {
            if (static_total >= n_work) break;
            if (tid == 0) sWork = static_total + atomicAdd(a.head, dq_next);
            __syncthreads();
            chunk_base = sWork;
            if (chunk_base >= n_work) break;
            chunk_left = chunk_len = dq_next;
            if (a.guided) { // the chunks shrink towards the end of the list: the launch's tail is one cluster, not one chunk
                const uint32_t g = (n_work - chunk_base) / (gridDim.x * 2u);
                dq_next = g < 1u ? 1u : (g > DQ ? DQ : g);
            }
        }
        const uint32_t work = chunk_base + (chunk_len - chunk_left);
        chunk_left--;
        if (work >= n_work) break;
        const uint32_t item = a.list[work];
        const int frame = (int)(item >> 20), ci = (int)(item & 0xFFFFFu);
        const ck_cluster_t cl = ws.d_clusters[(size_t)frame * ws.cluster_cap + ci];
        const ck_packed_point *pts = ws.d_points + (size_t)frame * ws.ext_cap + cl.start;
        // 8 bytes per point for the large class (sort scratch, then the maxima list): one fixed region per workgroup, so
        // the same few hundred KB are reused cluster after cluster and stay in L2
        unsigned long long *scratch8 = GK ? ws.d_hscratch + (size_t)blockIdx.x * 2 * capr + capr : ws.d_lscratch + (size_t)blockIdx.x * CK_LSCRATCH_PER_WG;
        const int sz0 = (int)cl.count;
        const uint16_t *wq = a.wimg + (size_t)frame * a.qw * a.qh;
        const uint8_t *im = a.im + (size_t)frame * a.pitch;
        if (sz0 > CAP) continue; // cannot happen: the class lists are built from the counts
        PROF_DECL;
        PROF(15);

        if (a.stop_after == 0) continue;
        // ---- 1. bounding box + border direction ----------------------------------------------------------
        int xmin = 1 << 30, xmax = -(1 << 30), ymin = 1 << 30, ymax = -(1 << 30);
        {   // one coalesced pass; all of a lane's loads are issued before the first use (one memory round trip, not EPL)
            constexpr int EPL = CAP / NTH;
            // packed point -> the staged form the sort reads: x | y << 16 | (u8)gx << 32 | (u8)gy << 40
            auto stage = [](ck_packed_point v) -> unsigned long long {
                const int k = (int)(v >> 1) & 3, sgn = (v & 1u) ? 1 : -1;
                const int dx = k == 2 ? -1 : (k == 1 ? 0 : 1), dy = k == 0 ? 0 : 1;
                return (unsigned long long)((v >> 16) & 0x1FFFu) | ((unsigned long long)((v >> 3) & 0x1FFFu) << 16) |
                       ((unsigned long long)(uint8_t)(int8_t)(dx * sgn) << 32) | ((unsigned long long)(uint8_t)(int8_t)(dy * sgn) << 40);
            };
            if constexpr (EPL <= 16) {
                ck_packed_point rawp[EPL];
#pragma unroll
                for (int e = 0; e < EPL; e++) { int i = tid + e * NTH; rawp[e] = (i < sz0) ? pts[i] : 0u; }
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    int i = tid + e * NTH;
                    if (i < sz0) {
                        const unsigned long long raw = stage(rawp[e]);
                        sKeys[i] = raw;
                        int px = (int)(raw & 0xFFFF), py = (int)((raw >> 16) & 0xFFFF);
                        xmin = min(xmin, px); xmax = max(xmax, px);
                        ymin = min(ymin, py); ymax = max(ymax, py);
                    }
                }
            } else {
                for (int i = tid; i < sz0; i += NTH) {
                    const unsigned long long raw = stage(pts[i]);
                    sKeys[i] = raw;
                    int px = (int)(raw & 0xFFFF), py = (int)((raw >> 16) & 0xFFFF);
                    xmin = min(xmin, px); xmax = max(xmax, px);
                    ymin = min(ymin, py); ymax = max(ymax, py);
                }
            }
        }
        int *iscr = reinterpret_cast<int *>(sScratch);
        {
            int bb[4] = {xmin, -xmax, ymin, -ymax};
            B::reduce_min4(bb, iscr);
            xmin = bb[0]; xmax = -bb[1]; ymin = bb[2]; ymax = -bb[3];
        }
        if ((xmax - xmin) * (ymax - ymin) < a.min_tag_width) continue;
        if (a.stop_after == 10) continue;
        long long dot;
        {   // pick the keys-per-thread count that matches the cluster (padded to a power of two, at least one per thread)
            constexpr int EPLS = CAP / NTH;
            int n2 = NTH;
            while (n2 < sz0) n2 <<= 1;
            const int epl = n2 / NTH;
            __syncthreads(); // raw points staged by other threads
            bool sorted = false;
            if constexpr (MLDS) // keys fit in registers (CAP / NTH <= 16)
                dot = keys_bucket_sort<NTH, EPLS>(sKeys, reinterpret_cast<uint32_t *>(sPraw), sScratch, sz0, xmin, xmax, ymin, ymax,
                                                  a.normal_ok, a.reversed_ok, a.stop_after != 11, &sorted);
            else
                dot = keys_bucket_sort_global<NTH, GK>(sKeys, reinterpret_cast<uint32_t *>(sPraw), sScratch,
                                                   scratch8, sz0, xmin, xmax, ymin, ymax,
                                                   a.normal_ok, a.reversed_ok, a.stop_after != 11, &sorted);
            if (sorted || GK) {}
            else if (EPLS >= 32 && epl == 32) dot = keys_sort<NTH, (EPLS >= 32 && !GK ? 32 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
            else if (EPLS >= 16 && epl == 16) dot = keys_sort<NTH, (EPLS >= 16 && !GK ? 16 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
            else if (EPLS >= 8 && epl == 8) dot = keys_sort<NTH, (EPLS >= 8 && !GK ? 8 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
            else if (EPLS >= 4 && epl == 4) dot = keys_sort<NTH, (EPLS >= 4 && !GK ? 4 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
            else if (EPLS >= 2 && epl == 2) dot = keys_sort<NTH, (EPLS >= 2 && !GK ? 2 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
            else dot = keys_sort<NTH, 1>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
        }
        const int reversed = dot < 0;
        PROF(0);
        if (reversed && !a.reversed_ok) continue;
        if (!reversed && !a.normal_ok) continue;
        if (a.stop_after == 1 || a.stop_after == 11) continue;

        // ---- 2. duplicate removal, packing to (x,y) ----------------------------------------------------------------
        __syncthreads();
        PROF(1);
        // compaction writes u32 (x<<13|y) at index <= i into the first half of the same buffer: a round only
        // overwrites bytes below 4*(base+NTH) while unread keys start at byte 8*(base+NTH)
        int sz = 0;
        for (int base = 0; base < sz0; base += NTH) {
            int i = base + tid;
            unsigned long long key = 0;
            int keep = 0;
            if (i < sz0) { key = sKeys[i]; keep = (i == 0) || (sKeys[i - 1] != key); }
            int tot = 0;
            const int incl = B::scan_flag(keep, iscr, &tot);
            __syncthreads();
            if (keep) sXY[sz + (int)incl - 1] = (uint32_t)(key & 0x3FFFFFFu);
            sz += (int)tot;
            __syncthreads();
        }
        PROF(2);
        if (sz < 24) continue;
        const int ksz = sz / 12 < 20 ? sz / 12 : 20;
        if (ksz < 2) continue;

        if (a.stop_after == 2) continue;
        // ---- 3. gradient weights: one image gather per point, all in flight together ---------------------------------
        {
            constexpr int EPL = CAP / NTH;
            if constexpr (EPL <= 16) { // issue every gather of the lane, then compute: one round trip
                uint16_t g[EPL];
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    int i = tid + e * NTH;
                    g[e] = 1;
                    if (i < sz) {
                        uint32_t xy = sXY[i];
                        int x = (int)(xy >> 13), y = (int)(xy & 0x1FFF);
                        int ix = (x + 1) >> 1, iy = (y + 1) >> 1;
                        if (ix >= 0 && ix < a.qw && iy >= 0 && iy < a.qh) g[e] = wq[(size_t)iy * a.qw + ix];
                    }
                }
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    int i = tid + e * NTH;
                    if (i < sz) sW[i] = g[e];
                }
            } else {
                for (int i = tid; i < sz; i += NTH) {
                    uint32_t xy = sXY[i];
                    int x = (int)(xy >> 13), y = (int)(xy & 0x1FFF);
                    int ix = (x + 1) >> 1, iy = (y + 1) >> 1;
                    sW[i] = (ix >= 0 && ix < a.qw && iy >= 0 && iy < a.qh) ? wq[(size_t)iy * a.qw + ix] : (uint16_t)1;
                }
            }
        }
        // the cluster's slice of the 8-byte scratch is free again (the sort is done); the large class keeps its maxima there
        double *gval = reinterpret_cast<double *>(scratch8);
        uint32_t *gidx = reinterpret_cast<uint32_t *>(gval + (sz0 + 1) / 2);
        if (tid == 0) { sNmax = 0; sFlag = 0; }
        if (tid < 6) sTot[0][tid] = 0;
        __syncthreads();

        PROF(3);
        if (a.stop_after == 3) continue;
        // ---- 4. chunk loop: moment prefix sums, windowed error, smoothing, maxima ----------------------------------------
        const int nch = (sz + CH - 1) / CH;
        for (int c = 0; c < nch; c++) {
            const int cbase = c * CH;
            const int chn = min(CH, sz - cbase);
            const int sl = chn + 2 * HALO;
            constexpr int EPT = (SL + NTH - 1) / NTH;
            // The errors of the points cbase-4 .. cbase+3 were formed by the previous chunk (its last eight: window sums are exact
            // integers, so the value of a point does not depend on the chunk that forms it).  Carried over, this chunk forms
            // exactly chn errors — one round of the lanes for a full chunk instead of two with the second nearly empty.
            double carry = 0.0;
            if (c > 0 && tid < 8) carry = sErr[HALO + CH - 4 + tid];
            M6 loc[EPT];
            M6 run = m6_zero();
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                int j = tid * EPT + e;
                if (j < sl) {
                    int gi = wrap_index(cbase - HALO + j, sz);
                    run = m6_add(run, moments_of(sXY[gi], sW[gi]));
                }
                loc[e] = run;
            }
            unsigned long long xa[3] = {(unsigned long long)run.Mxx, (unsigned long long)run.Mxy, (unsigned long long)run.Myy};
            uint32_t xb[3] = {(uint32_t)run.Mx, (uint32_t)run.My, (uint32_t)run.W};
            B::scan_incl_3x64_3x32(xa, xb, sScratch);
            xa[0] -= (unsigned long long)run.Mxx; xa[1] -= (unsigned long long)run.Mxy; xa[2] -= (unsigned long long)run.Myy;
            xb[0] -= (uint32_t)run.Mx; xb[1] -= (uint32_t)run.My; xb[2] -= (uint32_t)run.W;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                int j = tid * EPT + e;
                if (j < sl) {
                    sP64[0][j] = loc[e].Mxx + (long long)xa[0]; sP64[1][j] = loc[e].Mxy + (long long)xa[1]; sP64[2][j] = loc[e].Myy + (long long)xa[2];
                    sP32[0][j] = (uint32_t)loc[e].Mx + xb[0]; sP32[1][j] = (uint32_t)loc[e].My + xb[1]; sP32[2][j] = (uint32_t)loc[e].W + xb[2];
                }
            }
            __syncthreads();
            {   // cumulative moments at every G-th point of this chunk (and at its end, which closes the cluster's last block)
                constexpr int BPC = CH / G;
                if (tid < 6 * BPC) {
                    int q = tid % 6, bb = tid / 6;
                    int last = min((bb + 1) * G, chn);      // points of this chunk covered through block bb
                    if (bb * G < chn) {
                        const int jh = HALO + last - 1, jl = HALO - 1;
                        long long d;
                        if (q == 0) d = (long long)(sP32[0][jh] - sP32[0][jl]);
                        else if (q == 1) d = (long long)(sP32[1][jh] - sP32[1][jl]);
                        else if (q == 5) d = (long long)(sP32[2][jh] - sP32[2][jl]);
                        else d = sP64[q - 2][jh] - sP64[q - 2][jl];
                        sTot[c * BPC + bb + 1][q] = sTot[c * BPC][q] + d;
                    }
                }
            }
            if (c > 0 && tid < 8) sErr[HALO - 4 + tid] = carry;
            for (int j = (c > 0 ? HALO + 4 : HALO - 4) + tid; j < HALO + chn + 4; j += NTH) {
                int hi = j + ksz, lo = j - ksz - 1;
                uint32_t mx = sP32[0][hi], my = sP32[1][hi], mw = sP32[2][hi];
                M6 m;
                m.Mxx = sP64[0][hi]; m.Mxy = sP64[1][hi]; m.Myy = sP64[2][hi];
                if (lo >= 0) {
                    mx -= sP32[0][lo]; my -= sP32[1][lo]; mw -= sP32[2][lo];
                    m.Mxx -= sP64[0][lo]; m.Mxy -= sP64[1][lo]; m.Myy -= sP64[2][lo];
                }
                double e;
                {   // fit_line_m's error for this window, with the three first-order sums converted straight from 32 bits
                    const double inv = 1.0 / (double)mw;
                    const double Ex = (0.5 * (double)mx) * inv, Ey = (0.5 * (double)my) * inv;
                    const double Cxx = (0.25 * (double)m.Mxx) * inv - Ex * Ex;
                    const double Cxy = (0.25 * (double)m.Mxy) * inv - Ex * Ey;
                    const double Cyy = (0.25 * (double)m.Myy) * inv - Ey * Ey;
                    const double d = Cxx - Cyy, q4 = 4.0 * Cxy;
                    const double disc = sqrt(d * d + q4 * Cxy);
                    e = (double)(2 * ksz + 1) * (0.5 * ((Cxx + Cyy) - disc));
                }
                sErr[j] = e;
            }
            __syncthreads();
            // smoothing and maxima in one pass: each point smooths itself and its two neighbours from nine errors (same
            // operation order as the oracle's one-value-at-a-time loop, so the same bits)
            for (int j = HALO + tid; j < HALO + chn; j += NTH) {
                double e9[9];
#pragma unroll
                for (int q = 0; q < 9; q++) e9[q] = sErr[j + q - 4];
                double sp = 0.0, s = 0.0, sn = 0.0;
#pragma unroll
                for (int q = 0; q < 7; q++) { sp += e9[q] * k_smooth[q]; s += e9[q + 1] * k_smooth[q]; sn += e9[q + 2] * k_smooth[q]; }
                if (s > sn && s > sp) {
                    int pos = atomicAdd(&sNmax, 1); // pos < sz/2
                    if (MLDS) { sMaxVal[pos] = s; sMaxIdx[pos] = (uint16_t)(cbase + j - HALO); }
                    else { gval[pos] = s; gidx[pos] = (uint32_t)(cbase + j - HALO); }
                }
            }
            // no barrier here: the next chunk writes sErr only after the barrier that follows its prefix sums
        }
        __syncthreads();
        PROF(4);
        const int nmax_all = sNmax;
        if (nmax_all < 4) continue;
        auto mval = [&](int i) -> double { return MLDS ? sMaxVal[i] : gval[i]; };
        auto midx = [&](int i) -> int { return MLDS ? (int)sMaxIdx[i] : (int)gidx[i]; };

        if (a.stop_after == 4) continue;
        // ---- 5a. threshold = (max_nmaxima+1)-th largest smoothed error ---------------------------------------------------
        double thr = 0.0;
        bool use_thr = false;
        if (nmax_all > a.max_nmaxima) {
            use_thr = true;
            if (MLDS && nmax_all <= NTH) {
                // one value per thread: count the values above / not below it; the (max_nmaxima+1)-th largest is the one
                // with gt <= max_nmaxima < ge (every thread holding that value writes the same bits)
                const double v = tid < nmax_all ? sMaxVal[tid] : 0.0;
                int gt = 0, ge = 0;
                for (int j = 0; j < nmax_all; j++) {
                    const double u = sMaxVal[j];
                    gt += (u > v) ? 1 : 0;
                    ge += (u >= v) ? 1 : 0;
                }
                if (tid < nmax_all && gt <= a.max_nmaxima && a.max_nmaxima < ge) sRed[0] = v;
                __syncthreads();
                thr = sRed[0];
            } else {
            double cur = HUGE_VAL;
            int remaining = a.max_nmaxima + 1;
            for (int round = 0; round <= a.max_nmaxima; round++) {
                // largest value below `cur` and how many times it occurs, in one pass and one reduction
                double m = -HUGE_VAL;
                int cnt = 0;
                for (int i = tid; i < nmax_all; i += NTH) {
                    double v = mval(i);
                    if (v < cur) { if (v > m) { m = v; cnt = 1; } else if (v == m) cnt++; }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    double om = __shfl_xor(m, d, 64);
                    int oc = __shfl_xor(cnt, d, 64);
                    if (om > m) { m = om; cnt = oc; } else if (om == m) cnt += oc;
                }
                if (NTH > 64) {
                    if ((tid & 63) == 0) { sRed[tid >> 6] = m; sRedI[tid >> 6] = cnt; }
                    __syncthreads();
                    m = sRed[0]; cnt = sRedI[0];
                    for (int k = 1; k < NTH / 64; k++) {
                        if (sRed[k] > m) { m = sRed[k]; cnt = sRedI[k]; } else if (sRed[k] == m) cnt += sRedI[k];
                    }
                    __syncthreads();
                }
                if (cnt >= remaining) { thr = m; break; }
                remaining -= cnt;
                cur = m;
            }
            }
        }
        // survivors (at most max_nmaxima of them), then put them in increasing index order
        __syncthreads();
        if (tid == 0) sNmax = 0;
        __syncthreads();
        for (int i = tid; i < nmax_all; i += NTH) {
            if (use_thr && mval(i) <= thr) continue;
            int pos = atomicAdd(&sNmax, 1);
            if (pos < MAXSEL) sSelIdx[pos] = midx(i);
        }
        __syncthreads();
        const int nsel = sNmax;
        if (nsel < 4 || nsel > MAXSEL) continue;
        {   // rank sort (the indices are distinct): no serial chain of dependent LDS round trips
            int mine = 0, rank = 0;
            if (tid < nsel) {
                mine = sSelIdx[tid];
                for (int j = 0; j < nsel; j++) rank += (sSelIdx[j] < mine) ? 1 : 0;
            }
            __syncthreads();
            if (tid < nsel) sSelIdx[rank] = mine;
        }
        __syncthreads();

        PROF(6);
        if (a.stop_after == 5) continue;
        // ---- 5b. moment prefix sums at the selected maxima (chunk totals + partial sums inside the chunk) ------------------
        {
            // LPM lanes per selected maximum, G / LPM points each: the at most G points beyond the table entry
            if (tid < MAXSEL * LPM) {
                const int s = tid / LPM, part = tid % LPM;
                M6 m = m6_zero();
                int gi = 0, blk = 0;
                if (s < nsel) {
                    gi = sSelIdx[s];
                    blk = gi / G;
                    constexpr int PPL = G / LPM;
                    const int i0 = blk * G + part * PPL;
#pragma unroll
                    for (int e = 0; e < PPL; e++) {
                        const int i = i0 + e;
                        if (i <= gi) m = m6_add(m, moments_of(sXY[i], sW[i]));
                    }
                }
                long long pv[6] = {m.Mx, m.My, m.Mxx, m.Mxy, m.Myy, m.W};
#pragma unroll
                for (int q = 0; q < 6; q++)
#pragma unroll
                    for (int d = LPM / 2; d >= 1; d >>= 1) pv[q] += __shfl_xor(pv[q], d, 64);
                if (s < nsel && part == 0) {
                    M6 self = moments_of(sXY[gi], sW[gi]);
                    long long sv[6] = {self.Mx, self.My, self.Mxx, self.Mxy, self.Myy, self.W};
#pragma unroll
                    for (int q = 0; q < 6; q++) { long long v = sTot[blk][q] + pv[q]; sSelI[s][q] = v; sSelE[s][q] = v - sv[q]; }
                }
            }
        }
        __syncthreads();
        const int ntot = (sz + G - 1) / G;
        const M6 total = {sTot[ntot][0], sTot[ntot][1], sTot[ntot][2], sTot[ntot][3], sTot[ntot][4], sTot[ntot][5]};
        auto rangeM = [&](int sa, int sb, int *N) { // points from maximum sa to maximum sb inclusive, going forward
            M6 I = {sSelI[sb][0], sSelI[sb][1], sSelI[sb][2], sSelI[sb][3], sSelI[sb][4], sSelI[sb][5]};
            M6 E = {sSelE[sa][0], sSelE[sa][1], sSelE[sa][2], sSelE[sa][3], sSelE[sa][4], sSelE[sa][5]};
            int i0 = sSelIdx[sa], i1 = sSelIdx[sb];
            if (i0 < i1) { *N = i1 - i0 + 1; return m6_sub(I, E); }
            *N = sz - i0 + i1 + 1;
            return m6_add(m6_sub(total, E), I);
        };

        PROF(7);
        if (a.stop_after == 6) continue;
        // ---- 5c. one line fit per ordered pair of maxima; the 4-subset search is then table lookups ------------------------
        PairFit *sF = reinterpret_cast<PairFit *>(sPraw);
        {   // lane k takes the pair a < b from the table and fits both directions in one go (a -> b, and b -> a around the
            // end); the two f64 dependency chains overlap
            const int npf = nsel * (nsel - 1) / 2;
            if constexpr (NTH > 64) { // enough lanes for one fit each
                const int npairs = nsel * nsel;
                for (int pr = tid; pr < npairs; pr += NTH) {
                    const int sa = pr / nsel, sb = pr - sa * nsel;
                    if (sa == sb) continue;
                    int N;
                    double lp[4], e, ms;
                    const M6 m = rangeM(sa, sb, &N);
                    fit_line_m(m, N, lp, &e, &ms);
                    PairFit f;
                    f.err = e; f.mse = ms; f.nx = lp[2]; f.ny = lp[3];
                    sF[sa * MAXSEL + sb] = f;
                }
            } else
            for (int k = tid; k < npf; k += NTH) {
                const int pk = g_pair_table.v[k], sa = pk >> 4, sb = pk & 15;
                int Nf, Nw;
                const M6 mf = rangeM(sa, sb, &Nf), mw = rangeM(sb, sa, &Nw);
                double lp[4], lq[4], ef, msf, ew, msw;
                fit_line_m(mf, Nf, lp, &ef, &msf);
                fit_line_m(mw, Nw, lq, &ew, &msw);
                PairFit f;
                f.err = ef; f.mse = msf; f.nx = lp[2]; f.ny = lp[3];
                sF[sa * MAXSEL + sb] = f;
                f.err = ew; f.mse = msw; f.nx = lq[2]; f.ny = lq[3];
                sF[sb * MAXSEL + sa] = f;
            }
        }
        __syncthreads();
        double best = HUGE_VAL;
        int bestc = 1 << 30;
        {
            const int ncomb = nsel * (nsel - 1) * (nsel - 2) * (nsel - 3) / 24;
            for (int cb = tid; cb < ncomb; cb += NTH) { // every lane walks its own subsets; ties resolved by the packed index
                const int pk = g_combo_table.v[cb];
                const int m0 = pk >> 12, m1 = (pk >> 8) & 15, m2 = (pk >> 4) & 15, m3 = pk & 15;
                {
                    {
                        {
                            const PairFit f01 = sF[m0 * MAXSEL + m1];
                            if (f01.mse > a.max_mse) continue;
                            const PairFit f12 = sF[m1 * MAXSEL + m2];
                            if (f12.mse > a.max_mse) continue;
                            double dp = f01.nx * f12.nx + f01.ny * f12.ny;
                            if (fabs(dp) > a.cos_critical) continue;
                            const PairFit f23 = sF[m2 * MAXSEL + m3];
                            if (f23.mse > a.max_mse) continue;
                            const PairFit f30 = sF[m3 * MAXSEL + m0];
                            if (f30.mse > a.max_mse) continue;
                            double e = f01.err + f12.err + f23.err + f30.err;
                            if (e < best || (e == best && pk < bestc)) { best = e; bestc = pk; }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            double ob = __shfl_xor(best, d, 64);
            int oc = __shfl_xor(bestc, d, 64);
            if (ob < best || (ob == best && oc < bestc)) { best = ob; bestc = oc; }
        }
        if (NTH > 64) {
            if ((tid & 63) == 0) { sRed[tid >> 6] = best; sRedI[tid >> 6] = bestc; }
            __syncthreads();
            best = sRed[0]; bestc = sRedI[0];
            for (int k = 1; k < NTH / 64; k++)
                if (sRed[k] < best || (sRed[k] == best && sRedI[k] < bestc)) { best = sRed[k]; bestc = sRedI[k]; }
            __syncthreads();
        }
        PROF(8);
        if (best == HUGE_VAL) continue;
        if (best / (double)sz >= a.max_mse) continue;

        if (a.stop_after == 7) continue;
        // ---- 5d. lines, corners, geometric checks: lane i of the first wave owns side i / corner i, lanes 0 and 1 the two
        // triangles of the area; every value is formed by the same operations as in the oracle's sequential code, and the
        // verdict is the conjunction of all checks, so their order does not matter ---------------------------------------------
        if (tid < 64) {
            const int li = tid & 3;
            const int sel[4] = {(bestc >> 12) & 15, (bestc >> 8) & 15, (bestc >> 4) & 15, bestc & 15};
            int ok = 1;
            double line[4];
            {   // the side's line: normal and mse are in the pair table already (the same fit of the same sums), only the
                // point on the line (the weighted mean) is formed here
                int N;
                const int s0 = sel[li], s1 = sel[(li + 1) & 3];
                const M6 m = rangeM(s0, s1, &N);
                const PairFit pf = sF[s0 * MAXSEL + s1];
                const double inv = 1.0 / (double)m.W;
                line[0] = (0.5 * (double)m.Mx) * inv; line[1] = (0.5 * (double)m.My) * inv;
                line[2] = pf.nx; line[3] = pf.ny;
                if (pf.mse > a.max_mse) ok = 0;
            }
            double ln[4]; // the next side's line
#pragma unroll
            for (int q = 0; q < 4; q++) ln[q] = __shfl(line[q], (tid & ~3) | ((li + 1) & 3), 64);
            double Px, Py;
            {
                double A00 = line[3], A01 = -ln[3], A10 = -line[2], A11 = ln[2];
                double B0 = -line[0] + ln[0], B1 = -line[1] + ln[1];
                double det = A00 * A11 - A10 * A01;
                if (fabs(det) < 0.001) ok = 0;
                double W00 = A11 / det, W01 = -A01 / det;
                double L0 = W00 * B0 + W01 * B1;
                Px = line[0] + L0 * A00;
                Py = line[1] + L0 * A10;
            }
            // corners of the other lanes come by shuffle with a computed source lane (an indexed local array would live in
            // scratch memory)
            const int g0 = tid & ~3;
            auto corner_x = [&](int q) { return __shfl(Px, g0 | (q & 3), 64); };
            auto corner_y = [&](int q) { return __shfl(Py, g0 | (q & 3), 64); };
            {   // Heron: triangle (0,1,2) on even lanes, (2,3,0) on odd ones; area = first + second
                const int t = li & 1;
                const int va = t ? 2 : 0, vb = t ? 3 : 1, vc = t ? 0 : 2;
                const double ax = corner_x(va), ay = corner_y(va), bx = corner_x(vb), by = corner_y(vb), cx = corner_x(vc), cy = corner_y(vc);
                double len[3];
                { double ddx = bx - ax, ddy = by - ay; len[0] = sqrt(ddx * ddx + ddy * ddy); }
                { double ddx = cx - bx, ddy = cy - by; len[1] = sqrt(ddx * ddx + ddy * ddy); }
                { double ddx = ax - cx, ddy = ay - cy; len[2] = sqrt(ddx * ddx + ddy * ddy); }
                double pp = (len[0] + len[1] + len[2]) / 2.0;
                double term = sqrt(pp * (pp - len[0]) * (pp - len[1]) * (pp - len[2]));
                double t0 = __shfl(term, g0, 64), t1 = __shfl(term, g0 | 1, 64);
                double area = 0.0;
                area += t0; area += t1;
                double tw = (double)a.min_tag_width;
                if (area < 0.95 * tw * tw) ok = 0;
            }
            {
                const double x1 = corner_x(li + 1), y1 = corner_y(li + 1), x2 = corner_x(li + 2), y2 = corner_y(li + 2);
                double dx1 = x1 - Px, dy1 = y1 - Py;
                double dx2 = x2 - x1, dy2 = y2 - y1;
                double cs = (dx1 * dx2 + dy1 * dy2) / sqrt((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2));
                if (cs > a.cos_critical || cs < -a.cos_critical) ok = 0;
                if (dx1 * dy2 < dy1 * dx2) ok = 0;
            }
            const int all_ok = (__ballot(ok != 0) & 0xFull) == 0xFull;
            if (tid < 4 && all_ok) {
                double qx = Px, qy = Py;
                if (a.decimate > 1) { qx = (qx - 0.5) * (double)a.decimate + 0.5; qy = (qy - 0.5) * (double)a.decimate + 0.5; }
                sQuad[li][0] = qx; sQuad[li][1] = qy;
            }
            if (tid == 0) sFlag = all_ok;
        }
        __syncthreads();
        PROF(9);
        if (!sFlag) continue;
        if (a.refine) {
            // Edge refinement (oracle refine_edges).  The samples of an edge are independent: lane (edge, k) evaluates sample
            // 16*round + k (its 2*range+1 gradient probes are independent loads), then ONE lane per edge accumulates the
            // refined points in sample order, so every sum is formed exactly as in the sequential code.
            const int edge = (tid >> 4) & 3, k = tid & 15;
            const bool worker = tid < 64;
            const int ea = edge, eb = (edge + 1) & 3;
            double nx = 0, ny = 0, mag = 1;
            int nsamples = 0;
            if (worker) {
                nx = sQuad[eb][1] - sQuad[ea][1];
                ny = -sQuad[eb][0] + sQuad[ea][0];
                mag = sqrt(nx * nx + ny * ny);
                nx = nx / mag; ny = ny / mag;
                if (reversed) { nx = -nx; ny = -ny; }
                nsamples = (int)(mag / 8.0);
                if (nsamples < 16) nsamples = 16;
            }
            int max_samples = nsamples; // the same for the 16 lanes of an edge; rounds are driven by the largest edge
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) max_samples = max(max_samples, __shfl_xor(max_samples, d, 64));
            if (NTH > 64) {
                if (tid == 0) sRedI[0] = max_samples;
                __syncthreads();
                max_samples = sRedI[0];
                __syncthreads();
            }
            double Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, N = 0; // meaningful on lane k == 0 of every edge
            for (int base = 0; base < max_samples; base += 16) {
                if (worker) {
                    const int sidx = base + k;
                    double bx = __builtin_nan(""), by = 0;
                    if (sidx < nsamples) {
                        double alpha = (1.0 + (double)sidx) / ((double)nsamples + 1.0);
                        double x0 = alpha * sQuad[ea][0] + (1.0 - alpha) * sQuad[eb][0];
                        double y0 = alpha * sQuad[ea][1] + (1.0 - alpha) * sQuad[eb][1];
                        double Mn = 0, Mcount = 0;
                        const int range = a.decimate + 1;
                        for (int n = -range; n <= range; n++) {
                            double grange = 1.0;
                            int x1 = (int)(x0 + ((double)n + grange) * nx), y1 = (int)(y0 + ((double)n + grange) * ny);
                            if (x1 < 0 || x1 >= a.w || y1 < 0 || y1 >= a.h) continue;
                            int x2 = (int)(x0 + ((double)n - grange) * nx), y2 = (int)(y0 + ((double)n - grange) * ny);
                            if (x2 < 0 || x2 >= a.w || y2 < 0 || y2 >= a.h) continue;
                            int g1 = im[(size_t)y1 * a.stride + x1], g2 = im[(size_t)y2 * a.stride + x2];
                            if (g1 < g2) continue;
                            double weight = (double)((g2 - g1) * (g2 - g1));
                            Mn += weight * (double)n;
                            Mcount += weight;
                        }
                        if (Mcount != 0) {
                            double n0 = Mn / Mcount;
                            bx = x0 + n0 * nx; by = y0 + n0 * ny;
                        }
                    }
                    sRefine[edge][k][0] = bx; sRefine[edge][k][1] = by;
                }
                __syncthreads();
                if (worker && k == 0)
                    for (int q = 0; q < 16 && base + q < nsamples; q++) {
                        double bx = sRefine[edge][q][0], by = sRefine[edge][q][1];
                        if (bx != bx) continue; // no usable gradient at this sample
                        Mx += bx; My += by; Mxx += bx * bx; Mxy += bx * by; Myy += by * by; N += 1.0;
                    }
                __syncthreads();
            }
            if (worker && k == 0) {
                double line[4];
                if (N < 2.0) {
                    line[0] = 0.5 * (sQuad[ea][0] + sQuad[eb][0]); line[1] = 0.5 * (sQuad[ea][1] + sQuad[eb][1]);
                    line[2] = nx; line[3] = ny;
                } else {
                    double Ex = Mx / N, Ey = My / N;
                    double Cxx = Mxx / N - Ex * Ex, Cxy = Mxy / N - Ex * Ey, Cyy = Myy / N - Ey * Ey;
                    double d = Cxx - Cyy, q4 = 4.0 * Cxy;
                    double disc = sqrt(d * d + q4 * Cxy);
                    double eig = 0.5 * (Cxx + Cyy + disc);
                    double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
                    double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
                    double fx, fy, M;
                    if (M1 > M2) { fx = nx1; fy = ny1; M = M1; } else { fx = nx2; fy = ny2; M = M2; }
                    double len = sqrt(M);
                    line[0] = Ex; line[1] = Ey;
                    if (len < 1e-12) { line[2] = nx; line[3] = ny; }
                    else { line[2] = fx / len; line[3] = fy / len; }
                }
                for (int q = 0; q < 4; q++) sLines[edge][q] = line[q];
            }
            __syncthreads();
            if (tid == 0)
                for (int i = 0; i < 4; i++) {
                    int j = (i + 1) & 3;
                    double A00 = sLines[i][3], A01 = -sLines[j][3], A10 = -sLines[i][2], A11 = sLines[j][2];
                    double B0 = -sLines[i][0] + sLines[j][0], B1 = -sLines[i][1] + sLines[j][1];
                    double det = A00 * A11 - A10 * A01;
                    if (fabs(det) > 0.001) {
                        double W00 = A11 / det, W01 = -A01 / det;
                        double L0 = W00 * B0 + W01 * B1;
                        sQuad[j][0] = sLines[i][0] + L0 * A00;
                        sQuad[j][1] = sLines[i][1] + L0 * A10;
                    }
                }
        }
        if (tid == 0) {
            uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
            uint32_t qi = atomicAdd(&counters[CK_CNT_QUADS], 1u);
            if (qi < (uint32_t)ws.quad_cap) {
                ck_quad_t q;
                for (int i = 0; i < 4; i++) { q.p[i][0] = sQuad[i][0]; q.p[i][1] = sQuad[i][1]; }
                q.reversed_border = reversed; q.rep0 = cl.rep0; q.rep1 = cl.rep1;
                ws.d_quads[(size_t)frame * ws.quad_cap + qi] = q;
            } else atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_QUADS_OVERFLOW);
        }
        PROF(10);
    }
}

// ---- split fit, first kernel: bounding box, border direction, sort, duplicate removal -> the cluster's extended sequence -----------
// The phases are k_fit's first ones (same helpers).  What is new is how a workgroup gets at its clusters: per cluster the old loop
// pays a chain of memory round trips (list entry -> record -> points -> weights) with nothing else to do, and the split fit's first
// kernel is nothing but that chain plus a sort.  Here the HEADS of a chunk's clusters (list entry and record) are fetched by one
// thread each, all at once, and the points of cluster k + 1 are in flight in registers while cluster k is sorted; the weights
// are left to k_chunk, which streams.  NTH threads per cluster, up to CAP points; MLDS / GK as in k_fit.
template <int NTH, int CAP, bool MLDS, int WPS, bool GK>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(WPS, 8))) void k_seq(FitArgs a) {
    using B = Block<NTH>;
    constexpr int EPL = CAP / NTH;
    constexpr bool PF = EPL <= 16; // the next cluster's points travel in registers
    constexpr int HB = MLDS ? ((CAP < 2048 ? CAP : 2048) + 1) * 4 : 2 * (1024 + 1) * 4; // the sort's histogram
    __shared__ unsigned long long sKeysL[GK ? 1 : CAP];
    __shared__ __attribute__((aligned(16))) unsigned char sHist[HB];
    __shared__ long long sScratch[6 * (NTH / 64) + 2];
    __shared__ uint32_t sWork;
    __shared__ uint32_t sHItem[64], sHStart[64], sHCount[64];
    const int tid = threadIdx.x;
    const ck_stage_ws &ws = a.ws;
    const int capr = GK ? ws.hcap : CAP;
    unsigned long long *const sKeys = GK ? ws.d_hscratch + (size_t)blockIdx.x * 2 * capr : sKeysL;
    unsigned long long *scratch8 = GK ? ws.d_hscratch + (size_t)blockIdx.x * 2 * capr + capr : ws.d_lscratch + (size_t)blockIdx.x * CK_LSCRATCH_PER_WG;
    uint32_t *sXY = reinterpret_cast<uint32_t *>(sKeys);
    int *iscr = reinterpret_cast<int *>(sScratch);
    const uint32_t n_work = min(*a.list_count, (uint32_t)a.list_cap);
    // clusters per dequeue: an add on the one counter of a class costs ~17 ns however many workgroups wait for it (26 000 of them were
    // 0.45 of the 0.7 ms of the smallest class); the chunks shrink towards the end of the list (guided), so large ones cost no tail
    constexpr uint32_t DQMAX = CAP <= 512 ? 64u : (CAP <= 1024 ? 32u : (CAP <= 2048 ? 16u : (CAP <= 4096 ? 8u : (CAP <= 8192 ? 4u : 2u))));
    const uint32_t per_wg4 = n_work / (gridDim.x * 4u);
    const uint32_t DQ = per_wg4 < 1u ? 1u : (per_wg4 > DQMAX ? DQMAX : per_wg4);
    const uint32_t static_total = gridDim.x * DQ;
    uint32_t chunk_base = blockIdx.x * DQ, chunk_len = DQ, dq_next = DQ;
    bool first_chunk = true;
    // packed point -> the staged form the sort reads: x | y << 16 | (u8)gx << 32 | (u8)gy << 40
    auto stage = [](ck_packed_point v) -> unsigned long long {
        const int k = (int)(v >> 1) & 3, sgn = (v & 1u) ? 1 : -1;
        const int dx = k == 2 ? -1 : (k == 1 ? 0 : 1), dy = k == 0 ? 0 : 1;
        return (unsigned long long)((v >> 16) & 0x1FFFu) | ((unsigned long long)((v >> 3) & 0x1FFFu) << 16) |
               ((unsigned long long)(uint8_t)(int8_t)(dx * sgn) << 32) | ((unsigned long long)(uint8_t)(int8_t)(dy * sgn) << 40);
    };
    for (;;) {
        if (!first_chunk) {
            if (static_total >= n_work) break;
            B::sync();
            if (tid == 0) sWork = static_total + atomicAdd(a.head, dq_next);
            B::sync();
            chunk_base = sWork;
            chunk_len = dq_next;
            if (a.guided) {
                const uint32_t g = chunk_base < n_work ? (n_work - chunk_base) / (gridDim.x * 2u) : 0u;
                dq_next = g < 1u ? 1u : (g > DQ ? DQ : g);
            }
        }
        first_chunk = false;
        if (chunk_base >= n_work) break;
        const int len = (int)min(chunk_len, n_work - chunk_base);
        B::sync(); // the heads of the chunk before this one have been read
        if (tid < len) {
            const uint32_t item = a.list[chunk_base + (uint32_t)tid];
            const ck_cluster_t cl = ws.d_clusters[(size_t)(item >> 20) * ws.cluster_cap + (item & 0xFFFFFu)];
            sHItem[tid] = item; sHStart[tid] = cl.start; sHCount[tid] = cl.count;
        }
        B::sync();
        ck_packed_point nxt[PF ? EPL : 1];
        auto fetch = [&](int k) {
            if constexpr (PF) {
                const uint32_t item = sHItem[k], cnt = sHCount[k];
                const ck_packed_point *p = ws.d_points + (size_t)(item >> 20) * ws.ext_cap + sHStart[k];
#pragma unroll
                for (int e = 0; e < EPL; e++) { const uint32_t i = (uint32_t)(tid + e * NTH); nxt[e] = (i < cnt && cnt <= (uint32_t)CAP) ? p[i] : 0u; }
            }
        };
        fetch(0);
        for (int k = 0; k < len; k++) {
            const uint32_t item = sHItem[k];
            const int frame = (int)(item >> 20), ci = (int)(item & 0xFFFFFu);
            const int sz0 = (int)sHCount[k];
            const ck_packed_point *pts = ws.d_points + (size_t)frame * ws.ext_cap + sHStart[k];
            ck_packed_point rawp[PF ? EPL : 1];
            if constexpr (PF) {
                __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0); // vmcnt(0): this cluster's points are here before the next one's are asked for (see k_tail)
#pragma unroll
                for (int e = 0; e < EPL; e++) rawp[e] = nxt[e];
                if (k + 1 < len) fetch(k + 1);
            }
            B::sync(); // the cluster before this one is done with the key array
            uint32_t *cstate = ws.d_cstate + 2 * ((size_t)frame * ws.cluster_cap + ci);
            if (tid == 0) *cstate = 0u; // rejected unless the end says otherwise (same thread: ordered)
            if (sz0 > CAP || sz0 < 1) continue; // cannot happen: the class lists are built from the counts
            if (a.stop_after == 0) continue;
            // ---- 1. bounding box ---------------------------------------------------------------------------------------------
            int xmin = 1 << 30, xmax = -(1 << 30), ymin = 1 << 30, ymax = -(1 << 30);
            if constexpr (PF) {
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const int i = tid + e * NTH;
                    if (i < sz0) {
                        const unsigned long long raw = stage(rawp[e]);
                        sKeys[i] = raw;
                        const int px = (int)(raw & 0xFFFF), py = (int)((raw >> 16) & 0xFFFF);
                        xmin = min(xmin, px); xmax = max(xmax, px);
                        ymin = min(ymin, py); ymax = max(ymax, py);
                    }
                }
            } else {
                for (int i = tid; i < sz0; i += NTH) {
                    const unsigned long long raw = stage(pts[i]);
                    sKeys[i] = raw;
                    const int px = (int)(raw & 0xFFFF), py = (int)((raw >> 16) & 0xFFFF);
                    xmin = min(xmin, px); xmax = max(xmax, px);
                    ymin = min(ymin, py); ymax = max(ymax, py);
                }
            }
            {
                int bb[4] = {xmin, -xmax, ymin, -ymax};
                B::reduce_min4(bb, iscr);
                xmin = bb[0]; xmax = -bb[1]; ymin = bb[2]; ymax = -bb[3];
            }
            if ((xmax - xmin) * (ymax - ymin) < a.min_tag_width) continue;
            if (a.stop_after == 10) continue;
            // ---- 2. border direction, sort ---------------------------------------------------------------------------------------
            long long dot;
            {
                constexpr int EPLS = CAP / NTH;
                int n2 = NTH;
                while (n2 < sz0) n2 <<= 1;
                const int epl = n2 / NTH;
                B::sync(); // raw points staged by other threads
                bool sorted = false;
                if constexpr (MLDS)
                    dot = keys_bucket_sort<NTH, EPLS>(sKeys, reinterpret_cast<uint32_t *>(sHist), sScratch, sz0, xmin, xmax, ymin, ymax,
                                                      a.normal_ok, a.reversed_ok, a.stop_after != 11, &sorted);
                else
                    dot = keys_bucket_sort_global<NTH, GK>(sKeys, reinterpret_cast<uint32_t *>(sHist), sScratch, scratch8, sz0, xmin, xmax, ymin, ymax,
                                                           a.normal_ok, a.reversed_ok, a.stop_after != 11, &sorted);
                if (sorted || GK) {}
                else if (EPLS >= 32 && epl == 32) dot = keys_sort<NTH, (EPLS >= 32 && !GK ? 32 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
                else if (EPLS >= 16 && epl == 16) dot = keys_sort<NTH, (EPLS >= 16 && !GK ? 16 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
                else if (EPLS >= 8 && epl == 8) dot = keys_sort<NTH, (EPLS >= 8 && !GK ? 8 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
                else if (EPLS >= 4 && epl == 4) dot = keys_sort<NTH, (EPLS >= 4 && !GK ? 4 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
                else if (EPLS >= 2 && epl == 2) dot = keys_sort<NTH, (EPLS >= 2 && !GK ? 2 : 1)>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
                else dot = keys_sort<NTH, 1>(sKeys, sScratch, sz0, xmin, xmax, ymin, ymax, a.normal_ok, a.reversed_ok, a.stop_after != 11);
            }
            const int reversed = dot < 0;
            if (reversed && !a.reversed_ok) continue;
            if (!reversed && !a.normal_ok) continue;
            if (a.stop_after == 1 || a.stop_after == 11) continue;
            // ---- 3. duplicate removal, packing to (x, y) (k_fit's: the compaction writes below what is still unread) -------------
            B::sync();
            int sz = 0;
            for (int base = 0; base < sz0; base += NTH) {
                const int i = base + tid;
                unsigned long long key = 0;
                int keep = 0;
                if (i < sz0) { key = sKeys[i]; keep = (i == 0) || (sKeys[i - 1] != key); }
                int tot = 0;
                const int incl = B::scan_flag(keep, iscr, &tot);
                B::sync();
                if (keep) sXY[sz + (int)incl - 1] = (uint32_t)(key & 0x3FFFFFFu);
                sz += (int)tot;
                B::sync();
            }
            if (sz < 24) continue;
            const int ksz = sz / 12 < 20 ? sz / 12 : 20;
            if (ksz < 2) continue;
            if (a.stop_after == 2) continue;
            // ---- 4. the extended sequence.  Its place in the frame is handed out in the order the clusters get here (k_chunk and k_tail
            // do not care where a cluster lies), so the positions k_chunk works through are all live ones --------------------------
            if (tid == 0) sWork = atomicAdd(ws.d_counters + (size_t)frame * CK_CNT_STRIDE + CK_CNT_EXT, (uint32_t)(sz + CK_EXT_HALO));
            B::sync();
            const uint32_t es32 = sWork;
            uint32_t *exy = ws.d_ext_xy + (size_t)frame * ws.ext_cap + es32;
            for (int q = tid; q < sz + CK_EXT_HALO; q += NTH) {
                int src = q - CK_EXT_PRE; // sz >= 24: at most two steps bring it into [0, sz)
                if (src < 0) src += sz;
                if (src < 0) src += sz;
                if (src >= sz) src -= sz;
                exy[q] = sXY[src] | ((uint32_t)ksz << 26);
            }
            if (tid == 0) { cstate[1] = es32; cstate[0] = (uint32_t)sz | (reversed ? 0x80000000u : 0u); }
        }
    }
}

// ---- split fit, middle kernel: windowed line-fit error, smoothing and maxima of EVERY position of a frame's extended sequences --------
// One workgroup decides CK_SPAN = 960 consecutive positions (15 words of 64) from 1024 loaded ones; it knows nothing about clusters:
// a cluster's extended sequence carries its own neighbours (ck_internal.h), the window half-width travels in the point word, and
// what it finds is filed by POSITION — a bit per position, the smoothed errors of a span's maxima side by side in position
// order, moment sums of every aligned block of 32 positions — so k_tail picks its cluster's share with a few popcounts.  Positions
// nobody wrote (rejected clusters, the room duplicates left) compute garbage that nobody reads.  All sums are exact integers,
// the doubles are formed by the operations of k_fit's chunk loop in the same order: the same bits.
constexpr int KL = 1024, KOFF = 32;
static_assert(CK_SPAN == 960 && KOFF >= CK_EXT_PRE + 3 && KL - KOFF - CK_SPAN >= CK_EXT_POST + 4, "span geometry");
__device__ __forceinline__ void k_chunk_body(const ck_stage_ws &ws, int qw, int qh) {
    __shared__ __attribute__((aligned(16))) unsigned long long sP64[3][KL]; // inclusive sums from the first loaded position: Mxx, Mxy, Myy
    __shared__ __attribute__((aligned(16))) uint32_t sP32[3][KL];           // Mx, My, W (a window's sums stay below 2^32: differences are exact)
    __shared__ uint8_t sKsz[KL];
    __shared__ unsigned long long sScan64[3][4];
    __shared__ uint32_t sScan32[3][4];
    __shared__ uint32_t sCnt[16];
    // the errors and their smoothed values take the place of two of the sums once every window has been formed (the four errors of a
    // thread wait in registers for the barrier in between): 38 KB instead of 46, four workgroups per CU instead of three
    double *sS = reinterpret_cast<double *>(&sP64[0][0]), *sErr = reinterpret_cast<double *>(&sP64[1][0]);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int frame = blockIdx.y;
    const uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
    const uint32_t ext_total = min(counters[CK_CNT_EXT], (uint32_t)ws.ext_cap); // positions k_seq handed out
    const uint32_t *xy = ws.d_ext_xy + (size_t)frame * ws.ext_cap;
    uint16_t *w16 = ws.d_ext_w + (size_t)frame * ws.ext_cap;
    const uint16_t *wq = ws.d_wimg + (size_t)frame * qw * qh;
    double *mval = ws.d_maxval + (size_t)frame * (ws.ext_cap / 2);
    uint16_t *mpos = ws.d_maxpos + (size_t)frame * (ws.ext_cap / 2);
    unsigned long long *mmask = ws.d_maxmask + (size_t)frame * (ws.ext_cap / 64);
    uint16_t *mpre = ws.d_maxpre + (size_t)frame * (ws.ext_cap / 64);
    long long *blk = ws.d_blk + (size_t)frame * 6 * (ws.ext_cap / 32);
    // the gradient weight of every point: one gather from the weight image (k_fit's phase 3; a position nobody wrote gathers at
    // whatever 13-bit coordinates it holds, inside the image or refused), kept for k_tail in the sequence's weight array.  The
    // span after this one is fetched while this one is worked on: its points at the top, their weights once those have arrived
    auto load_xy = [&](uint32_t s) -> uint4 {
        const long long p0 = (long long)s * CK_SPAN - KOFF + 4 * tid;
        if ((unsigned long long)s * CK_SPAN < ext_total && p0 >= 0 && p0 + 4 <= (long long)ws.ext_cap) return *reinterpret_cast<const uint4 *>(xy + p0);
        return make_uint4(0, 0, 0, 0);
    };
    auto gather_w = [&](uint32_t v) -> uint32_t {
        const int x = (int)((v >> 13) & 0x1FFFu), y = (int)(v & 0x1FFFu);
        const int ix = (x + 1) >> 1, iy = (y + 1) >> 1;
        return (ix < qw && iy < qh) ? (uint32_t)wq[(size_t)iy * qw + ix] : 1u;
    };
    uint4 x4 = load_xy(blockIdx.x);
    uint32_t ww[4] = {gather_w(x4.x), gather_w(x4.y), gather_w(x4.z), gather_w(x4.w)};
    for (uint32_t s = blockIdx.x; (unsigned long long)s * CK_SPAN < ext_total; s += gridDim.x) {
        // 1. four consecutive positions per thread: moments, running sums, one scan over the workgroup
        __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0); // vmcnt(0): this span's points and weights (asked for during the span before) are here
        const long long p0 = (long long)s * CK_SPAN - KOFF + 4 * tid;
        const uint32_t xw[4] = {x4.x, x4.y, x4.z, x4.w};
        if (tid >= KOFF / 4 && tid < (KOFF + CK_SPAN) / 4) // the positions this span decides: their weights stay for k_tail
            *reinterpret_cast<uint2 *>(w16 + p0) = make_uint2(ww[0] | (ww[1] << 16), ww[2] | (ww[3] << 16));
        unsigned long long l64[4][3];
        uint32_t l32[4][3];
        unsigned long long a64[3] = {0, 0, 0};
        uint32_t a32[3] = {0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const M6 m = moments_of(xw[e] & 0x3FFFFFFu, ww[e]);
            a64[0] += (unsigned long long)m.Mxx; a64[1] += (unsigned long long)m.Mxy; a64[2] += (unsigned long long)m.Myy;
            a32[0] += (uint32_t)m.Mx; a32[1] += (uint32_t)m.My; a32[2] += (uint32_t)m.W;
#pragma unroll
            for (int q = 0; q < 3; q++) { l64[e][q] = a64[q]; l32[e][q] = a32[q]; }
        }
        unsigned long long x64[3];
        uint32_t x32[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { x64[q] = wave_scan_u64(a64[q]); x32[q] = wave_scan_u32(a32[q]); }
        lds_barrier(); // the previous span's readers are done with every array
        if (lane == 63)
#pragma unroll
            for (int q = 0; q < 3; q++) { sScan64[q][wv] = x64[q]; sScan32[q][wv] = x32[q]; }
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 3; q++) {
            unsigned long long b64 = 0;
            uint32_t b32 = 0;
            for (int k = 0; k < wv; k++) { b64 += sScan64[q][k]; b32 += sScan32[q][k]; }
            x64[q] += b64 - a64[q]; x32[q] += b32 - a32[q]; // exclusive of this thread's four
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = 4 * tid + e;
#pragma unroll
            for (int q = 0; q < 3; q++) { sP64[q][j] = l64[e][q] + x64[q]; sP32[q][j] = l32[e][q] + x32[q]; }
            const uint32_t k = (xw[e] >> 26) & 31u;
            sKsz[j] = (uint8_t)(k > 20u ? 20u : k);
        }
        // the next span's points: asked for here, behind this span's own wait (the compiler's waits are for "everything outstanding":
        // at the top of the loop the request would be waited for at once); their weights once the errors are done
        const uint4 nx4 = load_xy(s + gridDim.x);
        lds_barrier();
        // 2. moment sums of the span's 30 aligned blocks of 32 positions (Mx, My, Mxx, Mxy, Myy, W: the order of M6)
        if (tid < 180) {
            const int b = tid / 6, q = tid - 6 * b;
            const int jh = KOFF + 32 * b + 31, jl = jh - 32;
            long long d;
            if (q == 0) d = (long long)(uint32_t)(sP32[0][jh] - sP32[0][jl]);
            else if (q == 1) d = (long long)(uint32_t)(sP32[1][jh] - sP32[1][jl]);
            else if (q == 5) d = (long long)(uint32_t)(sP32[2][jh] - sP32[2][jl]);
            else d = (long long)(sP64[q - 2][jh] - sP64[q - 2][jl]);
            blk[((size_t)s * (CK_SPAN / 32) + b) * 6 + q] = d;
        }
        // 3. windowed line-fit error of the positions whose smoothed value is needed (k_fit's chunk loop, same operations)
        double ev[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = tid + 256 * r;
            ev[r] = 0.0;
            if (j >= KOFF - 4 && j < KOFF + CK_SPAN + 4) {
                const int ksz = (int)sKsz[j];
                const int hi = j + ksz, lo = j - ksz - 1;
                const uint32_t mx = sP32[0][hi] - sP32[0][lo], my = sP32[1][hi] - sP32[1][lo], mw = sP32[2][hi] - sP32[2][lo];
                M6 m;
                m.Mxx = (long long)(sP64[0][hi] - sP64[0][lo]); m.Mxy = (long long)(sP64[1][hi] - sP64[1][lo]); m.Myy = (long long)(sP64[2][hi] - sP64[2][lo]);
                const double inv = mw < (uint32_t)INV_N ? g_inv_table.v[mw] : 1.0 / (double)mw;
                const double Ex = (0.5 * (double)mx) * inv, Ey = (0.5 * (double)my) * inv;
                const double Cxx = (0.25 * f64_of_u52((unsigned long long)m.Mxx)) * inv - Ex * Ex;
                const double Cxy = (0.25 * f64_of_u52((unsigned long long)m.Mxy)) * inv - Ex * Ey;
                const double Cyy = (0.25 * f64_of_u52((unsigned long long)m.Myy)) * inv - Ey * Ey;
                const double d = Cxx - Cyy, q4 = 4.0 * Cxy;
                const double disc = sqrt(d * d + q4 * Cxy);
                ev[r] = (double)(2 * ksz + 1) * (0.5 * ((Cxx + Cyy) - disc));
            }
        }
        lds_barrier(); // every window has been read: the sums may go
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = tid + 256 * r;
            if (j >= KOFF - 4 && j < KOFF + CK_SPAN + 4) sErr[j] = ev[r];
        }
        const uint32_t nww[4] = {gather_w(nx4.x), gather_w(nx4.y), gather_w(nx4.z), gather_w(nx4.w)};
        lds_barrier();
        // 4. smoothed errors (the oracle's one-value-at-a-time sum, in its order)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = tid + 256 * r;
            if (j >= KOFF - 1 && j < KOFF + CK_SPAN + 1) {
                double sm = 0.0;
#pragma unroll
                for (int q = 0; q < 7; q++) sm += sErr[j + q - 3] * k_smooth[q];
                sS[j] = sm;
            }
        }
        lds_barrier();
        // 5. maxima: one word of 64 positions per wave and round
        unsigned long long bal[4];
        double myv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = KOFF + tid + 256 * r;
            bool ismax = false;
            myv[r] = 0.0;
            if (j < KOFF + CK_SPAN) {
                const double v = sS[j];
                myv[r] = v;
                ismax = v > sS[j + 1] && v > sS[j - 1];
            }
            bal[r] = __ballot(ismax);
            const int k = 4 * r + wv;
            if (lane == 0 && k < 16) sCnt[k] = (uint32_t)__popcll(bal[r]);
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = 4 * r + wv;
            if (k >= CK_SPAN / 64) continue;
            uint32_t pre = 0;
            for (int i = 0; i < k; i++) pre += sCnt[i];
            if (lane == 0) { mmask[(size_t)s * (CK_SPAN / 64) + k] = bal[r]; mpre[(size_t)s * (CK_SPAN / 64) + k] = (uint16_t)pre; }
            if ((bal[r] >> lane) & 1ull) {
                const uint32_t pos = pre + (uint32_t)__popcll(bal[r] & ((1ull << lane) - 1ull));
                mval[(size_t)s * (CK_SPAN / 2) + pos] = myv[r];
                mpos[(size_t)s * (CK_SPAN / 2) + pos] = (uint16_t)(tid + 256 * r);
            }
        }
        x4 = nx4;
#pragma unroll
        for (int e = 0; e < 4; e++) ww[e] = nww[e];
    }
}

__global__ __launch_bounds__(256) void k_chunk(ck_stage_ws ws, int qw, int qh) { k_chunk_body(ws, qw, qh); }

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ long long readlane_i64(long long v, int l) {
    const unsigned long long u = (unsigned long long)v;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// ---- split fit, last kernel: one wave per cluster ----------------------------------------------------------------------------------
// The cluster's maxima are runs of the position-ordered lists k_chunk left (one run per span the cluster touches); the (max_nmaxima + 1)-th
// largest smoothed error is the threshold, the survivors in index order are the candidate corners; their moment prefix sums come
// from the aligned block sums (a wave scan) plus at most 32 points each; from there on the phases are k_fit's (pair fits, 4-subsets,
// corners and checks, edge refinement) on 64 lanes.  Sums that start at the block boundary before the cluster's first point carry
// whatever the positions before it hold — the same addend in every prefix sum, so it leaves with the differences.
constexpr int MAXRUN = 72; // spans a cluster of CK_HUGE_CAP points can touch
#ifdef CK_FLAT_DEBUG
__device__ unsigned int g_flat_dbg[32];
#define FDBG(k) do { if (tid == 0) atomicAdd(&g_flat_dbg[k], 1u); } while (0)
#define FDBGV(k, v) do { if (tid == 0) atomicAdd(&g_flat_dbg[k], (unsigned)(v)); } while (0)
#else
#define FDBG(k)
#define FDBGV(k, v)
#endif
__device__ __forceinline__ void k_tail_body(const FitArgs &a) {
    constexpr int NTH = 64, T_TAIL = MAXSEL, T_HEAD = MAXSEL + 1;
    __shared__ __attribute__((aligned(16))) unsigned char sPraw[sizeof(PairFit) * MAXSEL * MAXSEL];
    __shared__ long long sSelI[2 * MAXSEL][6];
    __shared__ long long sFP[2 * (MAXSEL + 2)][6]; // block prefixes and partial sums of the targets (5b) ...
    long long (*sF6)[6] = sFP, (*sPart)[6] = sFP + (MAXSEL + 2);
    static_assert(sizeof(long long) * 6 * 2 * (MAXSEL + 2) >= 4 * 4 * MAXRUN, "the run tables live in the prefix tables' bytes");
    uint32_t *sRunStart = reinterpret_cast<uint32_t *>(sFP), *sRunN = sRunStart + MAXRUN, *sRunOff = sRunN + MAXRUN; // ... which hold the runs of the maxima lists until the selection is done
    int *sRunO = reinterpret_cast<int *>(sRunOff + MAXRUN);
    __shared__ int sSelIdx[MAXSEL];
    // The corner phase uses four lanes per cluster: the clusters that get that far leave a record per side here and are worked off
    // eight at a time (32 lanes) — an eighth of the instructions per cluster
    constexpr int BATCH = 8;
    struct SideRec { long long Mx, My, W; double nx, ny, mse; };
    __shared__ SideRec sB[BATCH][4];
    __shared__ uint32_t sBMeta[BATCH][4]; // frame, reversed border, rep0, rep1
    __shared__ uint32_t sWork;
    __shared__ int sBad;
    __shared__ double sLines[4][4];
    __shared__ double sQuad[4][2];
    long long (*sSelE)[6] = sSelI + MAXSEL;
    double (*sRefine)[16][2] = reinterpret_cast<double (*)[16][2]>(sPraw);
    const int tid = threadIdx.x, lane = tid;
    const ck_stage_ws &ws = a.ws;
    const uint32_t n_work = min(*a.list_count, (uint32_t)a.list_cap);
    constexpr uint32_t DQMAX = 64u; // (one lane per head; few adds on the one counter: see k_seq)
    const uint32_t per_wg4 = n_work / (gridDim.x * 4u);
    const uint32_t DQ = per_wg4 < 1u ? 1u : (per_wg4 > DQMAX ? DQMAX : per_wg4);
    const uint32_t static_total = gridDim.x * DQ;
    uint32_t chunk_base = blockIdx.x * DQ, chunk_len = DQ, dq_next = DQ;
    bool first_chunk = true;
    auto rank_excl = [](const unsigned long long *mmask, const uint16_t *mpre, uint32_t x, uint32_t s) -> uint32_t {
        // maxima of span s before position x (clamped: bounded even on garbage)
        if (x == s * CK_SPAN) return 0u;
        const uint32_t w = (x - 1) >> 6, nb = x - (w << 6);
        const unsigned long long m = mmask[w] & (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull));
        const uint32_t r = (uint32_t)mpre[w] + (uint32_t)__popcll(m);
        return r > CK_SPAN / 2 ? CK_SPAN / 2 : r;
    };
    for (;;) {
        // The wave's time is a chain of memory round trips unless they are shared: the HEADS of a chunk's clusters (list entry, state
        // word, record, the ranks that bound their maxima lists) are fetched by one lane each, all at once; a cluster's maxima and its
        // first block sums are fetched while the cluster before it is fitted.
        if (!first_chunk) {
            if (static_total >= n_work) break;
            wave_sync();
            if (tid == 0) sWork = static_total + atomicAdd(a.head, dq_next);
            wave_sync();
            chunk_base = sWork;
            chunk_len = dq_next;
            if (a.guided) {
                const uint32_t g = chunk_base < n_work ? (n_work - chunk_base) / (gridDim.x * 2u) : 0u;
                dq_next = g < 1u ? 1u : (g > DQ ? DQ : g);
            }
        }
        first_chunk = false;
        if (chunk_base >= n_work) break;
        const int len = (int)min(chunk_len, n_work - chunk_base);
        // ---- heads: lane l takes cluster l of the chunk ----------------------------------------------------------------------
        uint32_t h_item = 0, h_st = 0, h_e0 = 0, h_rep0 = 0, h_rep1 = 0, h_s0 = 0, h_n0 = 0, h_s1 = 0, h_n1 = 0;
        int h_nruns = 0;
        if (lane < len) {
            h_item = a.list[chunk_base + (uint32_t)lane];
            const int frame = (int)(h_item >> 20), ci = (int)(h_item & 0xFFFFFu);
            const uint2 st2 = *reinterpret_cast<const uint2 *>(ws.d_cstate + 2 * ((size_t)frame * ws.cluster_cap + ci));
            const uint32_t st = st2.x;
            const ck_cluster_t cl = ws.d_clusters[(size_t)frame * ws.cluster_cap + ci];
            const uint32_t sz = st & 0x7FFFFFFFu;
            // (below 24: rejected before the fit; above the count or beyond the buffer: a state nobody wrote in this call)
            if (sz >= 24u && sz <= cl.count && (unsigned long long)st2.y + sz + CK_EXT_HALO <= (unsigned long long)ws.ext_cap) {
                h_st = st; h_rep0 = cl.rep0; h_rep1 = cl.rep1;
                const uint32_t e0 = st2.y + CK_EXT_PRE, e1 = e0 + sz;
                h_e0 = e0;
                const uint32_t s_lo = e0 / CK_SPAN, s_hi = (e1 - 1) / CK_SPAN;
                h_nruns = (int)(s_hi - s_lo) + 1;
                if (h_nruns <= 2) {
                    const unsigned long long *mmask = ws.d_maxmask + (size_t)frame * (ws.ext_cap / 64);
                    const uint16_t *mpre = ws.d_maxpre + (size_t)frame * (ws.ext_cap / 64);
                    const uint32_t hi0 = min(e1, s_lo * CK_SPAN + CK_SPAN);
                    const uint32_t r0 = rank_excl(mmask, mpre, e0, s_lo), r1 = rank_excl(mmask, mpre, hi0, s_lo);
                    h_s0 = s_lo * (CK_SPAN / 2) + r0; h_n0 = r1 > r0 ? r1 - r0 : 0u;
                    if (h_nruns == 2) { h_s1 = s_hi * (CK_SPAN / 2); h_n1 = rank_excl(mmask, mpre, e1, s_hi); }
                }
            }
        }
        // what lane `lane` of cluster k needs first: its maximum (if the cluster has at most 64, in at most two runs).  (The sums of the
        // cluster's first 64 blocks travelled this way too: twelve more registers per stage, seven spilled — and a build of this kernel
        // that spills has returned wrong results, see the note at the kernel's definition)
        struct Pre { double v; uint32_t pos; };
        auto prefetch = [&](int k) -> Pre {
            Pre p;
            p.v = 0.0; p.pos = 0;
            if (k >= len) return p;
            const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)h_st, k);
            if (!st) return p;
            const uint32_t item = (uint32_t)__builtin_amdgcn_readlane((int)h_item, k);
            const int frame = (int)(item >> 20);
            const int nruns = __builtin_amdgcn_readlane(h_nruns, k);
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)h_n0, k), n1 = (uint32_t)__builtin_amdgcn_readlane((int)h_n1, k);
            // (every v_readlane at a place all lanes reach: what it returns for a lane that is switched off is not defined — a build
            // that spilled these registers restored only the active lanes around such a read, and lost detections)
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)h_s0, k), s1 = (uint32_t)__builtin_amdgcn_readlane((int)h_s1, k);
            if (nruns <= 2 && n0 + n1 <= 64u && (uint32_t)lane < n0 + n1) {
                const uint32_t idx = (uint32_t)lane < n0 ? s0 + (uint32_t)lane : s1 + ((uint32_t)lane - n0);
                p.v = (ws.d_maxval + (size_t)frame * (ws.ext_cap / 2))[idx];
                p.pos = (ws.d_maxpos + (size_t)frame * (ws.ext_cap / 2))[idx];
            }
            return p;
        };
        int nb = 0; // clusters waiting for the corner phase
        auto flush = [&]() {
            // ---- 5d. lines, corners, geometric checks: lane 4 c + i owns side i / corner i of waiting cluster c; every value is formed by
            // the same operations as in the oracle's sequential code, and the verdict is the conjunction of all checks ---------
            const int c = tid >> 2, li = tid & 3;
            const bool mine = c < nb;
            int ok = mine ? 1 : 0;
            double line[4] = {0, 0, 0, 0};
            if (mine) {
                const SideRec r = sB[c][li];
                const double inv = 1.0 / (double)r.W;
                line[0] = (0.5 * (double)r.Mx) * inv; line[1] = (0.5 * (double)r.My) * inv;
                line[2] = r.nx; line[3] = r.ny;
                if (r.mse > a.max_mse) ok = 0;
            }
            double ln[4];
#pragma unroll
            for (int q = 0; q < 4; q++) ln[q] = __shfl(line[q], (tid & ~3) | ((li + 1) & 3), 64);
            double Px, Py;
            {
                double A00 = line[3], A01 = -ln[3], A10 = -line[2], A11 = ln[2];
                double B0_ = -line[0] + ln[0], B1 = -line[1] + ln[1];
                double det = A00 * A11 - A10 * A01;
                if (fabs(det) < 0.001) ok = 0;
                double W00 = A11 / det, W01 = -A01 / det;
                double L0 = W00 * B0_ + W01 * B1;
                Px = line[0] + L0 * A00;
                Py = line[1] + L0 * A10;
            }
            const int g0 = tid & ~3;
            auto corner_x = [&](int q) { return __shfl(Px, g0 | (q & 3), 64); };
            auto corner_y = [&](int q) { return __shfl(Py, g0 | (q & 3), 64); };
            {
                const int t = li & 1;
                const int va = t ? 2 : 0, vb = t ? 3 : 1, vc = t ? 0 : 2;
                const double ax = corner_x(va), ay = corner_y(va), bx = corner_x(vb), by = corner_y(vb), cx = corner_x(vc), cy = corner_y(vc);
                double len[3];
                { double ddx = bx - ax, ddy = by - ay; len[0] = sqrt(ddx * ddx + ddy * ddy); }
                { double ddx = cx - bx, ddy = cy - by; len[1] = sqrt(ddx * ddx + ddy * ddy); }
                { double ddx = ax - cx, ddy = ay - cy; len[2] = sqrt(ddx * ddx + ddy * ddy); }
                double pp = (len[0] + len[1] + len[2]) / 2.0;
                double term = sqrt(pp * (pp - len[0]) * (pp - len[1]) * (pp - len[2]));
                double t0 = __shfl(term, g0, 64), t1 = __shfl(term, g0 | 1, 64);
                double area = 0.0;
                area += t0; area += t1;
                double tw = (double)a.min_tag_width;
                if (area < 0.95 * tw * tw) ok = 0;
            }
            {
                const double x1 = corner_x(li + 1), y1 = corner_y(li + 1), x2 = corner_x(li + 2), y2 = corner_y(li + 2);
                double dx1 = x1 - Px, dy1 = y1 - Py;
                double dx2 = x2 - x1, dy2 = y2 - y1;
                double cs = (dx1 * dx2 + dy1 * dy2) / sqrt((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2));
                if (cs > a.cos_critical || cs < -a.cos_critical) ok = 0;
                if (dx1 * dy2 < dy1 * dx2) ok = 0;
            }
            const unsigned long long okb = __ballot(ok != 0);
            wave_sync(); // every side record has been read: the accepted clusters' corners take their first bytes
            if (mine && ((okb >> (4 * c)) & 0xFull) == 0xFull) {
                double qx = Px, qy = Py;
                if (a.decimate > 1) { qx = (qx - 0.5) * (double)a.decimate + 0.5; qy = (qy - 0.5) * (double)a.decimate + 0.5; }
                double *dst = reinterpret_cast<double *>(&sB[c][li]);
                dst[0] = qx; dst[1] = qy;
            }
            wave_sync();
            for (int cc = 0; cc < nb; cc++) {
                if (((okb >> (4 * cc)) & 0xFull) != 0xFull) continue;
                FDBG(6);
                const int frame = (int)sBMeta[cc][0], reversed = (int)sBMeta[cc][1];
                const uint32_t rep0 = sBMeta[cc][2], rep1 = sBMeta[cc][3];
                const uint8_t *im = a.im + (size_t)frame * a.pitch;
                if (tid < 4) { const double *src = reinterpret_cast<const double *>(&sB[cc][tid]); sQuad[tid][0] = src[0]; sQuad[tid][1] = src[1]; }
                wave_sync();
                if (a.refine) { // edge refinement (oracle refine_edges), as in k_fit: lane (edge, k) evaluates sample 16 * round + k, one lane per edge accumulates in sample order
                    const int edge = (tid >> 4) & 3, k = tid & 15;
                    const int ea = edge, eb = (edge + 1) & 3;
                    double nx = sQuad[eb][1] - sQuad[ea][1];
                    double ny = -sQuad[eb][0] + sQuad[ea][0];
                    const double mag = sqrt(nx * nx + ny * ny);
                    nx = nx / mag; ny = ny / mag;
                    if (reversed) { nx = -nx; ny = -ny; }
                    int nsamples = (int)(mag / 8.0);
                    if (nsamples < 16) nsamples = 16;
                    int max_samples = nsamples;
        #pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) max_samples = max(max_samples, __shfl_xor(max_samples, d, 64));
                    double Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, N = 0;
                    wave_sync(); // the pair table's bytes become the sample buffer
                    for (int base = 0; base < max_samples; base += 16) {
                        {
                            const int sidx = base + k;
                            double bx = __builtin_nan(""), by = 0;
                            if (sidx < nsamples) {
                                double alpha = (1.0 + (double)sidx) / ((double)nsamples + 1.0);
                                double x0 = alpha * sQuad[ea][0] + (1.0 - alpha) * sQuad[eb][0];
                                double y0 = alpha * sQuad[ea][1] + (1.0 - alpha) * sQuad[eb][1];
                                double Mn = 0, Mcount = 0;
                                const int range = a.decimate + 1;
                                for (int n = -range; n <= range; n++) {
                                    double grange = 1.0;
                                    int x1 = (int)(x0 + ((double)n + grange) * nx), y1 = (int)(y0 + ((double)n + grange) * ny);
                                    if (x1 < 0 || x1 >= a.w || y1 < 0 || y1 >= a.h) continue;
                                    int x2 = (int)(x0 + ((double)n - grange) * nx), y2 = (int)(y0 + ((double)n - grange) * ny);
                                    if (x2 < 0 || x2 >= a.w || y2 < 0 || y2 >= a.h) continue;
                                    int g1 = im[(size_t)y1 * a.stride + x1], g2 = im[(size_t)y2 * a.stride + x2];
                                    if (g1 < g2) continue;
                                    double weight = (double)((g2 - g1) * (g2 - g1));
                                    Mn += weight * (double)n;
                                    Mcount += weight;
                                }
                                if (Mcount != 0) {
                                    double n0 = Mn / Mcount;
                                    bx = x0 + n0 * nx; by = y0 + n0 * ny;
                                }
                            }
                            sRefine[edge][k][0] = bx; sRefine[edge][k][1] = by;
                        }
                        wave_sync();
                        if (k == 0)
                            for (int q = 0; q < 16 && base + q < nsamples; q++) {
                                double bx = sRefine[edge][q][0], by = sRefine[edge][q][1];
                                if (bx != bx) continue;
                                Mx += bx; My += by; Mxx += bx * bx; Mxy += bx * by; Myy += by * by; N += 1.0;
                            }
                        wave_sync();
                    }
                    if (k == 0) {
                        double line[4];
                        if (N < 2.0) {
                            line[0] = 0.5 * (sQuad[ea][0] + sQuad[eb][0]); line[1] = 0.5 * (sQuad[ea][1] + sQuad[eb][1]);
                            line[2] = nx; line[3] = ny;
                        } else {
                            double Ex = Mx / N, Ey = My / N;
                            double Cxx = Mxx / N - Ex * Ex, Cxy = Mxy / N - Ex * Ey, Cyy = Myy / N - Ey * Ey;
                            double d = Cxx - Cyy, q4 = 4.0 * Cxy;
                            double disc = sqrt(d * d + q4 * Cxy);
                            double eig = 0.5 * (Cxx + Cyy + disc);
                            double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
                            double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
                            double fx, fy, M;
                            if (M1 > M2) { fx = nx1; fy = ny1; M = M1; } else { fx = nx2; fy = ny2; M = M2; }
                            double len = sqrt(M);
                            line[0] = Ex; line[1] = Ey;
                            if (len < 1e-12) { line[2] = nx; line[3] = ny; }
                            else { line[2] = fx / len; line[3] = fy / len; }
                        }
                        for (int q = 0; q < 4; q++) sLines[edge][q] = line[q];
                    }
                    wave_sync();
                    if (tid == 0)
                        for (int i = 0; i < 4; i++) {
                            int j = (i + 1) & 3;
                            double A00 = sLines[i][3], A01 = -sLines[j][3], A10 = -sLines[i][2], A11 = sLines[j][2];
                            double B0_ = -sLines[i][0] + sLines[j][0], B1 = -sLines[i][1] + sLines[j][1];
                            double det = A00 * A11 - A10 * A01;
                            if (fabs(det) > 0.001) {
                                double W00 = A11 / det, W01 = -A01 / det;
                                double L0 = W00 * B0_ + W01 * B1;
                                sQuad[j][0] = sLines[i][0] + L0 * A00;
                                sQuad[j][1] = sLines[i][1] + L0 * A10;
                            }
                        }
                }
                if (tid == 0) {
                    uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
                    uint32_t qi = atomicAdd(&counters[CK_CNT_QUADS], 1u);
                    if (qi < (uint32_t)ws.quad_cap) {
                        ck_quad_t q;
                        for (int i = 0; i < 4; i++) { q.p[i][0] = sQuad[i][0]; q.p[i][1] = sQuad[i][1]; }
                        q.reversed_border = reversed; q.rep0 = rep0; q.rep1 = rep1;
                        ws.d_quads[(size_t)frame * ws.quad_cap + qi] = q;
                    } else atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_QUADS_OVERFLOW);
                }
                wave_sync(); // the quad has been written out: the next accepted cluster may take the corner slots
            }
            nb = 0;
        };
        Pre nxt = prefetch(0);
        for (int k = 0; k < len; k++) {
        // what was fetched for this cluster has to be here BEFORE the next cluster's fetch is issued: the compiler's waits inside the
        // loops below are for "everything outstanding", and would otherwise sit on the fetch just issued
        __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0); // vmcnt(0)
        const Pre cur = nxt;
        nxt = prefetch(k + 1);
        wave_sync();
        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)h_st, k);
        FDBG(0);
        if (!st) continue;
        FDBG(1); FDBGV(13, st & 0x7FFFFFFFu);
        const uint32_t item = (uint32_t)__builtin_amdgcn_readlane((int)h_item, k);
        const int frame = (int)(item >> 20);
        const int sz = (int)(st & 0x7FFFFFFFu), reversed = (int)(st >> 31);
        const uint32_t rep0 = (uint32_t)__builtin_amdgcn_readlane((int)h_rep0, k), rep1 = (uint32_t)__builtin_amdgcn_readlane((int)h_rep1, k);
        if (a.stop_after <= 4) continue;
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)h_e0, k), e1 = e0 + (uint32_t)sz; // the cluster's points: positions [e0, e1)
        const uint32_t *xy = ws.d_ext_xy + (size_t)frame * ws.ext_cap;
        const uint16_t *w16 = ws.d_ext_w + (size_t)frame * ws.ext_cap;
        const double *mval = ws.d_maxval + (size_t)frame * (ws.ext_cap / 2);
        const uint16_t *mpos = ws.d_maxpos + (size_t)frame * (ws.ext_cap / 2);
        const unsigned long long *mmask = ws.d_maxmask + (size_t)frame * (ws.ext_cap / 64);
        const uint16_t *mpre = ws.d_maxpre + (size_t)frame * (ws.ext_cap / 64);
        const long long *blk = ws.d_blk + (size_t)frame * 6 * (ws.ext_cap / 32);

        // ---- the cluster's maxima: one run of k_chunk's lists per span it touches ------------------------------------------------
        const uint32_t s_lo = e0 / CK_SPAN, s_hi = (e1 - 1) / CK_SPAN;
        const int nruns = (int)(s_hi - s_lo) + 1;
        if (nruns > MAXRUN) continue; // cannot happen: sz <= CK_HUGE_CAP
        uint32_t nmax_u = 0;
        if (tid == 0) sBad = 0;
        bool prefetched = false; // this lane's maximum is in `cur`
        if (nruns <= 2) { // bounded by the head lane already
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)h_n0, k), n1 = (uint32_t)__builtin_amdgcn_readlane((int)h_n1, k);
            const uint32_t rs0 = (uint32_t)__builtin_amdgcn_readlane((int)h_s0, k), rs1 = (uint32_t)__builtin_amdgcn_readlane((int)h_s1, k);
            if (tid == 0) {
                sRunStart[0] = rs0; sRunN[0] = n0; sRunOff[0] = 0; sRunO[0] = (int)(s_lo * CK_SPAN) - (int)e0;
                sRunStart[1] = rs1; sRunN[1] = n1; sRunOff[1] = n0; sRunO[1] = (int)(s_hi * CK_SPAN) - (int)e0;
            }
            nmax_u = n0 + n1;
            prefetched = nmax_u <= 64u;
        } else
        for (int rb = 0; rb < nruns; rb += 64) {
            const int r = rb + lane;
            uint32_t n = 0, start = 0;
            int obase = 0;
            if (r < nruns) {
                const uint32_t s = s_lo + (uint32_t)r;
                const uint32_t lo = max(e0, s * CK_SPAN), hi = min(e1, s * CK_SPAN + CK_SPAN);
                const uint32_t r0 = rank_excl(mmask, mpre, lo, s), r1 = rank_excl(mmask, mpre, hi, s);
                n = r1 > r0 ? r1 - r0 : 0u;
                start = s * (CK_SPAN / 2) + r0;
                obase = (int)(s * CK_SPAN) - (int)e0;
            }
            const uint32_t incl = wave_scan_u32(n);
            if (r < nruns) { sRunStart[r] = start; sRunN[r] = n; sRunOff[r] = nmax_u + incl - n; sRunO[r] = obase; }
            nmax_u += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        wave_sync();
        const int nmax_all = (int)nmax_u;
        FDBGV(10, nmax_all);
        if (nmax_all < 4) continue;
        FDBG(2);

        // ---- 5a. threshold = (max_nmaxima+1)-th largest smoothed error; survivors in index order ----------------------------------
        int nsel = 0;
        const bool use_thr = nmax_all > a.max_nmaxima;
        if (nmax_all <= 64) { // one maximum per lane
            double myv = 0.0;
            int myo = -1;
            if (prefetched) {
                if (lane < nmax_all) { myv = cur.v; myo = sRunO[(uint32_t)lane < sRunN[0] ? 0 : 1] + (int)cur.pos; }
            } else {
                for (int r = 0; r < nruns; r++) {
                    const uint32_t off = sRunOff[r], n = sRunN[r];
                    if ((uint32_t)lane >= off && (uint32_t)lane < off + n) {
                        const uint32_t idx = sRunStart[r] + ((uint32_t)lane - off);
                        myv = mval[idx];
                        myo = sRunO[r] + (int)mpos[idx];
                    }
                }
                __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0); // (here, not inside the loop below, where it would also hold up the common path)
            }
            const bool has = lane < nmax_all;
            // the values are compared through their order-preserving integer image (one 64-bit integer compare against an f64 compare and
            // its wait states, twice per round of the loop): a cluster's smoothed errors are never NaN and never -0 (the sums start from +0)
            const unsigned long long myb = (unsigned long long)__double_as_longlong(myv);
            const unsigned long long myk = (myb >> 63) ? ~myb : (myb | (1ull << 63));
            unsigned long long thrk = 0ull;
            if (use_thr) {
                int gt = 0, ge = 0;
                for (int j = 0; j < nmax_all; j++) {
                    const unsigned long long u = (unsigned long long)readlane_i64((long long)myk, j);
                    gt += (u > myk) ? 1 : 0;
                    ge += (u >= myk) ? 1 : 0;
                }
                const unsigned long long hit = __ballot(has && gt <= a.max_nmaxima && a.max_nmaxima < ge);
                if (hit == 0ull) continue; // (values that do not order: garbage)
                thrk = (unsigned long long)readlane_i64((long long)myk, __builtin_ctzll(hit));
            }
            const bool keep = has && !(use_thr && myk <= thrk);
            const unsigned long long kb = __ballot(keep);
            nsel = __popcll(kb);
            if (keep) {
                const int pos = __popcll(kb & ((1ull << lane) - 1ull));
                if (pos < MAXSEL) sSelIdx[pos] = myo;
                if (myo < 0 || myo >= sz) sBad = 1;
            }
        } else {
            double thr = 0.0;
            if (use_thr) {
                double cur = HUGE_VAL;
                int remaining = a.max_nmaxima + 1;
                for (int round = 0; round <= a.max_nmaxima; round++) {
                    double m = -HUGE_VAL;
                    int cnt = 0;
                    for (int r = 0; r < nruns; r++) {
                        const uint32_t n = sRunN[r], st0 = sRunStart[r];
                        for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
                            const double v = mval[st0 + i];
                            if (v < cur) { if (v > m) { m = v; cnt = 1; } else if (v == m) cnt++; }
                        }
                    }
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        const double om = __shfl_xor(m, d, 64);
                        const int oc = __shfl_xor(cnt, d, 64);
                        if (om > m) { m = om; cnt = oc; } else if (om == m) cnt += oc;
                    }
                    if (cnt >= remaining) { thr = m; break; }
                    remaining -= cnt;
                    cur = m;
                }
            }
            for (int r = 0; r < nruns; r++) {
                const uint32_t n = sRunN[r], st0 = sRunStart[r];
                for (uint32_t ib = 0; ib < n; ib += 64) {
                    const uint32_t i = ib + (uint32_t)lane;
                    bool keep = false;
                    int myo = -1;
                    if (i < n) { keep = !(use_thr && mval[st0 + i] <= thr); myo = sRunO[r] + (int)mpos[st0 + i]; }
                    const unsigned long long kb = __ballot(keep);
                    if (keep) {
                        const int pos = nsel + __popcll(kb & ((1ull << lane) - 1ull));
                        if (pos < MAXSEL) sSelIdx[pos] = myo;
                        if (myo < 0 || myo >= sz) sBad = 1;
                    }
                    nsel += __popcll(kb);
                }
            }
        }
        wave_sync();
        FDBGV(11, nsel); FDBGV(12, sBad);
        if (nsel < 4 || nsel > MAXSEL || sBad) continue;
        FDBG(3);
        // (the lists are in position order and so are the runs: the survivors are sorted already)

        if (a.stop_after == 5) continue;
        // ---- 5b. moment prefix sums at the selected maxima: aligned block sums (wave scan) + the points beyond the last block ----------
        const uint32_t B0 = e0 >> 5, blk_l = (e1 - 1) >> 5; // blocks of the first and of the last point
        auto target_pos = [&](int t) -> uint32_t { return t < nsel ? e0 + (uint32_t)sSelIdx[t] : e1 - 1; };
        // four lanes per target, eight positions each: from the target's block boundary to the target (head: to the position before
        // the first point).  Their loads are issued first and are under way while the block sums are scanned
        const int pt = lane >> 2, ppart = lane & 3;
        const bool plive = pt < nsel || pt == T_TAIL || pt == T_HEAD;
        uint32_t pqs = 1, pqe = 0; // inclusive range; empty when pqe + 1 == pqs
        if (plive) {
            if (pt == T_HEAD) { pqs = B0 << 5; pqe = e0 - 1; }
            else { const uint32_t pm = target_pos(pt); pqs = pm & ~31u; pqe = pm; }
        }
        uint32_t pxy[8], pw[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t q = pqs + (uint32_t)(ppart * 8 + e);
            pxy[e] = 0; pw[e] = 0;
            if (plive && q <= pqe && q >= pqs) { pxy[e] = xy[q]; pw[e] = w16[q]; }
        }
        uint32_t sxy = 0, sw = 0; // the target itself (the exclusive sum is the inclusive one less its moments)
        if (pt < nsel && ppart == 0) { sxy = xy[pqe]; sw = w16[pqe]; }
        for (int i = tid; i < (MAXSEL + 2) * 6; i += NTH) sF6[i / 6][i % 6] = 0;
        wave_sync();
        {
            const int nfull = (int)(blk_l - B0); // blocks B0 .. blk_l - 1: what a target's prefix can need
            // lane t (< 13) knows where target t's prefix ends: the sums through the block before the target's
            const int my_kk = lane <= nsel ? (int)((target_pos(lane < nsel ? lane : T_TAIL) >> 5) - B0) - 1 : -2;
            long long carry[6] = {0, 0, 0, 0, 0, 0};
            for (int rb = 0; rb < nfull; rb += 64) {
                const int b = rb + lane;
                long long v[6] = {0, 0, 0, 0, 0, 0};
                if (b < nfull) {
                    const long long *src = blk + (size_t)(B0 + (uint32_t)b) * 6;
#pragma unroll
                    for (int q = 0; q < 6; q++) v[q] = src[q];
                }
#pragma unroll
                for (int q = 0; q < 6; q++) v[q] = (long long)wave_scan_u64((unsigned long long)v[q]) + carry[q];
                // lane t picks its target's prefix from the lane that holds it
                const int srcl = my_kk - rb;
                const bool mine = srcl >= 0 && srcl < 64;
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const long long got = __shfl(v[q], mine ? srcl : 0, 64);
                    if (mine) sF6[lane < nsel ? lane : T_TAIL][q] = got;
                }
#pragma unroll
                for (int q = 0; q < 6; q++) carry[q] = readlane_i64(v[q], 63);
            }
        }
        wave_sync(); // the block prefixes are in place
        {
            // eight points per lane: the first-order sums fit 32 bits (8 x 511 x 8192), a position outside the range holds weight 0
            uint32_t aMx = 0, aMy = 0, aW = 0;
            unsigned long long aMxx = 0, aMxy = 0, aMyy = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t X = ((pxy[e] >> 13) & 0x1FFFu) + 1, Y = (pxy[e] & 0x1FFFu) + 1, Wt = pw[e] & 0x1FFu; // as k_chunk forms them
                const uint32_t wx = Wt * X, wy = Wt * Y;
                aMx += wx; aMy += wy; aW += Wt;
                aMxx += (unsigned long long)wx * X; aMxy += (unsigned long long)wx * Y; aMyy += (unsigned long long)wy * Y;
            }
#pragma unroll
            for (int d = 2; d >= 1; d >>= 1) {
                aMx += __shfl_xor(aMx, d, 64); aMy += __shfl_xor(aMy, d, 64); aW += __shfl_xor(aW, d, 64);
                aMxx += __shfl_xor(aMxx, d, 64); aMxy += __shfl_xor(aMxy, d, 64); aMyy += __shfl_xor(aMyy, d, 64);
            }
            if (plive && ppart == 0) {
                const long long pv[6] = {(long long)aMx, (long long)aMy, (long long)aMxx, (long long)aMxy, (long long)aMyy, (long long)aW};
                if (pt < nsel) {
                    const M6 self = moments_of(sxy & 0x3FFFFFFu, sw & 0x1FFu);
                    const long long sv[6] = {self.Mx, self.My, self.Mxx, self.Mxy, self.Myy, self.W};
#pragma unroll
                    for (int q = 0; q < 6; q++) { const long long v = sF6[pt][q] + pv[q]; sSelI[pt][q] = v; sSelE[pt][q] = v - sv[q]; }
                } else {
#pragma unroll
                    for (int q = 0; q < 6; q++) sPart[pt][q] = pv[q];
                }
            }
        }
        wave_sync();
        M6 total; // the cluster's own points only
        {
            long long tv[6];
#pragma unroll
            for (int q = 0; q < 6; q++) tv[q] = sF6[T_TAIL][q] + sPart[T_TAIL][q] - sPart[T_HEAD][q];
            total.Mx = tv[0]; total.My = tv[1]; total.Mxx = tv[2]; total.Mxy = tv[3]; total.Myy = tv[4]; total.W = tv[5];
        }
        auto rangeM = [&](int sa, int sb, int *N) { // points from maximum sa to maximum sb inclusive, going forward
            M6 I = {sSelI[sb][0], sSelI[sb][1], sSelI[sb][2], sSelI[sb][3], sSelI[sb][4], sSelI[sb][5]};
            M6 E = {sSelE[sa][0], sSelE[sa][1], sSelE[sa][2], sSelE[sa][3], sSelE[sa][4], sSelE[sa][5]};
            int i0 = sSelIdx[sa], i1 = sSelIdx[sb];
            if (i0 < i1) { *N = i1 - i0 + 1; return m6_sub(I, E); }
            *N = sz - i0 + i1 + 1;
            return m6_add(m6_sub(total, E), I);
        };

        if (a.stop_after == 6) continue;
        // ---- 5c. one line fit per ordered pair of maxima; the 4-subset search is then table lookups ------------------------
        PairFit *sF = reinterpret_cast<PairFit *>(sPraw);
        {
            const int npf = nsel * (nsel - 1) / 2;
            for (int k = tid; k < npf; k += NTH) {
                const int pk = g_pair_table.v[k], sa = pk >> 4, sb = pk & 15;
                int Nf, Nw;
                const M6 mf = rangeM(sa, sb, &Nf), mw = rangeM(sb, sa, &Nw);
                double lp[4], lq[4], ef, msf, ew, msw;
                fit_line_m(mf, Nf, lp, &ef, &msf);
                fit_line_m(mw, Nw, lq, &ew, &msw);
                PairFit f;
                f.err = ef; f.mse = msf; f.nx = lp[2]; f.ny = lp[3];
                sF[sa * MAXSEL + sb] = f;
                f.err = ew; f.mse = msw; f.nx = lq[2]; f.ny = lq[3];
                sF[sb * MAXSEL + sa] = f;
            }
        }
        wave_sync();
        double best = HUGE_VAL;
        int bestc = 1 << 30;
        {
            const int ncomb = nsel * (nsel - 1) * (nsel - 2) * (nsel - 3) / 24;
            for (int cb = tid; cb < ncomb; cb += NTH) {
                const int pk = g_combo_table.v[cb];
                const int m0 = pk >> 12, m1 = (pk >> 8) & 15, m2 = (pk >> 4) & 15, m3 = pk & 15;
                const PairFit f01 = sF[m0 * MAXSEL + m1];
                if (f01.mse > a.max_mse) continue;
                const PairFit f12 = sF[m1 * MAXSEL + m2];
                if (f12.mse > a.max_mse) continue;
                double dp = f01.nx * f12.nx + f01.ny * f12.ny;
                if (fabs(dp) > a.cos_critical) continue;
                const PairFit f23 = sF[m2 * MAXSEL + m3];
                if (f23.mse > a.max_mse) continue;
                const PairFit f30 = sF[m3 * MAXSEL + m0];
                if (f30.mse > a.max_mse) continue;
                double e = f01.err + f12.err + f23.err + f30.err;
                if (e < best || (e == best && pk < bestc)) { best = e; bestc = pk; }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            double ob = __shfl_xor(best, d, 64);
            int oc = __shfl_xor(bestc, d, 64);
            if (ob < best || (ob == best && oc < bestc)) { best = ob; bestc = oc; }
        }
        if (best == HUGE_VAL) continue;
        FDBG(4);
        if (best / (double)sz >= a.max_mse) continue;
        FDBG(5);

        if (a.stop_after == 7) continue;
        // ---- 5d. the four sides' records; the corner phase runs when eight clusters wait (or the chunk ends) -----------------------
        if (tid < 4) {
            const int sel[4] = {(bestc >> 12) & 15, (bestc >> 8) & 15, (bestc >> 4) & 15, bestc & 15};
            int N;
            const int s0 = sel[tid], s1 = sel[(tid + 1) & 3];
            const M6 m = rangeM(s0, s1, &N);
            const PairFit pf = sF[s0 * MAXSEL + s1];
            SideRec r;
            r.Mx = m.Mx; r.My = m.My; r.W = m.W; r.nx = pf.nx; r.ny = pf.ny; r.mse = pf.mse;
            sB[nb][tid] = r;
        }
        if (tid == 0) { sBMeta[nb][0] = (uint32_t)frame; sBMeta[nb][1] = (uint32_t)reversed; sBMeta[nb][2] = rep0; sBMeta[nb][3] = rep1; }
        nb++;
        wave_sync();
        if (nb == BATCH) flush();
        } // clusters of the chunk
        if (nb) flush();
    }
}

// Register budget: four waves per SIMD (128 registers, nothing spilled).  A build for five (96 registers, 59 of them spilled) once
// looked 7 % faster and was WRONG (detections missing at 1920x1080 and 2448x2048, copies of one frame differing): two v_readlane
// reads of the chunk heads stood inside `if (lane < ...)` / `if (tid == 0)` blocks, where the lane they read may be switched off, and
// the spill code restores active lanes only.  With every v_readlane where all lanes are active that build is correct — and 3 %
// slower than this one (9.35 against 9.05 ms).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_tail(FitArgs a) { k_tail_body(a); }

// Gradient-magnitude weights of a whole batch in one streaming pass (1 B read, 2 B written per pixel): the fitter then needs
// ONE 2-byte gather per contour point instead of four byte gathers.  Four pixels per thread: three aligned 4-byte loads (row
// above, row, row below) plus the two bytes beside the group.  Image rows are padded to 16 bytes (staged frames and the decimated
// copy alike), so a group that starts inside a row is readable; the weight image's rows are qw entries long, so a width that is
// not a multiple of 4 ends in a partial group and its stores go out one by one.
__global__ __launch_bounds__(256) void k_weight_image(const uint8_t *__restrict__ qim, size_t qpitch, int qstride, int qw, int qh,
                                                      uint16_t *__restrict__ wimg) {
    // block = 64 groups of four pixels x 4 rows; the grid's y and z are row blocks and frames (no index arithmetic to undo)
    const int w4 = (qw + 3) >> 2;
    const int g = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (g >= w4 || iy >= qh) return;
    const size_t fr = blockIdx.z;
    const int ix = 4 * g;
    uint16_t out[4] = {1, 1, 1, 1};
    if (iy > 0 && iy + 1 < qh) {
        const uint8_t *row = qim + fr * qpitch + (size_t)iy * qstride + ix;
        const uint32_t mid = *reinterpret_cast<const uint32_t *>(row);
        const uint32_t up = *reinterpret_cast<const uint32_t *>(row - qstride), dn = *reinterpret_cast<const uint32_t *>(row + qstride);
        const int left = ix > 0 ? (int)row[-1] : 0, right = ix + 4 < qw ? (int)row[4] : 0;
        const int m[6] = {left, (int)(mid & 255u), (int)((mid >> 8) & 255u), (int)((mid >> 16) & 255u), (int)(mid >> 24), right};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = ix + k;
            if (x > 0 && x + 1 < qw) {
                const int gx = m[k + 2] - m[k];
                const int gy = (int)((dn >> (8 * k)) & 255u) - (int)((up >> (8 * k)) & 255u);
                out[k] = (uint16_t)(isqrt_u32((uint32_t)(gx * gx + gy * gy)) + 1);
            }
        }
    }
    uint16_t *dst = wimg + fr * ((size_t)qw * qh) + (size_t)iy * qw + ix;
    if ((qw & 3) == 0)
        *reinterpret_cast<uint2 *>(dst) = make_uint2((uint32_t)out[0] | ((uint32_t)out[1] << 16), (uint32_t)out[2] | ((uint32_t)out[3] << 16));
    else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (ix + k < qw) dst[k] = out[k];
    }
}

// builds the per-size-class work lists from the cluster tables: one workgroup per frame counts its clusters per class in
// LDS, reserves the four list ranges with four global atomics, then writes (the lists' internal order is irrelevant)
__global__ __launch_bounds__(1024) void k_classify(ck_stage_ws ws, int n, uint32_t *lists, uint32_t *list_counts, int list_cap, int split) {
    __shared__ uint32_t sCnt[CK_FIT_LISTS], sBase[CK_FIT_LISTS];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
    const uint32_t nc = counters[CK_CNT_CLUSTERS];
    const ck_cluster_t *cls = ws.d_clusters + (size_t)frame * ws.cluster_cap;
    auto class_of = [split](uint32_t c) { return (c <= 256 && (split & 2)) ? 7 : c <= 512 ? 0 : (c <= 1024 && (split & 1)) ? 6 : (c <= 2048 ? 1 : (c <= 4096 ? 2 : (c <= 8192 ? 3 : (c <= 16384 ? 4 : 5)))); };
    if (tid < CK_FIT_LISTS) sCnt[tid] = 0;
    __syncthreads();
    // (a cluster without points is one k_scan had no room for: skipped)
    for (uint32_t i = tid; i < nc; i += 1024) if (cls[i].count) { atomicAdd(&sCnt[class_of(cls[i].count)], 1u); atomicAdd(&sCnt[CK_FIT_CLASSES], 1u); }
    __syncthreads();
    if (tid < CK_FIT_LISTS) { sBase[tid] = sCnt[tid] ? atomicAdd(&list_counts[tid], sCnt[tid]) : 0u; }
    __syncthreads();
    if (tid < CK_FIT_LISTS) sCnt[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < nc; i += 1024) {
        if (!cls[i].count) continue;
        const int k = class_of(cls[i].count);
        const uint32_t pos = sBase[k] + atomicAdd(&sCnt[k], 1u);
        if (pos < (uint32_t)list_cap) lists[(size_t)k * list_cap + pos] = ((uint32_t)frame << 20) | i;
        // list CK_FIT_CLASSES: every cluster, whatever its size (the tail kernel of the split fit)
        const uint32_t pa = sBase[CK_FIT_CLASSES] + atomicAdd(&sCnt[CK_FIT_CLASSES], 1u);
        if (pa < (uint32_t)list_cap) lists[(size_t)CK_FIT_CLASSES * list_cap + pa] = ((uint32_t)frame << 20) | i;
    }
}
} // namespace

#ifdef CK_FLAT_DEBUG
extern "C" int ck_flat_debug_read(unsigned int *out, int reset) {
    unsigned int z[32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flat_dbg), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_flat_dbg), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
#ifdef CK_FIT_PROFILE
extern "C" int ck_fit_profile_read(unsigned long long *out, int reset) {
    unsigned long long z[3][16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fit_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fit_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif

int ck_launch_fit_quads(ck_handle *h, const uint8_t *qframes, int qstride, size_t qpitch, const uint8_t *frames, int stride,
                        size_t pitch, int n) {
    ck_stage_ws &ws = h->ws;
    if (n > 4095 || ws.cluster_cap > (1 << 20)) return CK_EINVAL;
    // work lists, their counts and the dequeue heads live in the fit scratch; counts and heads were zeroed with the cluster
    // tables (k_clusters.hip: k_clear)
    const ck_fit_layout fl = ck_fit_scratch_layout(ws, h->cfg.max_batch);
    const int list_cap = fl.list_cap;
    uint32_t *lists = fl.lists, *list_counts = fl.list_counts, *heads = fl.heads;
    static const int fit_split = CK_KNOB("CK_FIT_SPLIT", 3); // (diagnostics: bit 0 = 513..1024 points have their own class, bit 1 = up to 256 points have)
    // A small call (one frame per call is the reference's own pattern) gives every class a handful of workgroups whose time is
    // one cluster's dependency chain: the classes then run side by side on their own streams instead of one after the other —
    // and without the two youngest classes, which exist for throughput (more clusters in flight per CU) and would only add two
    // more chains to the handle's lane (0.64 -> 0.70 ms per 1280x800 frame at quad_decimate 2).
    static const int force_par = CK_KNOB("CK_FIT_PAR", 0);
    // (measured at 1280x800: side by side wins up to 16 frames at quad_decimate 1, up to 32 at 2, where the clusters are fewer)
    const bool side_by_side = n <= (h->cfg.quad_decimate > 1 ? 2 * CK_FIT_PARALLEL_MAX_FRAMES : CK_FIT_PARALLEL_MAX_FRAMES) || force_par;
    const int split = side_by_side ? 0 : fit_split;
    hipLaunchKernelGGL(k_classify, dim3((unsigned)n), dim3(1024), 0, h->stream, ws, n, lists, list_counts, list_cap, split);
    FitArgs a;
    a.qim = qframes; a.qw = h->qw; a.qh = h->qh; a.qstride = qstride; a.qpitch = qpitch;
    a.wimg = ws.d_wimg;
    auto launch_wimg = [&](hipStream_t st) {
        const int w4 = (h->qw + 3) / 4;
        hipLaunchKernelGGL(k_weight_image, dim3((unsigned)((w4 + 63) / 64), (unsigned)((h->qh + 3) / 4), (unsigned)n), dim3(256), 0, st, qframes,
                           qpitch, qstride, h->qw, h->qh, ws.d_wimg);
    };
    // The split fit needs the weights in its second kernel only: the weight image (a streaming kernel, bound by HBM) then runs on the
    // second side stream beside the first kernels (bound by their sort) instead of before them (CK_FIT_WIMG_ASIDE=0: as before)
    static const int flat_env0 = CK_KNOB("CK_FIT_FLAT", 1);
    static const int wimg_aside_env = CK_KNOB("CK_FIT_WIMG_ASIDE", 1);
    static const int tails_aside_ok0 = CK_KNOB("CK_FIT_TAILS_ASIDE", 1) && ck_streams_wanted() < 2;
    const bool flat0 = flat_env0 >= 2 || (flat_env0 == 1 && !side_by_side && (size_t)n * (size_t)h->qw * (size_t)h->qh >= ((size_t)100 << 20));
    const bool wimg_aside = flat0 && !side_by_side && tails_aside_ok0 && wimg_aside_env;
    if (!wimg_aside) launch_wimg(h->stream);
    a.im = frames; a.w = h->w; a.h = h->h; a.stride = stride; a.pitch = pitch;
    a.decimate = h->cfg.quad_decimate; a.refine = h->cfg.refine_edges; a.max_nmaxima = h->cfg.max_nmaxima;
    a.cos_critical = h->cfg.cos_critical_rad; a.max_mse = h->cfg.max_line_fit_mse;
    a.normal_ok = 0; a.reversed_ok = 0; a.min_tag_width = 1 << 30;
    for (int f = 0; f < h->cfg.n_families; f++) {
        const ck_family_t *fam = h->cfg.families[f];
        if (fam->width_at_border < a.min_tag_width) a.min_tag_width = fam->width_at_border;
        if (fam->reversed_border) a.reversed_ok = 1; else a.normal_ok = 1;
    }
    a.min_tag_width /= h->cfg.quad_decimate;
    if (a.min_tag_width < 3) a.min_tag_width = 3;
    a.ws = ws; a.list_cap = list_cap;
    a.stop_after = CK_KNOB("CK_FIT_STOP_AFTER", 99); // (read per call)
    a.guided = CK_KNOB("CK_FIT_GUIDED", 1);
    // chunk sizes: 512 points for the multi-wave classes (fewer scans and barriers per point, 9 % less halo work); their LDS
    // then sits at the occupancy steps — 52.9 KB (3 workgroups/CU), 80.5 KB (2/CU), ≈ 124 KB and ≈ 163 KB (1/CU each)
    int cus = 256;
    // three lanes of similar length for a typical frame: {S, M1} on the handle's stream, {L1} and {M2, L2} on the side streams
    // (more streams than that end up sharing hardware queues and wait for each other anyway)
    hipStream_t cs[CK_FIT_CLASSES] = {h->stream, h->stream, h->stream, h->stream, h->stream, h->stream, h->stream, h->stream};
    // A batch keeps one lane, except for the three classes with ONE 512-thread workgroup per CU (8193..16384 points: 2.2 clusters
    // per workgroup on average, so a quarter of the CUs do a third cluster while the rest idle; the largest; 4097..8192): one after
    // the other on a side stream, started first, their tails and barrier waits run under the small classes instead of before
    // them.  (Measured: the 2049..4096 class there as well, or the three on two side streams, is slower than this.)
    static const int tails_aside_ok = tails_aside_ok0;
    const bool tails_aside = !side_by_side && tails_aside_ok;
    if (side_by_side) { cs[3] = h->fit_stream[0]; cs[2] = h->fit_stream[1]; cs[4] = h->fit_stream[1]; }
    if (tails_aside) { cs[4] = h->fit_stream[0]; cs[5] = h->fit_stream[0]; cs[3] = h->fit_stream[0]; }
    // (diagnostics: CK_FIT_ASIDE: bit c set = class c on side stream 0, bit 8 + c = on side stream 1.  Split fit, measured: the three
    // largest classes aside (0x38) 8.59 ms · with the 2049-4096 class 8.68 · with the 1025-2048 class too 8.65 · the 4097-8192 class back
    // on the handle's stream 8.57 · or on the second side stream 8.52 · ...: all within the noise of one box)
    static const int aside_env = CK_KNOB0("CK_FIT_ASIDE", -1);
    if (tails_aside && aside_env >= 0)
        for (int c = 0; c < CK_FIT_CLASSES; c++) cs[c] = ((aside_env >> c) & 1) ? h->fit_stream[0] : (((aside_env >> (8 + c)) & 1) ? h->fit_stream[1] : h->stream);
    if (side_by_side || tails_aside) {
        CK_HIP(hipEventRecord(h->ev_fit_fork, h->stream));
        for (int k = 0; k < CK_FIT_SIDE_STREAMS; k++) CK_HIP(hipStreamWaitEvent(h->fit_stream[k], h->ev_fit_fork, 0));
        if (wimg_aside) launch_wimg(h->fit_stream[1]); // (joined with the side streams before k_chunk)
    }
    static const int gk_env = CK_KNOB("CK_FIT_GK", 0); // (experiment: bit c set = class c keeps its keys in global memory: small LDS, more workgroups per CU)
    // (not for a call whose classes run side by side: the classes with their keys in global memory index ONE scratch slice by workgroup)
    const int gk_mask = ws.d_hscratch && !side_by_side ? gk_env : 0;
    static const int skip_mask = CK_KNOB("CK_FIT_SKIP", 0); // (diagnostics: bit c set = class c is not launched)
    // The split fit (k_seq per class -> k_chunk over all positions -> k_tail over all clusters): CK_FIT_FLAT = 0 never, 2 always,
    // 1 (default): for calls that run their classes one after the other AND bring enough pixels — its three stages each ramp a
    // persistent grid up and down, which a quarter-size batch notices (1280x800 x 256 at quad_decimate 2: 2.85 against 2.52 ms
    // unsplit; at full resolution 9.75 against 9.94, 1920x1080 21.4 against 22.7, 2448x2048 x 128 27.8 against 31.5)
    static const int flat_env = flat_env0;
    const bool flat = flat_env >= 2 || (flat_env == 1 && !side_by_side && (size_t)n * (size_t)h->qw * (size_t)h->qh >= ((size_t)100 << 20));
    static const int seq_alt = CK_KNOB("CK_SEQ_ALT", 0); // (diagnostics: bit c = class c of k_seq with a tighter register budget)
    // (diagnostics: workgroups per CU of k_seq's three smallest classes)
    static const int seq_wgs0 = CK_KNOB("CK_SEQ_WGS0", 16), seq_wgs7 = CK_KNOB("CK_SEQ_WGS7", 16), seq_wgs6 = CK_KNOB("CK_SEQ_WGS6", 8);
    auto launch_split = [&](int c) {
        switch (c) {
        // (tighter register budgets for the four classes below — six / eight / five / five waves per SIMD — measured: no difference)
        case 0: hipLaunchKernelGGL((k_seq<64, 512, true, 4, false>), dim3((unsigned)(cus * seq_wgs0)), dim3(64), 0, cs[c], a); break;
        case 7: hipLaunchKernelGGL((k_seq<64, 256, true, 4, false>), dim3((unsigned)(cus * seq_wgs7)), dim3(64), 0, cs[c], a); break;
        case 6: hipLaunchKernelGGL((k_seq<128, 1024, true, 4, false>), dim3((unsigned)(cus * seq_wgs6)), dim3(128), 0, cs[c], a); break;
        case 1: hipLaunchKernelGGL((k_seq<256, 2048, true, 4, false>), dim3((unsigned)(cus * 4)), dim3(256), 0, cs[c], a); break;
        case 2: if (seq_alt & 16) hipLaunchKernelGGL((k_seq<256, 4096, true, 2, false>), dim3((unsigned)(cus * 2)), dim3(256), 0, cs[c], a);
                else hipLaunchKernelGGL((k_seq<256, 4096, true, 3, false>), dim3((unsigned)(cus * 3)), dim3(256), 0, cs[c], a); // (three workgroups per CU at 168 registers: 8.58 against 8.78 ms with two at 256)
                break;
        case 3: hipLaunchKernelGGL((k_seq<512, 8192, true, 2, false>), dim3((unsigned)cus), dim3(512), 0, cs[c], a); break;
        case 4: hipLaunchKernelGGL((k_seq<512, 16384, false, 2, false>), dim3((unsigned)cus), dim3(512), 0, cs[c], a); break;
        default:
            if (ws.d_hscratch) hipLaunchKernelGGL((k_seq<512, CK_HUGE_CAP, false, 2, true>), dim3((unsigned)CK_HUGE_WGS), dim3(512), 0, cs[c], a);
            break;
        }
    };
    auto launch = [&](int c) {
        if ((skip_mask >> c) & 1) return;
        a.list = lists + (size_t)c * list_cap; a.list_count = list_counts + c; a.head = heads + c;
        if (flat) { launch_split(c); return; }
        switch (c) {
        case 0: { static const int s_wgs = CK_KNOB("CK_FIT_S_WGS", 12); // (diagnostics: workgroups per CU of the small class)
            hipLaunchKernelGGL((k_fit<64, 512, 64, true, 3>), dim3((unsigned)(cus * s_wgs)), dim3(64), 0, cs[c], a); break; }
        case 7: hipLaunchKernelGGL((k_fit<64, 256, 64, true, 4>), dim3((unsigned)(cus * 16)), dim3(64), 0, cs[c], a); break;
        case 6: hipLaunchKernelGGL((k_fit<128, 1024, 128, true, 4>), dim3((unsigned)(cus * 7)), dim3(128), 0, cs[c], a); break;
        case 1: hipLaunchKernelGGL((k_fit<256, 2048, 224, true, 4>), dim3((unsigned)(cus * 4)), dim3(256), 0, cs[c], a); break;
        case 2: hipLaunchKernelGGL((k_fit<256, 4096, 512, true, 2>), dim3((unsigned)(cus * 2)), dim3(256), 0, cs[c], a); break;
        case 3:
            if (gk_mask & 8) hipLaunchKernelGGL((k_fit<512, 16384, 512, false, 4, true>), dim3((unsigned)CK_HUGE_WGS), dim3(512), 0, cs[c], a);
            else hipLaunchKernelGGL((k_fit<512, 8192, 896, true, 2>), dim3((unsigned)cus), dim3(512), 0, cs[c], a);
            break;
        case 4:
            if (gk_mask & 16) hipLaunchKernelGGL((k_fit<512, 16384, 512, false, 4, true>), dim3((unsigned)CK_HUGE_WGS), dim3(512), 0, cs[c], a);
            else hipLaunchKernelGGL((k_fit<512, 16384, 512, false, 2>), dim3((unsigned)cus), dim3(512), 0, cs[c], a);
            break;
        default: // more than 16384 points: only frames with more than 2730 pixels of half-perimeter have the buffer (and can have such clusters)
            if (ws.d_hscratch) hipLaunchKernelGGL((k_fit<512, CK_HUGE_CAP, 896, false, 2, true>), dim3((unsigned)CK_HUGE_WGS), dim3(512), 0, cs[c], a);
            break;
        }
    };
    if (side_by_side) { launch(3); launch(2); launch(4); launch(0); launch(1); launch(5); } // the side lanes first, then the handle's own
    else if (tails_aside) { launch(4); launch(5); launch(3); launch(7); launch(0); launch(6); launch(1); launch(2); }
    else { launch(7); launch(0); launch(6); for (int c = 1; c < 6; c++) launch(c); }
    if (side_by_side || tails_aside)
        for (int k = 0; k < CK_FIT_SIDE_STREAMS; k++) {
            CK_HIP(hipEventRecord(h->ev_fit_join[k], h->fit_stream[k]));
            CK_HIP(hipStreamWaitEvent(h->stream, h->ev_fit_join[k], 0));
        }
    if (flat) {
        // spans per frame and workgroups that share them: a batch gives every workgroup a few spans, a short call one each
        const unsigned spans = (unsigned)(ws.ext_cap / CK_SPAN);
        static const int chunk_wgs = CK_KNOB("CK_CHUNK_WGS", 64); // (diagnostics: k_chunk workgroups per CU over the batch)
        static const int tail_wgs = CK_KNOB("CK_TAIL_WGS", 16);    // (diagnostics: k_tail workgroups per CU)
        unsigned gx = (unsigned)((cus * chunk_wgs + n - 1) / n);
        if (gx < 8) gx = 8;
        if (gx > spans) gx = spans;
        // (k_chunk's register count capped at 96 / 80: no difference — three workgroups per CU either way, its LDS decides)
        if (a.stop_after > 3) hipLaunchKernelGGL(k_chunk, dim3(gx, (unsigned)n), dim3(256), 0, h->stream, ws, h->qw, h->qh);
        a.list = lists + (size_t)CK_FIT_CLASSES * list_cap; a.list_count = list_counts + CK_FIT_CLASSES; a.head = heads + CK_FIT_CLASSES;
        hipLaunchKernelGGL(k_tail, dim3((unsigned)(cus * tail_wgs)), dim3(64), 0, h->stream, a);
    }
    CK_HIP(hipGetLastError());
    return CK_OK;
}
