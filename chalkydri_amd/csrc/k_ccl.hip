// k_ccl.hip — adaptive threshold + union-find segmentation for gfx950 (the stage scored against the HBM
// roofline: 7 algorithmic bytes per pixel = 1 R image + 1 W thresholded + 1 R thresholded + 4 W label).
//
// Replaces, in the reference's production path, the threshold and connected-component stages of the external
// AprilTag-3 detector reached at crates/apriltags/src/lib.rs:301; the connectivity rule is the one CAT spells
// out in crates/chalkydri-apriltags/src/lib.rs:501-549 (4-connected black, 8-connected white, origin columns
// 1..w-2).
//
// Structure (one launch each, batched over frames):
//   k_tile   one workgroup per 64x128 tile: coalesced 16-byte loads of the tile + 4-px halo into LDS, 4x4
//            tile min/max, 3x3 dilation, tri-state threshold (written once, 16 B/lane), bit-parallel
//            union-find in LDS (one lane per 64-pixel row segment and colour, runs found with clz/ctz on u64
//            masks, u16 parents, saturating u8 sizes: 26 KB of LDS, six workgroups per CU), labels written once
//            (64 B/lane), ring-touching roots appended per wave.  HBM traffic 1.2 R + 1 W + 4 W bytes per pixel.
//   k_merge  one thread per tile-ring pixel: links reduced to pairs of tile-local roots, de-duplicated per tile in an
//            LDS hash set, joined with atomicMin on the label words of the roots involved.
//   k_roots  flattens the entries of ring-touching roots and accumulates their sizes into csize[].
// No full-frame relabel pass exists: interior components are final when k_tile writes them; ring-touching
// ones are resolved by consumers with one extra hop (label word format in ck_internal.h).
#include <stdlib.h>

#include "ck_internal.h"

namespace {

constexpr int TW = CK_TW, TH = CK_TH, NT = 256; // NT: merge / utility kernels
constexpr int SEGW = 64;                             // pixels per row segment = width of a lane's bit masks
using mask_t = uint64_t;                             // (32-bit segments with twice the lanes were measured: slightly slower)
constexpr int KNT = TH * (TW / SEGW) * 2;            // k_tile: one lane per (row segment, colour)
constexpr int IMG_PITCH = 160;          // 12 pad | 4 halo | 128 tile | 4 halo | 12 pad
constexpr int IMG_ROWS = TH + 8;
constexpr int T4X = TW / 4 + 2, T4Y = TH / 4 + 2;
constexpr int NSEG = TW / SEGW;
constexpr int MPIECES = SEGW / 16;                   // 16-pixel chunks per mask
constexpr mask_t MALL = ~mask_t(0), MONE = 1;
__device__ __forceinline__ int mctz(mask_t v) { return sizeof(mask_t) == 8 ? __builtin_ctzll(v) : __builtin_ctz((uint32_t)v); }
__device__ __forceinline__ int mclz(mask_t v) { return sizeof(mask_t) == 8 ? __builtin_clzll(v) : __builtin_clz((uint32_t)v); }
__device__ __forceinline__ int mpopc(mask_t v) { return sizeof(mask_t) == 8 ? __popcll(v) : __popc((uint32_t)v); }

// LDS of k_tile, 26 KB, so that six workgroups (24 waves) share a CU — the union-find is bound by LDS round-trip
// latency, and resident waves are what hides it:
//   parent  u16[TH*TW]  tile-local node index of the parent (8192 nodes fit 13 bits); while the threshold is computed
//                       the same bytes hold the staged image, the 4x4 min/max and the per-4x4 threshold words
//   size    u8[TH*TW]   at the roots: pixel count saturating at 127 (only "< min_component_px" is ever asked, and
//                       ck_create refuses min_component_px > 127) | bit 7 = component touches the tile ring
//   masks   mask_t[TH][NSEG][2]
constexpr int OFF_PARENT = 0;                              // u16[TH*TW] = 16384
constexpr int OFF_IMG = 0;                                 // IMG_ROWS*IMG_PITCH = 11520
constexpr int OFF_MINMAX = OFF_IMG + IMG_ROWS * IMG_PITCH; // u16[T4Y*T4X] (1224 -> 1280)
constexpr int OFF_THR = OFF_MINMAX + 1280;                 // u16[(TH/4)*(TW/4)] = 1024
constexpr int OFF_SIZE = TH * TW * 2;                      // u8[TH*TW] = 8192
constexpr int OFF_MASK = OFF_SIZE + TH * TW;               // mask_t[TH][NSEG][2]
constexpr int LDS_BYTES = OFF_MASK + TH * NSEG * 2 * (int)sizeof(mask_t);
constexpr uint32_t SIZE_SAT = 127;
static_assert(OFF_THR + 1024 <= OFF_SIZE, "threshold scratch must fit in the parent array");
static_assert(LDS_BYTES + 64 <= 27136, "keep six workgroups per CU");
static_assert(TH * NSEG * 2 == KNT, "k_tile thread mapping: colour x row x segment");
static_assert(TH * TW <= 65536, "node indices are u16");
static_assert(TW == 128, "label pass splits a node index with >> 7 / & 127");

#ifdef CK_TILE_PROFILE
#define TCNT_ARG , uint32_t &tcnt
#define TCNT_PASS , tcnt
#define TCNT_INC ++tcnt
#else
#define TCNT_ARG
#define TCNT_PASS
#define TCNT_INC
#endif
// find with path halving.  Plain stores race with the min-hooks of lds_union, but every value ever written to p[a]
// is an ancestor of a, so the forest stays valid (a lost hook is re-issued by its own union).
// two halving finds walked in lockstep: both chains have a read in flight at every step (the kernel is bound by LDS
// round-trip latency, not LDS bandwidth)
// (plain loads behind compiler barriers, not volatile ones: a volatile read is waited for before the next is issued, which
// would put the two chains' reads one after the other)
__device__ __forceinline__ void lds_find2(uint16_t *p, uint32_t &a, uint32_t &b TCNT_ARG) {
    for (;;) {
        TCNT_INC;
        __asm__ volatile("" ::: "memory");
        uint32_t na = p[a], nb = p[b];
        bool da = (na == a), db = (nb == b);
        if (da && db) return;
        __asm__ volatile("" ::: "memory");
        uint32_t ga = p[na], gb = p[nb];
        if (!da) { if (ga != na) p[a] = (uint16_t)ga; a = ga; }
        if (!db) { if (gb != nb) p[b] = (uint16_t)gb; b = gb; }
    }
}
// atomic min on one u16 entry (LDS has no 16-bit atomics): compare-and-swap on the word that holds it.  Returns the
// entry's previous value.  A concurrent halving store to the other half only makes the swap fail and retry.
__device__ __forceinline__ uint32_t lds_min16(uint16_t *p, uint32_t idx, uint32_t val) {
    uint32_t *wp = reinterpret_cast<uint32_t *>(p) + (idx >> 1);
    const uint32_t sh = (idx & 1u) * 16u;
    uint32_t wv = *reinterpret_cast<volatile uint32_t *>(wp);
    for (;;) {
        const uint32_t cur = (wv >> sh) & 0xFFFFu;
        if (cur <= val) return cur;
        const uint32_t prev = atomicCAS(wp, wv, (wv & ~(0xFFFFu << sh)) | (val << sh));
        if (prev == wv) return cur;
        wv = prev;
    }
}
// root = smaller index
__device__ __forceinline__ void lds_union(uint16_t *p, uint32_t a, uint32_t b TCNT_ARG) {
    for (;;) {
        lds_find2(p, a, b TCNT_PASS);
        if (a == b) return;
        if (a < b) { uint32_t t = a; a = b; b = t; }
        uint32_t old = lds_min16(p, a, b);
        if (old == a) return;
        a = old;
    }
}
// adds `add` pixels (and the ring flag) to the size byte of a root: saturating, and free once nothing would change —
// the one huge component of a noisy tile saturates after a few adds and every later run only reads
__device__ __forceinline__ void lds_size_add(uint8_t *sz, uint32_t root, uint32_t add, bool ring) {
    uint32_t *wp = reinterpret_cast<uint32_t *>(sz) + (root >> 2);
    const uint32_t sh = (root & 3u) * 8u;
    uint32_t wv = *reinterpret_cast<volatile uint32_t *>(wp);
    for (;;) {
        const uint32_t cur = (wv >> sh) & 0xFFu;
        uint32_t cnt = (cur & 0x7Fu) + add;
        cnt = cnt > SIZE_SAT ? SIZE_SAT : cnt;
        const uint32_t nv = cnt | (cur & 0x80u) | (ring ? 0x80u : 0u);
        if (nv == cur) return;
        const uint32_t prev = atomicCAS(wp, wv, (wv & ~(0xFFu << sh)) | (nv << sh));
        if (prev == wv) return;
        wv = prev;
    }
}
// start bit of the run that contains bit i, given the run-start mask S (bit i's run start is <= i)
__device__ __forceinline__ int run_start(mask_t S, int i) {
    return SEGW - 1 - mclz(S & (MALL >> (SEGW - 1 - i)));
}
__device__ __forceinline__ mask_t origin_mask(int x0, int w) {
    mask_t O = MALL;
    if (x0 == 0) O &= ~MONE;
    int last = (w - 1) - x0;
    if (last >= 0 && last < SEGW) O &= ~(MONE << last);
    return O;
}
// gathers bit 7 of each byte of v into a nibble (bit k = byte k)
__device__ __forceinline__ uint32_t msb_nibble(uint32_t v) {
    return (((v >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xFu;
}

// Diagnostic build only (-DCK_TILE_PROFILE): per-phase cycle totals of k_tile in a buffer of their own.
#ifdef CK_TILE_PROFILE
__device__ unsigned long long g_tile_prof[16];
__device__ unsigned long long g_tile_prof2[8]; // wave 0's cycles inside the union phase: setup, adoption, barrier, atomic unions, barrier
#define TPROF_DECL unsigned long long tp0 = __builtin_readcyclecounter()
#define TPROF(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_tile_prof[k], t_ - tp0); tp0 = t_; } while (0)
#else
#define TPROF_DECL
#define TPROF(k)
#endif

// PRE = false: `frames` are gray images and the tri-state threshold is computed here;
// PRE = true : `frames` already hold a tri-state map (0 / 127 / 255), e.g. CAT's class map, and only the
//              segmentation runs (the map is copied through to `thresh` for the merge kernel).
template <bool PRE>
__global__ __launch_bounds__(KNT) void k_tile(const uint8_t *__restrict__ frames, size_t frame_pitch, int stride,
                                             int w, int h, int tiles_x, int tiles_y, int min_diff, int min_comp,
                                             uint8_t *__restrict__ thresh, uint32_t *__restrict__ labels,
                                             ck_border_root *__restrict__ broots,
                                             uint32_t *__restrict__ broot_count, int broot_cap, uint32_t *__restrict__ csize, int stop_after) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int frame = blockIdx.x / tiles, tile = blockIdx.x - frame * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    const uint8_t *img = frames + (size_t)frame * frame_pitch;
    const size_t fbase = (size_t)frame * (size_t)w * (size_t)h;
    uint16_t *parent = reinterpret_cast<uint16_t *>(lds + OFF_PARENT);
    uint8_t *size8 = lds + OFF_SIZE;

    TPROF_DECL;
    // ---- P0: stage the tile and its 4-pixel halo ----------------------------------------------------------
    for (int item = tid; item < IMG_ROWS * 8; item += KNT) {
        int r = item >> 3, c = item & 7;
        int gy = ty0 - 4 + r, gx = tx0 + 16 * c;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gy >= 0 && gy < h) {
            const uint8_t *src = img + (size_t)gy * stride + gx;
            if (gx + 16 <= w) v = *reinterpret_cast<const uint4 *>(src);
            else {
                if (gx + 4 <= w) v.x = *reinterpret_cast<const uint32_t *>(src);
                if (gx + 8 <= w) v.y = *reinterpret_cast<const uint32_t *>(src + 4);
                if (gx + 12 <= w) v.z = *reinterpret_cast<const uint32_t *>(src + 8);
            }
        }
        *reinterpret_cast<uint4 *>(lds + OFF_IMG + r * IMG_PITCH + 16 + 16 * c) = v;
    }
    for (int item = tid; item < IMG_ROWS * 2; item += KNT) {
        int r = item >> 1, side = item & 1;
        int gy = ty0 - 4 + r, gx = side ? tx0 + TW : tx0 - 4;
        uint32_t v = 0;
        if (gy >= 0 && gy < h && gx >= 0 && gx + 4 <= w)
            v = *reinterpret_cast<const uint32_t *>(img + (size_t)gy * stride + gx);
        *reinterpret_cast<uint32_t *>(lds + OFF_IMG + r * IMG_PITCH + (side ? 16 + TW : 12)) = v;
    }
    __syncthreads();
    TPROF(0);

    if (stop_after == 0) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P1: min/max of every 4x4 tile of the staged region -------------------------------------------
    uint16_t *minmax = reinterpret_cast<uint16_t *>(lds + OFF_MINMAX);
    const int w4 = w >> 2, h4 = h >> 2;
    if (!PRE)
    for (int item = tid; item < T4Y * T4X; item += KNT) {
        int i = item / T4X, j = item - i * T4X;
        int g4x = (tx0 >> 2) - 1 + j, g4y = (ty0 >> 2) - 1 + i;
        uint32_t mn = 255, mx = 0;
        if (g4x >= 0 && g4x < w4 && g4y >= 0 && g4y < h4) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                uint32_t d = *reinterpret_cast<const uint32_t *>(lds + OFF_IMG + (4 * i + rr) * IMG_PITCH + 12 + 4 * j);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t v = (d >> (8 * b)) & 255u;
                    mn = min(mn, v); mx = max(mx, v);
                }
            }
        }
        minmax[item] = (uint16_t)(mn | (mx << 8)); // outside the frame: (255,0) is neutral for the dilation
    }
    __syncthreads();

    // ---- P2: 3x3 dilation -> per-4x4-tile threshold word (bit 8 = low contrast) ---------------------------
    uint16_t *thr = reinterpret_cast<uint16_t *>(lds + OFF_THR);
    if (!PRE)
    for (int item = tid; item < (TH / 4) * (TW / 4); item += KNT) {
        int i = item / (TW / 4), j = item - i * (TW / 4);
        uint32_t mn = 255, mx = 0;
#pragma unroll
        for (int di = 0; di < 3; di++)
#pragma unroll
            for (int dj = 0; dj < 3; dj++) {
                uint32_t m = minmax[(i + di) * T4X + j + dj];
                mn = min(mn, m & 255u); mx = max(mx, m >> 8);
            }
        int diff = (int)mx - (int)mn;
        thr[item] = (uint16_t)((diff < min_diff) ? 0x100u : (mn + (uint32_t)(diff >> 1)));
    }
    __syncthreads();
    TPROF(1);

    if (stop_after == 1) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P3: threshold 16 pixels per item, write them, build the per-row colour masks ----------------------
    uint16_t *mask16 = reinterpret_cast<uint16_t *>(lds + OFF_MASK); // [r][seg][colour][piece]
    for (int item = tid; item < TH * 8; item += KNT) {
        int r = item >> 3, c = item & 7;
        int gy = ty0 + r, gx = tx0 + 16 * c;
        uint4 px = *reinterpret_cast<const uint4 *>(lds + OFF_IMG + (r + 4) * IMG_PITCH + 16 + 16 * c);
        uint32_t in[4] = {px.x, px.y, px.z, px.w}, out[4];
        uint32_t wbits = 0, bbits = 0;
        // the one to three rows below the last whole 4x4 tile take that tile's threshold (the oracle's ragged-edge rule);
        // ck_create refuses the heights whose last whole tile row belongs to the workgroup above
        const int r4 = min((ty0 + r) >> 2, h4 - 1) - (ty0 >> 2);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t tw_ = PRE ? 0u : thr[r4 * (TW / 4) + 4 * c + k];
            uint32_t o;
            if (gy >= h || gx + 4 * k >= w) o = 0x7F7F7F7Fu;      // outside the frame: no colour
            else if (PRE) o = in[k];
            else if (tw_ & 0x100u) o = 0x7F7F7F7Fu;
            else {
                o = 0;
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (((in[k] >> (8 * b)) & 255u) > tw_) o |= 0xFFu << (8 * b);
            }
            out[k] = o;
            wbits |= msb_nibble(o) << (4 * k);                     // 255 -> bit 7 set
            bbits |= msb_nibble(~(o << 7)) << (4 * k);             // 0 -> bit 0 clear (127 and 255 have it set)
        }
        if (gy < h && gx < w) {
            uint8_t *dst = thresh + fbase + (size_t)gy * w + gx;
            if (gx + 16 <= w && (w & 15) == 0) *reinterpret_cast<uint4 *>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
            else
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (gx + 4 * k < w) *reinterpret_cast<uint32_t *>(dst + 4 * k) = out[k];
        }
        int seg = c / MPIECES, piece = c % MPIECES;
        mask16[((r * NSEG + seg) * 2 + 0) * MPIECES + piece] = (uint16_t)wbits;
        mask16[((r * NSEG + seg) * 2 + 1) * MPIECES + piece] = (uint16_t)bbits;
    }
    __syncthreads();
    TPROF(2);

    if (stop_after == 2) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P4: the image scratch is dead; it becomes the parent array (parent[i] = i), sizes start at zero -------------
    for (int i = tid * 8; i < TH * TW; i += KNT * 8) {
        const uint32_t lo = (uint32_t)i | ((uint32_t)(i + 1) << 16);
        *reinterpret_cast<uint4 *>(&parent[i]) = make_uint4(lo, lo + 0x00020002u, lo + 0x00040004u, lo + 0x00060006u);
    }
    // (the size bytes are zeroed after P5b: until then their 8 KB hold the waves' link pools)

    // ---- P5: unions.  thread = (colour, row, segment) --------------------------------------------------------------
    const mask_t *masks = reinterpret_cast<const mask_t *>(lds + OFF_MASK);
    const int color = tid / (KNT / 2), sitem = tid % (KNT / 2);
    const int r = sitem / NSEG, s = sitem - r * NSEG;
    const int x0 = tx0 + SEGW * s;
    const mask_t M = masks[(r * NSEG + s) * 2 + color];
    const mask_t O = origin_mask(x0, w);
    const mask_t S = M & ~((M << 1) & O); // segment-local run starts
    const uint32_t base = (uint32_t)(r * TW + SEGW * s);
    // Events of this segment: every link from one of its runs to a run that comes earlier in scan order
    //   hleft : bit 0 continues the run that ends the segment on the left
    //   Ev    : vertical links (first column of every stretch where this row and the row above overlap)
    //   DL/DR : white only, diagonal links not already implied by a vertical one
    mask_t Ev = 0, DL = 0, DR = 0, U = 0, Su = 0, Ul = 0;
    uint32_t left_node = 0;
    bool hleft = false;
    if (M) {
        if (s > 0 && (M & O & MONE)) {
            mask_t Ml = masks[(r * NSEG + s - 1) * 2 + color];
            if (Ml >> (SEGW - 1)) {
                mask_t Ol = origin_mask(x0 - SEGW, w);
                mask_t Sl = Ml & ~((Ml << 1) & Ol);
                left_node = base - SEGW + (uint32_t)(SEGW - 1 - mclz(Sl));
                hleft = true;
            }
        }
        if (r > 0) {
            U = masks[((r - 1) * NSEG + s) * 2 + color];
            Su = U & ~((U << 1) & O);
            mask_t V = M & U & O;
            Ev = V & ~(V << 1);
            if (color == 0) {
                mask_t Ur = 0;
                if (s > 0) Ul = masks[((r - 1) * NSEG + s - 1) * 2];
                if (s < NSEG - 1) Ur = masks[((r - 1) * NSEG + s + 1) * 2];
                int xn = x0 + SEGW; // origin flag of the column right of this segment
                mask_t On = (xn >= 1 && xn <= w - 2) ? MONE : (mask_t)0;
                mask_t MO = M & O;
                DL = MO & ((U << 1) | (Ul >> (SEGW - 1))) & ~U & ~(MO << 1);
                DR = MO & ((U >> 1) | (Ur << (SEGW - 1))) & ~(U & ((O >> 1) | (On << (SEGW - 1)))) & ~(MO >> 1);
            }
        }
    }
    auto up_left_node = [&](int i) -> uint32_t { // run of the pixel up-left of bit i
        if (i > 0) return base - TW + (uint32_t)run_start(Su, i - 1);
        mask_t Ol = origin_mask(x0 - SEGW, w);
        mask_t Sl = Ul & ~((Ul << 1) & Ol);
        return base - TW - SEGW + (uint32_t)(SEGW - 1 - mclz(Sl));
    };
    auto up_right_node = [&](int i) -> uint32_t { // bit 0 of the segment on the right always starts a run
        return (i < SEGW - 1) ? base - TW + (uint32_t)run_start(Su, i + 1) : base - TW + SEGW;
    };
#ifdef CK_TILE_PROFILE
    unsigned long long tu0 = __builtin_readcyclecounter();
#define TU(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_tile_prof2[k], t_ - tu0); tu0 = t_; } while (0)
#else
#define TU(k)
#endif
    __syncthreads(); // parent[] initialised everywhere before the first adoption lands
    TU(0);
    if (stop_after == 3) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P5a: every run adopts ONE earlier run as its parent with a plain store.  Only the owner writes the entry
    // and nothing reads parent[] in this phase, so no find and no atomic is needed for these links; the target always
    // has a smaller index, which keeps the forest invariant (parent <= self) the atomic phase relies on.
    {
        mask_t St = S;
        while (St) {
            const mask_t low = St & ((mask_t)0 - St); // this run's start bit
            const int i = mctz(low);
            St ^= low;
            const mask_t span = (St & ((mask_t)0 - St)) - low; // bits from this start up to the next one (or to the top: 0 - low)
            mask_t e;
            if ((e = Ev & span)) { int j = mctz(e); Ev &= ~(MONE << j); parent[base + i] = (uint16_t)(base - TW + (uint32_t)run_start(Su, j)); }
            else if ((e = DL & span)) { int j = mctz(e); DL &= ~(MONE << j); parent[base + i] = (uint16_t)up_left_node(j); }
            else if ((e = DR & span)) { int j = mctz(e); DR &= ~(MONE << j); parent[base + i] = (uint16_t)up_right_node(j); }
            else if (i == 0 && hleft) { hleft = false; parent[base] = (uint16_t)left_node; }
        }
    }
    TU(1);
    __syncthreads();
    TU(2);
    if (stop_after == 4 || stop_after == 5) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P5b: the remaining links (a run touching a second, third ... earlier run) go through the atomic union
#ifdef CK_TILE_PROFILE
    uint32_t tcnt = 0, tun = 0;
#define TUN ++tun
#else
#define TUN
#endif
    // The links of a wave are pooled and dealt out evenly: a lane with eight links no longer holds 63 others up — the wave runs
    // ceil(links / 64) union slots (3-4 in a noisy tile) instead of max-over-lanes(links) (7-8), and every slot costs the deepest
    // find in the wave.  The pool is the wave's quarter of the (not yet used) size array: POOL links of two u16 nodes; links
    // that do not fit stay with their lane and are joined the old way.
    {
        constexpr int POOL = TH * TW / (KNT / 64) / 4; // u32 entries per wave
        uint32_t *pool = reinterpret_cast<uint32_t *>(size8) + (tid >> 6) * POOL;
        const uint32_t mine = (hleft ? 1u : 0u) + (uint32_t)mpopc(Ev) + (uint32_t)mpopc(DL) + (uint32_t)mpopc(DR);
        const uint32_t incl = wave_scan_u32(mine);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t pos = incl - mine;
        while ((hleft || Ev || DL || DR) && pos < (uint32_t)POOL) {
            uint32_t ua, ub;
            if (hleft) { hleft = false; ua = base; ub = left_node; }
            else if (Ev) { int i = mctz(Ev); Ev &= Ev - 1; ua = base + run_start(S, i); ub = base - TW + run_start(Su, i); }
            else if (DL) { int i = mctz(DL); DL &= DL - 1; ua = base + run_start(S, i); ub = up_left_node(i); }
            else { int i = mctz(DR); DR &= DR - 1; ua = base + run_start(S, i); ub = up_right_node(i); }
            pool[pos++] = ua | (ub << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t pooled = total < (uint32_t)POOL ? total : (uint32_t)POOL;
        for (uint32_t j = (uint32_t)(tid & 63); j < pooled; j += 64) {
            const uint32_t e = pool[j];
            TUN;
            lds_union(parent, e & 0xFFFFu, e >> 16 TCNT_PASS);
        }
        while (hleft || Ev || DL || DR) { // overflow of the pool (pathological tiles only)
            uint32_t ua, ub;
            if (hleft) { hleft = false; ua = base; ub = left_node; }
            else if (Ev) { int i = mctz(Ev); Ev &= Ev - 1; ua = base + run_start(S, i); ub = base - TW + run_start(Su, i); }
            else if (DL) { int i = mctz(DL); DL &= DL - 1; ua = base + run_start(S, i); ub = up_left_node(i); }
            else { int i = mctz(DR); DR &= DR - 1; ua = base + run_start(S, i); ub = up_right_node(i); }
            TUN;
            lds_union(parent, ua, ub TCNT_PASS);
        }
    }
    TU(3);
    __syncthreads();
    TU(4);
    for (int i = tid * 16; i < TH * TW; i += KNT * 16) *reinterpret_cast<uint4 *>(size8 + i) = make_uint4(0, 0, 0, 0); // pools -> sizes
    TPROF(3);
#ifdef CK_TILE_PROFILE
    {   // [8] unions, [9] find2 iterations summed over lanes, [10] per-wave maximum of a lane's find2 iterations
        uint32_t su = tun, sc = tcnt, mx = tcnt;
        for (int o = 32; o; o >>= 1) { su += __shfl_xor(su, o); sc += __shfl_xor(sc, o); mx = max(mx, (uint32_t)__shfl_xor(mx, o)); }
        if ((tid & 63) == 0) { atomicAdd(&g_tile_prof[8], su); atomicAdd(&g_tile_prof[9], sc); atomicAdd(&g_tile_prof[10], mx); }
    }
    uint32_t wruns = 0, whops = 0, wit = 0;
#endif

    __syncthreads(); // halving stores must land before the owners publish final roots
    if (stop_after == 6) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P6: flatten run starts, accumulate sizes and ring flags at the roots ---------------------------------------
    {
        const bool ring_row = (r == 0 && ty0 > 0) || (r == TH - 1 && ty0 + TH < h);
        mask_t St = S;
        // consecutive runs of a segment usually end at the same root (in a noisy tile nearly every white run belongs to
        // the one spanning component): their pixels are summed in registers and flushed once per change of root
        uint32_t acc_root = 0xFFFFFFFFu, acc_add = 0;
        bool acc_ring = false;
#ifdef CK_TILE_PROFILE
        unsigned long long tq0 = __builtin_readcyclecounter(), tq_ext = 0, tq_walk = 0, tq_tail = 0;
#define TQ(acc) do { unsigned long long t_ = __builtin_readcyclecounter(); acc += t_ - tq0; tq0 = t_; } while (0)
#else
#define TQ(acc)
#endif
        while (St) { // four runs per round: their root walks proceed in lockstep
            uint32_t node[4], root[4], add[4];
            bool ring[4], live[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                live[q] = St != 0;
                node[q] = base; add[q] = 0; ring[q] = false;
                if (live[q]) {
                    const mask_t low = St & ((mask_t)0 - St); // this run's start bit
                    const int i = mctz(low);
                    St ^= low;
                    const mask_t span = (St & ((mask_t)0 - St)) - low; // up to the next start (or to the top: 0 - low)
                    mask_t run = M & span;
                    node[q] = base + i;
                    add[q] = (uint32_t)mpopc(run);
                    ring[q] = ring_row || (s == 0 && tx0 > 0 && (run & MONE)) || (s == NSEG - 1 && tx0 + TW < w && (run >> (SEGW - 1)));
                }
                root[q] = node[q];
            }
            TQ(tq_ext);
            // plain loads behind a compiler barrier: the four reads of a step are independent and go out together (as volatile
            // reads each one was waited for before the next was issued: four LDS round trips per step instead of one)
            const uint16_t *vp = parent;
            for (int it = 0; it < 8192; it++) {
                __asm__ volatile("" ::: "memory");
                uint32_t n0 = vp[root[0]], n1 = vp[root[1]], n2 = vp[root[2]], n3 = vp[root[3]];
#ifdef CK_TILE_PROFILE
                wit++; whops += (n0 != root[0]) + (n1 != root[1]) + (n2 != root[2]) + (n3 != root[3]);
#endif
                if (n0 == root[0] && n1 == root[1] && n2 == root[2] && n3 == root[3]) break;
                root[0] = n0; root[1] = n1; root[2] = n2; root[3] = n3;
            }
            TQ(tq_walk);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (live[q]) {
#ifdef CK_TILE_PROFILE
                    wruns++;
#endif
                    parent[node[q]] = (uint16_t)root[q];
                    if (root[q] == acc_root) { acc_add += add[q]; acc_ring = acc_ring || ring[q]; }
                    else {
                        if (acc_root != 0xFFFFFFFFu) lds_size_add(size8, acc_root, acc_add, acc_ring);
                        acc_root = root[q]; acc_add = add[q]; acc_ring = ring[q];
                    }
                }
            TQ(tq_tail);
        }
        if (acc_root != 0xFFFFFFFFu) lds_size_add(size8, acc_root, acc_add, acc_ring);
#ifdef CK_TILE_PROFILE
        if (tid == 0) { atomicAdd(&g_tile_prof[14], tq_ext); atomicAdd(&g_tile_prof[15], tq_walk); atomicAdd(&g_tile_prof[7], tq_tail); }
#endif
    }
    __syncthreads();
    TPROF(4);
#ifdef CK_TILE_PROFILE
    {   // [11] runs, [12] hops summed over lanes, [13] per-wave maximum of a lane's walk iterations
        uint32_t sr = wruns, sh = whops, mx = wit;
        for (int o = 32; o; o >>= 1) { sr += __shfl_xor(sr, o); sh += __shfl_xor(sh, o); mx = max(mx, (uint32_t)__shfl_xor(mx, o)); }
        if ((tid & 63) == 0) { atomicAdd(&g_tile_prof[11], sr); atomicAdd(&g_tile_prof[12], sh); atomicAdd(&g_tile_prof[13], mx); }
    }
#endif

    if (stop_after == 7) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P7: write label words (16 pixels per item) -------------------------------------------------------------------
    const uint32_t nitems = TH * 8;
    for (uint32_t item0 = 0; item0 < nitems; item0 += KNT) { // uniform trip count: the ring-root append below votes per wave
        const int item = (int)item0 + tid;
        int rr = item >> 3, c = item & 7;
        int gy = ty0 + rr, gx = tx0 + 16 * c;
        uint32_t nroots = 0;
        uint32_t roots_mask = 0;
        uint32_t outw[16];
        const bool inside = gy < h && gx < w;
        int seg = c / MPIECES, piece = c % MPIECES;
        uint32_t sbase = (uint32_t)(rr * TW + SEGW * seg);
        if (inside) {
            mask_t Wm = masks[(rr * NSEG + seg) * 2], Bm = masks[(rr * NSEG + seg) * 2 + 1];
            mask_t Oo = origin_mask(tx0 + SEGW * seg, w);
            mask_t SW = Wm & ~((Wm << 1) & Oo), SB = Bm & ~((Bm << 1) & Oo);
            // everything below works on the 16 bits of this chunk with compile-time shifts (the kernel is bound by
            // instruction issue): colour bits, run-start bits, and for each colour the node of the run that is already
            // open when the chunk begins
            const int sh = 16 * piece;
            const uint32_t w16 = (uint32_t)(Wm >> sh) & 0xFFFFu, b16 = (uint32_t)(Bm >> sh) & 0xFFFFu;
            const uint32_t sw16 = (uint32_t)(SW >> sh) & 0xFFFFu, sb16 = (uint32_t)(SB >> sh) & 0xFFFFu;
            const mask_t lowmask = (MONE << sh) - MONE; // sh <= SEGW - 16
            const mask_t lw = SW & lowmask, lb = SB & lowmask;
            const uint32_t carryW = sbase + (lw ? (uint32_t)(SEGW - 1 - mclz(lw)) : 0u);
            const uint32_t carryB = sbase + (lb ? (uint32_t)(SEGW - 1 - mclz(lb)) : 0u);
            const uint32_t cbase = sbase + (uint32_t)sh;
            const uint32_t gbase = (uint32_t)ty0 * (uint32_t)w + (uint32_t)tx0;
            // Straight-line code, no exec-mask regions: a pixel's run starts at the nearest start bit of EITHER colour at or
            // below it (a start of the other colour cannot lie inside a run), or before the chunk (then the run has the colour
            // of pixel 0 and the carried node of that colour).  Uncoloured pixels look up a harmless in-range node.
            const uint32_t any16 = w16 | b16, st16 = sw16 | sb16;
            uint32_t cur = (w16 & 1u) ? carryW : carryB;
            uint32_t nodev[16], rootv[16], sizev[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                cur = ((st16 >> k) & 1u) ? cbase + (uint32_t)k : cur;
                nodev[k] = cur;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) rootv[k] = parent[nodev[k]];
#pragma unroll
            for (int k = 0; k < 16; k++) sizev[k] = size8[rootv[k]];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t root = rootv[k], sw = sizev[k];
                const uint32_t gidx = gbase + (root >> 7) * (uint32_t)w + (root & (TW - 1));
                const uint32_t cls = (sw & 0x80u) ? CK_LBL_BORDER : ((int)(sw & 0x7Fu) < min_comp ? CK_LBL_SMALL : 0u);
                outw[k] = ((any16 >> k) & 1u) ? (gidx | cls) : CK_LBL_INVALID;
                // a ring-touching root: a run start whose label word points at itself
                roots_mask |= ((sw & 0x80u) && root == cbase + (uint32_t)k) ? (1u << k) : 0u; // roots are run starts: never an uncoloured pixel
            }
            uint32_t *dst = labels + fbase + (size_t)gy * w + gx;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (gx + 4 * q < w)
                    *reinterpret_cast<uint4 *>(dst + 4 * q) = make_uint4(outw[4 * q], outw[4 * q + 1], outw[4 * q + 2], outw[4 * q + 3]);
            nroots = (uint32_t)__popc(roots_mask);
        }
        // ---- P8: ring-touching roots go to the frame's list: one global reservation per wave and round -----------------
        const uint32_t incl = wave_scan_u32(nroots);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total) {
            uint32_t wbase = 0;
            if ((tid & 63) == 63) wbase = atomicAdd(&broot_count[frame], total);
            wbase = (uint32_t)__builtin_amdgcn_readlane((int)wbase, 63);
            uint32_t pos = wbase + incl - nroots;
            while (roots_mask) {
                int k = __builtin_ctz(roots_mask);
                roots_mask &= roots_mask - 1;
                if (pos < (uint32_t)broot_cap) {
                    uint32_t node = sbase + (uint32_t)(16 * piece + k);
                    ck_border_root br;
                    br.root = (uint32_t)gy * (uint32_t)w + (uint32_t)(tx0 + SEGW * seg + 16 * piece + k);
                    br.size = size8[node] & 0x7Fu;
                    broots[(size_t)frame * broot_cap + pos] = br;
                    csize[fbase + br.root] = 0; // k_roots accumulates the parts of a component at its global root
                }
                pos++;
            }
        }
    }

    TPROF(5);
    TPROF(6);
}

#ifdef CK_TILE_PROFILE
} // namespace
extern "C" int ck_tile_profile_read(unsigned long long *out, int reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_tile_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
extern "C" int ck_tile_profile2_read(unsigned long long *out, int reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_prof2), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_tile_prof2), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
namespace {
#endif

// ---- cross-tile merge ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t g_load(const uint32_t *L, uint32_t i) {
    return __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & CK_LBL_IDX_MASK;
}
// find over the root entries of ring-touching components, with path halving (same argument as lds_find)
__device__ __forceinline__ uint32_t g_find(uint32_t *L, uint32_t r) {
    for (;;) {
        uint32_t n = g_load(L, r);
        if (n == r) return r;
        uint32_t g = g_load(L, n);
        if (g == n) return n;
        __hip_atomic_store(&L[r], g | CK_LBL_BORDER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = g;
    }
}
__device__ __forceinline__ uint32_t g_find_ro(const uint32_t *L, uint32_t r) {
    for (;;) {
        uint32_t n = g_load(L, r);
        if (n == r) return r;
        r = n;
    }
}
// union of the components whose tile-local roots are a and b (both ring-touching, so their entries carry CK_LBL_BORDER)
__device__ __forceinline__ void g_union_roots(uint32_t *L, uint32_t a, uint32_t b) {
    for (;;) {
        a = g_find(L, a);
        b = g_find(L, b);
        if (a == b) return;
        if (a < b) { uint32_t t = a; a = b; b = t; }
        uint32_t old = atomicMin(&L[a], b | CK_LBL_BORDER) & CK_LBL_IDX_MASK;
        if (old == a) return;
        a = old;
    }
}

// One workgroup per tile, one thread per pixel of the tile's top row / left column / right column.  Every link that
// crosses a tile boundary is reduced to the pair of tile-local roots it joins; a small LDS hash set keeps one thread
// per distinct pair (a large component crosses an edge at dozens of places), and only those run the global union.
constexpr int MSET = 1024;
__device__ __forceinline__ void merge_link(uint32_t *L, unsigned long long *set, uint32_t p, uint32_t q) {
    uint32_t a = L[p] & CK_LBL_IDX_MASK, b = L[q] & CK_LBL_IDX_MASK; // written by k_tile, read-only until a root entry is hooked
    if (a == b) return;
    unsigned long long key = a < b ? ((unsigned long long)a << 32) | b : ((unsigned long long)b << 32) | a;
    uint32_t hsh = (uint32_t)((key >> 32) ^ key) * 2654435761u;
    uint32_t slot = (hsh >> 12) & (MSET - 1);
    for (int probe = 0; probe < 16; probe++) {
        unsigned long long prev = atomicCAS(&set[slot], 0ull, key);
        if (prev == key) return;          // another thread of this tile already owns the pair
        if (prev == 0ull) break;          // inserted: this thread does the union
        slot = (slot + 1) & (MSET - 1);
    }                                     // table crowded: fall through and union without de-duplication
    g_union_roots(L, a, b);
}

__global__ __launch_bounds__(NT) void k_merge(const uint8_t *__restrict__ thresh, uint32_t *__restrict__ labels, int w, int h,
                                              int tiles_x, int tiles_y) {
    __shared__ unsigned long long sSet[MSET];
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int frame = blockIdx.x / tiles, tile = blockIdx.x - frame * tiles;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    const size_t fbase = (size_t)frame * (size_t)w * (size_t)h;
    const uint8_t *T = thresh + fbase;
    uint32_t *L = labels + fbase;
    for (int i = tid; i < MSET; i += NT) sSet[i] = 0ull;
    __syncthreads();
    int x, y, kind;
    if (tid < TW) { x = tx0 + tid; y = ty0; kind = 0; }
    else if (tid < TW + TH) { x = tx0; y = ty0 + 1 + (tid - TW); kind = 1; }
    else { x = tx0 + TW - 1; y = ty0 + 1 + (tid - TW - TH); kind = 2; }
    if (y >= ty0 + TH || x >= w || y >= h) return;
    if (x < 1 || x > w - 2) return; // not an origin column
    const uint32_t p = (uint32_t)y * (uint32_t)w + (uint32_t)x;
    const uint8_t v = T[p];
    if (v == 127) return;
    if (kind == 0) {
        if (x == tx0 && tx0 > 0 && T[p - 1] == v) merge_link(L, sSet, p, p - 1);
        if (y >= 1) {
            if (T[p - w] == v) merge_link(L, sSet, p, p - w);
            if (v == 255) {
                if (T[p - w - 1] == 255) merge_link(L, sSet, p, p - w - 1);
                if (T[p - w + 1] == 255) merge_link(L, sSet, p, p - w + 1);
            }
        }
    } else if (kind == 1) {
        if (tx0 > 0) {
            if (T[p - 1] == v) merge_link(L, sSet, p, p - 1);
            if (v == 255 && T[p - w - 1] == 255) merge_link(L, sSet, p, p - w - 1);
        }
    } else {
        if (v == 255 && T[p - w + 1] == 255) merge_link(L, sSet, p, p - w + 1);
    }
}

// ---- ring-touching roots: flatten their entries and accumulate sizes at the global roots ---------------------------------------
// csize[] was zeroed at every ring-touching root by k_tile, so all parts of a component (the global root's own included)
// simply add up there.  csize[] is only ever compared with min_component_px: a part that is large enough on its own settles
// the answer with a plain store (thousands of parts of one frame-spanning component would otherwise queue on one address).
__global__ __launch_bounds__(NT) void k_roots(uint32_t *__restrict__ labels, uint32_t *__restrict__ csize,
                                              const ck_border_root *__restrict__ broots,
                                              const uint32_t *__restrict__ broot_count, int broot_cap, size_t npix, int min_comp) {
    const int frame = blockIdx.y;
    uint32_t n = min(broot_count[frame], (uint32_t)broot_cap);
    uint32_t *L = labels + (size_t)frame * npix;
    uint32_t *C = csize + (size_t)frame * npix;
    for (uint32_t k = blockIdx.x * NT + threadIdx.x; k < n; k += gridDim.x * NT) {
        ck_border_root br = broots[(size_t)frame * broot_cap + k];
        uint32_t g = g_find_ro(L, br.root);
        if (g != br.root) L[br.root] = g | CK_LBL_BORDER; // still a valid ancestor for concurrent finds
        if ((int)br.size >= min_comp) __hip_atomic_store(&C[g], SIZE_SAT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else atomicAdd(&C[g], br.size);
    }
}

// ---- parity / test path: canonical labels and exact sizes ---------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_canon(const uint32_t *__restrict__ labels, uint32_t *__restrict__ out, size_t npix, size_t total) {
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    size_t fb = (i / npix) * npix;
    uint32_t l = labels[i];
    uint32_t g = CK_LBL_INVALID;
    if (l != CK_LBL_INVALID) {
        g = l & CK_LBL_IDX_MASK;
        if (l & CK_LBL_BORDER) g = labels[fb + g] & CK_LBL_IDX_MASK;
    }
    out[i] = g;
}
__global__ __launch_bounds__(NT) void k_count(const uint32_t *__restrict__ canon, uint32_t *__restrict__ cnt, size_t npix, size_t total) {
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    uint32_t g = canon[i];
    if (g != CK_LBL_INVALID) atomicAdd(&cnt[(i / npix) * npix + g], 1u);
}
__global__ __launch_bounds__(NT) void k_sizes(const uint32_t *__restrict__ canon, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ sizes, size_t npix, size_t total) {
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    uint32_t g = canon[i];
    sizes[i] = (g == CK_LBL_INVALID) ? 0u : cnt[(i / npix) * npix + g];
}

__global__ __launch_bounds__(NT) void k_decimate(const uint8_t *__restrict__ src, size_t frame_pitch, int stride, int f, int qw, int qh,
                                                 uint8_t *__restrict__ dst, size_t total) {
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    size_t npix = (size_t)qw * qh;
    size_t fr = i / npix, rem = i - fr * npix;
    int y = (int)(rem / qw), x = (int)(rem - (size_t)y * qw);
    dst[i] = src[fr * frame_pitch + (size_t)(y * f) * stride + (size_t)x * f];
}

} // namespace

int ck_launch_threshold_segment(ck_handle *h, const uint8_t *frames, int stride, size_t frame_pitch, int n, bool precomputed) {
    const int tiles = h->tiles_x * h->tiles_y;
    static const int stop_after = getenv("CK_TILE_STOP_AFTER") ? atoi(getenv("CK_TILE_STOP_AFTER")) : 99;
    CK_HIP(hipMemsetAsync(h->d_broot_count, 0, sizeof(uint32_t) * (size_t)n, h->stream));
    if (precomputed)
        hipLaunchKernelGGL(k_tile<true>, dim3((unsigned)(tiles * n)), dim3(KNT), 0, h->stream, frames, frame_pitch, stride, h->qw, h->qh,
                           h->tiles_x, h->tiles_y, h->cfg.min_white_black_diff, h->cfg.min_component_px, h->d_thresh, h->d_labels,
                           h->d_broots, h->d_broot_count, h->broot_cap, h->d_csize, stop_after);
    else
        hipLaunchKernelGGL(k_tile<false>, dim3((unsigned)(tiles * n)), dim3(KNT), 0, h->stream, frames, frame_pitch, stride, h->qw, h->qh,
                           h->tiles_x, h->tiles_y, h->cfg.min_white_black_diff, h->cfg.min_component_px, h->d_thresh, h->d_labels,
                           h->d_broots, h->d_broot_count, h->broot_cap, h->d_csize, stop_after);
    hipLaunchKernelGGL(k_merge, dim3((unsigned)(tiles * n)), dim3(NT), 0, h->stream, h->d_thresh, h->d_labels, h->qw, h->qh,
                       h->tiles_x, h->tiles_y);
    int bx = (h->broot_cap + NT * 8 - 1) / (NT * 8);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_roots, dim3((unsigned)bx, (unsigned)n), dim3(NT), 0, h->stream, h->d_labels, h->d_csize, h->d_broots,
                       h->d_broot_count, h->broot_cap, h->npix, h->cfg.min_component_px);
    CK_HIP(hipGetLastError());
    return CK_OK;
}

int ck_launch_canonical_labels(ck_handle *h, int n, uint32_t *d_out, uint32_t *d_sizes) {
    size_t total = h->npix * (size_t)n;
    unsigned blocks = (unsigned)((total + NT - 1) / NT);
    hipLaunchKernelGGL(k_canon, dim3(blocks), dim3(NT), 0, h->stream, h->d_labels, d_out, h->npix, total);
    if (d_sizes) {
        // exact sizes by counting: test path only (the pipeline uses the SMALL flag / csize[] instead)
        uint32_t *cnt = nullptr;
        CK_HIP(hipMalloc(&cnt, total * sizeof(uint32_t)));
        CK_HIP(hipMemsetAsync(cnt, 0, total * sizeof(uint32_t), h->stream));
        hipLaunchKernelGGL(k_count, dim3(blocks), dim3(NT), 0, h->stream, d_out, cnt, h->npix, total);
        hipLaunchKernelGGL(k_sizes, dim3(blocks), dim3(NT), 0, h->stream, d_out, cnt, d_sizes, h->npix, total);
        CK_HIP(hipStreamSynchronize(h->stream));
        CK_HIP(hipFree(cnt));
    }
    CK_HIP(hipGetLastError());
    return CK_OK;
}

int ck_launch_decimate(ck_handle *h, const uint8_t *frames, int stride, size_t frame_pitch, int n) {
    size_t total = h->npix * (size_t)n;
    unsigned blocks = (unsigned)((total + NT - 1) / NT);
    hipLaunchKernelGGL(k_decimate, dim3(blocks), dim3(NT), 0, h->stream, frames, frame_pitch, stride, h->cfg.quad_decimate, h->qw,
                       h->qh, h->d_qframes, total);
    CK_HIP(hipGetLastError());
    return CK_OK;
}
