// k_ccl.hip — adaptive threshold + union-find segmentation for gfx950 (the stage scored against the HBM
// roofline on SURVEY §8d's accounting: 7 algorithmic bytes per pixel = 1 R image + 1 W thresholded + 1 R thresholded + 4 W
// label; the fused kernel below never re-reads the threshold map and writes 2-byte tile-local label words, so it moves fewer).
//
// Replaces, in the reference's production path, the threshold and connected-component stages of the external
// AprilTag-3 detector reached at crates/apriltags/src/lib.rs:301; the connectivity rule is the one CAT spells
// out in crates/chalkydri-apriltags/src/lib.rs:501-549 (4-connected black, 8-connected white, origin columns
// 1..w-2), stated as bit operations on 32-pixel row words in ck_links.h.
//
// Structure (one launch each, batched over frames):
//   k_tile   one workgroup (256 threads) per 32x128 tile.  Coalesced 16-byte loads of the tile + halo into LDS, 4x4
//            min/max, 3x3 dilation, tri-state threshold (written once, 16 B/lane) and the rows' colour bits as 32-pixel
//            words; a tile none of whose 4x4 tiles has contrast only stores constants.  The union-find works on NODES = the
//            components of 2 x 32 pixel blocks (a pair of rows, one word, one colour: ck_links.h), found with bit operations:
//            (a) one lane per pair word (two waves per colour) lists its nodes and lets every node adopt ONE node of the pair
//            above with a plain store; (b) the links that are left over, pooled in LDS, one lane per link, through an atomic-min
//            union; (c) flatten + exact sizes (ds_add into the roots' own entries); (d) ring-touching roots claimed from the 320
//            ring pixels and written to the tile's slice, white ids from 0 up and black ones from the top down; (e) the 16-bit
//            label word of every node formed once into a table, one lookup per pixel, 16-byte stores (8 pixels) per lane and
//            row.  19.6 KB of LDS: eight workgroups per CU.  HBM traffic per pixel: about 1.3 bytes read (tile + halo), 1 byte
//            threshold + 2 bytes label word + 0.16 bytes ring entries written (measured: profiles/traffic_latest.json).
//   k_fmerge the ring-touching roots of a frame's tiles are joined across the tile boundaries in LDS (ids and colours along the
//            boundaries come from k_tile; no label or threshold word is read) and the frame's slot tables get every such
//            component's frame-level root and size.  One workgroup per frame when its roots fit (both colours in one sweep),
//            else two, one per colour — white and black never join; a workgroup with more roots than parents + keys fit
//            keeps only the parents in LDS, and beyond that runs in global memory.
// No full-frame relabel pass exists: interior components are final when k_tile writes them; the label words of ring-touching
// ones carry a slot, which consumers resolve with two independent table reads (label word format in ck_internal.h).
#include <stdlib.h>
#include <type_traits>

#include "ck_internal.h"
#include "ck_links.h"

namespace {

constexpr int TW = CK_TW, TH = CK_TH, NT = 256; // NT: merge / utility kernels
constexpr int KNT = 256;                         // k_tile: one lane per (row, word, colour) where rows are handled, one per run elsewhere
constexpr int NWD = TW / 32;                     // 32-pixel words per tile row
constexpr int IMG_PITCH = 160;                   // 16 left halo | 128 tile | 16 right halo (only 8 + 8 of the halo are used)
constexpr int IMG_ROWS = TH + 8;
constexpr int T4X = TW / 4 + 2, T4Y = TH / 4 + 2; // 4x4-tile min/max grid of the staged region (one ring around the tile's own)
static_assert((TH / 2) * NWD == 64, "k_tile node phases: one wave per colour, one lane per (pair of rows, word)");
static_assert(TH * 8 == KNT, "threshold / label passes: one 16-pixel chunk per thread");
static_assert(TH * TW <= 4096, "list entries keep the node in 12 bits, the colour in bit 12");
static_assert(TW == 128, "a node index splits with >> 7 / & 127");

// LDS of k_tile, 19.6 KB, so that eight workgroups (32 waves: all a CU can hold) share a CU:
//   parent  u16[TH*TW]  the union-find's entries, indexed by tile pixel.  NODES are the components of 2 x 32 pixel blocks (a pair of
//                       rows, one word, one colour: ck_links.h, "nodes over PAIRS of rows" — about half as many as one-row runs
//                       on dense noise); a node owns the entry at its LOOKUP pixel (first column, top pixel if the column has
//                       both: what its pixels find with one count-leading-zeros) and, when that is another pixel, at its MIN
//                       pixel (the lookup entry then points at it); the pixels of the frame's two non-origin columns are nodes of
//                       their own.  An entry:
//                         non-root: tile-local index of the parent (bit 15 clear)
//                         root:     CK_ROOT | pixel count of the component in bits 0..12 (a tile has 4096 pixels) | CK_RING when it
//                                   touches the tile ring — the union-find's sizes live in the roots' own entries
//                       after the flatten pass an entry is the component's root entry or its root's node (the label pass
//                       looks up at most two);
//                       while the threshold is computed the same bytes hold the staged image and the 4x4 min/max (the
//                       per-4x4 threshold words sit in the list's bytes, which is not alive yet)
//   list    u32[LIST_CAP]  the tile's nodes: lookup pixel | colour << 12 | pixel count << 16; the white ones from the front, the black ones
//                       from the back (the two waves that build it need not know each other's counts); a tile with more nodes
//                       (one-pixel patterns) does without the list
//   pool    u32[POOL_CAP]  links that need an atomic union (two u16 entries each)
//   masks   u32[TH][NWD][2]  colour bits of every row word (0 white, 1 black)
//   starts  u32[TH/2][NWD][2]  node starts of every pair word
constexpr int OFF_PARENT = 0;                              // u16[TH*TW] = 8192
constexpr int OFF_IMG = 0;                                 // IMG_ROWS*IMG_PITCH = 6400
constexpr int OFF_MINMAX = OFF_IMG + IMG_ROWS * IMG_PITCH; // u32[T4Y*T4X] = 1360 (one dword per 4x4 tile: the dilation's neighbour reads stay 4-byte aligned)
constexpr int OFF_LIST = TH * TW * 2;                      // u32[LIST_CAP] = 8192
constexpr int LIST_CAP = 2048;                             // nodes the list holds (dense binary noise has about 800 per tile)
static_assert(LIST_CAP % (2 * KNT) == 0 && LIST_CAP * 4 == TH * TW * 2, "the list fills the 8 KB behind the parent array");
constexpr int OFF_THR = OFF_LIST;                          // u16[(TH/4)*(TW/4)] = 512 (the run list is not alive yet)
constexpr int POOL_CAP = 512;                              // dense binary noise leaves about 150 links per tile for the pool
constexpr int OFF_POOL = OFF_LIST + TH * TW * 2;           // u32[POOL_CAP] = 2048
constexpr int OFF_MASK = OFF_POOL + POOL_CAP * 4;          // u32[TH][NWD][2] = 1024
constexpr int OFF_S2 = OFF_MASK + TH * NWD * 2 * 4;        // u32[TH/2][NWD][2] = 512
constexpr int OFF_MISC = OFF_S2 + (TH / 2) * NWD * 2 * 4;  // u32[16]
constexpr int LDS_BYTES = OFF_MISC + 64;
constexpr int RING_CAP = CK_RING_CAP;                      // ring-touching roots of a tile: at most one per ring pixel
// the label table of P6c / P7 (32-bit words over the dead parent + list arrays and the first bytes of the dead pool): entries
// [0, 4096) by (pair word * 2 + colour) * 32 + node number, then the pixels of the frame's two non-origin columns (side * 32 + row),
// then one entry that reads "no component"
// (a (pair word, colour) owns TAB_STRIDE = 33 entries, not 32: with 32 the small node numbers of all pair words fell on the same
// dozen LDS banks — nine lanes per bank in a label-pass read against three with 33 (five for the table indexed by lookup pixel);
// same box: 0.9525 ms with the lookup-pixel table, 0.9631 with stride 32, 0.9141 with stride 33)
constexpr int TAB_STRIDE = 33, TAB_EDGE = (TH / 2) * NWD * 2 * TAB_STRIDE, TAB_NONE = TAB_EDGE + 64;
static_assert((TAB_NONE + 1) * 4 <= OFF_POOL + POOL_CAP * 4 && TAB_NONE < (1 << 13), "label table: parent + list arrays + the pool's first bytes; 13-bit indices");
constexpr uint32_t CK_ROOT = 0x8000u, CK_RING = 0x4000u, CK_CLAIM = 0x2000u, CK_COUNT = 0x1FFFu;
static_assert(OFF_MINMAX + T4Y * T4X * 4 <= OFF_LIST, "min/max scratch must fit in the parent array");
static_assert(LDS_BYTES <= 20480, "keep eight workgroups per CU");
static_assert(TH * TW <= CK_COUNT, "a component's pixel count fits below the flags of a root entry");

// find with path halving.  Plain stores race with the min-hooks of lds_union, but every value ever written to p[a]
// is an ancestor of a, so the forest stays valid (a lost hook is re-issued by its own union).
// Two halving finds walked in lockstep: both chains have a read in flight at every step
// (plain loads behind compiler barriers, not volatile ones: a volatile read is waited for before the next is issued, which
// would put the two chains' reads one after the other).  A root is an entry with CK_ROOT set.
// The byte offset of u16 entry i as i + i: the shift the compiler makes of `p[i]` is one of the vector instructions a SIMD takes
// 4.2 clocks for, an add 2.6 (tools/probes/valu_rate_probe.hip) — and the chains of the union-find are walked entry by entry.
__device__ __forceinline__ uint32_t twice(uint32_t i) { uint32_t r; __asm__("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(i)); return r; }
__device__ __forceinline__ uint32_t u16_at(const uint16_t *p, uint32_t i) { return *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(p) + twice(i)); }
__device__ __forceinline__ void lds_find2(uint16_t *p, uint32_t &a, uint32_t &b) {
    for (;;) {
        __asm__ volatile("" ::: "memory");
        const uint32_t na = u16_at(p, a), nb = u16_at(p, b);
        const bool da = (na & CK_ROOT) != 0, db = (nb & CK_ROOT) != 0;
        if (da && db) return;
        const uint32_t ia = da ? a : na, ib = db ? b : nb; // the parents (or the roots themselves)
        __asm__ volatile("" ::: "memory");
        const uint32_t ga = u16_at(p, ia), gb = u16_at(p, ib);
        if (!da) { if (!(ga & CK_ROOT)) { p[a] = (uint16_t)ga; a = ga; } else a = ia; }
        if (!db) { if (!(gb & CK_ROOT)) { p[b] = (uint16_t)gb; b = gb; } else b = ib; }
    }
}
// atomic min on one u16 entry (LDS has no 16-bit atomics): compare-and-swap on the word that holds it.  Returns the
// entry's previous value (a root's entry, CK_ROOT | ..., is larger than any index, so the min hooks it).  A concurrent halving
// store to the other half only makes the swap fail and retry.
__device__ __forceinline__ uint32_t lds_min16(uint16_t *p, uint32_t idx, uint32_t val) {
    uint32_t *wp = reinterpret_cast<uint32_t *>(p) + (idx >> 1);
    const uint32_t sh = (idx & 1u) * 16u;
    __asm__ volatile("" ::: "memory"); // (a fresh read, but a plain one: a volatile access through the cast pointer becomes a flat load, which waits for every outstanding global store)
    uint32_t wv = *wp;
    for (;;) {
        const uint32_t cur = (wv >> sh) & 0xFFFFu;
        if (cur <= val) return cur;
        const uint32_t prev = atomicCAS(wp, wv, (wv & ~(0xFFFFu << sh)) | (val << sh));
        if (prev == wv) return cur;
        wv = prev;
    }
}
// root = smaller index (unions run before any size is accumulated: the roots' entries are bare CK_ROOT)
__device__ __forceinline__ void lds_union(uint16_t *p, uint32_t a, uint32_t b) {
    for (;;) {
        lds_find2(p, a, b);
        if (a == b) return;
        if (a < b) { uint32_t t = a; a = b; b = t; }
        const uint32_t old = lds_min16(p, a, b);
        if (old & CK_ROOT) return; // a was still a root: hooked
        a = old;                   // somebody hooked it first: go on from its parent
    }
}
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
// per 16-bit half: a > t ? 1 : 0, as a saturating packed subtraction and a packed minimum (written as instructions: from the
// element-wise builtins the compiler makes four 16-bit compares, selects and a byte permute)
__device__ __forceinline__ uint32_t pk_gt_u16(uint32_t a, uint32_t t, uint32_t ones) {
    uint32_t d, r;
    __asm__("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(t));
    __asm__("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(d), "v"(ones));
    return r;
}
// gathers bit 7 of each byte of v into a nibble (bit k = byte k)
__device__ __forceinline__ uint32_t msb_nibble(uint32_t v) {
    return (((v >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xFu;
}

// v_ffbl_b32: index of the lowest set bit, 0xFFFFFFFF for 0 (what __builtin_ctz leaves undefined)
__device__ __forceinline__ uint32_t ffbl_raw(uint32_t v) {
    uint32_t r;
    __asm__("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// One pair word of the tile (rows 2p and 2p + 1, one 32-pixel word, one colour) as its lane sees it: its nodes, the nodes of the
// pair word above, and the links between them that have not been used yet (ck_links.h: the links of the two facing rows, keyed by
// the lower pixel; here every mask is already ANDed with the origin flags, so the rule's origin tests are all true).  Read from
// the row masks and node starts in LDS; words that do not exist (first pair / first or last word of the tile) are read at a
// safe index and masked to zero.
struct PairCtx {
    uint32_t Mt, S2;            // top-row pixels and node starts of this pair word
    uint32_t Ut, S2u;           // the same of the pair word above
    uint32_t base;              // tile pixel of bit 0 of this pair word's top row (the row below: + TW; the pair above: - 2 * TW)
    uint32_t left, up_l, up_r;  // lookup pixels: last node of the pair word on the left / of the upper-left one, first node of the upper-right one
    uint32_t Ev, DL, DR, flags; // links still to be used (DL without bit 0, DR without bit 31: those two cross a word boundary and are flags)
};
__device__ __forceinline__ uint32_t pair_entry(uint32_t Mt, uint32_t base, int s) { return base + (((Mt >> s) & 1u) ? 0u : (uint32_t)TW) + (uint32_t)s; }
__device__ __forceinline__ uint32_t lower_node(const PairCtx &v, int x) { return pair_entry(v.Mt, v.base, ck_run_start32(v.S2, x)); }
__device__ __forceinline__ uint32_t upper_node(const PairCtx &v, int x) { return pair_entry(v.Ut, v.base - 2u * TW, ck_run_start32(v.S2u, x)); }
__device__ __forceinline__ PairCtx load_pair(const uint32_t *mk, const uint32_t *s2w, int p, int wd, int c, int tx0, int w) {
    PairCtx v;
    const int mi = ((2 * p) * NWD + wd) * 2 + c; // the top row's word; the bottom row's: + 2 * NWD; the rows of the pair above: - 4 * NWD, - 2 * NWD
    const int si = (p * NWD + wd) * 2 + c;
    const bool has_l = wd > 0, has_u = p > 0, has_r = wd < NWD - 1, white = c == 0;
    const int x0 = tx0 + 32 * wd;
    const uint32_t O = ck_origin32(x0, w), Op = ck_origin32(x0 - 32, w), On = ck_origin32(x0 + 32, w);
    const uint32_t Mt = mk[mi] & O, Mb = mk[mi + 2 * NWD] & O;
    uint32_t Mtl = mk[has_l ? mi - 2 : mi] & Op, Mbl = mk[has_l ? mi + 2 * NWD - 2 : mi] & Op, S2l = s2w[has_l ? si - 2 : si];
    uint32_t Ut = mk[has_u ? mi - 4 * NWD : mi] & O, Ub = mk[has_u ? mi - 2 * NWD : mi] & O, S2u = s2w[has_u ? si - 2 * NWD : si];
    uint32_t Utl = mk[(has_u && has_l) ? mi - 4 * NWD - 2 : mi] & Op, Ubl = mk[(has_u && has_l) ? mi - 2 * NWD - 2 : mi] & Op;
    uint32_t S2ul = s2w[(has_u && has_l) ? si - 2 * NWD - 2 : si];
    uint32_t Utr = mk[(has_u && has_r) ? mi - 4 * NWD + 2 : mi] & On, Ubr = mk[(has_u && has_r) ? mi - 2 * NWD + 2 : mi] & On;
    Mtl = has_l ? Mtl : 0u; Mbl = has_l ? Mbl : 0u;
    Ut = has_u ? Ut : 0u; Ub = has_u ? Ub : 0u; S2u = has_u ? S2u : 0u;
    Ubl = (has_u && has_l) ? Ubl : 0u; Ubr = (has_u && has_r) ? Ubr : 0u;
    v.Mt = Mt; v.S2 = s2w[si]; v.Ut = Ut; v.S2u = S2u;
    v.base = (uint32_t)((2 * p) * TW + 32 * wd);
    v.left = pair_entry(Mtl, v.base - 32u, ck_last_start32(S2l));
    v.up_l = pair_entry(Utl, v.base - 2u * TW - 32u, ck_last_start32(S2ul));
    v.up_r = v.base - 2u * TW + 32u + ((Utr & 1u) ? 0u : (uint32_t)TW);
    const uint32_t hl = ck_pair_link32(white, Mt, Mb, (Mtl >> 31) != 0, (Mbl >> 31) != 0) & 1u;
    const ck_word_links K = ck_links_of_word(white, Mt, Ub, 0xFFFFFFFFu, false, (Ubl >> 31) != 0, (Ubr & 1u) != 0, true);
    v.Ev = K.Ev; v.DL = K.DL & ~1u; v.DR = K.DR & 0x7FFFFFFFu;
    v.flags = (hl ? CK_LINK_HLEFT : 0u) | ((K.DL & 1u) ? CK_LINK_CROSS_L : 0u) | ((K.DR >> 31) ? CK_LINK_CROSS_R : 0u);
    return v;
}
__device__ __forceinline__ uint32_t links_left(const PairCtx &v) {
    return (uint32_t)__popc(v.Ev) + (uint32_t)__popc(v.DL) + (uint32_t)__popc(v.DR) + (uint32_t)__popc(v.flags);
}
// takes one link out of the context: a = an entry of the lower node, b = an entry of the node at the other end (links_left(v) > 0).
// Order: Ev, DL, DR (the compares are on the masks, the rest is branch-free), then the flags in ascending order.
__device__ __forceinline__ void take_link(PairCtx &v, uint32_t &a, uint32_t &b) {
    const uint32_t E = v.Ev ? v.Ev : (v.DL ? v.DL : v.DR);
    if (E) {
        const int x = __builtin_ctz(E);
        const int ux = v.Ev ? x : (v.DL ? x - 1 : x + 1);
        const uint32_t clr = ~(1u << x);
        const bool ev = v.Ev != 0, dl = v.DL != 0;
        v.Ev &= ev ? clr : 0xFFFFFFFFu;
        v.DL &= (!ev && dl) ? clr : 0xFFFFFFFFu;
        v.DR &= (!ev && !dl) ? clr : 0xFFFFFFFFu;
        a = lower_node(v, x); b = upper_node(v, ux);
    } else {
        const uint32_t f = v.flags & (0u - v.flags);
        v.flags &= v.flags - 1u;
        a = f == CK_LINK_CROSS_R ? lower_node(v, 31) : pair_entry(v.Mt, v.base, 0); // column 0 of a word always starts a node
        // (as masks: from a chain of selects over the struct's fields the compiler makes an indexed load from a copy in scratch memory)
        b = (v.left & (0u - (f & 1u))) | (v.up_l & (0u - ((f >> 1) & 1u))) | (v.up_r & (0u - ((f >> 2) & 1u)));
    }
}
// tile pixel of the entry that origin pixel (row r, word wd, bit i) of colour c finds: the lookup pixel of its node
__device__ __forceinline__ uint32_t node_lookup(const uint32_t *mk, const uint32_t *s2w, int r, int wd, int i, uint32_t c, uint32_t O) {
    const int p = r >> 1;
    const uint32_t S2 = s2w[(p * NWD + wd) * 2 + (int)c];
    const uint32_t Mt = mk[((2 * p) * NWD + wd) * 2 + (int)c] & O;
    return pair_entry(Mt, (uint32_t)((2 * p) * TW + 32 * wd), ck_run_start32(S2, i));
}
// The frame's two non-origin columns (0 and w - 1) inside a tile: lanes 0..31 of `q` (a 6-bit index) stand for the rows of column
// 0, lanes 32..63 for those of column w - 1.  Returns false when the tile does not hold that pixel or it has no colour; else the
// pixel's tile-local column and colour.
__device__ __forceinline__ bool edge_pixel(const uint32_t *mk, int q, int tx0, int ty0, int w, int h, int &r, int &xl, uint32_t &c) {
    r = q & 31;
    const int side = q >> 5;
    const int xf = side ? w - 1 : 0;
    if ((side && w == 1) || xf < tx0 || xf >= tx0 + TW || ty0 + r >= h) return false;
    xl = xf - tx0;
    const uint32_t Wm = mk[(r * NWD + (xl >> 5)) * 2], Bm = mk[(r * NWD + (xl >> 5)) * 2 + 1];
    const uint32_t wh = (Wm >> (xl & 31)) & 1u, bl = (Bm >> (xl & 31)) & 1u;
    c = wh ? 0u : 1u;
    return (wh | bl) != 0;
}

// Diagnostic build only (-DCK_TILE_PROFILE): per-phase cycle totals of k_tile in a buffer of their own.
#ifdef CK_TILE_PROFILE
__device__ unsigned long long g_tile_prof[16];
#define TPROF_DECL unsigned long long tp0 = __builtin_readcyclecounter()
#define TPROF(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_tile_prof[k], t_ - tp0); tp0 = t_; } while (0)
#define TCOUNT(k, v) do { if (threadIdx.x == 0) atomicAdd(&g_tile_prof[k], (unsigned long long)(v)); } while (0)
#else
#define TPROF_DECL
#define TPROF(k)
#define TCOUNT(k, v)
#endif

// A barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which the compiler turns into a wait for
// EVERY outstanding memory operation: the barrier behind the threshold store then waits for HBM to acknowledge it.  Nothing a
// workgroup of k_tile writes to global memory is read by it again.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// PRE = false: `frames` are gray images and the tri-state threshold is computed here;
// PRE = true : `frames` already hold a tri-state map (0 / 127 / 255), e.g. CAT's class map, and only the
//              segmentation runs (the map is copied through to `thresh` for the merge kernel).
template <bool PRE>
__global__ __launch_bounds__(KNT) __attribute__((amdgpu_num_sgpr(80))) void k_tile(const uint8_t *__restrict__ frames, size_t frame_pitch, int stride,
                                             int w, int h, int tiles_x, int tiles_y, int frame0, int n_frames, int xcd_map, int min_diff, int min_comp,
                                             uint8_t *__restrict__ thresh, ck_label_t *__restrict__ labels,
                                             ck_border_root *__restrict__ broots, uint32_t *__restrict__ tile_count,
                                             uint16_t *__restrict__ ring, size_t ring_len, int stop_after, int sweeps) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    // The waves' ROLES rotate with the workgroup: wave k of every workgroup sits on SIMD k, and the roles are not equally heavy (the
    // second min/max round, the pooled unions, the second round of ring pixels fall to the first waves) — with fixed roles SIMD 0
    // of every CU carried the most vector work, and the kernel is bound by vector issue (tools/probes/valu_rate_probe.hip).  Every
    // index below derives from `tid`; lane numbers (tid & 63) are the hardware's.  Round 4, same box: 0.9608 -> 0.9495 ms.
    const int tid = (int)((threadIdx.x + 64u * ((blockIdx.x >> 3) & 3u)) & 255u);
    const int tiles = tiles_x * tiles_y;
    int frame, tile;
    if (xcd_map) { // workgroups b and b + 8 share an XCD (their L2): deal whole frames to XCDs, so that the halo rows a tile shares
                   // with its neighbours are read from HBM once
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        frame = frame0 + (j / tiles) * 8 + x; tile = j % tiles;
        if (frame >= n_frames) return;
    } else { frame = blockIdx.x / tiles; tile = blockIdx.x - frame * tiles; frame += frame0; } // (frames [frame0, n_frames) of the batch)
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int tx0 = tx * TW, ty0 = ty * TH;
    const uint8_t *img = frames + (size_t)frame * frame_pitch;
    const size_t fbase = (size_t)frame * (size_t)w * (size_t)h;
    uint16_t *parent = reinterpret_cast<uint16_t *>(lds + OFF_PARENT);
    uint32_t *parent32 = reinterpret_cast<uint32_t *>(lds + OFF_PARENT);
    uint32_t *list = reinterpret_cast<uint32_t *>(lds + OFF_LIST); // [LIST_CAP]
    uint32_t *mk = reinterpret_cast<uint32_t *>(lds + OFF_MASK);
    uint32_t *misc = reinterpret_cast<uint32_t *>(lds + OFF_MISC); // [0] / [1] white / black nodes, [5] pooled links, [6] / [7] white / black ring-touching roots, [8] some 4x4 tile has contrast, [9] some pixel has a colour
    if (tid == 0) { misc[5] = 0; misc[6] = 0; misc[7] = 0; misc[8] = 0; misc[9] = 0; } // (the barrier after P0 publishes them)

    // Whole 4x4 tiles only enter the min/max (the oracle's rule); pixels right of / below the last whole one take its
    // threshold.  Normally the threshold grid of a workgroup tile starts at its own first 4x4 column / row (c4x, c4y); when the
    // last workgroup tile has only 1..3 columns / rows, that 4x4 tile lies in the tile on the left / above and the grids
    // start one 4x4 tile earlier.  The staged window always starts 16 columns left of the tile (aligned 16-byte chunks) and
    // one 4x4 row above the min/max grid.
    const int w4 = w >> 2, h4 = h >> 2;
    const int c4x = min(tx0 >> 2, w4 - 1), c4y = min(ty0 >> 2, h4 - 1); // first 4x4 column / row of the threshold grid
    const int sx0 = tx0 - 16, sy0 = 4 * (c4y - 1);                       // frame coordinates of staged byte (0, 0)
    const int yoff = ty0 - sy0;                                          // staged row of tile row 0: 4 (8 in the ragged last tile row)
    const int xb = 4 * (c4x - 1) - sx0;                                  // staged byte of min/max column 0: 12 (8 when ragged)

    // Two workgroup-uniform shortcuts (scalar branches): a tile that lies wholly inside the frame (`full`: its pixels need no bounds
    // tests — every tile of a 1280 x 800 frame) and one whose staged window does, too (`inner`: 74 % of them).  The tests are
    // compares and exec-mask bookkeeping on the kernel's critical resource, vector and scalar issue.
    const bool full = (w & 15) == 0 && (h & 3) == 0 && tx0 + TW <= w && ty0 + TH <= h;
    const bool inner = full && tx0 >= 16 && tx0 + TW + 16 <= w && ty0 >= 4 && ty0 + TH + 4 <= h;
    TPROF_DECL;
    // ---- P0: stage the tile and its halo: IMG_ROWS rows of ten 16-byte chunks (16 left | 128 | 16 right) ------------
    auto p0 = [&](auto INNERC) { // a lane's chunks are asked for together, then stored: as a loop (load, wait, store, load, wait, store)
        constexpr bool INNER = decltype(INNERC)::value; // the second chunk's trip to HBM started only when the first had come back
        constexpr int P0N = (IMG_ROWS * 10 + KNT - 1) / KNT;
        uint4 v[P0N];
#pragma unroll
        for (int q = 0; q < P0N; q++) {
            const int item = tid + q * KNT;
            const int r = item / 10, c = item - r * 10;
            const int gy = sy0 + r, gx = sx0 + 16 * c;
            v[q] = make_uint4(0, 0, 0, 0);
            // rows are padded to 16 bytes (frame strides are multiples of 16): a chunk that starts inside the row is readable
            if (item < IMG_ROWS * 10 && (INNER || (gy >= 0 && gy < h && gx >= 0 && gx < w))) v[q] = *reinterpret_cast<const uint4 *>(img + (size_t)gy * stride + gx);
        }
#pragma unroll
        for (int q = 0; q < P0N; q++) {
            const int item = tid + q * KNT;
            const int r = item / 10, c = item - r * 10;
            if (item < IMG_ROWS * 10) *reinterpret_cast<uint4 *>(lds + OFF_IMG + r * IMG_PITCH + 16 * c) = v[q];
        }
    };
    if (inner) p0(std::true_type{}); else p0(std::false_type{});
    lds_barrier();
    TPROF(0);

    if (stop_after == 0) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P1: min/max of the 4x4 tiles: grid column j = 4x4 column c4x - 1 + j, row i = 4x4 row c4y - 1 + i ----------
    uint32_t *minmax = reinterpret_cast<uint32_t *>(lds + OFF_MINMAX);
    auto p1 = [&](auto INNERC) {
    constexpr bool INNER = decltype(INNERC)::value;
    for (int item = tid; item < T4Y * T4X; item += KNT) {
        const int i = item / T4X, j = item - i * T4X;
        const int g4x = c4x - 1 + j, g4y = c4y - 1 + i;
        uint32_t mn = 255, mx = 0;
        if (INNER || (g4x >= 0 && g4x < w4 && g4y >= 0 && g4y < h4)) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                uint32_t d = *reinterpret_cast<const uint32_t *>(lds + OFF_IMG + (4 * i + rr) * IMG_PITCH + xb + 4 * j);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t v = (d >> (8 * b)) & 255u;
                    mn = min(mn, v); mx = max(mx, v);
                }
            }
        }
        minmax[item] = mn | (mx << 8); // outside the frame: (255,0) is neutral for the dilation
    }
    };
    if (!PRE) { if (inner) p1(std::true_type{}); else p1(std::false_type{}); }
    lds_barrier();

    // ---- P2: 3x3 dilation -> per-4x4-tile threshold word (bit 8 = low contrast); column j = 4x4 column c4x + j -------
    uint16_t *thr = reinterpret_cast<uint16_t *>(lds + OFF_THR);
    int my_contrast = 0;
    if (!PRE)
    for (int item = tid; item < (TH / 4) * (TW / 4); item += KNT) {
        const int i = item / (TW / 4), j = item - i * (TW / 4);
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
        my_contrast |= diff >= min_diff ? 1 : 0;
    }
    // a tile none of whose 4x4 tiles has contrast (a smooth background) thresholds to 127 everywhere: the threshold pass below
    // then only stores constants (every threshold word a pixel of this tile can look up is one of the 256 just written)
    // (a flag word set by one lane per wave before the barrier: __syncthreads_or costs two more barriers)
    if (__builtin_amdgcn_ballot_w64(my_contrast != 0) && (tid & 63) == 0) misc[8] = 1;
    lds_barrier();
    const int tile_contrast = PRE ? 1 : (int)misc[8];
    TPROF(1);

    if (stop_after == 1) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P3: threshold 16 pixels per thread, write them, build the rows' colour words -------------------------------
    uint16_t *mask16 = reinterpret_cast<uint16_t *>(lds + OFF_MASK); // [r][word][colour][half]
    const bool packed_rows = (w & 3) == 0; // rows of thresh[] / labels[] start 4-pixel aligned: vector stores
    // the lane's 16 threshold bytes to global memory
    auto store_thresh = [&](auto FULLC, int gy, int gx, const uint32_t (&out)[4]) {
        constexpr bool FULL = decltype(FULLC)::value;
        if (FULL || (gy < h && gx < w)) {
            uint8_t *dst = thresh + fbase + (size_t)gy * w + gx;
            if (FULL || (packed_rows && gx + 16 <= w && (w & 15) == 0)) *reinterpret_cast<uint4 *>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
            else if (packed_rows) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (gx + 4 * k < w) *reinterpret_cast<uint32_t *>(dst + 4 * k) = out[k];
            } else {
                for (int k = 0; k < 16; k++)
                    if (gx + k < w) dst[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
            }
        }
    };
    uint32_t any_colour = 0;
    int tile_has_runs = 0;
    auto p3 = [&](auto FULLC) {
        constexpr bool FULL = decltype(FULLC)::value;
        const int r = tid >> 3, c = tid & 7;
        const int gy = ty0 + r, gx = tx0 + 16 * c;
        uint4 px = *reinterpret_cast<const uint4 *>(lds + OFF_IMG + (r + yoff) * IMG_PITCH + 16 + 16 * c);
        uint32_t in[4] = {px.x, px.y, px.z, px.w}, out[4];
        uint32_t wbits = 0, bbits = 0;
        const int r4 = (FULL ? (gy >> 2) : min(gy >> 2, h4 - 1)) - c4y;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = gx + 4 * k;
            uint32_t tw_ = PRE ? 0u : thr[r4 * (TW / 4) + (FULL ? (x >> 2) : min(x >> 2, w4 - 1)) - c4x];
            uint32_t o, wn, bn; // the four output bytes and the white / black bits of the group
            if (!FULL && (gy >= h || x >= w)) { o = 0x7F7F7F7Fu; wn = 0; bn = 0; }     // outside the frame: no colour
            else {
                if (PRE) { o = in[k]; wn = msb_nibble(o); bn = msb_nibble(~(o << 7)); } // 255 -> bit 7 set; 0 -> bit 0 clear (127 and 255 have it set)
                else {
                    // four byte compares "pixel > threshold" as two packed 16-bit saturating subtractions (even and odd bytes)
                    const uint32_t t2 = (tw_ & 0xFFu) * 0x00010001u;
                    const uint32_t me = pk_gt_u16(in[k] & 0x00FF00FFu, t2, 0x00010001u), mo = pk_gt_u16((in[k] >> 8) & 0x00FF00FFu, t2, 0x00010001u); // 0 / 1 per pixel
                    const bool lowc = (tw_ & 0x100u) != 0;
                    const uint32_t b01 = me | (mo << 8);                 // one bit per pixel at the bottom of its byte ...
                    o = lowc ? 0x7F7F7F7Fu : (b01 << 8) - b01;           // ... times 255 (no byte carries into the next)
                    // the white bits straight from the compare results (pixels 0, 1 in bits 0, 1; pixels 2, 3 in bits 16, 17); a pixel of
                    // a tile with contrast is black when it is not white
                    const uint32_t pm = me | (mo << 1);
                    const uint32_t t = (pm | (pm >> 14)) & 15u;
                    wn = lowc ? 0u : t; bn = lowc ? 0u : (t ^ 15u);
                }
                if (!FULL && __builtin_amdgcn_ballot_w64(x + 4 > w)) // (scalar branch: only the wave at the frame's right edge, and only when
                    if (x + 4 > w) {                         // the width is not a multiple of 4: the frame ends inside this group)
                        const uint32_t keep = 0xFFFFFFFFu >> (8 * (x + 4 - w));
                        o = (o & keep) | (0x7F7F7F7Fu & ~keep);
                        wn = msb_nibble(o); bn = msb_nibble(~(o << 7));
                    }
            }
            out[k] = o;
            wbits |= wn << (4 * k);
            bbits |= bn << (4 * k);
        }
        store_thresh(FULLC, gy, gx, out);
        const int wd = c >> 1, half = c & 1;
        mask16[(((r * NWD + wd) * 2 + 0) << 1) + half] = (uint16_t)wbits;
        mask16[(((r * NWD + wd) * 2 + 1) << 1) + half] = (uint16_t)bbits;
        any_colour = wbits | bbits;
        // a tile without a coloured pixel has nothing to segment
        if (__builtin_amdgcn_ballot_w64(any_colour != 0) && (tid & 63) == 0) misc[9] = 1;
        lds_barrier();
        tile_has_runs = (int)misc[9];
    };
    if (tile_contrast) { if (full) p3(std::true_type{}); else p3(std::false_type{}); }
    else {
        const uint32_t out[4] = {0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu};
        if (full) store_thresh(std::true_type{}, ty0 + (tid >> 3), tx0 + 16 * (tid & 7), out);
        else store_thresh(std::false_type{}, ty0 + (tid >> 3), tx0 + 16 * (tid & 7), out);
    }
    TPROF(2);

    if (stop_after == 2) return; // diagnostics (CK_TILE_STOP_AFTER)
    uint32_t nruns = 0; // nodes in the tile's list
    constexpr int NPL = LIST_CAP / KNT;          // nodes per lane a tile may have for its label words to go through the table
    constexpr uint32_t KEEP_NONE = 0xFFFFFFFFu;
    uint32_t keep[NPL], ekeep = KEEP_NONE;       // per node this lane flattened: label-table index | root pixel << 16 (ekeep: its pixel of a non-origin column)
#pragma unroll
    for (int k = 0; k < NPL; k++) keep[k] = KEEP_NONE;
    bool tabled = false;
    uint32_t ring_root[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, ring_white[2] = {0, 0}; // P6b: roots under this lane's ring pixels
    uint32_t nwhite = 0; // white nodes: list[0 .. nwhite); the black ones are list[LIST_CAP - (nruns - nwhite) .. LIST_CAP)
    auto list_at = [&](uint32_t j) -> uint32_t { return list[j < nwhite ? j : j + (uint32_t)LIST_CAP - nruns]; };
    uint32_t *s2w = reinterpret_cast<uint32_t *>(lds + OFF_S2);
    if (tile_has_runs) {
    // ---- P4: the nodes (ck_links.h, "nodes over PAIRS of rows").  One lane per (pair of rows, word) and one wave per colour: wave 0
    // the white pair words, wave 1 the black ones (which have no diagonal links: scalar branches).  A lane gets what it needs of its
    // neighbours (the word on its left: lane - 1; the pair above: lanes - 5, - 4, - 3) from their registers.  Meanwhile a lane per
    // pixel of the frame's non-origin columns (wave 2; only the tiles at the frame's left and right edge have such pixels) makes
    // those pixels nodes of their own.
    // (a) The node starts of every pair word go to LDS (the later phases find a pixel's node with them).
    // (b) Adoption: every node takes ONE node of the pair above as parent — the one its first link leads to; which link that is
    // comes out of the word's link masks without a loop (first set bit of every segment of a word: one subtraction) — with a plain
    // store to the node's min entry: only the owner writes an entry and nothing reads parent[] in this phase, so no find and no atomic
    // is needed; the target lies in an earlier row, which keeps the forest invariant parent < self.  The node's lookup entry, when it
    // is another pixel, points at the min entry; the lookup pixel goes to the tile's node list.
    // (c) The links that are left over go to the pool for the atomic unions of P5c: one reservation per wave.
    uint32_t *pool = reinterpret_cast<uint32_t *>(lds + OFF_POOL);
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform
    const int p = (tid >> 2) & (TH / 2 - 1), wd = tid & 3;
    {
        // All four waves work: waves 0 and 2 on the white pair words, 1 and 3 on the black ones.  The two waves of a colour form the
        // same masks (what was two idle waves' time) and share the per-node loop — the even-numbered nodes of every word to the
        // first, the odd-numbered to the second — and the leftover links (vertical ones and the word-boundary flags to the first,
        // the diagonal ones to the second): the loop runs to HALF the busiest word's node count.
        // (Round 4, with the roles rotating: one wave per colour walking ALL of a word's nodes and links while the other two skip the
        // phase — 550 fewer wave-instructions per workgroup — measured 3.6 % SLOWER, 0.990 against 0.955 ms: the longer chain of the
        // two working waves costs more than the duplicated mask work.  And at the round's end: the first wave of a colour forming the masks
        // alone and handing the second its first links, diagonal masks and list offset through four words per pair word in the middle of
        // the list array, behind one more barrier — 2/5 of the phase's instructions once per pair word instead of twice — measured
        // 0.7 % slower, 0.8713 against 0.8655 ms: tools/wip/p4_prolog_handoff.patch.)
        const int c = __builtin_amdgcn_readfirstlane((tid >> 6) & 1), half = __builtin_amdgcn_readfirstlane(tid >> 7);
        const bool white = c == 0, has_l = wd > 0, has_u = p > 0, has_r = wd < NWD - 1;
        const int mi = ((2 * p) * NWD + wd) * 2 + c;
        const uint32_t O = ck_origin32(tx0 + 32 * wd, w);
        const uint32_t Mt = mk[mi] & O, Mb = mk[mi + 2 * NWD] & O;
        const uint32_t base = (uint32_t)((2 * p) * TW + 32 * wd), upbase = base - 2u * TW;
        const int lane4 = (tid & 63) * 4;
        uint32_t l31 = dpp0<0x111, 0xF>((Mt >> 31) | ((Mb >> 31) << 1)); // lane - 1 (the same group of four lanes when it is used)
        l31 = has_l ? l31 : 0u;
        const uint32_t link = ck_pair_link32(white, Mt, Mb, (l31 & 1u) != 0, (l31 & 2u) != 0);
        const uint32_t S2 = ck_pair_starts32(Mt, Mb, link);
        s2w[(p * NWD + wd) * 2 + c] = S2;
        // what the neighbours need of this word: the lookup pixel of its last node, pixel 31 / pixel 0 of its bottom row, pixel 0 of its top row
        const uint32_t info = pair_entry(Mt, base, ck_last_start32(S2)) | ((Mb >> 31) << 12) | ((Mb & 1u) << 13) | ((Mt & 1u) << 14);
        const uint32_t info_l = dpp0<0x111, 0xF>(info);
        uint32_t info_ul = (uint32_t)__builtin_amdgcn_ds_bpermute(lane4 - 20, (int)info), info_ur = (uint32_t)__builtin_amdgcn_ds_bpermute(lane4 - 12, (int)info);
        uint32_t Ut = (uint32_t)__builtin_amdgcn_ds_bpermute(lane4 - 16, (int)Mt), Ub = (uint32_t)__builtin_amdgcn_ds_bpermute(lane4 - 16, (int)Mb);
        uint32_t S2u = (uint32_t)__builtin_amdgcn_ds_bpermute(lane4 - 16, (int)S2);
        Ut = has_u ? Ut : 0u; Ub = has_u ? Ub : 0u; S2u = has_u ? S2u : 0u;
        info_ul = (has_u && has_l) ? info_ul : 0u; info_ur = (has_u && has_r) ? info_ur : 0u;
        const ck_word_links K = ck_links_of_word(white, Mt, Ub, 0xFFFFFFFFu, false, ((info_ul >> 12) & 1u) != 0, ((info_ur >> 13) & 1u) != 0, true);
        const uint32_t DL = K.DL & ~1u, DR = K.DR & 0x7FFFFFFFu; // (the two links across a word boundary are flags)
        // Several one-row runs of a node can face the same node of the pair above: of the vertical links between one pair of nodes
        // only the first is kept — the first set bit of every segment of the two start masks taken together.  (First set bit of
        // every segment [start, next start) of a word: with a stopper at every segment's last column, subtracting the start bits
        // runs a borrow up to exactly that bit.)
        const uint32_t Sc = S2 | S2u;
        const uint32_t Evp = K.Ev | (Sc >> 1) | 0x80000000u;
        const uint32_t Ev = Evp & ~(Evp - Sc) & K.Ev;
        // the first link of every node: the one it adopts its parent through
        const uint32_t E = Ev | DL | DR;
        const uint32_t Ep = E | (S2 >> 1) | 0x80000000u;
        const uint32_t F = Ep & ~(Ep - S2) & E;
        const uint32_t Fm = F & DL, Fp = F & DR & ~DL; // ... goes up-left / up-right (else straight up)
        // (a), (b)
        const uint32_t cnt = (uint32_t)__popc(S2);
        const uint32_t incl = wave_scan_u32(cnt);
        if ((tid & 63) == 63) misc[c] = incl;
        uint32_t li = c ? (uint32_t)(LIST_CAP - 1) - (incl - cnt) : incl - cnt; // white from the front, black from the back
        // (list entry: lookup pixel | label-table index << 12 | pixel count << 25; the table index of a node is
        // ((pair word * 2 + colour) * TAB_STRIDE + its number among the word's nodes of that colour): what a pixel of P7 forms with one popcount)
        const uint32_t lstep = c ? 0xFFFFFFFFu : 1u;
        uint32_t lcol = base | ((uint32_t)((((p * NWD + wd) * 2 + c) * TAB_STRIDE) + half) << 12);
        const uint32_t nMt = ~Mt, nUt = ~Ut;
        uint32_t St = S2;
        if (half) { St &= St - 1u; li += lstep; } // the second wave of the colour starts at the word's second node
        for (; St; St &= St - 1u, li += 2u * lstep, lcol += 2u << 12) { // (the step skips the other wave's node)
            const uint32_t low = St & (0u - St);
            const uint32_t s = ffbl_raw(low);
            St ^= low;
            const uint32_t span = ((St & (0u - St)) - 1u) & ~(low - 1u);     // the node's columns: bit s up to the next start
            const uint32_t lk = s | (((nMt >> s) & 1u) << 7);                 // word-local: + TW for the bottom row
            const uint32_t mn = min(ffbl_raw(Mt & span), lk);                 // its first top pixel, if it has one (ffbl of 0 is 0xFFFFFFFF)
            const uint32_t e = F & span;                                      // the link it adopts through (at most one bit)
            const uint32_t ue = (e & ~(Fm | Fp)) | ((e & Fm) >> 1) | ((e & Fp) << 1); // the pixel of the row above at its other end
            const uint32_t su = 31u - (uint32_t)__builtin_clz((S2u & ((ue << 1) - 1u)) | 1u);
            const uint32_t t_up = upbase + su + (((nUt >> su) & 1u) << 7);
            parent[base + lk] = (uint16_t)(base + mn);           // (in this order: the two are one entry when the node's first column has a top pixel)
            parent[base + mn] = (uint16_t)(e ? t_up : CK_ROOT);  // no link to an earlier node: a root (count 0 for now)
            // the list entry: lookup pixel | colour << 12 | pixel count << 16.  (A tile with more nodes than the list holds — one-pixel
            // patterns — does without it: the index then runs past either end and nothing is stored.)
            if (li < (uint32_t)LIST_CAP) list[li] = lk | lcol | (((uint32_t)__popc(Mt & span) + (uint32_t)__popc(Mb & span)) << 25);
        }
        // (c) what is left: the vertical links one loop, the two diagonal kinds one loop each, then the three links that cross a
        // word boundary (a generic "take the next link" loop cost twice the instructions per link)
        const uint32_t lEv = half ? 0u : Ev & ~F, lDL = half ? DL & ~F : 0u, lDR = half ? DR & ~(F & ~DL) : 0u;
        const uint32_t fl = half ? 0u : ((link & 1u) ? CK_LINK_HLEFT : 0u) | ((K.DL & 1u) ? CK_LINK_CROSS_L : 0u) | ((K.DR >> 31) ? CK_LINK_CROSS_R : 0u);
        const uint32_t extra = (uint32_t)__popc(lEv) + (uint32_t)__popc(lDL) + (uint32_t)__popc(lDR) + (uint32_t)__popc(fl);
        const uint32_t xincl = wave_scan_u32(extra);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)xincl, 63);
        if (total) {
            uint32_t wb = 0;
            if ((tid & 63) == 63) wb = atomicAdd(&misc[5], total);
            wb = (uint32_t)__builtin_amdgcn_readlane((int)wb, 63);
            uint32_t pos = wb + xincl - extra;
            // an entry of the node that holds column x of this pair word / of the pair word above
            auto lower = [&](uint32_t x) -> uint32_t { const uint32_t st = 31u - (uint32_t)__builtin_clz((S2 & ((2u << x) - 1u)) | 1u); return base + st + (((nMt >> st) & 1u) << 7); };
            auto upper = [&](uint32_t x) -> uint32_t { const uint32_t st = 31u - (uint32_t)__builtin_clz((S2u & ((2u << x) - 1u)) | 1u); return upbase + st + (((nUt >> st) & 1u) << 7); };
            auto push = [&](uint32_t a_, uint32_t b_) { if (pos < (uint32_t)POOL_CAP) pool[pos] = a_ | (b_ << 16); pos++; };
            for (uint32_t m = lEv; m; m &= m - 1u) { const uint32_t x = (uint32_t)__builtin_ctz(m); push(lower(x), upper(x)); }
            for (uint32_t m = lDL; m; m &= m - 1u) { const uint32_t x = (uint32_t)__builtin_ctz(m); push(lower(x), upper(x - 1u)); }
            for (uint32_t m = lDR; m; m &= m - 1u) { const uint32_t x = (uint32_t)__builtin_ctz(m); push(lower(x), upper(x + 1u)); }
            if (fl & CK_LINK_HLEFT) push(base + (((nMt >> 0) & 1u) << 7), info_l & 0xFFFu);          // (column 0 of a word always starts a node)
            if (fl & CK_LINK_CROSS_L) push(base + (((nMt >> 0) & 1u) << 7), info_ul & 0xFFFu);
            if (fl & CK_LINK_CROSS_R) push(lower(31u), upbase + 32u + (((info_ur >> 14) & 1u) ? 0u : (uint32_t)TW));
        }
    }
    if (tid >= 2 * 64 && tid < 3 * 64 && (tx0 == 0 || tx0 + TW >= w)) { // (a lane per pixel of the frame's non-origin columns: entries no pair-word node owns)
        int r, xl; uint32_t ec;
        if (edge_pixel(mk, tid & 63, tx0, ty0, w, h, r, xl, ec)) parent[r * TW + xl] = (uint16_t)CK_ROOT;
    }
    lds_barrier();
    nwhite = misc[0];
    nruns = nwhite + misc[1];
    TPROF(4);
    if (stop_after == 4) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P5b: one pointer-jumping sweep over the (static) adoption forest: a node's parent lies in an earlier pair (or earlier in
    // its own), so the chains down a tag edge (one hop per pair, two where a lookup entry sits in between) are shortened before
    // the finds of the next two phases walk them
    tabled = nruns <= (uint32_t)LIST_CAP;
    if (tabled)
    for (int sweep = 0; sweep < sweeps; sweep++)
    for (uint32_t j = (uint32_t)tid; j < nruns; j += KNT) {
        const uint32_t pe = list_at(j) & 0xFFFu;
        const uint32_t q = parent[pe];
        const uint32_t g = parent[q & 0xFFFu]; // (a root's own entry has no parent to look at: the read is harmless, the result unused)
        if (!(q & CK_ROOT) && !(g & CK_ROOT)) parent[pe] = (uint16_t)g;
    }
    if (sweeps) lds_barrier(); // (uniform)
    TPROF(5);
    if (stop_after == 5) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P5c: the pooled links, one lane per link, through the atomic union ------------------------------------------
    const uint32_t npool = misc[5];
    {
        const uint32_t pooled = npool < (uint32_t)POOL_CAP ? npool : (uint32_t)POOL_CAP;
        for (uint32_t j = (uint32_t)tid; j < pooled; j += KNT) {
            const uint32_t e = pool[j];
            lds_union(parent, e & 0xFFFFu, e >> 16);
        }
        if (npool > (uint32_t)POOL_CAP) // more links than the pool holds (pathological maps): every pair word joins all its links again
            if (tid < 2 * 64) {
                PairCtx v = load_pair(mk, s2w, p, wd, c, tx0, w);
                for (uint32_t nl = links_left(v); nl; nl--) {
                    uint32_t a_, b_;
                    take_link(v, a_, b_);
                    lds_union(parent, a_, b_);
                }
            }
        // the pixels of the frame's non-origin columns: joined by what the rule lets their origin neighbours do — (1, y) joins its left
        // neighbour (0, y); white (1, y + 1) joins up-left (0, y); white (w - 2, y + 1) joins up-right (w - 1, y).  (A neighbour in the
        // next tile is the merge stage's business.)
        if (tid >= 3 * 64 && (tx0 == 0 || tx0 + TW >= w) && w >= 3) {
            int r, xl; uint32_t ec;
            if (edge_pixel(mk, tid & 63, tx0, ty0, w, h, r, xl, ec)) {
                const uint32_t me = (uint32_t)(r * TW + xl);
                const int xn = xl == 0 && tx0 == 0 ? 1 : xl - 1; // the origin column next to it (tile-local; -1: in the tile on the left)
                if (xn >= 0) {
                    const uint32_t On = ck_origin32(tx0 + (xn & ~31), w);
                    if (xl == 0 && tx0 == 0 && ((mk[(r * NWD) * 2 + (int)ec] >> 1) & 1u)) lds_union(parent, me, node_lookup(mk, s2w, r, 0, 1, ec, On));
                    if (ec == 0u && r + 1 < TH && ty0 + r + 1 < h && ((mk[((r + 1) * NWD + (xn >> 5)) * 2] >> (xn & 31)) & 1u))
                        lds_union(parent, me, node_lookup(mk, s2w, r + 1, xn >> 5, xn & 31, 0u, On));
                }
            }
        }
    }
    lds_barrier(); // halving stores must land before the owners publish final roots; the pool is dead
    TCOUNT(9, nruns); TCOUNT(10, npool);
    TPROF(6);
    if (stop_after == 6) return; // diagnostics (CK_TILE_STOP_AFTER)
    // ---- P6: flatten the nodes' lookup entries and add their pixels into their roots' entries; two nodes per lane and round so
    // that two root walks are in flight (a lane past the end walks node 0 again and writes nothing).  The lane keeps each node's
    // lookup pixel and root in registers: the label words are formed once per NODE after the ring-touching roots have their ids
    // (P6c) and the label pass reads them from a table (P7).  A tile with more nodes than the list holds (LIST_CAP = NPL nodes per
    // lane: one-pixel patterns; dense binary noise has four per lane) walks its nodes per pair word instead, and its label pass
    // looks every pixel's root up itself.
    auto flatten2 = [&](uint32_t j0, uint32_t &keep0, uint32_t &keep1) {
        uint32_t node[2], root[2], add[2], tix[2];
        bool live[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t j = j0 + (uint32_t)(q * KNT + tid);
            live[q] = j < nruns;
            const uint32_t e = list_at(live[q] ? j : 0u);
            node[q] = e & 0xFFFu; root[q] = node[q];
            add[q] = e >> 25; tix[q] = (e >> 12) & 0x1FFFu;
        }
        for (int it = 0; it < TH * TW; it++) { // plain loads behind a compiler barrier: the two reads of a step go out together
            __asm__ volatile("" ::: "memory");
            const uint32_t n0 = u16_at(parent, root[0]), n1 = u16_at(parent, root[1]);
            const bool d0 = (n0 & CK_ROOT) != 0, d1 = (n1 & CK_ROOT) != 0;
            if (d0 && d1) break;
            root[0] = d0 ? root[0] : n0; root[1] = d1 ? root[1] : n1;
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
            if (live[q]) {
                if (root[q] != node[q]) parent[node[q]] = (uint16_t)root[q];
                // the root's own entry takes the count (bits 0..12: a tile has 4096 pixels, so no carry reaches the flags or the
                // other half of the word)
                atomicAdd(&parent32[root[q] >> 1], add[q] << ((root[q] & 1u) * 16u));
            }
        keep0 = live[0] ? (tix[0] | (root[0] << 16)) : KEEP_NONE;
        keep1 = live[1] ? (tix[1] | (root[1] << 16)) : KEEP_NONE;
    };
    if (tabled) {
#pragma unroll
        for (int k = 0; k < NPL / 2; k++)
            if ((uint32_t)(k * 2 * KNT) < nruns) flatten2((uint32_t)(k * 2 * KNT), keep[2 * k], keep[2 * k + 1]);
    } else if (tid < 2 * 64) { // one lane per pair word, as in P4
        const uint32_t O = ck_origin32(tx0 + 32 * wd, w);
        const int mi = ((2 * p) * NWD + wd) * 2 + c;
        const uint32_t Mt = mk[mi] & O, Mb = mk[mi + 2 * NWD] & O, S2 = s2w[(p * NWD + wd) * 2 + c];
        const uint32_t base = (uint32_t)((2 * p) * TW + 32 * wd);
        for (uint32_t St = S2; St; St &= St - 1u) {
            const int s = __builtin_ctz(St);
            const uint32_t span = ck_span32(S2, s);
            const uint32_t node = pair_entry(Mt, base, s);
            uint32_t root = node;
            for (;;) { __asm__ volatile("" ::: "memory"); const uint32_t n0 = parent[root]; if (n0 & CK_ROOT) break; root = n0; }
            if (root != node) parent[node] = (uint16_t)root;
            atomicAdd(&parent32[root >> 1], ((uint32_t)__popc(Mt & span) + (uint32_t)__popc(Mb & span)) << ((root & 1u) * 16u));
        }
    }
    if (tid >= 3 * 64 && (tx0 == 0 || tx0 + TW >= w)) { // the non-origin columns' pixels: one pixel each
        int r, xl; uint32_t ec;
        if (edge_pixel(mk, tid & 63, tx0, ty0, w, h, r, xl, ec)) {
            const uint32_t me = (uint32_t)(r * TW + xl);
            uint32_t root = me;
            for (;;) { __asm__ volatile("" ::: "memory"); const uint32_t n0 = parent[root]; if (n0 & CK_ROOT) break; root = n0; }
            if (root != me) parent[me] = (uint16_t)root;
            atomicAdd(&parent32[root >> 1], 1u << ((root & 1u) * 16u));
            ekeep = (uint32_t)(TAB_EDGE + (tid & 63)) | (root << 16); // (its own table entry: side * 32 + row)
        }
    }
    lds_barrier();
    if (stop_after == 65) return; // diagnostics (CK_TILE_STOP_AFTER): flatten + sizes done, ids not yet
    // ---- P6b: the components that touch the tile ring get their tile-local id (= place in the tile's slice of the frame's list,
    // which takes the root's pixel and the component's pixel count).  One lane per RING PIXEL (320 of them, not one per run): it
    // finds its component's root and claims it (atomic OR of CK_CLAIM on the root's entry); the lane that wins the claim draws
    // the id and replaces the count in the root's entry by it (the flags stay).  The lane keeps the pixel's ring entry
    // (id | colour << 15) for the merge stage in a register; ids are complete after the barrier.
    {
        ck_border_root *slice = broots + ((size_t)frame * 2 * tiles + tile) * RING_CAP; // [frame][slices | packed copy]
        // (which ring pixel a lane looks at is uniform per wave in the first round — waves 0 and 1 the tile's top row, 2 and 3 its
        // bottom row — and one compare in the second, which only wave 0 runs: the left column on lanes 0..31, the right one on 32..63;
        // as a chain of per-lane selects over a running item number this bookkeeping was a fifth of the phase)
        const int wv6 = __builtin_amdgcn_readfirstlane(tid >> 6);
        auto ring_pixel = [&](const int rnd, const int rr, const int xx, const bool side) {
            if (!full && (ty0 + rr >= h || tx0 + xx >= w)) return;
            const int wd = xx >> 5, i = xx & 31;
            const uint32_t Wm = mk[(rr * NWD + wd) * 2], Bm = mk[(rr * NWD + wd) * 2 + 1];
            const uint32_t white = (Wm >> i) & 1u;
            // a pixel on a side behind which another tile lies makes its component ring-touching (that is the definition)
            if (!side || !(((Wm | Bm) >> i) & 1u)) return;
            const uint32_t Oo = ck_origin32(tx0 + 32 * wd, w);
            const uint32_t node = ((Oo >> i) & 1u) ? node_lookup(mk, s2w, rr, wd, i, white ^ 1u, Oo) : (uint32_t)(rr * TW + xx); // (a non-origin column's pixel is a node of its own)
            const uint32_t e = parent[node];
            const uint32_t root = (e & CK_ROOT) ? node : e; // flat since P6
            ring_root[rnd] = root; ring_white[rnd] = white;
            const uint32_t sh = (root & 1u) * 16u;
            const uint32_t re = (parent32[root >> 1] >> sh) & 0xFFFFu;
            if (re & CK_CLAIM) return;                        // claimed already
            const uint32_t old = atomicOr(&parent32[root >> 1], (CK_CLAIM | CK_RING) << sh);
            if ((old >> sh) & CK_CLAIM) return;               // another lane won
            // white roots take the ids 0, 1, ... and black ones RING_CAP - 1, RING_CAP - 2, ... (the merge stage joins the two
            // colours in separate workgroups); they cannot meet: a ring-touching root owns at least one of the RING_CAP ring pixels
            const uint32_t id = white ? atomicAdd(&misc[6], 1u) : (uint32_t)(RING_CAP - 1) - atomicAdd(&misc[7], 1u);
            ck_border_root br;
            br.root = (uint32_t)(ty0 + (int)(root >> 7)) * (uint32_t)w + (uint32_t)(tx0 + (int)(root & (TW - 1)));
            br.size = (old >> sh) & CK_COUNT;
            slice[id] = br;
            parent[root] = (uint16_t)(CK_ROOT | CK_RING | CK_CLAIM | id);
        };
        ring_pixel(0, wv6 >= 2 ? TH - 1 : 0, tid & (TW - 1), wv6 >= 2 ? ty0 + TH < h : ty0 > 0);
        if (wv6 == 0) ring_pixel(1, tid & 31, (tid & 32) ? TW - 1 : 0, (tid & 32) ? tx0 + TW < w : tx0 > 0);
    }
    lds_barrier();
    } // tile_has_runs
    if (tid == 0) tile_count[(size_t)frame * tiles + tile] = misc[6] | (misc[7] << 16); // white | black << 16
    TPROF(7);

    if (stop_after == 7) return; // diagnostics (CK_TILE_STOP_AFTER)
    uint16_t *ring_f = ring + (size_t)frame * ring_len;
    // the ring entries of the pixels this lane looked at in P6b (rows: HT / HB, columns: VL / VR; layout in ck_internal.h)
    {
        const int wv6 = __builtin_amdgcn_readfirstlane(tid >> 6);
        auto ring_out = [&](const int rnd, const int rr, const int xx, const size_t dst) {
            if (!full && (ty0 + rr >= h || tx0 + xx >= w)) return;
            uint32_t val = 0xFFFFu;
            if (tile_has_runs && ring_root[rnd] != 0xFFFFFFFFu) val = ((uint32_t)parent[ring_root[rnd]] & 0x1FFu) | (ring_white[rnd] << 15);
            ring_f[dst] = (uint16_t)val;
        };
        const int xx0 = tid & (TW - 1);
        ring_out(0, wv6 >= 2 ? TH - 1 : 0, xx0, (wv6 >= 2 ? (size_t)tiles_y * w : (size_t)0) + (size_t)ty * w + (size_t)(tx0 + xx0));
        if (wv6 == 0) {
            const int rr1 = tid & 31;
            const bool right = (tid & 32) != 0;
            ring_out(1, rr1, right ? TW - 1 : 0, 2 * (size_t)tiles_y * w + (right ? (size_t)tiles_x * h : (size_t)0) + (size_t)tx * h + (size_t)(ty0 + rr1));
        }
    }
    if (!tile_has_runs || tabled) {
    // ---- P6c: the label word of every node (16 bits: ck_internal.h), formed once per node from its root's entry (interior component:
    // the root's tile pixel, final; ring-touching: the component's tile-local id) and, behind a barrier — every read of the union-find is done —
    // written to a table of 32-bit words over the parent and list arrays, indexed by pair word, colour and the node's number among
    // the word's nodes of that colour (TAB_EDGE / TAB_NONE above).
    uint32_t *tab32 = reinterpret_cast<uint32_t *>(lds);
    if (tile_has_runs) {
        auto label_word = [&](uint32_t kp) -> uint32_t {
            const uint32_t root = kp >> 16;
            const uint32_t ce = parent[root & (uint32_t)(TH * TW - 1)]; // (KEEP_NONE: a harmless in-range read, the word is not used)
            return (ce & CK_RING) ? CK_LBL_BORDER | (ce & CK_LBL_ID_MASK)
                                  : ((root & CK_LBL_LOCAL_MASK) | ((int)(ce & CK_COUNT) < min_comp ? CK_LBL_SMALL : 0u));
        };
        uint32_t lw[NPL], elw;
#pragma unroll
        for (int k = 0; k < NPL; k++) { // (keep[k] holds list entries k * KNT ..: the rounds past the tile's node count are skipped as a whole)
            lw[k] = 0;
            if ((uint32_t)(k * KNT) < nruns) lw[k] = label_word(keep[k]);
        }
        elw = label_word(ekeep);
        lds_barrier();
#pragma unroll
        for (int k = 0; k < NPL; k++)
            if ((uint32_t)(k * KNT) < nruns && keep[k] != KEEP_NONE) tab32[keep[k] & 0x1FFFu] = lw[k];
        if (ekeep != KEEP_NONE) tab32[ekeep & 0x1FFFu] = elw;
        if (tid == 0) tab32[TAB_NONE] = 0xFFFFu;
        lds_barrier();
    }
    // ---- P7: write label words.  Lane L owns the 8-column group L & 15 of pair L >> 4, both rows: the search for a column's node
    // (per colour: the nearest node start at or below it, then the row of that node's lookup pixel) is shared by the column's two
    // pixels, ONE table lookup per pixel, and a lane's eight 16-bit words of a row leave in one 16-byte store.
    {
        const int g = tid & 15, pr = tid >> 4, wd = g >> 2, sh = 8 * (g & 3);
        const int gx = tx0 + 8 * g, gy = ty0 + 2 * pr;
        const uint32_t Oo = ck_origin32(tx0 + 32 * wd, w);
        const uint32_t edge8 = (~Oo >> sh) & 255u;    // columns of the group that are non-origin columns of the frame: their pixels are nodes of their own
        if (gx < w && gy < h) {
            uint32_t outw[2][4]; // two 16-bit words each
#pragma unroll
            for (int k = 0; k < 4; k++) outw[0][k] = outw[1][k] = 0xFFFFFFFFu;
            if (tile_has_runs) {
                const uint2 m4 = *reinterpret_cast<const uint2 *>(&mk[((2 * pr) * NWD + wd) * 2]);          // top row: white, black word ...
                const uint2 b4 = *reinterpret_cast<const uint2 *>(&mk[((2 * pr + 1) * NWD + wd) * 2]);      // ... and the bottom row's
                const uint32_t any_t = ((m4.x | m4.y) >> sh) & 255u, any_b = ((b4.x | b4.y) >> sh) & 255u;
                if (any_t | any_b) {
                    const uint2 s2 = *reinterpret_cast<const uint2 *>(&s2w[(pr * NWD + wd) * 2]);
                    // A pixel's table entry is its node's: (pair word * 2 + colour) * TAB_STRIDE + the node's number among the word's nodes of
                    // that colour = the node starts at or below the pixel's column, less one: ONE popcount per column and colour, on the
                    // start mask shifted to the group (the masks of the eight columns are constants; v_bcnt adds the base itself).
                    // (Round 4: the table had been indexed by the node's lookup pixel — per column and colour a count-leading-zeros
                    // for the start, the row of the lookup pixel from a second mask, three shifts and two three-operand ors.)
                    const uint32_t lowm = (1u << sh) - 1u;
                    const uint32_t tb = (uint32_t)(((pr * NWD + wd) * 2) * TAB_STRIDE);
                    const uint32_t bw = tb + (uint32_t)__popc(s2.x & lowm) - 1u, bb = tb + (uint32_t)TAB_STRIDE + (uint32_t)__popc(s2.y & lowm) - 1u;
                    const uint32_t sw8 = s2.x >> sh, sb8 = s2.y >> sh;
                    const uint32_t wt8 = m4.x >> sh, bt8 = m4.y >> sh, wb8 = b4.x >> sh, bb8 = b4.y >> sh;
                    uint32_t at[2][8];
                    // Which entry a pixel reads — its column's white node, its black node or the one that says "no colour" — is chosen
                    // without compares: a pixel's colour bit spread over a register (one signed bit-field extract) picks between two
                    // values through one three-input bit operation, x ^ (mask & (x ^ y)).  (Round 4: as ?: it was a mask, a compare and
                    // a select per level, two levels per pixel.)  A lane all of whose sixteen pixels have a colour — every lane of a
                    // frame of dense noise — needs one level.
                    // (as an instruction: from the shifts the compiler recognises a select and makes the compare and v_cndmask again)
                    auto spread = [](uint32_t bits, int k) -> uint32_t { uint32_t m; __asm__("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(k)); return m; };
                    if ((any_t & any_b) == 255u) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const uint32_t upto = (2u << k) - 1u;
                            const uint32_t iw = bw + (uint32_t)__popc(sw8 & upto), ib = bb + (uint32_t)__popc(sb8 & upto);
                            const uint32_t x = iw ^ ib;
                            at[0][k] = ib ^ (spread(wt8, k) & x);
                            at[1][k] = ib ^ (spread(wb8, k) & x);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const uint32_t upto = (2u << k) - 1u;
                            const uint32_t iw = bw + (uint32_t)__popc(sw8 & upto), ib = bb + (uint32_t)__popc(sb8 & upto);
                            // (a pixel without a colour reads the entry that says so; a pixel has at most one colour)
                            const uint32_t xw = iw ^ (uint32_t)TAB_NONE, xb = ib ^ (uint32_t)TAB_NONE;
                            at[0][k] = (uint32_t)TAB_NONE ^ (spread(wt8, k) & xw) ^ (spread(bt8, k) & xb);
                            at[1][k] = (uint32_t)TAB_NONE ^ (spread(wb8, k) & xw) ^ (spread(bb8, k) & xb);
                        }
                    }
                    if (edge8) { // (only the lanes at the frame's first and last column: those pixels are nodes of their own)
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if ((edge8 >> k) & 1u) { // (the frame's column 0 is side 0 of the edge entries, its last column side 1)
                                const uint32_t eb = (uint32_t)TAB_EDGE + (gx + k == 0 ? 0u : 32u) + (uint32_t)(2 * pr);
                                at[0][k] = ((any_t >> k) & 1u) ? eb : (uint32_t)TAB_NONE;
                                at[1][k] = ((any_b >> k) & 1u) ? eb + 1u : (uint32_t)TAB_NONE;
                            }
                    }
                    uint32_t lwv[2][8];
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int k = 0; k < 8; k++) lwv[r][k] = tab32[at[r][k]];
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int k = 0; k < 4; k++) outw[r][k] = lwv[r][2 * k] | (lwv[r][2 * k + 1] << 16);
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (gy + r >= h || stop_after == 98) continue; // (98: diagnostics, everything but the label stores)
                ck_label_t *dst = labels + fbase + (size_t)(gy + r) * w + gx;
                if ((w & 7) == 0) { // rows of labels[] start 8-pixel aligned: one 16-byte store
                    __asm__ volatile("" ::: "memory"); // (keeps the compiler from merging this store with the per-pixel ones of the other branch)
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(outw[r][0], outw[r][1], outw[r][2], outw[r][3]);
                    __asm__ volatile("" ::: "memory");
                } else {
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (gx + k < w) dst[k] = (ck_label_t)(outw[r][k >> 1] >> (16 * (k & 1)));
                }
            }
        }
    }
    } else {
    // a tile with more nodes than its lanes keep (one-pixel patterns): the label pass looks every pixel's root up itself
    // ---- P7 (direct): Two passes; in pass q lane L owns the 4-column group L & 31 of pair q * 8 + (L >> 5): both rows, so
    // that the search for a column's node (per colour: the nearest node start at or below it, then the row of that node's lookup
    // entry) is shared by the column's two pixels; a wave's store instruction covers two stretches of 512 contiguous bytes.  A
    // pixel's entry is either its component's root entry (CK_ROOT | flags | count or id) or the root's pixel: at most two lookups.
    {
        const int g = tid & 31, wd = g >> 3, sh = 4 * (g & 7);       // the same for the lane's two passes
        const int gx = tx0 + 4 * g;
        const uint32_t Oo = ck_origin32(tx0 + 32 * wd, w);
        const uint32_t below = (1u << sh) - 1u;
        const uint32_t edge4 = (~Oo >> sh) & 15u;     // columns of the group that are non-origin columns of the frame: their pixels are nodes of their own
        if (gx < w)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int pr = q * (KNT / 32) + (tid >> 5);
            const int gy = ty0 + 2 * pr;
            if (gy >= h) continue;
            const uint32_t pbase = (uint32_t)((2 * pr) * TW + 32 * wd), cbase = pbase + (uint32_t)sh;
            uint32_t outw[2][4];
#pragma unroll
            for (int k = 0; k < 4; k++) outw[0][k] = outw[1][k] = CK_LBL_NONE;
            if (tile_has_runs) {
                const uint2 m4 = *reinterpret_cast<const uint2 *>(&mk[((2 * pr) * NWD + wd) * 2]);          // top row: white, black word ...
                const uint2 b4 = *reinterpret_cast<const uint2 *>(&mk[((2 * pr + 1) * NWD + wd) * 2]);      // ... and the bottom row's
                const uint32_t col4[2][2] = {{(m4.x >> sh) & 15u, (m4.y >> sh) & 15u}, {(b4.x >> sh) & 15u, (b4.y >> sh) & 15u}}; // [row][colour]
                if (col4[0][0] | col4[0][1] | col4[1][0] | col4[1][1]) {
                    const uint2 s2 = *reinterpret_cast<const uint2 *>(&s2w[(pr * NWD + wd) * 2]);
                    const uint32_t Wt = m4.x & Oo, Bt = m4.y & Oo;
                    uint32_t cw = (uint32_t)ck_last_start32(s2.x & below), cb = (uint32_t)ck_last_start32(s2.y & below);
                    const uint32_t sw4 = (s2.x >> sh) & 15u, sb4 = (s2.y >> sh) & 15u;
                    uint32_t nodev[2][4], ev[2][4], rv[2][4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        cw = ((sw4 >> k) & 1u) ? (uint32_t)(sh + k) : cw;
                        cb = ((sb4 >> k) & 1u) ? (uint32_t)(sh + k) : cb;
                        const uint32_t atw = pbase + cw + (((Wt >> cw) & 1u) ? 0u : (uint32_t)TW), atb = pbase + cb + (((Bt >> cb) & 1u) ? 0u : (uint32_t)TW);
                        const bool edge = ((edge4 >> k) & 1u) != 0;
#pragma unroll
                        for (int r = 0; r < 2; r++) { // an uncoloured pixel looks up a harmless in-range entry; its word is not used
                            const uint32_t at = ((col4[r][0] >> k) & 1u) ? atw : atb;
                            nodev[r][k] = edge ? cbase + (uint32_t)(r * TW + k) : at;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int k = 0; k < 4; k++) ev[r][k] = parent[nodev[r][k]];
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int k = 0; k < 4; k++) rv[r][k] = parent[ev[r][k] & (uint32_t)(TH * TW - 1)]; // the root's entry when ev is a pixel (harmless otherwise)
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool is_root = (ev[r][k] & CK_ROOT) != 0;
                            const uint32_t node = is_root ? nodev[r][k] : ev[r][k]; // the component's root pixel
                            const uint32_t ce = is_root ? ev[r][k] : rv[r][k];      // its entry
                            // interior component: the root's pixel index, final; ring-touching: the component's slot in the frame's tables
                            // (as a branch: formed both ways and bit-selected it measured 1.2 % slower)
                            const uint32_t word = (ce & CK_RING) ? CK_LBL_BORDER | (ce & CK_LBL_ID_MASK)
                                                                 : ((node & CK_LBL_LOCAL_MASK) | ((int)(ce & CK_COUNT) < min_comp ? CK_LBL_SMALL : 0u));
                            outw[r][k] = (((col4[r][0] | col4[r][1]) >> k) & 1u) ? word : CK_LBL_NONE;
                        }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (gy + r >= h) continue;
                ck_label_t *dst = labels + fbase + (size_t)(gy + r) * w + gx;
                if (packed_rows) {
                    __asm__ volatile("" ::: "memory");
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(outw[r][0] | (outw[r][1] << 16), outw[r][2] | (outw[r][3] << 16));
                    __asm__ volatile("" ::: "memory");
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (gx + k < w) dst[k] = (ck_label_t)outw[r][k];
                }
            }
        }
    }
    }
    TPROF(8);
}

#ifdef CK_TILE_PROFILE
} // namespace
extern "C" int ck_tile_profile_read(unsigned long long *out, int reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_tile_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
namespace {
#endif

// ---- cross-tile merge: one workgroup per frame ------------------------------------------------------------------------------------
// k_tile left, for every frame: per tile the list of its ring-touching roots (broots slices: pixel index + pixel count; a
// root's place in its tile's slice is its tile-local id, slice * CK_RING_CAP + id its SLOT) and, along every tile boundary, the
// id and colour of the component each boundary pixel belongs to (ring).  A frame's workgroup numbers the roots consecutively
// (scan of the tiles' counts), replays the connectivity rule across the boundaries as unions over those numbers, and fills
// the frame's two tables, indexed by slot: groot[slot] = pixel index of the component's frame-level root (its smallest pixel:
// the canonical label), gsize[slot] = the component's pixel count (exact while below min_component_px).  The label words of
// ring-touching components carry the slot (ck_internal.h), so a consumer resolves one with two independent table reads.
// No label, threshold or per-pixel size word is read or written here.
//
// The unions run in LDS (u16 parents: up to FM_CAP roots per frame; dense noise at 1280x800 has about 22 000); a frame with
// more takes the same steps in global memory (fm_global_begin / fm_global_end).  The frame-level root must be the component's smallest
// pixel, so a union hooks the root with the larger PIXEL under the other one (compare-and-swap on the root's own entry).
constexpr int FM_NT = 1024;
constexpr int FM_CAP = 30000;                 // roots a workgroup's LDS path holds with 16-bit pixel keys beside the parents (twice as many without)
// Bands of tile rows: a frame with more than FM_BAND_MIN tiles is joined in bands of at most FM_BAND_TILES tiles.  What bands buy is the
// PATH, not parallelism (a batch fills the chip either way): with about 48 black ring-touching roots per tile of dense noise a colour's
// workgroup keeps parents and keys in LDS up to some 600 tiles (1920 x 1080: 510, stays in one piece: in two bands its k_fmerge took 0.66
// against 0.75 ms per 512 frames, and the joining of the bands 0.7 on top); 2448 x 2048 (1280 tiles) ran on the parents-only path.
constexpr int FM_BAND_MIN = 600, FM_BAND_TILES = 448;
constexpr int FM_WQ = 384;                    // joins a wave's queue holds (one round of boundary pixels adds at most 192)
constexpr int FM_WQS = FM_WQ + 8;             // a queue's stride in LDS: entry FM_WQ takes the joins that are none (stores without a branch)
constexpr uint32_t NOJ = 0xFFFFFFFFu;         // "no join"

struct FmFrame {
    const uint16_t *HT, *HB, *VL, *VR; // ring entries: tile-local id | colour << 15 (1 = white), 0xFFFF = no colour
    const uint32_t *base;              // first number of every tile's roots that this workgroup joins (LDS)
    const uint16_t *boff;              // mode 2: the tile's white roots, which its black ones are numbered behind; else 0 (LDS)
    const uint32_t *numtab;            // [tile][2]: what a white id is added to / a black id subtracted from to give the root's number (LDS)
    const uint16_t *hseg, *vseg;       // the tile edges worth sweeping (LDS): tile row << 5 | tile column of the tile below / on the right
    int nhs, nvs;
    int w, h, tiles_x, tiles_y;        // h: the frame's height (the pitch of the ring columns); tiles_y: tile rows of the BAND
    int hb;                            // pixel rows of the band this workgroup joins (the frame's height for a frame in one band)
    uint32_t mode;                     // what this workgroup joins: 0 the black roots, 1 the white ones, 2 both (a frame with few roots)
    int diag;                          // diagnostics (CK_FMERGE_STOP_AFTER 20 / 21): 1 the sweep without its loads, 2 without its joins
    // k_tile hands out white ids from 0 up and black ones from RING_CAP - 1 down; among the roots this workgroup joins, a tile's
    // white ones come first (in id order), then its black ones (in reverse id order)
    __device__ __forceinline__ bool acc(uint32_t e) const { return e != 0xFFFFu && (mode == 2u || (e >> 15) == mode); }
    __device__ __forceinline__ bool has(int t) const { return base[t + 1] != base[t]; }
    __device__ __forceinline__ uint32_t num(int t, uint32_t e) const {
        const uint32_t black = (e >> 15) ^ 1u, id = e & 0x7FFFu; // white: base + id; black: base + boff + RING_CAP - 1 - id
        return numtab[2 * t + black] + (black ? 0u - id : id);
    }
    // tile-local id of the l-th of the tile's cnt roots in that order
    __device__ __forceinline__ uint32_t id_of(int t, uint32_t l, uint32_t cnt) const {
        const uint32_t nw = mode == 2u ? (uint32_t)boff[t] : (mode == 1u ? cnt : 0u);
        return l < nw ? l : (uint32_t)(RING_CAP - 1) - (l - nw);
    }
    // slot (tile * RING_CAP + tile-local id) of root number r: its tile is the last one whose first number is <= r
    __device__ __forceinline__ uint32_t slot_of(uint32_t r, int tiles) const {
        int lo = 0, hi = tiles; // base[lo] <= r < base[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (base[mid] <= r) lo = mid; else hi = mid;
        }
        const uint32_t b0 = base[lo];
        return (uint32_t)lo * (uint32_t)RING_CAP + id_of(lo, r - b0, base[lo + 1] - b0);
    }
};
// Calls emit(j0, j1, j2) once per boundary pixel slot and lane (uniformly: every lane of the workgroup makes the same number of
// calls); a join is number | number << 16 when numbers fit 16 bits (LDS path) — the global path passes WIDE = true and gets
// the two numbers in separate calls of emit2.
constexpr int FM_GRP = 8;   // boundary pixels per lane and round: one 16-byte load per ring row
// eight consecutive ring entries (u16) in one load instruction: a sweep that fetched them two bytes at a time issued a million
// load instructions per workgroup at 2448 x 2048.  The rows are 2-byte aligned only (any frame width), which global memory takes.
struct __attribute__((packed, aligned(2))) FmEntries8 { unsigned long long lo, hi; };
struct FmRow8 { unsigned long long lo, hi; };
__device__ __forceinline__ FmRow8 fm_load8(const uint16_t *row, int i, int n) { // entries i .. i + 7 of a row of n (0xFFFF beyond it)
    FmRow8 r;
    if (i + FM_GRP <= n) {
        const FmEntries8 t = *reinterpret_cast<const FmEntries8 *>(row + i);
        r.lo = t.lo; r.hi = t.hi;
    } else {
        r.lo = r.hi = ~0ull;
        for (int q = 0; q < FM_GRP && i + q < n; q++) {
            const unsigned long long v = row[i + q];
            if (q < 4) r.lo = (r.lo & ~(0xFFFFull << (16 * q))) | (v << (16 * q));
            else r.hi = (r.hi & ~(0xFFFFull << (16 * (q - 4)))) | (v << (16 * (q - 4)));
        }
    }
    return r;
}
// the loop over a group's pixels looks at entry 0 (and 1) of the row and shifts the row down by one entry per round (a row
// indexed by the loop counter would live in scratch memory)
__device__ __forceinline__ uint32_t fm_first(const FmRow8 &e) { return (uint32_t)e.lo & 0xFFFFu; }
__device__ __forceinline__ uint32_t fm_second(const FmRow8 &e) { return (uint32_t)(e.lo >> 16) & 0xFFFFu; }
__device__ __forceinline__ void fm_shift(FmRow8 &e) { e.lo = (e.lo >> 16) | (e.hi << 48); e.hi = (e.hi >> 16) | (0xFFFFull << 48); }
// The sweep visits the tile edges of the two lists only: the top edge of a tile that has roots for this workgroup, the left edge of
// a tile when it or its left neighbour has (fm_edges).  A frame whose background thresholds to "no colour" has few.
__device__ __forceinline__ void fm_edges(const FmFrame &f, int tiles, uint16_t *hseg, uint16_t *vseg, uint32_t *segn, int tid) {
    if (tid < 2) segn[tid] = 0;
    __syncthreads();
    for (int t = tid; t < tiles; t += FM_NT) {
        const int ty = t / f.tiles_x, tx = t - ty * f.tiles_x;
        if (ty >= 1 && f.has(t)) hseg[atomicAdd(&segn[0], 1u)] = (uint16_t)(ty << 5 | tx);
        if (tx >= 1 && f.base[t + 1] != f.base[t - 1]) vseg[atomicAdd(&segn[1], 1u)] = (uint16_t)(ty << 5 | tx);
    }
    __syncthreads();
}
// The loops are kept rolled (one group of eight pixels per lane and round, the eight worked off by a loop): unrolled, the sweep was
// 140 KB of code.  The work on a pixel is written WITHOUT branches — every candidate join is formed and carries a flag: as nested
// ifs it compiled to a dozen exec-mask save / branch / restore sequences per pixel, and the sweep, sixteen waves walking a serial
// program, was bound by those scalar round trips (80 of the loop's 190 instructions).
// emit_q(v0, j0, v1, j1, v2, j2): up to three joins (number | number << 16) with their flags; emit_wide: the global-memory path's
// form (numbers beyond 16 bits), used when `wide`.
template <typename EmitQ, typename EmitWide>
__device__ __forceinline__ void fm_boundaries(const FmFrame &f, int tid, bool wide, EmitQ &&emit_q, EmitWide &&emit_wide) {
    const int w = f.w, h = f.h, hb = f.hb, tiles_x = f.tiles_x, tiles_y = f.tiles_y;
    const FmRow8 none = {~0ull, ~0ull};
    const uint32_t mode = f.mode;
    auto acc = [mode](uint32_t e) -> bool { return (e != 0xFFFFu) & ((mode == 2u) | ((e >> 15) == mode)); };
    auto num_of = [](const uint2 &nb, uint32_t e) -> uint32_t { const uint32_t id = e & 0x7FFFu; return (e & 0x8000u) ? nb.x + id : nb.y - id; };
    auto pick = [](bool c, const uint2 &a_, const uint2 &b_) -> uint2 { return make_uint2(c ? a_.x : b_.x, c ? a_.y : b_.y); };
    auto tabs = [&f](int t) -> uint2 { return *reinterpret_cast<const uint2 *>(&f.numtab[2 * t]); };
    // horizontal edges: the top row of tile row ty against the bottom row of tile row ty - 1 (up, and for white up-left / up-right)
    const int nh = f.nhs * (TW / FM_GRP);
#pragma unroll 1
    for (int item0 = 0; item0 < nh; item0 += FM_NT) {
        const int item = item0 + tid;
        FmRow8 P = none, Q = none;
        uint32_t QL = 0xFFFFu, QR = 0xFFFFu, PL = 0xFFFFu;
        int X = 0, TY = 1;
        if (item < nh) {
            const uint32_t sg = f.hseg[item / (TW / FM_GRP)];
            const int tyi = (int)(sg >> 5), x = (int)(sg & 31u) * TW + (item % (TW / FM_GRP)) * FM_GRP;
            X = x; TY = tyi;
            if (x < w && f.diag != 1) {
                const uint16_t *up = f.HB + (size_t)(tyi - 1) * w, *lo = f.HT + (size_t)tyi * w;
                P = fm_load8(lo, x, w); Q = fm_load8(up, x, w);
                if (x > 0) { QL = up[x - 1]; PL = lo[x - 1]; }
                if (x + FM_GRP < w) QR = up[x + FM_GRP];
            }
        }
        uint32_t q0 = QL, pl = PL;                     // the entries on the left of the pixel at hand
        uint32_t q1 = fm_first(Q);
        // the group lies inside one tile column: the numbers of its two tiles' roots start at the same table entries for all eight
        // pixels; only a diagonal neighbour at the group's very ends can lie in the next tile column (whose entries are loaded, too)
        const int trow = (TY - 1) * tiles_x, tcol = X >> 7;
        const uint2 nlo = tabs(trow + tiles_x + tcol), nup = tabs(trow + tcol);
        const uint2 nupL = tabs(trow + max(tcol - 1, 0)), nupR = tabs(trow + min(tcol + 1, tiles_x - 1));
#pragma unroll 1
        for (int j = 0; j < FM_GRP; j++) {
            const int x = X + j;
            const bool origin = (x >= 1) & (x <= w - 2); // only origin columns join
            const uint32_t praw = fm_first(P), p = origin ? praw : 0xFFFFu;
            const uint32_t q2 = j < FM_GRP - 1 ? fm_second(Q) : QR;
            // the pixel on the left made the same joins when it is the same component over the same component (and, for
            // white, the new diagonal neighbour up-right is that component again): nothing to add
            // (ring entries are tile-local ids: comparable inside one tile column only)
            const bool pw = (p & 0x8000u) != 0;
            const bool same = (x > 1) & ((x & (TW - 1)) != 0) & ((x & (TW - 1)) != TW - 1) & (pl == p) & (q0 == q1) & ((q2 == q1) | !pw);
            const bool valid = acc(p) & !same;
            const bool v0 = valid & (q1 != 0xFFFFu) & (((q1 ^ p) & 0x8000u) == 0);
            const bool v1 = valid & pw & ((q0 & 0x8000u) != 0) & (q0 != 0xFFFFu) & (q0 != q1);
            const bool v2 = valid & pw & ((q2 & 0x8000u) != 0) & (q2 != 0xFFFFu) & (q2 != q1);
            const uint32_t a0 = num_of(nlo, p), b0 = num_of(nup, q1);
            const uint32_t b1 = num_of(pick(((x - 1) >> 7) == tcol, nup, nupL), q0), b2 = num_of(pick(((x + 1) >> 7) == tcol, nup, nupR), q2);
            if (wide) emit_wide(v0 ? a0 : NOJ, b0, v1 ? a0 : NOJ, b1, v2 ? a0 : NOJ, b2); // (uniform)
            else emit_q(v0, a0 | (b0 << 16), v1, a0 | (b1 << 16), v2, a0 | (b2 << 16));
            q0 = q1; q1 = q2; pl = praw;
            fm_shift(P); fm_shift(Q);
        }
    }
    // vertical edges: the left column of tile column tx against the right column of tile column tx - 1
    const int nv = f.nvs * (TH / FM_GRP);
#pragma unroll 1
    for (int item0 = 0; item0 < nv; item0 += FM_NT) {
        const int item = item0 + tid;
        FmRow8 P = none, Q = none;
        uint32_t ql = 0xFFFFu, pu = 0xFFFFu;           // the entries above the pixels at hand
        int Y = 0, txi = 1;
        if (item < nv) {
            const uint32_t sg = f.vseg[item / (TH / FM_GRP)];
            txi = (int)(sg & 31u);
            Y = (int)(sg >> 5) * TH + (item % (TH / FM_GRP)) * FM_GRP;
            if (Y < hb && f.diag != 1) { // (rows are the band's own: its ring columns start at its first pixel row, their pitch is the frame's height)
                const uint16_t *lf = f.VR + (size_t)(txi - 1) * h, *rt = f.VL + (size_t)txi * h;
                P = fm_load8(rt, Y, hb); Q = fm_load8(lf, Y, hb);   // pixels (x, y ..) and (x - 1, y ..)
                if (Y > 0) { ql = lf[Y - 1]; pu = rt[Y - 1]; }
            }
        }
        const int x = txi * TW;                        // >= 1; an origin unless it is the frame's last column
        const int trow_i = Y / TH;
        const int tp = trow_i * tiles_x + txi;         // tile of (x, y) for the group's eight rows; (x - 1, y) lies in tp - 1
        const int tpu = max(trow_i - 1, 0) * tiles_x + txi; // the tiles above them (the group's first row can be a tile's top row)
        const uint2 nrt = tabs(tp), nlf = tabs(tp - 1), nrtU = tabs(tpu), nlfU = tabs(tpu - 1);
        const bool xin = x <= w - 2;
        (void)tiles_y;
#pragma unroll 1
        for (int j = 0; j < FM_GRP; j++) {
            const int y = Y + j;
            const bool ypos = y > 0;
            const bool up_same = ((y - 1) / TH) == trow_i; // the row above lies in the same tile row (all but the group's first row at a tile's top)
            const uint32_t praw = fm_first(P), qraw = fm_first(Q); // (x - 1, y) is always an origin column (1 <= x - 1 <= w - 2); beyond the frame's last row both are 0xFFFF
            const bool p_ok = acc(praw), q_ok = acc(qraw);      // (the other colour's pixels: another workgroup's)
            const uint32_t ap = num_of(nrt, praw), aq = num_of(nlf, qraw);
            const bool pin = p_ok & xin;
            const bool v0 = pin & q_ok & (((qraw ^ praw) & 0x8000u) == 0);
            const bool v1 = pin & ((praw & 0x8000u) != 0) & ypos & ((ql & 0x8000u) != 0) & (ql != 0xFFFFu);          // white: up-left
            const bool v2 = q_ok & ((qraw & 0x8000u) != 0) & ypos & ((pu & 0x8000u) != 0) & (pu != 0xFFFFu);        // white pixel (x - 1, y): up-right is (x, y - 1)
            const uint32_t b1 = num_of(pick(up_same, nlf, nlfU), ql), b2 = num_of(pick(up_same, nrt, nrtU), pu);
            if (wide) emit_wide(v0 ? ap : NOJ, aq, v1 ? ap : NOJ, b1, v2 ? aq : NOJ, b2); // (uniform)
            else emit_q(v0, ap | (aq << 16), v1, ap | (b1 << 16), v2, aq | (b2 << 16));
            ql = qraw; pu = praw;
            fm_shift(P); fm_shift(Q);
        }
    }
}

// two halving finds walked in lockstep (as lds_find2 of k_tile: both chains have a read in flight at every step, and one loop's
// worth of exec-mask bookkeeping instead of two); a root points at itself
__device__ __forceinline__ void fm_find2(uint16_t *p, uint32_t &a, uint32_t &b) {
    for (;;) {
        __asm__ volatile("" ::: "memory");
        const uint32_t na = p[a], nb = p[b];
        const bool da = na == a, db = nb == b;
        if (da && db) return;
        __asm__ volatile("" ::: "memory");
        const uint32_t ga = p[na], gb = p[nb]; // (a root's parent is itself: the read is harmless)
        if (!da) { if (ga != na) { p[a] = (uint16_t)ga; a = ga; } else a = na; }
        if (!db) { if (gb != nb) { p[b] = (uint16_t)gb; b = gb; } else b = nb; }
    }
}
// swaps parent[idx] from `expect` to `val` (u16 entry inside a 32-bit word); false when the entry no longer holds `expect`
__device__ __forceinline__ bool fm_cas16(uint16_t *p, uint32_t idx, uint32_t expect, uint32_t val) {
    uint32_t *wp = reinterpret_cast<uint32_t *>(p) + (idx >> 1);
    const uint32_t sh = (idx & 1u) * 16u;
    __asm__ volatile("" ::: "memory"); // (a fresh read, but a plain one: a volatile access through the cast pointer becomes a flat load, which waits for every outstanding global store)
    uint32_t wv = *wp;
    for (;;) {
        if (((wv >> sh) & 0xFFFFu) != expect) return false;
        const uint32_t prev = atomicCAS(wp, wv, (wv & ~(0xFFFFu << sh)) | (val << sh));
        if (prev == wv) return true;
        wv = prev; // the other half changed (a halving store or another hook): try again
    }
}
// key[] holds the top 16 bits of every root's pixel index (LDS): nearly every comparison is settled there, only two roots in the
// same key bucket read their exact pixels from the list in global memory
__device__ __forceinline__ void fm_union(uint16_t *p, const uint16_t *key, const ck_border_root *__restrict__ br, uint32_t a0, uint32_t b0) {
    uint32_t a = a0, b = b0;
    for (;;) {
        fm_find2(p, a, b);
        if (a == b) break;
        // both are roots right now; a root's pixel never changes.  Without a key array (more roots than parents + keys fit) the hook
        // goes by the roots' NUMBERS — any fixed order keeps the forest acyclic — and the component's smallest pixel is found
        // afterwards with one atomic minimum per root (reading the pixels from the packed list here, in L2, was what the
        // 2448 x 2048 frames waited for)
        uint32_t pa = key ? key[a] : a, pb = key ? key[b] : b;
        if (key && pa == pb) { pa = br[a].root; pb = br[b].root; }
        const uint32_t hi = pa > pb ? a : b, lo = pa > pb ? b : a;
        if (fm_cas16(p, hi, hi, lo)) { a = lo; break; } // hooked while still a root: parent pixel < child pixel, so no cycle can form
    }
    // the two ends now know an ancestor that is (or just was) the root: point them at it, so that the next join of the same
    // pair of components — boundaries cross a large component at many places — finds it in one step
    if (a0 != a && p[a0] != a) p[a0] = (uint16_t)a;
    if (b0 != a && p[b0] != a) p[b0] = (uint16_t)a;
}
// saturating add on a u16 entry: the sum is only ever compared with min_component_px (<= 32767 on this path)
__device__ __forceinline__ void fm_size_add(uint16_t *sz, uint32_t idx, uint32_t add, uint32_t enough) {
    uint32_t *wp = reinterpret_cast<uint32_t *>(sz) + (idx >> 1);
    const uint32_t sh = (idx & 1u) * 16u;
    __asm__ volatile("" ::: "memory"); // (a fresh read, but a plain one: a volatile access through the cast pointer becomes a flat load, which waits for every outstanding global store)
    uint32_t wv = *wp;
    for (;;) {
        const uint32_t cur = (wv >> sh) & 0xFFFFu;
        if (cur >= enough) return;                       // already "large": nothing a further part could change
        uint32_t nv = cur + add;
        nv = nv > 0x7FFFu ? 0x7FFFu : nv;
        const uint32_t prev = atomicCAS(wp, wv, (wv & ~(0xFFFFu << sh)) | (nv << sh));
        if (prev == wv) return;
        wv = prev;
    }
}

// ---- the same steps in global memory, for a frame with more roots than LDS holds (large frames, pathological maps) ----------------
__device__ __forceinline__ uint32_t gm_load(const uint32_t *a, uint32_t i) { return __hip_atomic_load(&a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t gm_find(uint32_t *p, uint32_t a) {
    for (;;) {
        const uint32_t n = gm_load(p, a);
        if (n == a) return a;
        const uint32_t g = gm_load(p, n);
        if (g == n) return n;
        __hip_atomic_store(&p[a], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = g;
    }
}
__device__ __forceinline__ void gm_union(uint32_t *p, const ck_border_root *__restrict__ br, uint32_t a, uint32_t b) {
    if (a == NOJ || b == NOJ) return;
    for (;;) {
        a = gm_find(p, a); b = gm_find(p, b);
        if (a == b) return;
        const uint32_t pa = br[a].root, pb = br[b].root;
        const uint32_t hi = pa > pb ? a : b, lo = pa > pb ? b : a;
        if (atomicCAS(&p[hi], hi, lo) == hi) return;
    }
}
// The global-memory path's steps either side of the (shared) boundary sweep
__device__ __forceinline__ void fm_global_begin(uint32_t n, uint32_t *gparent, uint32_t *gsz) {
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < n; i += FM_NT) { gparent[i] = i; gsz[i] = 0; }
    __threadfence();
    __syncthreads();
}
__device__ __forceinline__ void fm_global_end(const FmFrame &f, const ck_border_root *__restrict__ br, uint32_t n, uint32_t *gparent, uint32_t *gsz,
                                              uint32_t *groot, uint32_t *gsize, int tiles, uint32_t *bslot, uint32_t *xpar, uint32_t slot_lo) {
    const int tid = threadIdx.x;
    __threadfence();
    __syncthreads();
    for (uint32_t i = tid; i < n; i += FM_NT) {
        uint32_t r = i;
        for (;;) { const uint32_t nx = gm_load(gparent, r); if (nx == r) break; r = nx; }
        if (r != i) __hip_atomic_store(&gparent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&gsz[r], br[i].size);
    }
    __threadfence();
    __syncthreads();
    for (int t = tid >> 6; t < tiles; t += FM_NT / 64) {
        const uint32_t b0 = f.base[t], cnt = f.base[t + 1] - b0;
        for (uint32_t l = (uint32_t)(tid & 63); l < cnt; l += 64) {
            const uint32_t r = gm_load(gparent, b0 + l);
            const size_t slot = (size_t)t * RING_CAP + f.id_of(t, l, cnt);
            groot[slot] = br[r].root;
            gsize[slot] = gm_load(gsz, r);
            if (bslot) { // a frame in bands: which band root the slot belongs to (frame-level slot numbers)
                if (r == b0 + l) { bslot[slot] = slot_lo + (uint32_t)slot; xpar[slot] = slot_lo + (uint32_t)slot; }
                else bslot[slot] = slot_lo + f.slot_of(r, tiles);
            }
        }
    }
}

// Diagnostic build only (-DCK_FM_PROFILE): per-phase shader-clock totals of k_fmerge's first wave, in a buffer of their own.
#ifdef CK_FM_PROFILE
__device__ unsigned long long g_fm_prof[16];
#define FPROF_DECL unsigned long long fp0 = __builtin_readcyclecounter()
#define FPROF(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_fm_prof[k], t_ - fp0); fp0 = t_; } while (0)
#else
#define FPROF_DECL
#define FPROF(k)
#endif

// One workgroup per (frame, colour): white and black components never join, so the two colours of a frame are two independent
// union-finds over half the roots each — two workgroups per CU instead of one, each with half the chain of dependent steps.
// Dynamic LDS: parent u16[cap] | size / key u16[cap] | base u32[tiles + 1] | queue u32[16][FM_WQ].
__global__ __launch_bounds__(FM_NT, 4) __attribute__((amdgpu_num_sgpr(80))) void k_fmerge(ck_border_root *__restrict__ broots, const uint32_t *__restrict__ tile_count,
                                                  const uint16_t *__restrict__ ring, size_t ring_len, uint32_t *__restrict__ groot_all,
                                                  uint32_t *__restrict__ gsize_all, uint32_t *__restrict__ gscratch, size_t npix, int w, int h,
                                                  int tiles_x, int tiles_y_frame, int frame0, int n_frames, int min_comp, int lds_cap, int stop_after,
                                                  int band_rows, uint32_t *__restrict__ xband) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fm_lds[];
    __shared__ uint32_t wsum[2 * (FM_NT / 64)];
    __shared__ uint32_t segn[2];
    // The first half of the grid are the white workgroups, the second half the black ones (which have nothing to do for a frame
    // whose roots fit one workgroup: they must not hold the LDS of a CU while white ones wait for it).  Workgroups b and b + 8
    // share an XCD: a frame's workgroups run where k_tile wrote its ring entries and root slices (XCD f % 8 when frames are
    // dealt to XCDs).
    // Round 4: a frame with more tile rows than `band_rows` is joined in BANDS of whole tile rows — one (or two) workgroups per band, each
    // with a band's roots in the keyed LDS path (2448 x 2048 of dense noise: 55 000 black roots per frame, which one workgroup had to join
    // on the parents-only path with its sizes and smallest pixels in global memory) — and k_fseam / k_fapply join the bands.  A band
    // works in its own coordinates: tile rows, pixel rows, tile indices and slots count from the band's first tile row (the pointers are
    // advanced), so everything below reads as it did for a whole frame.
    const int half = (int)(gridDim.x >> 1); // a multiple of 8
    const uint32_t col = (int)blockIdx.x < half ? 1u : 0u; // 1 = white
    const int bands = (tiles_y_frame + band_rows - 1) / band_rows, f8 = half / bands;
    const int rel = (int)blockIdx.x - (col ? 0 : half), band = rel / f8;
    const int frame = frame0 + rel - band * f8, tid = threadIdx.x; // (frames [frame0, n_frames) of the batch)
    if (frame >= n_frames) return;
    const int ty_lo = band * band_rows, tiles_y = min(tiles_y_frame - ty_lo, band_rows);
    const int tiles = tiles_x * tiles_y, t_lo = ty_lo * tiles_x;                 // the band's tiles; its first tile in the frame
    const int hb = min(h - ty_lo * TH, tiles_y * TH);                              // its pixel rows
    const size_t frame_slots = (size_t)tiles_x * tiles_y_frame * RING_CAP, slot_lo = (size_t)t_lo * RING_CAP;
    tile_count += (size_t)frame * tiles_x * tiles_y_frame + t_lo;
    uint16_t *parent = reinterpret_cast<uint16_t *>(fm_lds);
    uint16_t *size16 = parent + lds_cap;                                   // (lds_cap is even)
    uint32_t *base = reinterpret_cast<uint32_t *>(size16 + lds_cap);
    uint32_t *queue = base + ((tiles + 1 + 3) & ~3);                       // per wave: joins waiting to be worked off
    uint16_t *boff = reinterpret_cast<uint16_t *>(queue + (FM_NT / 64) * FM_WQS);
    uint16_t *hseg = boff + ((tiles + 7) & ~7), *vseg = hseg + ((tiles + 7) & ~7);
    uint32_t *numtab = reinterpret_cast<uint32_t *>(vseg + ((tiles + 7) & ~7));
    const size_t slots = (size_t)tiles * RING_CAP; // of the band
    const ck_border_root *slice = broots + (size_t)frame * 2 * frame_slots + slot_lo;
    uint32_t *groot = groot_all + (size_t)frame * frame_slots + slot_lo, *gsize = gsize_all + (size_t)frame * frame_slots + slot_lo;
    // what the joining of the bands needs of every slot (frame-level slot numbers; null for a frame in one band): bslot[slot] = slot of
    // the component's root WITHIN ITS BAND; xpar[slot] = the band root's parent in the union-find over band roots (itself to begin with)
    uint32_t *bslot = xband ? xband + (size_t)frame * 2 * frame_slots + slot_lo : nullptr, *xpar = xband ? bslot + frame_slots : nullptr;
    const uint16_t *fr = ring + (size_t)frame * ring_len;
    FmFrame f;
    f.HT = fr + (size_t)ty_lo * w; f.HB = fr + (size_t)tiles_y_frame * w + (size_t)ty_lo * w;
    f.VL = fr + 2 * (size_t)tiles_y_frame * w + (size_t)ty_lo * TH; f.VR = f.VL + (size_t)tiles_x * h;
    f.base = base; f.boff = boff; f.numtab = numtab; f.w = w; f.h = h; f.hb = hb; f.tiles_x = tiles_x; f.tiles_y = tiles_y;
    f.diag = stop_after == 20 ? 1 : (stop_after == 21 ? 2 : (stop_after == 22 ? 1 : 0));
    // numbers: the tiles' counts, scanned (up to four tiles per thread).  Both colours are counted first: when all of a frame's
    // roots fit the LDS path together, the white workgroup joins both colours in one sweep and the black one has nothing to do.
    uint32_t n;
    {
        const int per = (tiles + FM_NT - 1) / FM_NT; // <= 4
        uint32_t cw[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0}, sw = 0, sb = 0;
        for (int k = 0; k < per; k++) {
            const int t = tid * per + k;
            const uint32_t tc = t < tiles ? tile_count[t] : 0u; // white | black << 16
            cw[k] = tc & 0xFFFFu; cb[k] = tc >> 16;
            sw += cw[k]; sb += cb[k];
        }
        const uint32_t iw = wave_scan_u32(sw), ib = wave_scan_u32(sb);
        if ((tid & 63) == 63) { wsum[tid >> 6] = iw; wsum[FM_NT / 64 + (tid >> 6)] = ib; }
        __syncthreads();
        uint32_t ow = iw - sw, ob = ib - sb, nw = 0, nb = 0;
        for (int k = 0; k < FM_NT / 64; k++) {
            if (k < (tid >> 6)) { ow += wsum[k]; ob += wsum[FM_NT / 64 + k]; }
            nw += wsum[k]; nb += wsum[FM_NT / 64 + k];
        }
        const bool both = nw + nb <= (uint32_t)lds_cap && min_comp <= 0x7FFF;
        if (both && !col) return;
        f.mode = both ? 2u : col;
        n = both ? nw + nb : (col ? nw : nb);
        uint32_t off = both ? ow + ob : (col ? ow : ob);
        for (int k = 0; k < per; k++) {
            const int t = tid * per + k;
            if (t < tiles) {
                const uint32_t bo = both ? cw[k] : 0u;
                base[t] = off; boff[t] = (uint16_t)bo;
                numtab[2 * t] = off; numtab[2 * t + 1] = off + bo + (uint32_t)(RING_CAP - 1);
            }
            off += both ? cw[k] + cb[k] : (col ? cw[k] : cb[k]);
            if (t == tiles - 1) base[tiles] = off;
        }
        __syncthreads();
    }
    if (n == 0) return;
    FPROF_DECL;
    fm_edges(f, tiles, hseg, vseg, segn, tid);
    FPROF(0);
    f.hseg = hseg; f.vseg = vseg; f.nhs = (int)segn[0]; f.nvs = (int)segn[1];
    // the same entries packed, a root's index = its number: from the front of the frame's second half — the black workgroup of a
    // frame whose colours are joined separately packs at its end (the two colours' roots together are at most `slots`)
    ck_border_root *br = broots + (size_t)frame * 2 * frame_slots + frame_slots + slot_lo + (f.mode == 0u ? slots - n : 0);
    uint32_t *sc = gscratch + (size_t)frame * 2 * frame_slots + slot_lo + (f.mode == 0u ? slots - n : 0); // global-memory path: parents; the sizes a frame's slots further on
    // More roots than parents + keys fit: up to twice as many (and at most 65 535) still run their unions in LDS, on the parents
    // alone — hooked by root number, with the components' smallest pixels and sizes settled afterwards by atomics in global memory.
    const bool keyless = n > (uint32_t)lds_cap;
    // ... and beyond that (or with a min_component_px the 16-bit sizes cannot express) the same steps run in global memory; the
    // boundary sweep below is the same code for both (one instantiation: its joins go to the wave's queue or straight to gm_union)
    const bool gmode = n > 2u * (uint32_t)lds_cap || n > 0xFFFFu || min_comp > 0x7FFF;
    uint32_t *gsz = sc + frame_slots;
    // while the unions run, the size array holds the roots' pixel keys: pixel index >> key_shift, 16 bits
    int key_shift = 0;
    while ((npix - 1) >> key_shift > 0xFFFFu) key_shift++;
    // Pack: a wave takes FM_PT tiles per round and copies their slices' entries; the loads of a round (two per lane and tile) are
    // all issued before the first store — one memory round trip per round instead of one per tile (the old loop's chain of load,
    // store, load ... was a sixth of the kernel at 1280 x 800).  The entry's number is also where its parent, key and, on the
    // keyless path, its size and smallest-pixel cells start out: written here, from the registers that hold the entry.
    constexpr int FM_PT = 4;
    for (int t0 = tid >> 6; t0 < tiles; t0 += FM_PT * (FM_NT / 64)) {
        ck_border_root e[FM_PT][2];
        uint32_t at[FM_PT][2];
#pragma unroll
        for (int u = 0; u < FM_PT; u++) {
            const int t = t0 + u * (FM_NT / 64);
            const uint32_t b0 = t < tiles ? base[t] : 0u, cnt = t < tiles ? base[t + 1] - b0 : 0u;
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const uint32_t l = (uint32_t)(tid & 63) + 64u * v;
                at[u][v] = l < cnt ? b0 + l : 0xFFFFFFFFu;
                if (l < cnt) e[u][v] = slice[(size_t)t * RING_CAP + f.id_of(t, l, cnt)];
            }
        }
#pragma unroll
        for (int u = 0; u < FM_PT; u++)
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const uint32_t i = at[u][v];
                if (i == 0xFFFFFFFFu) continue;
                br[i] = e[u][v];
                if (!gmode) {
                    parent[i] = (uint16_t)i; // (without keys the parents run on into the key array's bytes)
                    if (keyless) { gsz[i] = 0; sc[i] = e[u][v].root; } // (sc[]: the smallest pixel of the component a root ends up heading)
                    else size16[i] = (uint16_t)(e[u][v].root >> key_shift);
                }
            }
        for (int u = 0; u < FM_PT; u++) { // a tile with more than 128 ring-touching roots of this workgroup's colours: the rest, one by one
            const int t = t0 + u * (FM_NT / 64);
            if (t >= tiles) break;
            const uint32_t b0 = base[t], cnt = base[t + 1] - b0;
            for (uint32_t l = (uint32_t)(tid & 63) + 128u; l < cnt; l += 64) {
                const ck_border_root x = slice[(size_t)t * RING_CAP + f.id_of(t, l, cnt)];
                const uint32_t i = b0 + l;
                br[i] = x;
                if (!gmode) {
                    parent[i] = (uint16_t)i;
                    if (keyless) { gsz[i] = 0; sc[i] = x.root; } else size16[i] = (uint16_t)(x.root >> key_shift);
                }
            }
        }
    }
    // the packed list and the cells in global memory were written by this workgroup and are read (and updated by atomics, which
    // execute at L2, where this CU's stores are once they are counted done) by this workgroup only: a workgroup-scope release
    // and the barrier order them — no agent-scope fence, which is a cache write-back and invalidate per thread
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    FPROF(1);
    if (stop_after == 0) return; // diagnostics (CK_FMERGE_STOP_AFTER)
    if (gmode) fm_global_begin(n, sc, gsz);
    FPROF(2);
    const uint16_t *key = keyless ? nullptr : size16;
    // Every boundary pixel yields up to three joins; most lanes have fewer, and a join is a chain of dependent LDS reads.  So the
    // joins of a wave are queued in LDS (wave prefix sums) and then worked off one per lane, all lanes busy, instead of every
    // lane running its own zero to three joins while the others wait.
    uint32_t *wq = queue + (tid >> 6) * FM_WQS;
    const int lane = tid & 63;
    uint32_t qn = 0; // joins waiting in the wave's queue (wave-uniform)
    auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (stop_after != 10) // (10: diagnostics, the sweep without its unions)
        for (uint32_t j = (uint32_t)lane; j < qn; j += 64) {
            const uint32_t e = wq[j];
            fm_union(parent, key, br, e & 0xFFFFu, e >> 16);
        }
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };
    fm_boundaries(f, tid, gmode,
        [&](bool v0, uint32_t j0, bool v1, uint32_t j1, bool v2, uint32_t j2) {
            // into the wave's queue without a branch: a join that is not one goes to the queue's spare slot
            const bool live = (f.diag != 2) & (stop_after != 22);
            v0 &= live; v1 &= live; v2 &= live;
            const uint32_t cnt = (uint32_t)v0 + (uint32_t)v1 + (uint32_t)v2;
            const uint32_t incl = wave_scan_u32(cnt);
            const uint32_t pos = qn + incl - cnt;
            wq[v0 ? pos : (uint32_t)FM_WQ] = j0;
            wq[v1 ? pos + (uint32_t)v0 : (uint32_t)FM_WQ] = j1;
            wq[v2 ? pos + (uint32_t)v0 + (uint32_t)v1 : (uint32_t)FM_WQ] = j2;
            qn += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (qn > FM_WQ - 192) drain();
        },
        [&](uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) { gm_union(sc, br, a0, b0); gm_union(sc, br, a1, b1); gm_union(sc, br, a2, b2); });
    FPROF(3);
    if (gmode) { fm_global_end(f, br, n, sc, gsz, groot, gsize, tiles, bslot, xpar, (uint32_t)slot_lo); return; }
    drain();
    __syncthreads();
    FPROF(4);
    if (stop_after == 1) return;
    for (uint32_t i = tid; i < n; i += FM_NT) { // flatten (walks only read: other lanes' entries may still be mid-chain)
        uint32_t r = i;
        for (;;) { const uint32_t nx = parent[r]; if (nx == r) break; r = nx; }
        if (r != i) parent[i] = (uint16_t)r; // a non-root entry: rewriting it with its root keeps every other walk valid
        if (!keyless) size16[i] = 0;         // the keys are dead: the array becomes the sizes
    }
    __syncthreads();
    FPROF(5);
    if (stop_after == 2) return;
    const uint32_t enough = (uint32_t)min_comp;
    // sizes (and, without keys, the components' smallest pixels): four roots per lane and round, their list entries requested
    // together.  On the keyless path the cells are in global memory and a frame's giant component is most of its roots: every add
    // and every minimum to ONE address, served one by one at L2 — so a cell is read first, and an add to a size already at
    // min_component_px (sizes are exact below it only) or a minimum that would not lower the cell is not issued.
    for (uint32_t i0 = 0; i0 < n; i0 += 4 * FM_NT) {
        ck_border_root e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * FM_NT + tid; if (i < n) e[u] = br[i]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * FM_NT + tid;
            if (i >= n) continue;
            const uint32_t r = parent[i], sz = e[u].size;
            if (keyless) {
                if (gm_load(gsz, r) < enough) atomicAdd(&gsz[r], sz);
                if (r != i && gm_load(sc, r) > e[u].root) atomicMin(&sc[r], e[u].root);
            } else fm_size_add(size16, r, sz > 0x7FFFu ? 0x7FFFu : sz, enough);
        }
    }
    if (keyless) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // (the cells are read back by this workgroup only, with L2 loads)
    __syncthreads();
    FPROF(6);
    if (stop_after == 3) return;
    // the tables, tile by tile: FM_PT tiles per wave and round, the round's lookups in flight together
    for (int t0 = tid >> 6; t0 < tiles; t0 += FM_PT * (FM_NT / 64)) {
        uint32_t rr[FM_PT][2], rootv[FM_PT][2], sizev[FM_PT][2];
        size_t slot[FM_PT][2];
#pragma unroll
        for (int u = 0; u < FM_PT; u++) {
            const int t = t0 + u * (FM_NT / 64);
            const uint32_t b0 = t < tiles ? base[t] : 0u, cnt = t < tiles ? base[t + 1] - b0 : 0u;
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const uint32_t l = (uint32_t)(tid & 63) + 64u * v;
                rr[u][v] = 0xFFFFFFFFu;
                if (l < cnt) { rr[u][v] = parent[b0 + l]; slot[u][v] = (size_t)t * RING_CAP + f.id_of(t, l, cnt); }
            }
        }
#pragma unroll
        for (int u = 0; u < FM_PT; u++)
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const uint32_t r = rr[u][v];
                if (r == 0xFFFFFFFFu) continue;
                rootv[u][v] = keyless ? gm_load(sc, r) : br[r].root;
                sizev[u][v] = keyless ? gm_load(gsz, r) : size16[r];
            }
#pragma unroll
        for (int u = 0; u < FM_PT; u++)
#pragma unroll
            for (int v = 0; v < 2; v++)
                if (rr[u][v] != 0xFFFFFFFFu) { groot[slot[u][v]] = rootv[u][v]; gsize[slot[u][v]] = sizev[u][v]; }
        if (bslot) { // a frame in bands: which band root every slot belongs to (frame-level slot numbers); a band root is its own parent
#pragma unroll
            for (int u = 0; u < FM_PT; u++)
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    const uint32_t r = rr[u][v];
                    if (r == 0xFFFFFFFFu) continue;
                    const int t = t0 + u * (FM_NT / 64);
                    const uint32_t own = base[t] + (uint32_t)(tid & 63) + 64u * v;
                    const uint32_t rs = r == own ? (uint32_t)slot[u][v] : f.slot_of(r, tiles);
                    bslot[slot[u][v]] = (uint32_t)slot_lo + rs;
                    if (r == own) xpar[slot[u][v]] = (uint32_t)slot_lo + rs;
                }
        }
        for (int u = 0; u < FM_PT; u++) { // (more than 128 roots in a tile: the rest)
            const int t = t0 + u * (FM_NT / 64);
            if (t >= tiles) break;
            const uint32_t b0 = base[t], cnt = base[t + 1] - b0;
            for (uint32_t l = (uint32_t)(tid & 63) + 128u; l < cnt; l += 64) {
                const uint32_t r = parent[b0 + l];
                const size_t sl = (size_t)t * RING_CAP + f.id_of(t, l, cnt);
                groot[sl] = keyless ? gm_load(sc, r) : br[r].root;
                gsize[sl] = keyless ? gm_load(gsz, r) : size16[r];
                if (bslot) {
                    const uint32_t rs = r == b0 + l ? (uint32_t)sl : f.slot_of(r, tiles);
                    bslot[sl] = (uint32_t)slot_lo + rs;
                    if (r == b0 + l) xpar[sl] = (uint32_t)slot_lo + rs;
                }
            }
        }
    }
    FPROF(7);
}

#ifdef CK_FM_PROFILE
} // namespace
extern "C" int ck_fm_profile_read(unsigned long long *out, int reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fm_prof), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fm_prof), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
namespace {
#endif

// ---- a frame in bands (round 4): the bands' components joined across the band boundaries ---------------------------------------
// k_fmerge has joined every band of tile rows by itself and left, per slot, groot / gsize at the BAND's level, bslot[slot] = the slot
// of the component's root within its band, and every band root its own parent in xpar[].  k_fseam (one workgroup per frame) replays
// the connectivity rule across the band boundaries — the top row of a band's first tile row against the bottom row of the tile row
// above it: up for both colours, up-left and up-right for white (crates/chalkydri-apriltags/src/lib.rs:501-549; the diagonal joins
// of the tile corners on that row are among them, which is why k_fmerge's vertical sweep leaves a band's first pixel row alone) — as
// unions over band roots in global memory (a few thousand joins per frame: hooked by slot number), then folds every joined band
// root's smallest pixel and pixel count into its final root (each band root once: a claim bit in its xpar entry).  k_fapply, one
// thread per slot, gives every slot the values of its final root.
constexpr uint32_t XCLAIM = 0x80000000u;
__device__ __forceinline__ uint32_t x_find(uint32_t *xpar, uint32_t a, bool halve) {
    for (;;) {
        const uint32_t n = gm_load(xpar, a) & ~XCLAIM;
        if (n == a) return a;
        const uint32_t g = gm_load(xpar, n) & ~XCLAIM;
        if (g == n) return n;
        if (halve) __hip_atomic_store(&xpar[a], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (no claim bits yet while the unions run)
        a = g;
    }
}
// One thread per pixel of a band boundary's lower row (256 per workgroup: a thread's work is a chain of a dozen memory round trips,
// so the grid is cut fine and many workgroups share a CU).  FOLD = false: the unions; FOLD = true, a launch later: every band root
// that took part hands its smallest pixel and its pixel count to its final root.  A join the pixel on the left (or, for up-right,
// the pixel on the right) makes as its own vertical join is left to it: along a run of one component over one component only the
// run's ends join.
template <bool FOLD>
__global__ __launch_bounds__(NT) void k_fseam(const uint16_t *__restrict__ ring, size_t ring_len, uint32_t *__restrict__ groot_all,
                                              uint32_t *__restrict__ gsize_all, uint32_t *__restrict__ xband, int w, int tiles_x, int tiles_y,
                                              int band_rows, int frame0) {
    const int frame = frame0 + (int)blockIdx.y;
    const int item = (int)blockIdx.x * NT + (int)threadIdx.x;
    const int bands = (tiles_y + band_rows - 1) / band_rows;
    if (item >= (bands - 1) * w) return;
    const size_t slots = (size_t)tiles_x * tiles_y * RING_CAP;
    uint32_t *groot = groot_all + (size_t)frame * slots, *gsize = gsize_all + (size_t)frame * slots;
    const uint32_t *bslot = xband + (size_t)frame * 2 * slots;
    uint32_t *xpar = xband + (size_t)frame * 2 * slots + slots;
    const uint16_t *HT = ring + (size_t)frame * ring_len, *HB = HT + (size_t)tiles_y * w;
    const int b = item / w + 1, x = item - (b - 1) * w, ty = b * band_rows;
    if (x < 1 || x > w - 2) return; // only origin columns join
    const uint16_t *lo = HT + (size_t)ty * w, *up = HB + (size_t)(ty - 1) * w;
    const uint32_t p = lo[x];
    if (p == 0xFFFFu) return;
    const uint32_t q[3] = {up[x], up[x - 1], up[x + 1]};
    const bool pw = (p & 0x8000u) != 0;
    // (ring entries are tile-local ids: comparable inside one tile column only)
    const bool run_l = x - 1 >= 1 && (x & (TW - 1)) != 0 && lo[x - 1] == p, run_r = x + 1 <= w - 2 && ((x + 1) & (TW - 1)) != 0 && lo[x + 1] == p;
    const bool dup[3] = {run_l && q[1] == q[0], run_l, run_r};
    const uint32_t A = bslot[(size_t)(ty * tiles_x + (x >> 7)) * RING_CAP + (p & 0x7FFFu)];
    bool a_done = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int xq = k == 0 ? x : (k == 1 ? x - 1 : x + 1);
        const bool v = q[k] != 0xFFFFu && (k == 0 ? ((q[k] ^ p) & 0x8000u) == 0 : (pw && (q[k] & 0x8000u) != 0));
        if (!v || dup[k]) continue;
        const uint32_t B = bslot[(size_t)((ty - 1) * tiles_x + (xq >> 7)) * RING_CAP + (q[k] & 0x7FFFu)];
        if (!FOLD) { // union: the band root with the larger slot number goes under the other
            uint32_t a = A, c = B;
            for (;;) {
                a = x_find(xpar, a, true); c = x_find(xpar, c, true);
                if (a == c) break;
                const uint32_t hi = a > c ? a : c, lw = a > c ? c : a;
                if (atomicCAS(&xpar[hi], hi, lw) == hi) break;
            }
        } else { // fold: every band root that took part, once (the claim bit)
            for (int e = a_done ? 1 : 0; e < 2; e++) {
                const uint32_t P = e ? B : A;
                if (gm_load(xpar, P) & XCLAIM) continue;
                if (atomicOr(&xpar[P], XCLAIM) & XCLAIM) continue;
                const uint32_t R = x_find(xpar, P, false);
                if (R == P) continue;
                atomicMin(&groot[R], groot[P]);
                atomicAdd(&gsize[R], gsize[P]);
            }
            a_done = true;
        }
    }
}
// one wave per tile: its ring-touching components' slots (white ids from 0 up, black ones from RING_CAP - 1 down)
__global__ __launch_bounds__(NT) void k_fapply(const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ groot_all, uint32_t *__restrict__ gsize_all,
                                               const uint32_t *__restrict__ xband, int tiles, int frame0) {
    const int frame = frame0 + (int)blockIdx.y;
    const int t = (int)blockIdx.x * (NT / 64) + (int)(threadIdx.x >> 6);
    if (t >= tiles) return;
    const size_t slots = (size_t)tiles * RING_CAP;
    const uint32_t tc = tile_count[(size_t)frame * tiles + t]; // white | black << 16
    const uint32_t nw = tc & 0xFFFFu, cnt = nw + (tc >> 16);
    const uint32_t *bslot = xband + (size_t)frame * 2 * slots, *xpar = bslot + slots;
    uint32_t *groot = groot_all + (size_t)frame * slots, *gsize = gsize_all + (size_t)frame * slots;
    for (uint32_t l = threadIdx.x & 63u; l < cnt; l += 64u) {
        const uint32_t s = (uint32_t)t * (uint32_t)RING_CAP + (l < nw ? l : (uint32_t)(RING_CAP - 1) - (l - nw));
        uint32_t R = bslot[s];
        if (xpar[R] == R) continue; // its band root took part in no join across a band boundary (no claim bit): the band's values are final
        for (;;) { const uint32_t n = xpar[R] & ~XCLAIM; if (n == R) break; R = n; }
        if (R == s) continue; // the final root holds its values already
        groot[s] = groot[R];
        gsize[s] = gsize[R];
    }
}

// ---- parity / test path: canonical labels and exact sizes ---------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_canon(const ck_label_t *__restrict__ labels, const uint32_t *__restrict__ groot, size_t slots,
                                              uint32_t *__restrict__ out, size_t npix, size_t total, int w, int tiles_x) {
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const uint32_t l = labels[i];
    uint32_t g = CK_LBL_INVALID;
    if (l != CK_LBL_NONE) {
        const size_t fr = i / npix, pi = i - fr * npix;
        const int y = (int)(pi / (size_t)w), x = (int)(pi - (size_t)y * w);
        g = (l & CK_LBL_BORDER) ? groot[fr * slots + ck_label_slot(l, x, y, tiles_x)] // a ring-touching component: the word carries its tile-local id
                                : ck_label_interior_root(l, x, y, w);
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

// rows of the decimated copy are padded to 16 bytes like every staged frame (k_tile reads 16-byte chunks): qstride >= qw
__global__ __launch_bounds__(NT) void k_decimate(const uint8_t *__restrict__ src, size_t frame_pitch, int stride, int f, int qw, int qh,
                                                 int qstride, uint8_t *__restrict__ dst, size_t total) {
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    size_t npix = (size_t)qw * qh;
    size_t fr = i / npix, rem = i - fr * npix;
    int y = (int)(rem / qw), x = (int)(rem - (size_t)y * qw);
    dst[fr * (size_t)qstride * qh + (size_t)y * qstride + x] = src[fr * frame_pitch + (size_t)(y * f) * stride + (size_t)x * f];
}

} // namespace

int ck_launch_threshold_segment(ck_handle *h, const uint8_t *frames, int stride, size_t frame_pitch, int n, bool precomputed) {
    const int tiles = h->tiles_x * h->tiles_y;
    static const int stop_after = CK_KNOB("CK_TILE_STOP_AFTER", 99);
    static const int sweeps = CK_KNOB("CK_TILE_SWEEPS", 0); // (pointer-jumping sweeps before the pooled unions: 0, 1 and 2 time the same since the nodes are pair components)
    // frames dealt to XCDs (a frame's tiles share one L2): worth it once there are frames for all eight (CK_TILE_XCD=0/1 forces it)
    static const int xcd_env = CK_KNOB("CK_TILE_XCD", -1);
    static const int fm_stop = CK_KNOB("CK_FMERGE_STOP_AFTER", 99);
    // CK_SEG_CHUNKS=k cuts the batch into k chunks of frames (multiples of 8: the XCD dealing) and runs chunk i's k_fmerge — one
    // workgroup per frame, bound by chains of dependent steps — on a side stream beside chunk i + 1's k_tile.  Measured on the
    // bench batch (1280 x 800 x 256, same box): 1.07 ms whole, 1.10 in two chunks, 1.22 in four, 1.77 in eight: the merge's
    // workgroups (16 waves and most of a CU's LDS each) displace more of k_tile than their waiting hides.  So the default is one chunk.
    static const int chunks_env = CK_KNOB("CK_SEG_CHUNKS", 0);
    // (not together with CK_STREAMS=2: the views of a split batch share the handle's one side stream and its events)
    int chunks = chunks_env > 0 && ck_streams_wanted() < 2 ? chunks_env : 1;
    if (chunks > CK_SEG_CHUNKS_MAX) chunks = CK_SEG_CHUNKS_MAX;
    const int per = ((n + chunks - 1) / chunks + 7) & ~7;
    const int cap_env = CK_KNOB("CK_FMERGE_CAP", 0); // the path-forcing tests (diag build) force the global-memory path with a small value (read per call)
    // roots the LDS path of one workgroup holds (dense binary noise has about 90 per tile).  A frame whose roots fit is joined by ONE
    // workgroup, both colours in one sweep; a larger one by two, one per colour (1920 x 1080 of dense noise: 23 000 each); a
    // workgroup with more takes the global-memory path.
    // Bands of tile rows (round 4): a frame with more than FM_BAND_MIN tiles is joined band by band (k_fmerge), then across the
    // bands (k_fseam twice, k_fapply) — every band's roots then fit the keyed LDS path.  2448 x 2048 (1280 tiles, three bands of 22
    // tile rows), per 256 frames: k_fmerge 0.93 + k_fseam 0.15 + k_fapply 0.18 ms against 2.4 in one piece per colour (parents only
    // in LDS, sizes and smallest pixels by atomics in global memory).  CK_FMERGE_BAND_ROWS (diagnostics build, read per call) forces
    // the band height: the path-forcing tests.
    const int band_env = CK_KNOB("CK_FMERGE_BAND_ROWS", 0);
    int band_rows = h->tiles_y;
    if (tiles > FM_BAND_MIN) {
        const int rows = FM_BAND_TILES / h->tiles_x > 0 ? FM_BAND_TILES / h->tiles_x : 1, nb = (h->tiles_y + rows - 1) / rows;
        band_rows = (h->tiles_y + nb - 1) / nb; // bands of equal height
    }
    if (band_env > 0 && band_env < h->tiles_y) band_rows = band_env;
    const int bands = (h->tiles_y + band_rows - 1) / band_rows, btiles = band_rows * h->tiles_x; // tiles of a (full) band
    int cap = btiles * 120;
    cap = cap < 4096 ? 4096 : (cap > FM_CAP ? FM_CAP : cap);
    if (cap_env > 0 && cap_env < cap) cap = cap_env;
    // (the per-tile arrays and the join queues come first: very large frames leave less room for roots)
    const size_t fixed = (size_t)((btiles + 1 + 3) & ~3) * 4 + (size_t)(FM_NT / 64) * FM_WQS * 4 + (size_t)((btiles + 7) & ~7) * 2 * 3 + (size_t)btiles * 8;
    const size_t lds_max = 160 * 1024 - 512;
    if ((size_t)cap * 4 + fixed > lds_max) cap = (int)((lds_max - fixed) / 4);
    cap &= ~1;
    const size_t lds = (size_t)cap * 4 + fixed;
    if (!h->fmerge_lds_allowed) { // per handle, i.e. per device: a process may hold handles on several GPUs
        CK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fmerge), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
        h->fmerge_lds_allowed = true;
    }
    bool forked = false;
    for (int f0 = 0, ci = 0; f0 < n; f0 += per, ci++) {
        const int f1 = f0 + per < n ? f0 + per : n, cn = f1 - f0;
        const int xcd_map = xcd_env >= 0 ? xcd_env : (cn >= 16 ? 1 : 0);
        const unsigned grid = xcd_map ? (unsigned)(((cn + 7) / 8) * 8 * tiles) : (unsigned)(tiles * cn);
        if (precomputed)
            hipLaunchKernelGGL(k_tile<true>, dim3(grid), dim3(KNT), 0, h->stream, frames, frame_pitch, stride, h->qw, h->qh,
                               h->tiles_x, h->tiles_y, f0, f1, xcd_map, h->cfg.min_white_black_diff, h->cfg.min_component_px, h->d_thresh, h->d_labels,
                               h->d_broots, h->d_tile_count, h->d_ring, h->ring_len, stop_after, sweeps);
        else
            hipLaunchKernelGGL(k_tile<false>, dim3(grid), dim3(KNT), 0, h->stream, frames, frame_pitch, stride, h->qw, h->qh,
                               h->tiles_x, h->tiles_y, f0, f1, xcd_map, h->cfg.min_white_black_diff, h->cfg.min_component_px, h->d_thresh, h->d_labels,
                               h->d_broots, h->d_tile_count, h->d_ring, h->ring_len, stop_after, sweeps);
        if (stop_after < 98) continue; // (a k_tile cut short by the diagnostics knob leaves tile counts or ring entries unwritten: nothing for the merge to read)
        hipStream_t ms = h->stream;
        if (f1 < n) { // not the last chunk: its merge goes beside the next chunk's k_tile
            CK_HIP(hipEventRecord(h->ev_seg[ci], h->stream));
            CK_HIP(hipStreamWaitEvent(h->seg_stream, h->ev_seg[ci], 0));
            ms = h->seg_stream;
            forked = true;
        }
        hipLaunchKernelGGL(k_fmerge, dim3((unsigned)(((cn + 7) / 8) * 16 * bands)), dim3(FM_NT), lds, ms, h->d_broots, h->d_tile_count, h->d_ring, h->ring_len,
                           h->d_groot, h->d_gsize, h->d_gscratch, h->npix, h->qw, h->qh, h->tiles_x, h->tiles_y, f0, f1, h->cfg.min_component_px, cap,
                           fm_stop, band_rows, bands > 1 ? h->d_xband : nullptr);
        if (bands > 1 && fm_stop >= 99) {
            const dim3 sg((unsigned)(((bands - 1) * h->qw + NT - 1) / NT), (unsigned)cn);
            hipLaunchKernelGGL(k_fseam<false>, sg, dim3(NT), 0, ms, h->d_ring, h->ring_len, h->d_groot, h->d_gsize, h->d_xband, h->qw, h->tiles_x, h->tiles_y, band_rows, f0);
            hipLaunchKernelGGL(k_fseam<true>, sg, dim3(NT), 0, ms, h->d_ring, h->ring_len, h->d_groot, h->d_gsize, h->d_xband, h->qw, h->tiles_x, h->tiles_y, band_rows, f0);
            hipLaunchKernelGGL(k_fapply, dim3((unsigned)((tiles + NT / 64 - 1) / (NT / 64)), (unsigned)cn), dim3(NT), 0, ms, h->d_tile_count, h->d_groot,
                               h->d_gsize, h->d_xband, tiles, f0);
        }
    }
    if (forked) {
        CK_HIP(hipEventRecord(h->ev_seg_join, h->seg_stream));
        CK_HIP(hipStreamWaitEvent(h->stream, h->ev_seg_join, 0));
    }
    CK_HIP(hipGetLastError());
    return CK_OK;
}

int ck_launch_canonical_labels(ck_handle *h, int n, uint32_t *d_out, uint32_t *d_sizes) {
    size_t total = h->npix * (size_t)n;
    unsigned blocks = (unsigned)((total + NT - 1) / NT);
    hipLaunchKernelGGL(k_canon, dim3(blocks), dim3(NT), 0, h->stream, h->d_labels, h->d_groot, (size_t)h->broot_cap, d_out, h->npix, total, h->qw, h->tiles_x);
    if (d_sizes) {
        // exact sizes by counting: test path only (the pipeline uses the SMALL flag of the label words and the slot tables instead).
        // One exit: the count array is released on every path, and through the handle's allocator (guard pages under CK_POISON=3).
        uint32_t *cnt = nullptr;
        hipError_t e = ck_malloc_dev(&cnt, total * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, total * sizeof(uint32_t), h->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_count, dim3(blocks), dim3(NT), 0, h->stream, d_out, cnt, h->npix, total);
            hipLaunchKernelGGL(k_sizes, dim3(blocks), dim3(NT), 0, h->stream, d_out, cnt, d_sizes, h->npix, total);
            e = hipStreamSynchronize(h->stream);
        }
        (void)ck_free_dev(cnt);
        if (e != hipSuccess) {
            snprintf(ck_err_text, sizeof ck_err_text, "canonical labels: %s", hipGetErrorString(e));
            (void)hipGetLastError();
            return e == hipErrorOutOfMemory ? CK_ENOMEM : CK_EDEVICE;
        }
    }
    CK_HIP(hipGetLastError());
    return CK_OK;
}

int ck_launch_decimate(ck_handle *h, const uint8_t *frames, int stride, size_t frame_pitch, int n) {
    size_t total = h->npix * (size_t)n;
    unsigned blocks = (unsigned)((total + NT - 1) / NT);
    hipLaunchKernelGGL(k_decimate, dim3(blocks), dim3(NT), 0, h->stream, frames, frame_pitch, stride, h->cfg.quad_decimate, h->qw,
                       h->qh, (h->qw + 15) / 16 * 16, h->d_qframes, total);
    CK_HIP(hipGetLastError());
    return CK_OK;
}
