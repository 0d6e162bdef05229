// ck_links.h — the connectivity rule of the segmentation stage as bit operations on 32-pixel row words.
//
// The rule is CAT's connected_components (crates/chalkydri-apriltags/src/lib.rs:501-549): origin pixels are the columns
// 1..w-2 of a frame; an origin pixel joins its left and its upper neighbour when they have its colour, and a WHITE origin
// pixel also joins its upper-left and upper-right neighbours (4-connected black, 8-connected white).
//
// k_tile (k_ccl.hip) works on RUNS: a run is a maximal stretch of same-coloured pixels of one row inside one 32-pixel word
// whose pixels are joined left-to-right by the rule above (so a run always ends at a word boundary and at a non-origin
// column; the piece in the next word is a run of its own, joined to this one by a `hleft` link).  A run's node is the pixel
// of its first bit.  Everything here is plain integer arithmetic shared by the kernel and by the host-side check
// (tests/cpp/links_check.cpp), which replays the links through a sequential union-find and compares with the oracle.
#ifndef CK_LINKS_H
#define CK_LINKS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CK_HD __host__ __device__ __forceinline__
#else
#define CK_HD static inline
#endif

CK_HD int ck_ctz32(uint32_t v) { return __builtin_ctz(v); }   // v != 0
CK_HD int ck_clz32(uint32_t v) { return __builtin_clz(v); }   // v != 0
CK_HD int ck_popc32(uint32_t v) { return __builtin_popcount(v); }

// origin flags of the 32 pixels that start at frame column x0 (columns 1..w-2 are origins)
CK_HD uint32_t ck_origin32(int x0, int w) {
    uint32_t O = 0xFFFFFFFFu;
    if (x0 == 0) O &= ~1u;
    const int last = (w - 1) - x0;
    if (last >= 0 && last < 32) O &= ~(1u << last);
    return O;
}
// run starts of a word: a coloured pixel that the rule does not join to its left neighbour inside the word
CK_HD uint32_t ck_starts32(uint32_t M, uint32_t O) { return M & ~((M << 1) & O); }
// bits [0..j] of a word (j = 31 gives all ones)
CK_HD uint32_t ck_upto32(int j) { return (2u << j) - 1u; }
// bit position of the start of the run that holds pixel j (S = run starts, bit j belongs to a run)
CK_HD int ck_run_start32(uint32_t S, int j) { return 31 - ck_clz32((S & ck_upto32(j)) | 1u); }
// bit position of the last run start of a word (0 for an empty word)
CK_HD int ck_last_start32(uint32_t S) { return 31 - ck_clz32(S | 1u); }
// the pixels of the run that starts at bit i
CK_HD uint32_t ck_run_bits32(uint32_t M, uint32_t S, int i) {
    const uint32_t next = S & ~ck_upto32(i);                         // starts above i
    const uint32_t lim = next ? ((next & (0u - next)) - 1u) : 0xFFFFFFFFu;
    return M & lim & ~((1u << i) - 1u);
}

// Links of one run to runs that come EARLIER in scan order (row above, or the word on its left in the same row).
struct ck_run_links {
    uint32_t R;      // the run's pixels
    uint32_t G;      // bit j set: a link to the run of the row above that holds pixel j of the same word (one bit per such run)
    uint32_t flags;  // CK_LINK_HLEFT | CK_LINK_CROSS_L | CK_LINK_CROSS_R
};
#define CK_LINK_HLEFT 1u   /* continues the last run of the word on the left (same row) */
#define CK_LINK_CROSS_L 2u /* white: joined to the last run of the upper-left word through its pixel 31 */
#define CK_LINK_CROSS_R 4u /* white: joined to the first run of the upper-right word through its pixel 0 */
// M, U: this row's / the upper row's word of the run's colour (U = 0 on the tile's first row).  O: origin flags of the word.
// m_prev31: pixel 31 of the word on the left, same row, same colour (false for the tile's first word).
// u_prev31 / u_next0: pixel 31 / pixel 0 of the upper-left / upper-right word (false outside the tile: those links cross a
// tile boundary and belong to the merge stage).  o_next0: origin flag of the first pixel of the word on the right.
// Written without branches: on the GPU every lane of a wave evaluates it for a different run.
CK_HD ck_run_links ck_links_of_run(bool white, uint32_t M, uint32_t U, uint32_t O, int i, bool m_prev31, bool u_prev31,
                                   bool u_next0, bool o_next0) {
    ck_run_links L;
    const uint32_t S = ck_starts32(M, O);
    L.R = ck_run_bits32(M, S, i);
    const uint32_t RO = L.R & O;                                      // only origin pixels reach out
    const uint32_t diag = white ? ((RO << 1) | (RO >> 1)) : 0u;       // white also touches the two diagonals
    const uint32_t T = U & (RO | diag);                               // touched pixels of the row above
    const bool hleft = (i == 0) & ((RO & 1u) != 0) & m_prev31;
    // pixel 0's upper-left neighbour lives in the word on the left; when U has pixel 0 that run already continues it
    const bool cross_l = white & ((RO & 1u) != 0) & u_prev31 & ((U & 1u) == 0);
    // pixel 31's upper-right neighbour is pixel 0 of the word on the right; when U has pixel 31 and that pixel 0 is an
    // origin, the rule has already joined the two upper runs
    const bool cross_r = white & ((RO >> 31) != 0) & u_next0 & !(((U >> 31) != 0) & o_next0);
    L.flags = (hleft ? CK_LINK_HLEFT : 0u) | (cross_l ? CK_LINK_CROSS_L : 0u) | (cross_r ? CK_LINK_CROSS_R : 0u);
    // the touched pixels of one upper run are contiguous: keep the first of each run (a touched pixel whose left neighbour
    // is not touched, or that starts an upper run itself)
    const uint32_t Su = ck_starts32(U, O);
    L.G = T & (~(T << 1) | Su);
    return L;
}

// The same links for ALL runs of one word at once, keyed by the LOWER pixel that carries the link (what k_tile's adoption pass
// uses: one lane per word walks its runs with everything in registers):
//   Ev  pixel x joins the run that holds pixel x of the row above; one bit per pair of vertically overlapping runs (the first
//       column of the overlap)
//   DL  white pixel x joins the run that holds pixel x - 1 of the row above, which no vertical link of its run implies
//       (pixel -1 = pixel 31 of the upper-left word)
//   DR  white pixel x joins the run that STARTS at pixel x + 1 of the row above (pixel 32 = pixel 0 of the upper-right word)
//   hleft  pixel 0 continues the last run of the word on the left
// Every bit lies inside the run it belongs to; a run may carry several.  The sets are complete and free of duplicates
// (tests/cpp/links_check.cpp replays them against the oracle, too).
struct ck_word_links {
    uint32_t Ev, DL, DR;
    bool hleft;
};
CK_HD ck_word_links ck_links_of_word(bool white, uint32_t M, uint32_t U, uint32_t O, bool m_prev31, bool u_prev31, bool u_next0,
                                     bool o_next0) {
    ck_word_links L;
    const uint32_t MO = M & O, wm = white ? 0xFFFFFFFFu : 0u;
    const uint32_t V = MO & U;
    L.Ev = V & ~(V << 1);
    // up-left: the pixel above-left is set, the pixel above is not (else the vertical link — or, for a run start, the upper run
    // itself — already covers it), and the left neighbour is not an origin pixel of this run (else ITS vertical link does)
    L.DL = MO & ((U << 1) | (u_prev31 ? 1u : 0u)) & ~U & ~(MO << 1) & wm;
    // up-right: the pixel above-right is set and not already joined to a set pixel above (that needs the pixel above-right to be
    // an origin), and the right neighbour is not an origin pixel of this run
    L.DR = MO & ((U >> 1) | (u_next0 ? 0x80000000u : 0u)) & ~(U & ((O >> 1) | (o_next0 ? 0x80000000u : 0u))) & ~(MO >> 1) & wm;
    L.hleft = ((MO & 1u) != 0) & m_prev31;
    return L;
}

// ---- nodes over PAIRS of rows ----------------------------------------------------------------------------------------------
// k_tile's union-find nodes are the components of 2 x 32 pixel blocks (rows 2p and 2p + 1 of a tile, one 32-pixel word, one
// colour): dense binary noise has about half as many of them as it has one-row runs, and every per-node phase of the kernel
// (adoption, sweep, unions, flatten, sizes) shrinks with them.  Inside two rows the components are a chain: with the pixels
// of the frame's two NON-ORIGIN columns (0 and w - 1, which initiate no join) taken out of the masks, the two pixels of a
// column are always joined (the lower one joins up), so a column is a unit and the block's components are the maximal
// stretches of columns in which every column is joined to the one on its left:
//     black (4-connected): the two columns share a row                       (Mt & Mt<<1) | (Mb & Mb<<1)
//     white (8-connected): both are occupied (any two pixels of neighbouring columns are at most one row apart)
// A node's entries in the kernel's parent array sit at two pixels: its LOOKUP pixel — the pixel of its first column, the top
// one when that column has both — which every pixel of the node finds with a count-leading-zeros on the start mask, and its
// MIN pixel (its first top-row pixel when it has one, else the lookup pixel), which orders the union-find: the root of a
// component is the node with the smallest min pixel, i.e. the component's smallest pixel index, the canonical label.  Where the
// two differ the lookup pixel's entry simply points at the min pixel (it lies in the row above: parent < self holds).
// The pixels of the non-origin columns are single-pixel nodes of their own, joined by what the rule lets their origin
// neighbours do; only tiles at the frame's left / right edge have them.
// All masks passed to these functions are already ANDed with the word's origin flags.
CK_HD uint32_t ck_pair_link32(bool white, uint32_t Mt, uint32_t Mb, bool lt31, bool lb31) { // bit x: column x continues column x - 1 (bit 0: pixel 31 of the word on the left)
    const uint32_t Lt = (Mt << 1) | (lt31 ? 1u : 0u), Lb = (Mb << 1) | (lb31 ? 1u : 0u);
    return white ? ((Mt | Mb) & (Lt | Lb)) : ((Mt & Lt) | (Mb & Lb));
}
CK_HD uint32_t ck_pair_starts32(uint32_t Mt, uint32_t Mb, uint32_t link) { return (Mt | Mb) & ~(link & ~1u); } // column 0 starts a node of the word (bit 0 of link: joined across the word boundary)
// the columns of the node that starts at column s: up to the next start (trailing empty columns included: they hold no pixel)
CK_HD uint32_t ck_span32(uint32_t S, int s) {
    const uint32_t next = S & ~ck_upto32(s);
    const uint32_t lim = next ? ((next & (0u - next)) - 1u) : 0xFFFFFFFFu;
    return lim & ~((1u << s) - 1u);
}
// word-local pixel of a node's lookup entry / min entry: row (0 top, 1 bottom) << 5 | column
CK_HD int ck_pair_lookup(uint32_t Mt, int s) { return (((Mt >> s) & 1u) ? 0 : 32) + s; }
CK_HD int ck_pair_min(uint32_t Mt, uint32_t span, int s) {
    const uint32_t T = Mt & span;
    return T ? ck_ctz32(T) : 32 + s;
}

#endif
