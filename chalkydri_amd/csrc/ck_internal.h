// Internal declarations shared by the HIP translation units of libchalkydri_hip.so.
#ifndef CK_INTERNAL_H
#define CK_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "chalkydri_hip.h"

// ---- label word format (segment stage output; DESIGN.md §Data layout) --------------------------------
// 16 bits per pixel, TILE-LOCAL (the consumer knows the pixel's 32 x 128 tile from its coordinates):
// bit  15     CK_LBL_BORDER: the tile-local component touches its tile's outer ring, so it may continue in a neighbouring tile;
//             bits 0..8 are its tile-local id, SLOT = tile index * CK_RING_CAP + id in the frame's tables:
//             groot[slot] = pixel index of the frame-level root, gsize[slot] = the component's pixel count (exact while below
//             min_component_px) — two independent reads
// bit  14     CK_LBL_SMALL : tile-interior component with fewer than min_component_px pixels (final)
// bits 0..11  (neither BORDER nor all ones) tile pixel (row << 7 | column) of the component's root inside the pixel's own tile:
//             its pixel index in the frame is (tile_y0 + row) * width + tile_x0 + column
// 0xFFFF      pixel thresholded to 127 (no component)
// (Round 3: the words were 32 bits with frame-level indices — 4 of the stage's 6 bytes per pixel of HBM traffic.)
typedef uint16_t ck_label_t;
#define CK_LBL_BORDER 0x8000u
#define CK_LBL_SMALL 0x4000u
#define CK_LBL_LOCAL_MASK 0x0FFFu
#define CK_LBL_ID_MASK 0x01FFu
#define CK_LBL_NONE 0xFFFFu
#define CK_LBL_INVALID 0xFFFFFFFFu /* "no component" in CANONICAL label arrays (32-bit pixel indices: the C ABI's label output) */

// CCL tile geometry (one workgroup per tile)
#define CK_TW 128
#define CK_TH 32
#define CK_RING_CAP (2 * (CK_TW + CK_TH)) /* ring-touching roots a tile can have: one per ring pixel at most */

// frame-level resolution of a label word at pixel (x, y): the root's pixel index for an interior component, CK_LBL_INVALID for none;
// a ring-touching component (CK_LBL_BORDER) is resolved by the caller through the slot tables (ck_label_slot)
__host__ __device__ inline uint32_t ck_label_slot(uint32_t l, int x, int y, int ccl_tiles_x) {
    return (uint32_t)((y / CK_TH) * ccl_tiles_x + x / CK_TW) * (uint32_t)CK_RING_CAP + (l & CK_LBL_ID_MASK);
}
__host__ __device__ inline uint32_t ck_label_interior_root(uint32_t l, int x, int y, int w) {
    return (uint32_t)((y & ~(CK_TH - 1)) + (int)((l >> 7) & (CK_TH - 1))) * (uint32_t)w + (uint32_t)((x & ~(CK_TW - 1)) + (int)(l & (CK_TW - 1)));
}

struct ck_border_root {
    uint32_t root; // pixel index of a tile-local root whose component touches the tile ring
    uint32_t size; // its tile-local pixel count
};

struct ck_dev_family {
    uint32_t nbits, ncodes, n_upstream;
    int32_t width_at_border, total_width, reversed_border;
    const uint64_t *codes; // device
    uint32_t bit_x[64], bit_y[64];
};

// One boundary point while it waits to be grouped (k_clusters.hip)
// size classes of the quad fit (k_quads.hip), by list index: 0: 257..512 points per cluster, 1: 1025..2048, 2: <= 4096, 3: <= 8192,
// 4: <= 16384, 5: the rest (up to 3 * (2w + 2h): only frames with more than 2730 pixels of half-perimeter can have such clusters),
// 6: 513..1024, 7: <= 256 (the two youngest classes took the free indices: a cluster's cost is mostly per cluster, not per
// point, so the small ones run on small workgroups, many per CU)
constexpr int CK_FIT_CLASSES = 8;
constexpr int CK_FIT_LISTS = 9;    // work lists: one per size class + (index 8) every cluster of the batch, which the tail kernel of the split fit walks
// The split quad fit (k_quads.hip: k_fit<..., SPLIT> -> k_chunk -> k_tail) passes a cluster's sorted, de-duplicated points on as an
// EXTENDED sequence: its last CK_EXT_PRE points, the points, its first CK_EXT_POST points again — so that the windowed line-fit
// error, its smoothing and the maxima test of every point read neighbours at plain offsets and a kernel can stream over all
// clusters of a frame without knowing where one ends.  A cluster's place in the frame's sequence is handed out when it gets there
// (a per-frame counter); it needs its point count after duplicate removal + CK_EXT_HALO positions.
constexpr int CK_EXT_PRE = 25, CK_EXT_POST = 24, CK_EXT_HALO = CK_EXT_PRE + CK_EXT_POST;
constexpr int CK_SPAN = 960;       // positions one k_chunk workgroup decides (15 words of 64), CK_SPAN / 2 = most maxima it can find
constexpr int CK_HUGE_CAP = 65536, CK_HUGE_WGS = 256; // largest class: points per cluster (3 * 4 * 4095 < 65536), workgroups in its grid
constexpr int CK_FIT_PARALLEL_MAX_FRAMES = 16; // calls with at most this many frames (twice as many at quad_decimate >= 2) run the classes side by side
constexpr int CK_SEG_CHUNKS_MAX = 8;           // pieces a batch's threshold + segmentation is cut into at most
constexpr int CK_FIT_SIDE_STREAMS = 2;         // ... on the handle's stream and this many more (a process has few hardware queues)
constexpr int CK_LSCRATCH_PER_WG = 16384, CK_LSCRATCH_WGS = 1024; // large class: points per cluster, workgroups in its grid (at most)

// A boundary point inside the pipeline, packed into 32 bits: [x:13][y:13][direction:2][sign:1] at bits 28..16, 15..3, 2..1, 0.
// x/y are half-pixel coordinates, direction = which of the four forward neighbours (1,0),(0,1),(-1,1),(1,1) the pair
// spans, sign = 1 when the gradient points along it.  k_emit writes the points of one (tile, cluster) run next to each
// other; k_scatter moves whole runs to their place inside the cluster; k_fit unpacks.
typedef uint32_t ck_packed_point;
__host__ __device__ inline ck_cluster_point_t ck_unpack_point(ck_packed_point v) {
    const int k = (int)(v >> 1) & 3, sgn = (v & 1u) ? 1 : -1;
    const int dx = k == 2 ? -1 : (k == 1 ? 0 : 1), dy = k == 0 ? 0 : 1;
    ck_cluster_point_t p;
    p.x = (uint16_t)((v >> 16) & 0x1FFFu); p.y = (uint16_t)((v >> 3) & 0x1FFFu);
    p.gx = (int8_t)(dx * sgn); p.gy = (int8_t)(dy * sgn); p.pad = 0;
    return p;
}
// One (tile, cluster) run of boundary points in the temp array
struct ck_run {
    uint32_t slot;      // hash-table slot of the cluster
    uint32_t base;      // unused (the run's place inside the cluster is handed out by k_scatter)
    uint32_t tmp_start; // where the run starts in the frame's temp array
    uint32_t count;
};

// Workspace of the irregular stages, sized for cfg.max_batch frames
struct ck_stage_ws {
    int ht_size;               // hash-table slots per frame (power of two)
    int point_cap, cluster_cap, quad_cap, det_cap;
    int max_cluster_points;
    unsigned long long *d_ht_keys; // [n][ht_size]  (rep0<<32 | rep1), 0 = empty
    uint32_t *d_ht_count;      // [n][ht_size]
    uint32_t *d_ht_off;        // [n][ht_size] start of the cluster in d_points (k_scatter advances it as it fills), or 0xFFFFFFFF
    ck_packed_point *d_tmp;    // [n][ext_cap] (point_cap used) points in emission order (runs)
    ck_packed_point *d_points; // [n][ext_cap] (point_cap used) points grouped by cluster
    // split quad fit: extended point sequences and what k_chunk leaves for k_tail.  d_tmp and d_points are dead by then and carry
    // the two largest arrays (same frame pitch, so a frame's bytes never overlap another frame's)
    int ext_cap;               // positions per frame: point_cap + CK_EXT_HALO * cluster_cap, rounded up to a multiple of CK_SPAN; the frame pitch of d_tmp / d_points
    uint32_t *d_ext_xy;        // = d_tmp: [n][ext_cap] x << 13 | y (half-pixel) | window half-width << 26
    uint16_t *d_ext_w;         // [n][ext_cap] gradient weight
    double *d_maxval;          // = d_points: [n][ext_cap / CK_SPAN][CK_SPAN / 2] smoothed errors at the maxima of a span, in position order
    uint16_t *d_maxpos;        // same shape: the maximum's position inside its span
    unsigned long long *d_maxmask; // [n][ext_cap / 64] bit = position is a maximum
    uint16_t *d_maxpre;        // [n][ext_cap / 64] maxima of the span before this word
    long long *d_blk;          // [n][ext_cap / 32][6] moment sums of every aligned block of 32 positions (Mx, My, Mxx, Mxy, Myy, W)
    uint32_t *d_cstate;        // [n][cluster_cap][2] points left after duplicate removal | reversed border << 31 (0: rejected before the fit), first position of the cluster's extended sequence
    ck_run *d_runs;            // [n][run_cap]
    int run_cap;
    unsigned long long *d_lscratch; // [CK_LSCRATCH_WGS][CK_LSCRATCH_PER_WG]: sort scratch / maxima list of the large fit class, per workgroup
    unsigned long long *d_hscratch; // [2][CK_HUGE_WGS][2][hcap]: keys + sort scratch / maxima list of the largest class (null when no cluster can be that large)
    int hcap;                       // points per cluster that buffer is laid out for: max_cluster_points rounded up to 1024 (<= CK_HUGE_CAP)
    ck_cluster_t *d_clusters;  // [n][cluster_cap]
    uint32_t *d_counters;      // [n][8]: 0 tmp points, 1 clusters, 2 kept points, 3 quads, 4 detections, 5 status
    ck_quad_t *d_quads;        // [n][quad_cap]
    ck_detection_t *d_dets;    // [n][det_cap]
    uint16_t *d_wimg;          // [n][qh][qw] gradient-magnitude weight of every pixel (isqrt(gx^2+gy^2)+1; 1 on the frame ring)
    void *d_fit_scratch;       // work lists of the quad-fit classes + decode candidates
    size_t fit_scratch_bytes;
    // pose stage (glue + SQPnP), sized for max_batch frames and det_cap tags per frame
    ck_field_tag_t *d_field; int field_cap;
    double *d_gyro; uint8_t *d_has_gyro;
    ck_sqpnp_problem_t *d_problems;
    ck_iso3_t *d_pose_tags;    // [n][det_cap]
    double *d_bearings;        // [n][det_cap][4][3]
    double *d_world;           // [n][det_cap][4][3]
    ck_sqpnp_result_t *d_results;
    ck_vision_measurement_t *d_meas;
    int32_t *d_valid;
};
#define CK_CNT_TMP 0
#define CK_CNT_CLUSTERS 1
#define CK_CNT_POINTS 2
#define CK_CNT_QUADS 3
#define CK_CNT_DETS 4
#define CK_CNT_STATUS 5
#define CK_CNT_RUNS 6
#define CK_CNT_EXT 7 /* split quad fit: positions of the frame's extended sequences handed out so far */
#define CK_CNT_STRIDE 8

struct ck_handle {
    ck_config_t cfg;
    int device;
    hipStream_t stream;
    hipStream_t stream2;  // with CK_STREAMS=2 the later pieces of a batch run their irregular stages here (ck_stages.hip: run_pipeline)
    hipEvent_t ev[16];
    hipEvent_t ev_fork, ev_join;
    // a call with a few frames runs the size classes of the quad fit side by side (k_quads.hip): each class is then a handful of
    // workgroups whose time is one cluster's dependency chain, and five chains in a row were a quarter of a one-frame call
    hipStream_t fit_stream[CK_FIT_SIDE_STREAMS];
    // threshold + segmentation of a large batch runs in chunks of frames: k_fmerge of chunk i (one workgroup per frame, bound by
    // chains of dependent steps) on seg_stream beside k_tile of chunk i + 1 on the handle's stream (k_ccl.hip)
    hipStream_t seg_stream;
    hipEvent_t ev_seg[CK_SEG_CHUNKS_MAX], ev_seg_join;
    hipEvent_t ev_fit_fork, ev_fit_join[CK_FIT_SIDE_STREAMS];
    int w, h;            // full-resolution frame
    int qw, qh;          // geometry of the image the quad stages run on (w/decimate)
    int tiles_x, tiles_y;
    size_t npix;         // qw*qh
    // device buffers, sized for cfg.max_batch frames
    uint8_t *d_frames;   // staged input frames, pitch = frame_stride
    int frame_stride;
    size_t frame_pitch;
    uint8_t *d_qframes;  // decimated copy (== d_frames when quad_decimate == 1)
    uint8_t *d_thresh;   // [n][qh][qw]
    ck_label_t *d_labels; // [n][qh][qw] label words
    uint32_t *d_groot;   // [n][broot_cap] slot table: frame-level root (pixel index) of every ring-touching tile-local component
    uint32_t *d_gsize;   // [n][broot_cap] slot table: its pixel count (exact while below min_component_px)
    uint32_t *d_gscratch; // [n][2][broot_cap] parents and sizes of k_fmerge's global-memory path
    uint32_t *d_xband;    // [n][2][broot_cap] a frame joined in bands of tile rows: every slot's root within its band, the band roots' parents (k_fseam)
    ck_border_root *d_broots; // [n][2][broot_cap]: per tile a slice of CK_RING_CAP entries (k_tile), then the same entries packed (k_fmerge)
    uint32_t *d_tile_count;   // [n][tiles]: entries used in every tile's slice
    int broot_cap;            // tiles * CK_RING_CAP
    // ids of the ring-touching roots along the tile boundaries (k_tile writes them, k_fmerge joins across them), per frame:
    // HT[tiles_y][qw] top rows | HB[tiles_y][qw] bottom rows | VL[tiles_x][qh] left columns | VR[tiles_x][qh] right columns;
    // an entry = index into the frame's d_broots list | colour << 15 (1 = white), 0xFFFF = no colour
    uint16_t *d_ring;         // [n][ring_len]
    size_t ring_len;
    // later stages
    ck_stage_ws ws;      // workspace of clusters / quads / decode
    ck_stage_ms_t last_ms;
    ck_dev_family *d_fams;
    int n_staged;        // frames currently staged in d_frames
    int n_last_pose;     // records the last ck_process_* call left in ws.d_meas (what ck_gather_poses may send); -1: none yet
    bool fmerge_lds_allowed; // k_fmerge's dynamic LDS limit has been raised on this handle's device
};

extern thread_local char ck_err_text[512];
#define CK_HIP(call)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            snprintf(ck_err_text, sizeof ck_err_text, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                                   \
            (void)hipGetLastError(); /* the runtime keeps the error per thread: the next launch check must not meet it */ \
            return CK_EDEVICE;                                                                              \
        }                                                                                                   \
    } while (0)

// Diagnostic knobs (CK_TILE_STOP_AFTER, CK_FIT_SKIP, CK_FMERGE_CAP, ...: kernels cut short, classes skipped, paths forced, launch
// geometry varied) exist only in the -DCK_DIAG build (`make diag` -> chalkydri_amd/lib/diag/libchalkydri_hip.so), which the
// measurement scripts under tools/ and the path-forcing tests load.  In the product build every knob IS its default: the macro
// drops the name, so neither the getenv call nor the string is in the library a host links — the environment cannot change what
// the drop-in returns.  The product library reads exactly two variables: CK_POISON (allocation fill / guard pages, below) and
// CK_STREAMS (ck_stages.hip: the post-segmentation stages on two streams; the bytes do not depend on it).
#ifdef CK_DIAG
static inline int ck_knob_(const char *name, int dflt, int base) { const char *e = getenv(name); return e ? (int)strtol(e, nullptr, base) : dflt; }
#define CK_KNOB(name, dflt) ck_knob_(name, dflt, 10)
#define CK_KNOB0(name, dflt) ck_knob_(name, dflt, 0)   /* hexadecimal masks allowed */
#define CK_KNOB_SET(name) (getenv(name) != nullptr)
#else
#define CK_KNOB(name, dflt) (dflt)
#define CK_KNOB0(name, dflt) (dflt)
#define CK_KNOB_SET(name) (false)
#endif
int ck_streams_wanted(); // ck_stages.hip: CK_STREAMS (1 or 2), read once

// Device allocation of the handle's buffers.  CK_POISON=1 (tests) fills every buffer with 0xA5 bytes, so that a kernel which
// reads an entry nobody wrote in this call — what an undersized capacity once made of the cluster and run tables — meets the
// same garbage on every run instead of whatever the allocator happens to hand back.  CK_POISON=2 also lists the buffers.
// CK_POISON=3 gives every buffer a virtual-address range of its own, mapped so that the buffer ENDS at the end of the mapping
// and the next granule is reserved but unmapped: an access past a buffer's end is a GPU memory fault at once, wherever the
// allocator would have put its neighbours (guard pages; hipMemAddressReserve / hipMemCreate / hipMemMap).
struct ck_guarded_alloc { void *va; size_t va_bytes; hipMemGenericAllocationHandle_t mem; size_t map_bytes; };
hipError_t ck_guarded_malloc(void **p, size_t bytes); // ck_api.hip
bool ck_guarded_free(void *p);                        // true when p was a guarded allocation (and is released now)
template <typename T>
static inline hipError_t ck_malloc_dev_at(T **p, size_t bytes, const char *what, int line) {
    const char *pe = getenv("CK_POISON"); // (read per allocation: a test sets it for the handles it creates)
    const int poison = pe ? atoi(pe) : 0;
    hipError_t e = (poison >= 3 && bytes && bytes < ((size_t)1 << 30)) ? ck_guarded_malloc( // (buffers of a gigabyte and more keep the plain allocator)
                   reinterpret_cast<void **>(p), bytes) : hipMalloc(reinterpret_cast<void **>(p), bytes);
    if (e == hipSuccess && poison && bytes) {
        e = hipMemset(*p, 0xA5, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize(); // (the fill is asynchronous, and the handle's streams do not wait for the null stream)
    }
    if (poison >= 2) fprintf(stderr, "ck_alloc %p..%p %zu %s:%d\n", (void *)*p, (void *)((char *)*p + bytes), bytes, what, line);
    return e;
}
#define ck_malloc_dev(p, bytes) ck_malloc_dev_at(p, bytes, #p, __LINE__)
static inline hipError_t ck_free_dev(const void *p) {
    if (p && ck_guarded_free(const_cast<void *>(p))) return hipSuccess;
    return hipFree(const_cast<void *>(p));
}

// ---- stage launchers (k_*.hip) ------------------------------------------------------------------------
// threshold + tile-local CCL + cross-tile merge + border-root flatten, on frames [0,n) of `frames`
int ck_launch_threshold_segment(ck_handle *h, const uint8_t *frames, int stride, size_t frame_pitch, int n, bool precomputed = false);
// canonical labels (min pixel index, flags stripped) and exact sizes — parity/test path, not the hot path
int ck_launch_canonical_labels(ck_handle *h, int n, uint32_t *d_labels_out, uint32_t *d_sizes_out);
int ck_launch_decimate(ck_handle *h, const uint8_t *frames, int stride, size_t frame_pitch, int n);
// workspace of the irregular stages (clusters / quads / decode)
int ck_stage_alloc(ck_handle *h);
void ck_stage_free(ck_handle *h);
int ck_stage_device_frames(ck_handle *h, const uint8_t *d_frames, int n, int stride, int64_t frame_pitch, const uint8_t **use,
                           int *use_stride, size_t *use_pitch);
int ck_run_threshold_segment(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n);
// gradient clusters from thresh/labels/csize of frames [0,n)
int ck_launch_clusters(ck_handle *h, int n);
// quad fit (+ edge refinement) of every cluster; qframes = image the clusters came from, frames = full resolution
int ck_launch_fit_quads(ck_handle *h, const uint8_t *qframes, int qstride, size_t qpitch, const uint8_t *frames, int stride,
                        size_t pitch, int n);
int ck_launch_decode(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n);
// whole pipeline on device-resident frames (16-byte aligned rows): detections to the host / pose records
int ck_detect_frames(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n, ck_detection_t *dets, int cap, int32_t *counts,
                     uint32_t *status);
int ck_process_frames(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n, const ck_process_params_t *pp, const double *gyro,
                      const uint8_t *has_gyro, ck_vision_measurement_t *out, int32_t *valid);
// glue + SQPnP + measurement on the detections left on the device by the last pipeline run
int ck_run_pose(ck_handle *h, int n, const ck_process_params_t *pp, const double *gyro, const uint8_t *has_gyro,
                ck_vision_measurement_t *out, int32_t *valid, bool upload_field = true, bool sync = true);

// Layout of the fit scratch (ck_stage_ws::d_fit_scratch): [CK_FIT_LISTS][list_cap] work-list entries, 16 list counts,
// 16 dequeue heads, then (256-byte aligned) the decode candidates' per-frame counts and the candidates themselves.
struct ck_fit_layout {
    int list_cap;
    uint32_t *lists, *list_counts, *heads, *cand_count;
    ck_detection_t *cands;
};
static inline ck_fit_layout ck_fit_scratch_layout(const ck_stage_ws &ws, int max_batch) {
    ck_fit_layout L;
    L.list_cap = ws.cluster_cap * max_batch;
    L.lists = reinterpret_cast<uint32_t *>(ws.d_fit_scratch);
    L.list_counts = L.lists + (size_t)CK_FIT_LISTS * L.list_cap;
    L.heads = L.list_counts + 16;
    const size_t list_bytes = ((size_t)CK_FIT_LISTS * L.list_cap + 32) * sizeof(uint32_t);
    uint8_t *base = reinterpret_cast<uint8_t *>(ws.d_fit_scratch) + ((list_bytes + 255) / 256) * 256;
    L.cand_count = reinterpret_cast<uint32_t *>(base);
    L.cands = reinterpret_cast<ck_detection_t *>(base + (((size_t)max_batch * 4 + 255) / 256) * 256);
    return L;
}

#ifdef __HIPCC__
// Wave-wide inclusive sums on the DPP path (no LDS crossbar, no lane-index arithmetic): four row_shr steps inside each
// row of 16 lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Lanes without a source add 0.
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, true); }
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned long long dpp0_64(unsigned long long v) {
    return ((unsigned long long)dpp0<CTRL, ROWS>((uint32_t)(v >> 32)) << 32) | dpp0<CTRL, ROWS>((uint32_t)v);
}
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t x) {
    x += dpp0<0x111, 0xF>(x); x += dpp0<0x112, 0xF>(x); x += dpp0<0x114, 0xF>(x); x += dpp0<0x118, 0xF>(x);
    x += dpp0<0x142, 0xA>(x); x += dpp0<0x143, 0xC>(x);
    return x;
}
__device__ __forceinline__ unsigned long long wave_scan_u64(unsigned long long x) {
    x += dpp0_64<0x111, 0xF>(x); x += dpp0_64<0x112, 0xF>(x); x += dpp0_64<0x114, 0xF>(x); x += dpp0_64<0x118, 0xF>(x);
    x += dpp0_64<0x142, 0xA>(x); x += dpp0_64<0x143, 0xC>(x);
    return x;
}

__device__ __forceinline__ int wave_min_i32(int x) { // same DPP ladder with min; lanes without a source keep their own value
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x111, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x112, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x114, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x118, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(x, 63);
}

#endif // __HIPCC__

#endif
