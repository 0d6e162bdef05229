// k_clusters.hip — gradient clusters: one point per 8-neighbour pair of opposite colour whose components both
// have >= min_component_px pixels, grouped by the unordered pair of component ids.
//
// Replaces the "gradient clusters" stage of the external AprilTag-3 detector (crates/apriltags/src/lib.rs:301);
// the hash-map-of-component-pairs formulation is the one CAT was converging on (stub `ClusterHash`, `u64hash_2`
// and the 0.2*w*h cluster map at crates/chalkydri-apriltags/src/lib.rs:115-124,551-557).
//
//   k_emit     one workgroup per 16x64 pixel tile: resolves every pixel's component once into LDS (one extra
//              hop for ring-touching components, see ck_internal.h), aggregates the tile's points per cluster key
//              in an LDS hash table, then makes ONE global insert + one add per (tile, key), writes the tile's points to
//              the temp array as one contiguous run per key (4-byte packed points) and one run record per key.
//   k_scan     one workgroup per frame: prefix sums over the frame's hash table -> cluster table + point offsets
//              (clusters outside [min_cluster_pixels, max_cluster_points] are dropped here).
//   k_scatter  one wave per run: temp array -> points grouped by cluster.
#include "ck_internal.h"

namespace {

constexpr int NT = 256;
constexpr int ETW = 64, ETH = 16;         // emit tile
constexpr int LW = ETW + 2, LH = ETH + 1; // staged region: one column each side, one row below
constexpr int LHT = 512;                  // LDS hash slots (= 2 * NT: the reservation pass gives every thread two)
constexpr uint32_t SKIP = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t key_hash(unsigned long long k) {
    uint32_t x = (uint32_t)((k >> 32) ^ k); // u64hash_2 of the reference's stub
    x *= 2654435761u;
    return x ^ (x >> 15);
}

struct EmitArgs {
    const uint8_t *thresh;
    const ck_label_t *labels;
    const uint32_t *groot, *gsize; // slot tables of the ring-touching components (k_ccl.hip: k_fmerge)
    size_t slots;
    int w, h, tiles_x, tiles_y, min_comp;
    int ccl_tiles_x; // tiles per row of the segmentation stage (the label words are local to its 32 x 128 tiles)
    int stop_after; // diagnostics (CK_EMIT_STOP_AFTER): 0 staging only, 1 +count, 2 +reserve; 99 = everything
    ck_stage_ws ws;
};

// A barrier that orders LDS traffic only: __syncthreads() carries a workgroup-scope fence, i.e. a wait for EVERY outstanding memory
// operation — the barrier before the point writes would wait for the acknowledgement of the cluster-count adds nobody reads.
// Nothing a workgroup of k_emit writes to global memory is read by it again.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(NT) void k_emit(EmitArgs a) {
    __shared__ uint8_t sT[LH][LW + 2];
    __shared__ uint32_t sR[LH][LW];
    __shared__ unsigned long long sKey[LHT];
    __shared__ uint32_t sCnt[LHT], sSlot[LHT], sTBase[LHT];
    __shared__ uint32_t sWave[NT / 64 + 1], sRunW[NT / 64 + 1];
    const int tid = threadIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    const int frame = blockIdx.x / tiles, tile = blockIdx.x - frame * tiles;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * ETW, y0 = ty * ETH;
    const int w = a.w, h = a.h;
    const size_t npix = (size_t)w * h;
    const uint8_t *T = a.thresh + (size_t)frame * npix;
    const ck_label_t *L = a.labels + (size_t)frame * npix;
    const uint32_t *GR = a.groot + (size_t)frame * a.slots, *GS = a.gsize + (size_t)frame * a.slots;
    const ck_stage_ws &ws = a.ws;
    unsigned long long *gkeys = ws.d_ht_keys + (size_t)frame * ws.ht_size;
    uint32_t *gcount = ws.d_ht_count + (size_t)frame * ws.ht_size;
    uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
    ck_packed_point *tmp = ws.d_tmp + (size_t)frame * ws.ext_cap;
    ck_run *runs = ws.d_runs + (size_t)frame * ws.run_cap;

    for (int i = tid; i < LHT; i += NT) { sKey[i] = 0ull; sCnt[i] = 0; }
    {   // staging: every thread's pixels go through the (up to three) dependent loads side by side, so a thread waits for
        // three memory round trips, not three per pixel
        constexpr int SPT = (LH * LW + NT - 1) / NT;
        uint32_t lab[SPT], hop[SPT], csz[SPT], slot[SPT];
        uint8_t tv[SPT];
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const int i = tid + q * NT;
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = y0 + ly, gx = x0 - 1 + lx;
            tv[q] = 127; lab[q] = CK_LBL_NONE; slot[q] = 0;
            if (i < LH * LW && gy < h && gx >= 0 && gx < w) {
                const uint32_t p = (uint32_t)gy * (uint32_t)w + (uint32_t)gx;
                tv[q] = T[p]; lab[q] = L[p];
                slot[q] = ck_label_slot(lab[q], gx, gy, a.ccl_tiles_x);
            }
        }
        // (a word with CK_LBL_SMALL: an interior component below min_component_px — or no component: CK_LBL_NONE has every bit set)
#pragma unroll
        for (int q = 0; q < SPT; q++) { // ring-touching components: the word holds a tile-local id; root and size come from the frame's tables
            const bool two = tv[q] != 127 && !(lab[q] & CK_LBL_SMALL) && (lab[q] & CK_LBL_BORDER);
            hop[q] = two ? GR[slot[q]] : CK_LBL_INVALID;
            csz[q] = two ? GS[slot[q]] : 0u;
        }
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const int i = tid + q * NT;
            if (i >= LH * LW) continue;
            const int ly = i / LW, lx = i - ly * LW;
            uint32_t r = SKIP;
            if (tv[q] != 127 && !(lab[q] & CK_LBL_SMALL)) { // SMALL covers CK_LBL_NONE too
                if (lab[q] & CK_LBL_BORDER) r = ((int)csz[q] < a.min_comp) ? SKIP : hop[q];
                else r = ck_label_interior_root(lab[q], x0 - 1 + lx, y0 + ly, w);
            }
            sT[ly][lx] = tv[q];
            sR[ly][lx] = r;
        }
    }
    lds_barrier();
    if (a.stop_after == 0) return;

    const int lx = (tid & 63) + 1; // staged column of this thread's pixels
    const int gx = x0 + (tid & 63);
    const int row0 = (tid >> 6) * 4;
    const int dxs[4] = {1, 0, -1, 1}, dys[4] = {0, 1, 1, 1};
    const bool colok = gx >= 1 && gx <= w - 2;

    // pass 1: count points per key in the LDS table; the add's return value is the point's rank inside (tile, key),
    // kept in registers (slot << 16 | rank) so that the write pass neither probes nor counts again
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t cand[16];
#pragma unroll
    for (int q = 0; q < 16; q++) cand[q] = NONE;
    if (colok) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            int ly = row0 + rr, gy = y0 + ly;
            if (gy < 1 || gy > h - 2) continue;
            int v0 = sT[ly][lx];
            uint32_t r0 = sR[ly][lx];
            if (r0 == SKIP) continue;
            // the (up to four) points of a pixel often share their key — the three neighbours below are side by side, so when
            // they are white they are one component: one slot search and ONE add per distinct key, the first point of the key
            // doing it for the others (fewer LDS atomics, and fewer lanes on one address in each)
            uint32_t r1v[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int v1 = sT[ly + dys[k]][lx + dxs[k]];
                r1v[k] = sR[ly + dys[k]][lx + dxs[k]];
                ok[k] = (v0 + v1 == 255) && r1v[k] != SKIP;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bool leader = ok[k];
#pragma unroll
                for (int j = 0; j < k; j++) leader = leader && !(ok[j] && r1v[j] == r1v[k]);
                if (!leader) continue;
                uint32_t n = 1;
#pragma unroll
                for (int j = k + 1; j < 4; j++) n += (ok[j] && r1v[j] == r1v[k]) ? 1u : 0u;
                const uint32_t r1 = r1v[k];
                unsigned long long key = r0 < r1 ? ((unsigned long long)r0 << 32) | r1 : ((unsigned long long)r1 << 32) | r0;
                uint32_t s = key_hash(key) & (LHT - 1);
                bool placed = false;
                for (int probe = 0; probe < LHT; probe++) {
                    // most points meet their key already in place: a plain read (lanes with one address are served together)
                    // finds that out, and only a slot seen empty costs a compare-and-swap (same-address lanes one by one)
                    __asm__ volatile("" ::: "memory");
                    unsigned long long prev = sKey[s];
                    if (prev == 0ull) prev = atomicCAS(&sKey[s], 0ull, key);
                    if (prev == 0ull || prev == key) {
                        uint32_t c = (s << 16) | atomicAdd(&sCnt[s], n);
                        cand[rr * 4 + k] = c;
#pragma unroll
                        for (int j = k + 1; j < 4; j++)
                            if (ok[j] && r1v[j] == r1) cand[rr * 4 + j] = ++c;
                        placed = true;
                        break;
                    }
                    s = (s + 1) & (LHT - 1);
                }
                if (!placed) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_CLUSTERS_OVERFLOW);
            }
        }
    }
    lds_barrier();
    if (a.stop_after == 1) return;

    // pass 2: one reservation of temp space and of run records per tile (exclusive scans of the per-key counts / used
    // slots), and per (tile, key) one global insert (waited for: it yields the slot) + one add to the cluster's point
    // count that nobody waits for.
    const unsigned long long k0 = sKey[2 * tid], k1 = sKey[2 * tid + 1];
    const uint32_t c0 = k0 ? sCnt[2 * tid] : 0u, c1 = k1 ? sCnt[2 * tid + 1] : 0u;
    uint32_t found0 = SKIP, found1 = SKIP, ridx0;
    {
        const uint32_t incl = wave_scan_u32(c0 + c1);
        const unsigned long long b0 = __ballot(k0 != 0ull), b1 = __ballot(k1 != 0ull);
        const unsigned long long below = (1ull << (tid & 63)) - 1ull;
        ridx0 = (uint32_t)(__popcll(b0 & below) + __popcll(b1 & below)); // rank of this thread's first used slot in its wave
        if ((tid & 63) == 63) sWave[tid >> 6] = incl;
        if ((tid & 63) == 0) sRunW[tid >> 6] = (uint32_t)(__popcll(b0) + __popcll(b1));
        lds_barrier();
        uint32_t before = 0, total = 0, rbefore = 0, rtotal = 0;
#pragma unroll
        for (int wv = 0; wv < NT / 64; wv++) {
            uint32_t t = sWave[wv], r = sRunW[wv];
            if (wv < (tid >> 6)) { before += t; rbefore += r; }
            total += t; rtotal += r;
        }
        ridx0 += rbefore;
        lds_barrier(); // every thread has read the per-wave totals before thread 0 reuses the arrays' last entries
        if (tid == 0) {
            sWave[NT / 64] = total ? atomicAdd(&counters[CK_CNT_TMP], total) : 0u;
            sRunW[NT / 64] = rtotal ? atomicAdd(&counters[CK_CNT_RUNS], rtotal) : 0u;
        }
        const uint32_t excl = before + incl - (c0 + c1);
        sTBase[2 * tid] = excl; sTBase[2 * tid + 1] = excl + c0;
        {   // both first probes of the frame table in flight together; only a probe that met another key walks on
            const uint32_t hm = (uint32_t)(ws.ht_size - 1);
            const uint32_t g0 = key_hash(k0) & hm, g1 = key_hash(k1) & hm;
            const unsigned long long p0 = k0 ? atomicCAS(&gkeys[g0], 0ull, k0) : 0ull;
            const unsigned long long p1 = k1 ? atomicCAS(&gkeys[g1], 0ull, k1) : 0ull;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const unsigned long long key = q ? k1 : k0, pv = q ? p1 : p0;
                if (key == 0ull) continue;
                uint32_t g = q ? g1 : g0, found = SKIP;
                if (pv == 0ull || pv == key) found = g;
                else {
                    for (int probe = 1; probe < ws.ht_size; probe++) {
                        g = (g + 1) & hm;
                        unsigned long long prev = atomicCAS(&gkeys[g], 0ull, key);
                        if (prev == 0ull || prev == key) { found = g; break; }
                    }
                }
                if (found == SKIP) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_CLUSTERS_OVERFLOW);
                sSlot[2 * tid + q] = found;
                if (q) found1 = found; else found0 = found;
            }
        }
        if (found0 != SKIP) atomicAdd(&gcount[found0], c0); // result unused: nothing waits for these
        if (found1 != SKIP) atomicAdd(&gcount[found1], c1);
    }
    lds_barrier();
    if (a.stop_after == 2) return;

    // pass 3: write the points
    const uint32_t tile_base = sWave[NT / 64];
    if (colok) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            int ly = row0 + rr, gy = y0 + ly;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t cd = cand[rr * 4 + k];
                if (cd == NONE) continue;
                const uint32_t s = cd >> 16, lr = cd & 0xFFFFu;
                uint32_t slot = sSlot[s];
                if (slot == SKIP) continue;
                uint32_t ti = tile_base + sTBase[s] + lr;
                if (ti >= (uint32_t)ws.point_cap) { atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_POINTS_OVERFLOW); continue; }
                int v0 = sT[ly][lx], v1 = sT[ly + dys[k]][lx + dxs[k]];
                tmp[ti] = ((uint32_t)(2 * gx + dxs[k]) << 16) | ((uint32_t)(2 * gy + dys[k]) << 3) | ((uint32_t)k << 1) | (v1 > v0 ? 1u : 0u);
            }
        }
    }
    // run records (their place inside the cluster is decided by k_scatter)
    {
        uint32_t ri = sRunW[NT / 64] + ridx0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if ((q ? k1 : k0) == 0ull) continue;
            const uint32_t found = q ? found1 : found0;
            // (a key the frame's table had no room for still owns its place in the run list: it gets an empty record — left
            // as it was, k_scatter would follow whatever an earlier call had written there)
            if (ri < (uint32_t)ws.run_cap) {
                ck_run r;
                r.slot = found != SKIP ? found : 0u; r.base = 0; r.tmp_start = tile_base + sTBase[2 * tid + q];
                r.count = found != SKIP ? (q ? c1 : c0) : 0u;
                runs[ri] = r;
            } else if (found != SKIP) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_CLUSTERS_OVERFLOW);
            ri++;
        }
    }
}

// ---- k_emit2 (round 4): the same function as k_emit, re-cut around what the counters said about it ---------------------------------
// k_emit on the bench batch (rocprofv3 --pmc): 1 226 vector, 1 128 scalar and 171 LDS instructions per wave; by phase (stop-after knob)
// staging 0.5 ms · count pass 1.16 · reservation 0.44 · point writes 0.35.  Every pixel of the count pass re-read its centre and four
// neighbours (ten LDS reads: threshold byte and root each) before it probed.  Here a staged pixel is ONE word — the frame pixel of its
// component's root | a colour bit (the colour test and the gradient sign come from it: the threshold bytes are not staged) — and a
// thread reads the 5 x 3 words around its four pixels ONCE (15 LDS reads instead of 40) and works on them from registers; the staging
// loads are branch-free (clamped addresses, masked results).
// With frames dealt to XCDs (a frame's tiles and slot tables in one L2: -0.11 ms) and a thread remembering its last key's place (the
// pixel below often continues the same boundary: -0.05 ms) the clusters stage takes 3.35 ms against k_emit's 3.50 (same box).  What the
// count pass spends its 1.2 ms on, by ablation: the key probe (64-bit read, compare-and-swap on an empty slot, the divergent loop
// around them) 0.63 ms, the counting add 0.08, reads and arithmetic 0.45.
// Tried on the way and measured slower (1280 x 800 x 256, clusters stage, same box; k_emit: 3.51-3.56 ms):
//  * the count and write passes as straight-line code over all 16 candidate slots of a thread (a slot without a point counts 0 into
//    a per-lane dummy entry) and the tile's points ordered in LDS and copied out in one piece: fewer scalar instructions, but 32 slot
//    reads / adds per thread instead of the 7 its 3.5 real keys need, and 2.6 x the bank conflicts: 4.19 ms;
//  * ring-touching components keyed by their SLOT (CCL tile, tile-local id) in the tile's table and resolved to the frame-level root
//    once per key in the reservation pass, instead of two gathers per staged pixel: staging 0.76 -> 0.49 ms, but a component that is
//    one piece in the frame is several pieces inside a 32 x 128 tile (the white giant component of dense noise above all), so every
//    black blob beside two of its pieces became two keys, two runs, two inserts: count pass 1.18 -> 1.75 ms, 4.05 ms in all.
constexpr uint32_t ID_WHITE = 0x40000000u, ID_VALUE = 0x3FFFFFFFu; // a staged pixel: frame pixel of its root (< 2^24) | colour
constexpr int SR_PAD = ((LH * LW + NT - 1) / NT) * NT; // ids: every thread stores SPT of them unconditionally

__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(80))) void k_emit2(EmitArgs a, int xcd_map, int n_frames) {
    __shared__ __attribute__((aligned(16))) uint32_t sRaw[SR_PAD + LHT * 2 + LHT];
    __shared__ uint32_t sSlot[LHT], sTBase[LHT];
    __shared__ uint32_t sWave[NT / 64 + 1], sRunW[NT / 64 + 1];
    uint32_t *sR = sRaw;                                                             // [LH][LW] ids
    unsigned long long *sKey = reinterpret_cast<unsigned long long *>(sRaw + SR_PAD); // [LHT]
    uint32_t *sCnt = sRaw + SR_PAD + LHT * 2;                                         // [LHT]
    const int tid = threadIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    int frame, tile;
    if (xcd_map) { // workgroups b and b + 8 share an XCD: a frame's tiles (and its slot tables) stay in one L2
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        frame = (j / tiles) * 8 + x; tile = j % tiles;
        if (frame >= n_frames) return;
    } else { frame = blockIdx.x / tiles; tile = blockIdx.x - frame * tiles; }
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * ETW, y0 = ty * ETH;
    const int w = a.w, h = a.h;
    const size_t npix = (size_t)w * h;
    const uint8_t *T = a.thresh + (size_t)frame * npix;
    const ck_label_t *L = a.labels + (size_t)frame * npix;
    const uint32_t *GR = a.groot + (size_t)frame * a.slots, *GS = a.gsize + (size_t)frame * a.slots;
    const ck_stage_ws &ws = a.ws;
    unsigned long long *gkeys = ws.d_ht_keys + (size_t)frame * ws.ht_size;
    uint32_t *gcount = ws.d_ht_count + (size_t)frame * ws.ht_size;
    uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
    ck_packed_point *tmp = ws.d_tmp + (size_t)frame * ws.ext_cap;
    ck_run *runs = ws.d_runs + (size_t)frame * ws.run_cap;

    for (int i = tid; i < LHT; i += NT) { sKey[i] = 0ull; sCnt[i] = 0; }
    {   // staging: every thread's pixels go through the (up to two) dependent loads side by side (addresses clamped into the frame,
        // results masked): a pixel becomes the frame pixel of its component's root | its colour, or SKIP
        constexpr int SPT = SR_PAD / NT;
        uint32_t lab[SPT], tv[SPT], slot[SPT], hop[SPT], csz[SPT];
        bool inb[SPT];
        int gxs[SPT], gys[SPT];
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const int i = tid + q * NT;
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = y0 + ly, gx = x0 - 1 + lx;
            inb[q] = i < LH * LW && gy < h && gx >= 0 && gx < w;
            const int cy = min(gy, h - 1), cx = min(max(gx, 0), w - 1);
            const uint32_t p = (uint32_t)cy * (uint32_t)w + (uint32_t)cx;
            tv[q] = T[p]; lab[q] = L[p];
            gxs[q] = cx; gys[q] = cy;
        }
#pragma unroll
        for (int q = 0; q < SPT; q++) { // ring-touching components: the word holds a tile-local id; root and size come from the frame's tables
            // (a word with CK_LBL_SMALL: an interior component below min_component_px — or no component: CK_LBL_NONE has every bit set)
            const bool two = inb[q] && tv[q] != 127u && !(lab[q] & CK_LBL_SMALL) && (lab[q] & CK_LBL_BORDER);
            slot[q] = two ? ck_label_slot(lab[q], gxs[q], gys[q], a.ccl_tiles_x) : 0u; // (entry 0: a harmless read, unused)
            hop[q] = GR[slot[q]]; csz[q] = GS[slot[q]];
        }
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const bool brd = (lab[q] & CK_LBL_BORDER) != 0u;
            const bool none = !inb[q] || tv[q] == 127u || (lab[q] & CK_LBL_SMALL) || (brd && (int)csz[q] < a.min_comp);
            const uint32_t root = brd ? hop[q] : ck_label_interior_root(lab[q], gxs[q], gys[q], w);
            sR[tid + q * NT] = none ? SKIP : (root | (tv[q] == 255u ? ID_WHITE : 0u));
        }
    }
    lds_barrier();
    if (a.stop_after == 0) return;

    const int lx = (tid & 63) + 1; // staged column of this thread's pixels
    const int gx = x0 + (tid & 63);
    const int row0 = (tid >> 6) * 4;
    const bool colok = gx >= 1 && gx <= w - 2;

    // pass 1: count points per key in the LDS table; the add's return value is the point's rank inside (tile, key), kept in
    // registers (sign << 31 | slot << 16 | rank) so that the write pass neither probes nor counts again.  The thread's 5 x 3 ids
    // are read once; the (up to four) points of a pixel often share their key — the three neighbours below are side by side, so
    // when they are white they are one component — and then make ONE add, the first point of the key doing it for the others.
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t cand[16];
    {
        uint32_t R[5][3];
        unsigned long long last_key = 0ull; // the thread's last key and where it is: the pixel below often continues the same boundary
        uint32_t last_s = 0;
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) R[r][c] = sR[(row0 + r) * LW + lx - 1 + c];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int gy = y0 + row0 + rr;
            const uint32_t r0 = (colok && gy >= 1 && gy <= h - 2) ? R[rr][1] : SKIP;
            const uint32_t r1v[4] = {R[rr][2], R[rr + 1][1], R[rr + 1][0], R[rr + 1][2]}; // (1,0) (0,1) (-1,1) (1,1)
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; k++) ok[k] = r0 != SKIP && r1v[k] != SKIP && ((r0 ^ r1v[k]) & ID_WHITE) != 0u;
            uint32_t lead[4], nth[4], mult[4]; // a point's leader (first point of its key), its place among the key's points, how many the leader stands for
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lead[k] = (uint32_t)k; nth[k] = 0; mult[k] = 1;
#pragma unroll
                for (int j = k - 1; j >= 0; j--) { const bool eq = ok[j] && r1v[j] == r1v[k]; lead[k] = eq ? (uint32_t)j : lead[k]; nth[k] += eq ? 1u : 0u; }
#pragma unroll
                for (int j = k + 1; j < 4; j++) mult[k] += (ok[j] && r1v[j] == r1v[k]) ? 1u : 0u;
            }
            uint32_t cl[4] = {NONE, NONE, NONE, NONE};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!(ok[k] && lead[k] == (uint32_t)k)) continue; // only the first point of a key probes and adds (LDS traffic is what this pass is bound by)
                const uint32_t ra = r0 & ID_VALUE, rb = r1v[k] & ID_VALUE; // (the roots without their colour bits: the cluster's key)
                const unsigned long long key = ra < rb ? ((unsigned long long)ra << 32) | rb : ((unsigned long long)rb << 32) | ra;
                uint32_t s = key_hash(key) & (LHT - 1);
                bool placed = false;
                if (key == last_key) { placed = true; s = last_s; } // (the probe is the expensive part of this pass: 0.6 of its 1.2 ms)
                else
                for (int probe = 0; probe < LHT; probe++) {
                    // most points meet their key already in place: a plain read (lanes with one address are served together)
                    // finds that out, and only a slot seen empty costs a compare-and-swap (same-address lanes one by one)
                    __asm__ volatile("" ::: "memory");
                    unsigned long long prev = sKey[s];
                    if (prev == 0ull) prev = atomicCAS(&sKey[s], 0ull, key);
                    if (prev == 0ull || prev == key) { placed = true; break; }
                    s = (s + 1) & (LHT - 1);
                }
                if (placed) { last_key = key; last_s = s; }
                if (placed) cl[k] = (s << 16) | atomicAdd(&sCnt[s], mult[k]);
                else atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_CLUSTERS_OVERFLOW);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t lk = lead[k];
                const uint32_t c = lk == 0u ? cl[0] : (lk == 1u ? cl[1] : (lk == 2u ? cl[2] : cl[3]));
                const uint32_t sign = (r1v[k] & ID_WHITE) ? 0x80000000u : 0u; // gradient sign: the neighbour is the white one
                cand[rr * 4 + k] = (ok[k] && c != NONE) ? ((c + nth[k]) | sign) : NONE;
            }
        }
    }
    lds_barrier();
    if (a.stop_after == 1) return;

    // pass 2: one reservation of temp space and of run records per tile (exclusive scans of the per-key counts / used entries),
    // and per (tile, key) one insert into the frame's table (waited for: it yields the slot) + one add to the cluster's point
    // count that nobody waits for.
    const unsigned long long k0 = sKey[2 * tid], k1 = sKey[2 * tid + 1];
    const uint32_t c0 = k0 ? sCnt[2 * tid] : 0u, c1 = k1 ? sCnt[2 * tid + 1] : 0u;
    uint32_t found0 = SKIP, found1 = SKIP, ridx0, tile_total;
    {
        const uint32_t incl = wave_scan_u32(c0 + c1);
        const unsigned long long b0 = __ballot(k0 != 0ull), b1 = __ballot(k1 != 0ull);
        const unsigned long long below = (1ull << (tid & 63)) - 1ull;
        ridx0 = (uint32_t)(__popcll(b0 & below) + __popcll(b1 & below)); // rank of this thread's first used entry in its wave
        if ((tid & 63) == 63) sWave[tid >> 6] = incl;
        if ((tid & 63) == 0) sRunW[tid >> 6] = (uint32_t)(__popcll(b0) + __popcll(b1));
        lds_barrier();
        uint32_t before = 0, total = 0, rbefore = 0, rtotal = 0;
#pragma unroll
        for (int wv = 0; wv < NT / 64; wv++) {
            uint32_t t = sWave[wv], r = sRunW[wv];
            if (wv < (tid >> 6)) { before += t; rbefore += r; }
            total += t; rtotal += r;
        }
        ridx0 += rbefore;
        tile_total = total;
        if (tid == 0) { // (entries NT / 64 of the two arrays are not among those the loop above reads)
            sWave[NT / 64] = total ? atomicAdd(&counters[CK_CNT_TMP], total) : 0u;
            sRunW[NT / 64] = rtotal ? atomicAdd(&counters[CK_CNT_RUNS], rtotal) : 0u;
        }
        const uint32_t excl = before + incl - (c0 + c1);
        sTBase[2 * tid] = excl; sTBase[2 * tid + 1] = excl + c0;
        {   // both first probes of the frame table in flight together; only a probe that met another key walks on
            const uint32_t hm = (uint32_t)(ws.ht_size - 1);
            const uint32_t g0 = key_hash(k0) & hm, g1 = key_hash(k1) & hm;
            const unsigned long long p0 = k0 ? atomicCAS(&gkeys[g0], 0ull, k0) : 0ull;
            const unsigned long long p1 = k1 ? atomicCAS(&gkeys[g1], 0ull, k1) : 0ull;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const unsigned long long key = q ? k1 : k0, pv = q ? p1 : p0;
                uint32_t found = SKIP;
                if (key != 0ull) {
                    uint32_t g = q ? g1 : g0;
                    if (pv == 0ull || pv == key) found = g;
                    else {
                        for (int probe = 1; probe < ws.ht_size; probe++) {
                            g = (g + 1) & hm;
                            unsigned long long prev = atomicCAS(&gkeys[g], 0ull, key);
                            if (prev == 0ull || prev == key) { found = g; break; }
                        }
                    }
                    if (found == SKIP) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_CLUSTERS_OVERFLOW);
                }
                sSlot[2 * tid + q] = found;
                if (q) found1 = found; else found0 = found;
            }
        }
        if (found0 != SKIP) atomicAdd(&gcount[found0], c0); // result unused: nothing waits for these
        if (found1 != SKIP) atomicAdd(&gcount[found1], c1);
    }
    lds_barrier();
    if (a.stop_after == 2) return;

    // pass 3: write the points (one contiguous run per key in the frame's temp array)
    const uint32_t tile_base = sWave[NT / 64];
    (void)tile_total;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t cd = cand[q];
        if (cd == NONE) continue;
        const int rr = q >> 2, k = q & 3;
        const int dxk = k == 1 ? 0 : (k == 2 ? -1 : 1), dyk = k == 0 ? 0 : 1;
        const uint32_t s = (cd >> 16) & 0x3FFu, lr = cd & 0xFFFFu;
        if (sSlot[s] == SKIP) continue; // (a key dropped in pass 2: a member below min_component_px, or no room in the frame's table)
        const uint32_t ti = tile_base + sTBase[s] + lr;
        if (ti >= (uint32_t)ws.point_cap) { atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_POINTS_OVERFLOW); continue; }
        const int gy = y0 + row0 + rr;
        tmp[ti] = ((uint32_t)(2 * gx + dxk) << 16) | ((uint32_t)(2 * gy + dyk) << 3) | ((uint32_t)k << 1) | (cd >> 31);
    }
    // run records (their place inside the cluster is decided by k_scatter)
    {
        uint32_t ri = sRunW[NT / 64] + ridx0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if ((q ? k1 : k0) == 0ull) continue;
            const uint32_t found = q ? found1 : found0;
            // (a key the frame's table had no room for still owns its place in the run list: it gets an empty record — left
            // as it was, k_scatter would follow whatever an earlier call had written there)
            if (ri < (uint32_t)ws.run_cap) {
                ck_run r;
                r.slot = found != SKIP ? found : 0u; r.base = 0; r.tmp_start = tile_base + (q ? sTBase[2 * tid + 1] : sTBase[2 * tid]);
                r.count = found != SKIP ? (q ? c1 : c0) : 0u;
                runs[ri] = r;
            } else if (found != SKIP) atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_CLUSTERS_OVERFLOW);
            ri++;
        }
    }
}

// zeroes the frames' key and count tables (16 bytes per thread) and, with its first threads, three short arrays
__global__ __launch_bounds__(NT) void k_clear(uint4 *keys4, size_t nkeys4, uint4 *cnt4, size_t ncnt4, uint32_t *c0, uint32_t n0, uint32_t *c1,
                                              uint32_t n1, uint32_t *c2, uint32_t n2) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (i < nkeys4) keys4[i] = z;
    else if (i - nkeys4 < ncnt4) cnt4[i - nkeys4] = z;
    if (i < n0) c0[i] = 0;
    if (i < n1) c1[i] = 0;
    if (i < n2) c2[i] = 0;
}

// ---- per-frame scan of the hash table -------------------------------------------------------------------------
constexpr int SNT = 1024;
// One workgroup per frame; every wave owns a contiguous 1/16 of the table and walks it 64 slots at a time (coalesced
// reads, DPP wave scans, running offsets in registers): one pass for the wave totals, one to hand out the offsets.
__global__ __launch_bounds__(SNT) void k_scan(ck_stage_ws ws, int min_cluster, int max_cluster) {
    __shared__ uint32_t sPts[SNT / 64], sCl[SNT / 64];
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long *gkeys = ws.d_ht_keys + (size_t)frame * ws.ht_size;
    const uint32_t *gcount = ws.d_ht_count + (size_t)frame * ws.ht_size;
    uint32_t *goff = ws.d_ht_off + (size_t)frame * ws.ht_size;
    uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
    ck_cluster_t *clusters = ws.d_clusters + (size_t)frame * ws.cluster_cap;
    const int seg = ws.ht_size / (SNT / 64); // ht_size is a multiple of SNT, so seg is a multiple of 64
    const int base = wv * seg;
    uint32_t pts = 0, cl = 0;
#pragma unroll 8
    for (int r = 0; r < seg; r += 64) { // (unrolled: eight loads in flight; as a plain loop every 64 slots cost a trip to memory)
        uint32_t c = gcount[base + r + lane];
        bool ok = (int)c >= min_cluster && (int)c <= max_cluster;
        pts += ok ? c : 0u; cl += ok ? 1u : 0u;
    }
    pts = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_u32(pts), 63);
    cl = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_u32(cl), 63);
    if (lane == 0) { sPts[wv] = pts; sCl[wv] = cl; }
    __syncthreads();
    uint32_t po = 0, co = 0;
    for (int k = 0; k < wv; k++) { po += sPts[k]; co += sCl[k]; }
    // (the counts of four rounds are fetched together, then worked off in order; seg is a multiple of 64, not always of 256)
    for (int r4 = 0; r4 < seg; r4 += 256) {
        uint32_t c4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) c4[q] = r4 + 64 * q < seg ? gcount[base + r4 + 64 * q + lane] : 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = r4 + 64 * q;
        if (r >= seg) continue; // (uniform)
        const int e = base + r + lane;
        const uint32_t c = c4[q];
        const bool ok = (int)c >= min_cluster && (int)c <= max_cluster;
        const uint32_t ip = wave_scan_u32(ok ? c : 0u), ic = wave_scan_u32(ok ? 1u : 0u);
        uint32_t off = SKIP;
        if (ok) {
            const uint32_t my_po = po + ip - c, my_co = co + ic - 1u;
            if (my_co < (uint32_t)ws.cluster_cap && my_po + c <= (uint32_t)ws.point_cap) {
                off = my_po;
                unsigned long long key = gkeys[e];
                ck_cluster_t ck;
                ck.rep0 = (uint32_t)(key >> 32); ck.rep1 = (uint32_t)key; ck.start = my_po; ck.count = c;
                clusters[my_co] = ck;
            } else {
                // no room for the cluster's points (or for its record): it is dropped.  Its record must not stay what an earlier
                // call left there — the count of clusters below includes it — so it becomes an empty one, which k_classify skips
                if (my_co < (uint32_t)ws.cluster_cap) {
                    ck_cluster_t ck;
                    ck.rep0 = 0; ck.rep1 = 0; ck.start = 0; ck.count = 0;
                    clusters[my_co] = ck;
                }
                atomicOr(&counters[CK_CNT_STATUS], (uint32_t)CK_FRAME_CLUSTERS_OVERFLOW);
            }
        }
        goff[e] = off;
        po += (uint32_t)__builtin_amdgcn_readlane((int)ip, 63);
        co += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63);
    }
    }
    if (tid == SNT - 1) {
        counters[CK_CNT_CLUSTERS] = min(co, (uint32_t)ws.cluster_cap);
        counters[CK_CNT_POINTS] = min(po, (uint32_t)ws.point_cap);
    }
}

// One wave per run: a (tile, cluster) run of the temp array goes to its place inside the cluster, coalesced on both sides.
// RPW: run records a wave takes per round — 64 for batches; a call with a few frames has too few runs to keep the chip busy
// that way (8 k runs = 128 waves), so it hands them out 8 at a time over eight times as many waves
template <int RPW>
__global__ __launch_bounds__(NT) void k_scatter(ck_stage_ws ws) {
    const int frame = blockIdx.y;
    const uint32_t *counters = ws.d_counters + (size_t)frame * CK_CNT_STRIDE;
    const uint32_t nruns = min(counters[CK_CNT_RUNS], (uint32_t)ws.run_cap);
    const uint32_t ntmp = min(counters[CK_CNT_TMP], (uint32_t)ws.point_cap);
    const ck_packed_point *tmp = ws.d_tmp + (size_t)frame * ws.ext_cap;
    const ck_run *runs = ws.d_runs + (size_t)frame * ws.run_cap;
    uint32_t *goff = ws.d_ht_off + (size_t)frame * ws.ht_size;
    ck_packed_point *pts = ws.d_points + (size_t)frame * ws.ext_cap;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (NT / 64);
    // 64 runs per wave and round: every lane fetches one run record and its cluster offset (two dependent loads paid once
    // for 64 runs), then the wave copies the runs one after the other (runs of up to 64 points: 16 lanes each, four runs at a time)
    for (uint32_t r0 = wave * RPW; r0 < nruns; r0 += nwaves * RPW) {
        ck_run mine = {0, 0, 0, 0};
        uint32_t moff = SKIP;
        if (lane < RPW && r0 + lane < nruns) { mine = runs[r0 + lane]; moff = goff[mine.slot]; }
        if (moff == SKIP) mine.count = 0; // cluster dropped by k_scan (too small / too large / no room)
        // goff[slot] starts as the cluster's offset and serves as its fill cursor: the add returns where this run goes
        const uint32_t dst0 = mine.count ? atomicAdd(&goff[mine.slot], mine.count) : 0u;
        // short runs first, four per step
        const unsigned long long small = __ballot(mine.count > 0 && mine.count <= 64);
        unsigned long long todo = small;
        const int sub = lane >> 4, sl = lane & 15;
        while (todo) {
            // pick up to four set bits of `todo`; sub-group `sub` takes the sub-th of them
            unsigned long long t = todo;
            int pick = -1;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int bit = t ? __builtin_ctzll(t) : -1;
                if (q == sub) pick = bit;
                if (t) t &= t - 1;
            }
            todo = t;
            const int srcl = pick < 0 ? 0 : pick;
            const uint32_t cnt = (uint32_t)__shfl((int)mine.count, srcl, 64), ts = (uint32_t)__shfl((int)mine.tmp_start, srcl, 64),
                           ds = (uint32_t)__shfl((int)dst0, srcl, 64);
            ck_packed_point v[4];
            bool ok[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { // all loads of the step are in flight before the first store
                const uint32_t j = (uint32_t)sl + 16u * q;
                ok[q] = pick >= 0 && j < cnt && ts + j < ntmp;
                v[q] = ok[q] ? tmp[ts + j] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (ok[q]) pts[ds + (uint32_t)sl + 16u * q] = v[q];
        }
        unsigned long long big = __ballot(mine.count > 64);
        while (big) {
            const int srcl = __builtin_ctzll(big);
            big &= big - 1;
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)mine.count, srcl), ts = (uint32_t)__builtin_amdgcn_readlane((int)mine.tmp_start, srcl),
                           ds = (uint32_t)__builtin_amdgcn_readlane((int)dst0, srcl);
            for (uint32_t j0 = 0; j0 < cnt; j0 += 256) {
                ck_packed_point v[4];
                bool ok[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t j = j0 + (uint32_t)lane + 64u * q;
                    ok[q] = j < cnt && ts + j < ntmp;
                    v[q] = ok[q] ? tmp[ts + j] : 0u;
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (ok[q]) pts[ds + j0 + (uint32_t)lane + 64u * q] = v[q];
            }
        }
    }
}

} // namespace

int ck_launch_clusters(ck_handle *h, int n) {
    ck_stage_ws &ws = h->ws;
    {   // one launch clears everything the rest of the call counts into: the frames' key / count tables and counters, the fit's
        // list counts and dequeue heads, the decode candidates' counts (five fills before; their launches were a twentieth of
        // a one-frame call)
        const ck_fit_layout fl = ck_fit_scratch_layout(ws, h->cfg.max_batch);
        const size_t nkeys4 = (size_t)ws.ht_size * n / 2, ncnt4 = (size_t)ws.ht_size * n / 4; // ht_size is a multiple of 1024
        const size_t total4 = nkeys4 + ncnt4;
        hipLaunchKernelGGL(k_clear, dim3((unsigned)((total4 + NT - 1) / NT)), dim3(NT), 0, h->stream,
                           reinterpret_cast<uint4 *>(ws.d_ht_keys), nkeys4, reinterpret_cast<uint4 *>(ws.d_ht_count), ncnt4, ws.d_counters,
                           (uint32_t)(CK_CNT_STRIDE * n), fl.list_counts, 32u, fl.cand_count, (uint32_t)n);
    }
    EmitArgs a;
    a.thresh = h->d_thresh; a.labels = h->d_labels; a.groot = h->d_groot; a.gsize = h->d_gsize; a.slots = (size_t)h->broot_cap;
    a.w = h->qw; a.h = h->qh; a.tiles_x = (h->qw + ETW - 1) / ETW; a.tiles_y = (h->qh + ETH - 1) / ETH;
    a.min_comp = h->cfg.min_component_px; a.ws = ws; a.ccl_tiles_x = h->tiles_x;
    { static const int stop_after = CK_KNOB("CK_EMIT_STOP_AFTER", 99); a.stop_after = stop_after; }
    static const int emit_v = CK_KNOB("CK_EMIT_V", 2);       // (diagnostics: 1 = the round-1..3 kernel, for A/B)
    static const int emit_xcd = CK_KNOB("CK_EMIT_XCD", -1);  // (diagnostics: frames dealt to XCDs 0 / 1; default: batches of 16 frames and more)
    if (emit_v == 1) hipLaunchKernelGGL(k_emit, dim3((unsigned)(a.tiles_x * a.tiles_y * n)), dim3(NT), 0, h->stream, a);
    else {
        const int xcd_map = emit_xcd >= 0 ? emit_xcd : (n >= 16 ? 1 : 0);
        const unsigned grid = xcd_map ? (unsigned)(((n + 7) / 8) * 8 * a.tiles_x * a.tiles_y) : (unsigned)(a.tiles_x * a.tiles_y * n);
        hipLaunchKernelGGL(k_emit2, dim3(grid), dim3(NT), 0, h->stream, a, xcd_map, n);
    }
    int min_cluster = h->cfg.min_cluster_pixels < 24 ? 24 : h->cfg.min_cluster_pixels;
    hipLaunchKernelGGL(k_scan, dim3((unsigned)n), dim3(SNT), 0, h->stream, ws, min_cluster, ws.max_cluster_points);
    if (n <= 4) hipLaunchKernelGGL(k_scatter<8>, dim3(256u, (unsigned)n), dim3(NT), 0, h->stream, ws);
    else {
        static const unsigned sc_wgs = (unsigned)CK_KNOB("CK_SCATTER_WGS", 32); // (diagnostics: workgroups per frame)
        hipLaunchKernelGGL(k_scatter<64>, dim3(sc_wgs, (unsigned)n), dim3(NT), 0, h->stream, ws);
    }
    CK_HIP(hipGetLastError());
    return CK_OK;
}
