// ck_stages.hip — workspace of the irregular stages and the detect / clusters / quads entry points.
#include <string.h>

#include <vector>

#include "ck_internal.h"

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// an allocation that fails for lack of memory is CK_ENOMEM, as in ck_create (not the CK_EDEVICE of a runtime failure)
#define CK_ALLOC(call)                                                                                     \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            snprintf(ck_err_text, sizeof ck_err_text, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                                   \
            (void)hipGetLastError();                                                                        \
            return e_ == hipErrorOutOfMemory ? CK_ENOMEM : CK_EDEVICE;                                      \
        }                                                                                                   \
    } while (0)
int ck_stage_alloc(ck_handle *h) {
    ck_stage_ws &ws = h->ws;
    const ck_config_t &cfg = h->cfg;
    const size_t nb = (size_t)cfg.max_batch;
    const int npix = (int)h->npix;
    ws.point_cap = cfg.max_points_per_frame > 0 ? cfg.max_points_per_frame : 4 * npix; // a pixel has four forward neighbours: no frame has more
    ws.cluster_cap = cfg.max_clusters_per_frame > 0 ? cfg.max_clusters_per_frame : npix / 32;
    if (ws.cluster_cap < 1024) ws.cluster_cap = 1024;
    if (ws.cluster_cap > (1 << 19)) ws.cluster_cap = 1 << 19; // keeps the hash table (2x) at most 2^20 slots per frame
    ws.quad_cap = cfg.max_quads_per_frame > 0 ? cfg.max_quads_per_frame : 1024;
    // k_finalize ranks a frame's decode candidates (quad_cap per family) in LDS, 4 bytes each: refuse what would not launch
    if ((size_t)ws.quad_cap * (size_t)cfg.n_families * sizeof(int) > 64 * 1024) return CK_EINVAL;
    ws.det_cap = 256;
    ws.ht_size = next_pow2(2 * ws.cluster_cap);
    if (ws.ht_size < 1024) ws.ht_size = 1024;
    ws.max_cluster_points = 3 * (2 * h->qw + 2 * h->qh); // AprilTag-3's bound; <= 3 * 4 * 4095 < CK_HUGE_CAP (ck_create bounds the sides)
    if (cfg.max_nmaxima < 4 || cfg.max_nmaxima > 12) return CK_EINVAL;
    CK_ALLOC(ck_malloc_dev(&ws.d_ht_keys, sizeof(unsigned long long) * (size_t)ws.ht_size * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_ht_count, sizeof(uint32_t) * (size_t)ws.ht_size * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_ht_off, sizeof(uint32_t) * (size_t)ws.ht_size * nb));
    {   // frame pitch of the point arrays = positions of the split fit's extended sequences (ck_internal.h)
        const size_t e = (size_t)ws.point_cap + (size_t)CK_EXT_HALO * ws.cluster_cap;
        const size_t r = (e + CK_SPAN - 1) / CK_SPAN * CK_SPAN;
        if (r > 0x7FFFFFFFu - 4096) return CK_EINVAL;
        ws.ext_cap = (int)r;
    }
    CK_ALLOC(ck_malloc_dev(&ws.d_tmp, sizeof(ck_packed_point) * (size_t)ws.ext_cap * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_points, sizeof(ck_packed_point) * (size_t)ws.ext_cap * nb));
    ws.d_ext_xy = ws.d_tmp; ws.d_maxval = reinterpret_cast<double *>(ws.d_points);
    CK_ALLOC(ck_malloc_dev(&ws.d_ext_w, sizeof(uint16_t) * (size_t)ws.ext_cap * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_maxpos, sizeof(uint16_t) * (size_t)(ws.ext_cap / 2) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_maxmask, sizeof(unsigned long long) * (size_t)(ws.ext_cap / 64) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_maxpre, sizeof(uint16_t) * (size_t)(ws.ext_cap / 64) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_blk, sizeof(long long) * 6 * (size_t)(ws.ext_cap / 32) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_cstate, sizeof(uint32_t) * 2 * (size_t)ws.cluster_cap * nb));
    ws.run_cap = 4 * ws.cluster_cap;
    CK_ALLOC(ck_malloc_dev(&ws.d_runs, sizeof(ck_run) * (size_t)ws.run_cap * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_lscratch, 2 * sizeof(unsigned long long) * (size_t)CK_LSCRATCH_PER_WG * CK_LSCRATCH_WGS));
    ws.d_hscratch = nullptr;
    ws.hcap = (ws.max_cluster_points + 1023) & ~1023; // (1920 x 1080: 18 432 points, 144 MiB instead of the 512 MiB of the class's template capacity)
    if (ws.hcap < 16384 && CK_KNOB_SET("CK_FIT_GK")) ws.hcap = 16384;
    if (ws.max_cluster_points > 16384 || CK_KNOB_SET("CK_FIT_GK")) // (one copy per stream of a split batch)
        CK_ALLOC(ck_malloc_dev(&ws.d_hscratch, 2 * sizeof(unsigned long long) * 2 * (size_t)ws.hcap * CK_HUGE_WGS));
    CK_ALLOC(ck_malloc_dev(&ws.d_clusters, sizeof(ck_cluster_t) * (size_t)ws.cluster_cap * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_counters, sizeof(uint32_t) * CK_CNT_STRIDE * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_quads, sizeof(ck_quad_t) * (size_t)ws.quad_cap * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_dets, sizeof(ck_detection_t) * (size_t)ws.det_cap * nb));
    // fit scratch: one work list per size class + counters, then the decode candidates
    size_t list_bytes = ((size_t)CK_FIT_LISTS * ws.cluster_cap * nb + 32) * sizeof(uint32_t);
    size_t cand_bytes = 256 + ((nb * 4 + 255) / 256) * 256 + sizeof(ck_detection_t) * (size_t)ws.quad_cap * cfg.n_families * nb;
    ws.fit_scratch_bytes = ((list_bytes + 255) / 256) * 256 + cand_bytes;
    CK_ALLOC(ck_malloc_dev(&ws.d_fit_scratch, 2 * ws.fit_scratch_bytes)); // second copy: the half-batch that runs on stream2
    CK_ALLOC(ck_malloc_dev(&ws.d_wimg, sizeof(uint16_t) * h->npix * nb));
    ws.field_cap = 1024;
    CK_ALLOC(ck_malloc_dev(&ws.d_field, sizeof(ck_field_tag_t) * (size_t)ws.field_cap));
    CK_ALLOC(ck_malloc_dev(&ws.d_gyro, sizeof(double) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_has_gyro, nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_problems, sizeof(ck_sqpnp_problem_t) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_pose_tags, sizeof(ck_iso3_t) * nb * ws.det_cap));
    CK_ALLOC(ck_malloc_dev(&ws.d_bearings, sizeof(double) * 12 * nb * ws.det_cap));
    CK_ALLOC(ck_malloc_dev(&ws.d_world, sizeof(double) * 12 * nb * ws.det_cap));
    CK_ALLOC(ck_malloc_dev(&ws.d_results, sizeof(ck_sqpnp_result_t) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_meas, sizeof(ck_vision_measurement_t) * nb));
    CK_ALLOC(ck_malloc_dev(&ws.d_valid, sizeof(int32_t) * nb));
    // family tables
    std::vector<ck_dev_family> fams((size_t)cfg.n_families);
    for (int f = 0; f < cfg.n_families; f++) {
        const ck_family_t *src = cfg.families[f];
        ck_dev_family &d = fams[(size_t)f];
        memset(&d, 0, sizeof d);
        d.nbits = src->nbits; d.ncodes = src->ncodes; d.n_upstream = src->n_upstream ? src->n_upstream : src->ncodes; /* 0 (a table built against ABI v1, or zero-initialised): the caller vouches for all of it */ d.width_at_border = src->width_at_border;
        d.total_width = src->total_width; d.reversed_border = src->reversed_border;
        for (uint32_t i = 0; i < src->nbits; i++) { d.bit_x[i] = src->bit_x[i]; d.bit_y[i] = src->bit_y[i]; }
        uint64_t *dc = nullptr;
        CK_ALLOC(ck_malloc_dev(&dc, sizeof(uint64_t) * src->ncodes));
        CK_HIP(hipMemcpy(dc, src->codes, sizeof(uint64_t) * src->ncodes, hipMemcpyHostToDevice));
        d.codes = dc;
    }
    CK_ALLOC(ck_malloc_dev(&h->d_fams, sizeof(ck_dev_family) * fams.size()));
    CK_HIP(hipMemcpy(h->d_fams, fams.data(), sizeof(ck_dev_family) * fams.size(), hipMemcpyHostToDevice));
    return CK_OK;
}
#undef CK_ALLOC

void ck_stage_free(ck_handle *h) {
    ck_stage_ws &ws = h->ws;
    if (h->d_fams) {
        std::vector<ck_dev_family> fams((size_t)h->cfg.n_families);
        if (hipMemcpy(fams.data(), h->d_fams, sizeof(ck_dev_family) * fams.size(), hipMemcpyDeviceToHost) == hipSuccess)
            for (auto &f : fams) (void)ck_free_dev(const_cast<uint64_t *>(f.codes));
        (void)ck_free_dev(h->d_fams);
    }
    (void)ck_free_dev(ws.d_ht_keys); (void)ck_free_dev(ws.d_ht_count); (void)ck_free_dev(ws.d_ht_off); (void)ck_free_dev(ws.d_tmp);
    (void)ck_free_dev(ws.d_ext_w); (void)ck_free_dev(ws.d_maxpos); (void)ck_free_dev(ws.d_maxmask); (void)ck_free_dev(ws.d_maxpre); (void)ck_free_dev(ws.d_blk);
    (void)ck_free_dev(ws.d_cstate);
    (void)ck_free_dev(ws.d_points); (void)ck_free_dev(ws.d_runs); (void)ck_free_dev(ws.d_lscratch); (void)ck_free_dev(ws.d_hscratch); (void)ck_free_dev(ws.d_clusters); (void)ck_free_dev(ws.d_counters); (void)ck_free_dev(ws.d_quads);
    (void)ck_free_dev(ws.d_dets); (void)ck_free_dev(ws.d_fit_scratch); (void)ck_free_dev(ws.d_wimg);
    (void)ck_free_dev(ws.d_field); (void)ck_free_dev(ws.d_gyro); (void)ck_free_dev(ws.d_has_gyro); (void)ck_free_dev(ws.d_problems);
    (void)ck_free_dev(ws.d_pose_tags); (void)ck_free_dev(ws.d_bearings); (void)ck_free_dev(ws.d_world); (void)ck_free_dev(ws.d_results);
    (void)ck_free_dev(ws.d_meas); (void)ck_free_dev(ws.d_valid);
}

// the whole detector on n frames resident on the device
// The stages after segmentation can run as consecutive pieces of the batch on two streams (CK_STREAMS=2), each piece filling
// the other's gaps (one-workgroup-per-frame kernels, latency chains).  Threshold + segmentation always run for the whole
// batch on the handle's stream (that is the stage the HBM roofline is quoted on); then the later pieces continue
// on stream2 through a VIEW of the handle — a copy whose per-frame pointers are advanced by n0 frames and whose scratch
// regions are the second copies allocated for it.  Frames are independent, so the results do not depend on the split.
int ck_streams_wanted() {
    // default 1: since the fit kernels dequeue their clusters in chunks (no long tail left to fill) two dense kernels side by
    // side only get in each other's way (21.8 vs 22.4 ms per 1280x800x256 batch); CK_STREAMS=2 keeps the split available
    static const int v = getenv("CK_STREAMS") ? atoi(getenv("CK_STREAMS")) : 1;
    return v;
}
static ck_handle make_view(const ck_handle *h, int f0, bool second_stream = true) {
    ck_handle v = *h;
    ck_stage_ws &w = v.ws;
    const size_t f = (size_t)f0, npix = h->npix;
    if (second_stream) v.stream = h->stream2;
    v.d_frames += f * h->frame_pitch;
    if (h->cfg.quad_decimate > 1) v.d_qframes += f * (size_t)((h->qw + 15) / 16 * 16) * h->qh;
    else v.d_qframes = v.d_frames;
    v.d_thresh += f * npix; v.d_labels += f * npix;
    v.d_groot += f * h->broot_cap; v.d_gsize += f * h->broot_cap; v.d_gscratch += f * 2 * h->broot_cap; v.d_xband += f * 2 * h->broot_cap;
    v.d_broots += f * 2 * h->broot_cap; v.d_tile_count += f * (size_t)(h->tiles_x * h->tiles_y); v.d_ring += f * h->ring_len;
    w.d_ht_keys += f * w.ht_size; w.d_ht_count += f * w.ht_size; w.d_ht_off += f * w.ht_size;
    w.d_tmp += f * w.ext_cap; w.d_points += f * w.ext_cap; w.d_runs += f * w.run_cap;
    w.d_ext_xy += f * w.ext_cap; w.d_ext_w += f * w.ext_cap; w.d_maxval += f * (size_t)(w.ext_cap / 2); w.d_maxpos += f * (size_t)(w.ext_cap / 2);
    w.d_maxmask += f * (size_t)(w.ext_cap / 64); w.d_maxpre += f * (size_t)(w.ext_cap / 64); w.d_blk += f * 6 * (size_t)(w.ext_cap / 32);
    w.d_cstate += f * 2 * w.cluster_cap;
    if (second_stream) w.d_lscratch += (size_t)CK_LSCRATCH_PER_WG * CK_LSCRATCH_WGS;
    if (second_stream && w.d_hscratch) w.d_hscratch += 2 * (size_t)w.hcap * CK_HUGE_WGS;
    w.d_clusters += f * w.cluster_cap; w.d_counters += f * CK_CNT_STRIDE; w.d_quads += f * w.quad_cap; w.d_dets += f * w.det_cap;
    w.d_wimg += f * npix;
    if (second_stream) w.d_fit_scratch = static_cast<uint8_t *>(w.d_fit_scratch) + w.fit_scratch_bytes;
    w.d_gyro += f; w.d_has_gyro += f; w.d_problems += f; w.d_pose_tags += f * w.det_cap; w.d_bearings += f * w.det_cap * 12;
    w.d_world += f * w.det_cap * 12; w.d_results += f; w.d_meas += f; w.d_valid += f;
    return v;
}

// clusters -> quad fit -> decode of n frames on h->stream
static int run_tail(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n, int upto, bool events) {
    hipEvent_t *ev = h->ev;
    int rc = ck_launch_clusters(h, n);
    if (rc != CK_OK) return rc;
    if (events) CK_HIP(hipEventRecord(ev[3], h->stream));
    if (upto >= 2) {
        const uint8_t *q = frames; int qs = stride; size_t qp = pitch;
        if (h->cfg.quad_decimate > 1) { q = h->d_qframes; qs = (h->qw + 15) / 16 * 16; qp = (size_t)qs * h->qh; }
        rc = ck_launch_fit_quads(h, q, qs, qp, frames, stride, pitch, n);
        if (rc != CK_OK) return rc;
    }
    if (events) CK_HIP(hipEventRecord(ev[4], h->stream));
    if (upto >= 3) {
        rc = ck_launch_decode(h, frames, stride, pitch, n);
        if (rc != CK_OK) return rc;
    }
    if (events) CK_HIP(hipEventRecord(ev[5], h->stream));
    return CK_OK;
}

struct ck_split {
    int parts = 1;     // the batch is cut into `parts` consecutive pieces; piece p runs on stream (p & 1)
    int first[9] = {0}; // piece p = frames [first[p], first[p + 1])
    ck_handle view[8]; // view[p] for p >= 1 (piece 0 uses the handle itself)
    bool split() const { return parts > 1; }
};
static int parts_wanted() {
    static const int v = CK_KNOB("CK_PARTS", 2);
    return v < 1 ? 1 : (v > 8 ? 8 : v);
}

static int run_pipeline(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n, int upto /*1 clusters, 2 quads, 3 all*/,
                        ck_split *split = nullptr) {
    hipEvent_t *ev = h->ev;
    CK_HIP(hipEventRecord(ev[1], h->stream));
    int rc = ck_run_threshold_segment(h, frames, stride, pitch, n);
    if (rc != CK_OK) return rc;
    CK_HIP(hipEventRecord(ev[2], h->stream));
    int parts = parts_wanted();
    if (parts > n) parts = n;
    if (!split || parts < 2 || ck_streams_wanted() < 2) {
        if (split) { split->parts = 1; split->first[0] = 0; split->first[1] = n; }
        return run_tail(h, frames, stride, pitch, n, upto, true);
    }
    split->parts = parts;
    for (int p = 0; p <= parts; p++) split->first[p] = (int)((long long)n * p / parts);
    CK_HIP(hipEventRecord(h->ev_fork, h->stream));
    CK_HIP(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    for (int p = 0; p < parts; p++) {
        const int f0 = split->first[p], cnt = split->first[p + 1] - f0;
        ck_handle *hp = h;
        if (p > 0) { split->view[p] = make_view(h, f0, (p & 1) != 0); hp = &split->view[p]; }
        rc = run_tail(hp, frames + (size_t)f0 * pitch, stride, pitch, cnt, upto, p == 0);
        if (rc != CK_OK) return rc;
    }
    return CK_OK;
}
// the handle's stream continues only after stream2 has finished its half
static int join_split(ck_handle *h, const ck_split &sp, int n) {
    (void)n;
    if (!sp.split()) return CK_OK;
    CK_HIP(hipEventRecord(h->ev_join, h->stream2));
    CK_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return CK_OK;
}

static int fetch_detections(ck_handle *h, int n, ck_detection_t *dets, int cap, int32_t *counts, uint32_t *status) {
    ck_stage_ws &ws = h->ws;
    std::vector<uint32_t> counters((size_t)n * CK_CNT_STRIDE);
    std::vector<ck_detection_t> all((size_t)n * ws.det_cap);
    CK_HIP(hipMemcpyAsync(counters.data(), ws.d_counters, counters.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipMemcpyAsync(all.data(), ws.d_dets, all.size() * sizeof(ck_detection_t), hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipEventRecord(h->ev[6], h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) {
        uint32_t nd = counters[(size_t)i * CK_CNT_STRIDE + CK_CNT_DETS];
        uint32_t st = counters[(size_t)i * CK_CNT_STRIDE + CK_CNT_STATUS];
        if ((int)nd > cap) { nd = (uint32_t)cap; st |= CK_FRAME_DETS_OVERFLOW; }
        memcpy(dets + (size_t)i * cap, all.data() + (size_t)i * ws.det_cap, sizeof(ck_detection_t) * nd);
        counts[i] = (int32_t)nd;
        if (status) status[i] = st;
    }
    ck_stage_ms_t &ms = h->last_ms;
    float t;
    auto el = [&](int a, int b) { t = 0; (void)hipEventElapsedTime(&t, h->ev[a], h->ev[b]); return t; };
    ms.h2d = el(0, 1); ms.threshold = el(1, 2); ms.segment = 0; ms.clusters = el(2, 3); ms.quads = el(3, 4);
    ms.decode = el(4, 5); ms.d2h = el(5, 6); ms.total = el(0, 6);
    return CK_OK;
}

extern "C" int ck_detect_uploaded(ck_handle_t *h, int32_t n, ck_detection_t *dets, int32_t cap, int32_t *counts, uint32_t *status) {
    if (!h || !dets || !counts || cap < 1 || n < 0 || n > h->n_staged) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipSetDevice(h->device));
    CK_HIP(hipEventRecord(h->ev[0], h->stream));
    ck_split sp;
    int rc = run_pipeline(h, h->d_frames, h->frame_stride, h->frame_pitch, n, 3, &sp);
    if (rc == CK_OK) rc = join_split(h, sp, n);
    if (rc != CK_OK) return rc;
    return fetch_detections(h, n, dets, cap, counts, status);
}

extern "C" int ck_detect_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, ck_detection_t *dets, int32_t cap,
                               int32_t *counts, uint32_t *status) {
    if (!h || !dets || !counts || cap < 1 || n < 0) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipSetDevice(h->device));
    CK_HIP(hipEventRecord(h->ev[0], h->stream));
    int rc = ck_upload_frames(h, imgs, n);
    if (rc != CK_OK) return rc;
    ck_split sp;
    rc = run_pipeline(h, h->d_frames, h->frame_stride, h->frame_pitch, n, 3, &sp);
    if (rc == CK_OK) rc = join_split(h, sp, n);
    if (rc != CK_OK) return rc;
    return fetch_detections(h, n, dets, cap, counts, status);
}

extern "C" int ck_detect_batch_device(ck_handle_t *h, const uint8_t *d_frames, int32_t n, int32_t stride, int64_t frame_pitch,
                                      ck_detection_t *dets, int32_t cap, int32_t *counts, uint32_t *status) {
    if (!h || !dets || !counts || cap < 1 || n < 0) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipSetDevice(h->device));
    CK_HIP(hipEventRecord(h->ev[0], h->stream));
    const uint8_t *use; int us; size_t up;
    int rc = ck_stage_device_frames(h, d_frames, n, stride, frame_pitch, &use, &us, &up);
    if (rc != CK_OK) return rc;
    ck_split sp;
    rc = run_pipeline(h, use, us, up, n, 3, &sp);
    if (rc == CK_OK) rc = join_split(h, sp, n);
    if (rc != CK_OK) return rc;
    return fetch_detections(h, n, dets, cap, counts, status);
}

int ck_detect_frames(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n, ck_detection_t *dets, int cap, int32_t *counts,
                     uint32_t *status) {
    if (!h || !dets || !counts || cap < 1 || n < 0 || n > h->cfg.max_batch) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipEventRecord(h->ev[0], h->stream));
    ck_split sp;
    int rc = run_pipeline(h, frames, stride, pitch, n, 3, &sp);
    if (rc == CK_OK) rc = join_split(h, sp, n);
    if (rc != CK_OK) return rc;
    return fetch_detections(h, n, dets, cap, counts, status);
}

extern "C" int ck_clusters_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, ck_cluster_t *clusters, int32_t cluster_cap,
                                 int32_t *n_clusters, ck_cluster_point_t *points, int32_t point_cap, int32_t *n_points) {
    if (!h || !clusters || !n_clusters || !points || !n_points || n < 0 || cluster_cap < 0 || point_cap < 0) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipSetDevice(h->device));
    int rc = imgs ? ck_upload_frames(h, imgs, n) : (n <= h->n_staged ? CK_OK : CK_EINVAL);
    if (rc != CK_OK) return rc;
    rc = run_pipeline(h, h->d_frames, h->frame_stride, h->frame_pitch, n, 1);
    if (rc != CK_OK) return rc;
    ck_stage_ws &ws = h->ws;
    CK_HIP(hipStreamSynchronize(h->stream));
    std::vector<uint32_t> counters((size_t)n * CK_CNT_STRIDE);
    CK_HIP(hipMemcpy(counters.data(), ws.d_counters, counters.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        uint32_t nc = counters[(size_t)i * CK_CNT_STRIDE + CK_CNT_CLUSTERS], np = counters[(size_t)i * CK_CNT_STRIDE + CK_CNT_POINTS];
        if ((int)nc > cluster_cap || (int)np > point_cap) return CK_ECAPACITY;
        CK_HIP(hipMemcpy(clusters + (size_t)i * cluster_cap, ws.d_clusters + (size_t)i * ws.cluster_cap, sizeof(ck_cluster_t) * nc, hipMemcpyDeviceToHost));
        std::vector<ck_packed_point> packed(np);
        CK_HIP(hipMemcpy(packed.data(), ws.d_points + (size_t)i * ws.ext_cap, sizeof(ck_packed_point) * np, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < np; k++) points[(size_t)i * point_cap + k] = ck_unpack_point(packed[k]);
        n_clusters[i] = (int32_t)nc; n_points[i] = (int32_t)np;
    }
    return CK_OK;
}

extern "C" int ck_quads_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, ck_quad_t *quads, int32_t quad_cap, int32_t *n_quads) {
    if (!h || !quads || !n_quads || n < 0 || quad_cap < 0) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipSetDevice(h->device));
    int rc = imgs ? ck_upload_frames(h, imgs, n) : (n <= h->n_staged ? CK_OK : CK_EINVAL);
    if (rc != CK_OK) return rc;
    rc = run_pipeline(h, h->d_frames, h->frame_stride, h->frame_pitch, n, 2);
    if (rc != CK_OK) return rc;
    ck_stage_ws &ws = h->ws;
    CK_HIP(hipStreamSynchronize(h->stream));
    std::vector<uint32_t> counters((size_t)n * CK_CNT_STRIDE);
    CK_HIP(hipMemcpy(counters.data(), ws.d_counters, counters.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        uint32_t nq = counters[(size_t)i * CK_CNT_STRIDE + CK_CNT_QUADS];
        if (nq > (uint32_t)ws.quad_cap) nq = (uint32_t)ws.quad_cap; // the counter keeps counting past the capacity (status bit set)
        if ((int)nq > quad_cap) return CK_ECAPACITY;
        CK_HIP(hipMemcpy(quads + (size_t)i * quad_cap, ws.d_quads + (size_t)i * ws.quad_cap, sizeof(ck_quad_t) * nq, hipMemcpyDeviceToHost));
        n_quads[i] = (int32_t)nq;
    }
    return CK_OK;
}

static int process_common(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n, const ck_process_params_t *pp,
                          const double *gyro, const uint8_t *has_gyro, ck_vision_measurement_t *out, int32_t *valid) {
    if (!pp || !gyro || !has_gyro || !out || !valid || (pp->n_field > 0 && !pp->field) || pp->n_field < 0) return CK_EINVAL;
    CK_HIP(hipEventRecord(h->ev[0], h->stream));
    if (pp->n_field > h->ws.field_cap) return CK_ECAPACITY;
    // the field layout is shared by both halves: upload it once, ahead of the fork
    if (pp->n_field) CK_HIP(hipMemcpyAsync(h->ws.d_field, pp->field, sizeof(ck_field_tag_t) * (size_t)pp->n_field, hipMemcpyDefault, h->stream));
    ck_split sp;
    int rc = run_pipeline(h, frames, stride, pitch, n, 3, &sp);
    if (rc != CK_OK) return rc;
    for (int p = 0; p < sp.parts && rc == CK_OK; p++) {
        const int f0 = sp.first[p], cnt = sp.first[p + 1] - f0;
        rc = ck_run_pose(p ? &sp.view[p] : h, cnt, pp, gyro + f0, has_gyro + f0, out + f0, valid + f0, false, false);
    }
    if (rc == CK_OK) rc = join_split(h, sp, n);
    h->n_last_pose = rc == CK_OK ? n : -1;
    CK_HIP(hipEventRecord(h->ev[6], h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    ck_stage_ms_t &ms = h->last_ms;
    float t;
    auto el = [&](int a, int b) { t = 0; (void)hipEventElapsedTime(&t, h->ev[a], h->ev[b]); return t; };
    ms.h2d = el(0, 1); ms.threshold = el(1, 2); ms.segment = 0; ms.clusters = el(2, 3); ms.quads = el(3, 4);
    ms.decode = el(4, 5); ms.d2h = el(5, 6); ms.total = el(0, 6); // d2h slot = glue + SQPnP + 64-byte records back
    return rc;
}

int ck_process_frames(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n, const ck_process_params_t *pp, const double *gyro,
                      const uint8_t *has_gyro, ck_vision_measurement_t *out, int32_t *valid) {
    if (!h || n < 0 || n > h->cfg.max_batch) return CK_EINVAL;
    if (n == 0) return CK_OK;
    return process_common(h, frames, stride, pitch, n, pp, gyro, has_gyro, out, valid);
}

extern "C" int ck_process_uploaded(ck_handle_t *h, int32_t n, const ck_process_params_t *pp, const double *gyro, const uint8_t *has_gyro,
                                   ck_vision_measurement_t *out, int32_t *valid) {
    if (!h || n < 0 || n > h->n_staged) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipSetDevice(h->device));
    return process_common(h, h->d_frames, h->frame_stride, h->frame_pitch, n, pp, gyro, has_gyro, out, valid);
}

extern "C" int ck_process_batch_device(ck_handle_t *h, const uint8_t *d_frames, int32_t n, int32_t stride, int64_t frame_pitch,
                                       const ck_process_params_t *pp, const double *gyro, const uint8_t *has_gyro,
                                       ck_vision_measurement_t *out, int32_t *valid) {
    if (!h) return CK_EINVAL;
    if (n == 0) return CK_OK;
    CK_HIP(hipSetDevice(h->device));
    const uint8_t *use; int us; size_t up;
    int rc = ck_stage_device_frames(h, d_frames, n, stride, frame_pitch, &use, &us, &up);
    if (rc != CK_OK) return rc;
    return process_common(h, use, us, up, n, pp, gyro, has_gyro, out, valid);
}
