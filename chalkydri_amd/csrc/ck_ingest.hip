// ck_ingest.hip — pinned host slots and asynchronous upload in front of the detector.
//
// Mirrors what reaches `AprilTags::process` in the reference: pooled host buffers holding one 8-bit-luma frame each with a
// stride that may exceed the width (crates/chalkydri/src/cameras/gst_to_cu.rs:49-72,131-188; consumed through
// image_from_cuimage, crates/apriltags/src/lib.rs:197-213).  Here a slot is a whole batch: pinned on the host so the copy
// engine reads it directly, laid out exactly like its device twin (16-byte aligned rows) so that ONE asynchronous copy
// moves the batch; the copy runs on its own stream and the compute stream only waits on the slot's event.
#include <string.h>

#include <new>

#include "ck_internal.h"

struct ck_ingest {
    ck_handle *h;
    int nslots;
    size_t slot_bytes;
    uint8_t *host[8];
    uint8_t *dev[8];
    hipEvent_t ready[8];
    int staged[8];
    hipStream_t copy;
};

static bool luma_first(uint32_t fourcc) {
    auto cc = [](const char *s) { return (uint32_t)(uint8_t)s[0] | ((uint32_t)(uint8_t)s[1] << 8) | ((uint32_t)(uint8_t)s[2] << 16) | ((uint32_t)(uint8_t)s[3] << 24); };
    return fourcc == cc("GREY") || fourcc == cc("GRAY") || fourcc == cc("Y800") || fourcc == cc("NV12") || fourcc == cc("NV21") ||
           fourcc == cc("I420") || fourcc == cc("YV12");
}

extern "C" int ck_ingest_create(ck_handle_t *h, int32_t n_slots, ck_ingest_t **out) {
    if (!h || !out || n_slots < 1 || n_slots > 8) return CK_EINVAL;
    *out = nullptr;
    CK_HIP(hipSetDevice(h->device));
    ck_ingest *g = new (std::nothrow) ck_ingest();
    if (!g) return CK_ENOMEM;
    memset(g, 0, sizeof *g);
    g->h = h; g->nslots = n_slots;
    g->slot_bytes = h->frame_pitch * (size_t)h->cfg.max_batch;
    hipError_t e = hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking);
    for (int s = 0; s < n_slots && e == hipSuccess; s++) {
        e = hipHostMalloc(reinterpret_cast<void **>(&g->host[s]), g->slot_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&g->dev[s]), g->slot_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ready[s], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        snprintf(ck_err_text, sizeof ck_err_text, "ingest ring allocation failed: %s", hipGetErrorString(e));
        ck_ingest_destroy(g);
        return CK_ENOMEM;
    }
    *out = g;
    return CK_OK;
}

extern "C" void ck_ingest_destroy(ck_ingest_t *g) {
    if (!g) return;
    (void)hipSetDevice(g->h->device);
    if (g->copy) (void)hipStreamSynchronize(g->copy);
    for (int s = 0; s < g->nslots; s++) {
        if (g->host[s]) (void)hipHostFree(g->host[s]);
        if (g->dev[s]) (void)hipFree(g->dev[s]);
        if (g->ready[s]) (void)hipEventDestroy(g->ready[s]);
    }
    if (g->copy) (void)hipStreamDestroy(g->copy);
    delete g;
}

extern "C" int32_t ck_ingest_stride(const ck_ingest_t *g) { return g ? g->h->frame_stride : 0; }

extern "C" uint8_t *ck_ingest_frame(ck_ingest_t *g, int32_t slot, int32_t index) {
    if (!g || slot < 0 || slot >= g->nslots || index < 0 || index >= g->h->cfg.max_batch) return nullptr;
    return g->host[slot] + (size_t)index * g->h->frame_pitch;
}

extern "C" int ck_ingest_write(ck_ingest_t *g, int32_t slot, int32_t index, const ck_image_u8_t *img, uint32_t fourcc) {
    uint8_t *dst = ck_ingest_frame(g, slot, index);
    if (!dst || !img || !img->buf) return CK_EINVAL;
    if (!luma_first(fourcc)) return CK_EUNSUPPORTED;
    const ck_handle *h = g->h;
    if (img->width != h->w || img->height != h->h || img->stride < img->width) return CK_EINVAL;
    for (int y = 0; y < h->h; y++) memcpy(dst + (size_t)y * h->frame_stride, img->buf + (size_t)y * img->stride, (size_t)h->w);
    return CK_OK;
}

extern "C" int ck_ingest_submit(ck_ingest_t *g, int32_t slot, int32_t n) {
    if (!g || slot < 0 || slot >= g->nslots || n < 0 || n > g->h->cfg.max_batch) return CK_EINVAL;
    CK_HIP(hipSetDevice(g->h->device));
    if (n) CK_HIP(hipMemcpyAsync(g->dev[slot], g->host[slot], g->h->frame_pitch * (size_t)n, hipMemcpyHostToDevice, g->copy));
    CK_HIP(hipEventRecord(g->ready[slot], g->copy));
    g->staged[slot] = n;
    return CK_OK;
}

// `n` is the caller's statement of how many frames its output arrays hold: it must be the count the slot was submitted with
// (the kernels write one entry per staged frame).
extern "C" int ck_detect_ingested(ck_ingest_t *g, int32_t slot, int32_t n, ck_detection_t *dets, int32_t cap, int32_t *counts, uint32_t *status) {
    if (!g || slot < 0 || slot >= g->nslots || n != g->staged[slot]) return CK_EINVAL;
    if (n == 0) return CK_OK;
    ck_handle *h = g->h;
    CK_HIP(hipSetDevice(h->device));
    CK_HIP(hipStreamWaitEvent(h->stream, g->ready[slot], 0));
    return ck_detect_frames(h, g->dev[slot], h->frame_stride, h->frame_pitch, n, dets, cap, counts, status);
}

extern "C" int ck_process_ingested(ck_ingest_t *g, int32_t slot, int32_t n, const ck_process_params_t *pp, const double *gyro,
                                   const uint8_t *has_gyro, ck_vision_measurement_t *out, int32_t *valid) {
    if (!g || slot < 0 || slot >= g->nslots || n != g->staged[slot]) return CK_EINVAL;
    if (n == 0) return CK_OK;
    ck_handle *h = g->h;
    CK_HIP(hipSetDevice(h->device));
    CK_HIP(hipStreamWaitEvent(h->stream, g->ready[slot], 0));
    return ck_process_frames(h, g->dev[slot], h->frame_stride, h->frame_pitch, n, pp, gyro, has_gyro, out, valid);
}
