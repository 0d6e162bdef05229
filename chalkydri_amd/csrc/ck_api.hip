// ck_api.hip — C ABI entry points: handle lifecycle, frame staging, stage orchestration.
// See include/chalkydri_hip.h for the contract and the reference interfaces each entry replaces.
#include <string.h>

#include <new>
#include <vector>

#include "ck_internal.h"

thread_local char ck_err_text[512] = "";

// ---- CK_POISON=3: guard-page allocations (ck_internal.h) --------------------------------------------------------------------------
#include <map>
#include <mutex>
namespace {
std::mutex g_guard_mu;
std::map<void *, ck_guarded_alloc> g_guarded;
}
hipError_t ck_guarded_malloc(void **p, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorUnknown;
    ck_guarded_alloc g = {};
    const size_t used = (bytes + 15) / 16 * 16;                    // the buffer keeps 16-byte alignment (its widest accesses); at most 15 bytes of slack
    g.map_bytes = (used + gran - 1) / gran * gran;
    g.va_bytes = g.map_bytes + gran;                               // ... and one granule that stays unmapped
    e = hipMemAddressReserve(&g.va, g.va_bytes, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&g.mem, g.map_bytes, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(g.va, g.va_bytes); return e; }
    e = hipMemMap(g.va, g.map_bytes, 0, g.mem, 0);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (e == hipSuccess) e = hipMemSetAccess(g.va, g.map_bytes, &acc, 1);
    if (e != hipSuccess) { (void)hipMemRelease(g.mem); (void)hipMemAddressFree(g.va, g.va_bytes); return e; }
    *p = static_cast<char *>(g.va) + (g.map_bytes - used);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guarded[*p] = g;
    return hipSuccess;
}
bool ck_guarded_free(void *p) {
    ck_guarded_alloc g;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guarded.find(p);
        if (it == g_guarded.end()) return false;
        g = it->second;
        g_guarded.erase(it);
    }
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(g.va, g.map_bytes);
    (void)hipMemRelease(g.mem);
    // The address range stays reserved for the life of the process.  Freed, the runtime hands the same range to the next
    // reservation, and kernels then still reach the OLD, released pages through it while the copy engine sees the new ones
    // (tools/probes/vmm_reuse_probe.hip shows it with runtime calls alone: every reuse reads back zeros / the fill pattern):
    // that was the fp64 probe's "zeros" under CK_POISON=3.  A test process reserves a few thousand ranges of 47-bit address
    // space at most; and an access after free faults now, too.
    return true;
}

extern "C" const char *ck_last_error(void) { return ck_err_text; }

extern "C" int ck_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int ck_create(const ck_config_t *cfg, ck_handle_t **out) {
    if (!cfg || !out) return CK_EINVAL;
    *out = nullptr;
    if (cfg->width < 16 || cfg->height < 16 || cfg->width > 4095 || cfg->height > 4095) return CK_EINVAL;
    if (cfg->max_batch < 1 || cfg->n_families < 1 || cfg->n_families > CK_MAX_FAMILIES) return CK_EINVAL;
    if (cfg->quad_decimate != 1 && cfg->quad_decimate != 2) return CK_EUNSUPPORTED;
    for (int i = 0; i < cfg->n_families; i++)
        if (!cfg->families[i] || cfg->families[i]->nbits > 64 || cfg->families[i]->total_width > 16 ||
            cfg->families[i]->n_upstream > cfg->families[i]->ncodes) return CK_EINVAL;
    int qw = cfg->width / cfg->quad_decimate, qh = cfg->height / cfg->quad_decimate;
    if (qw < 8 || qh < 8) return CK_EINVAL;
    if (cfg->min_component_px < 1 || cfg->min_component_px > 0x3FFFFFFF) return CK_EINVAL;
    if (ck_device_count() <= 0) return CK_ENODEVICE;
    ck_handle *h = new (std::nothrow) ck_handle();
    if (!h) return CK_ENOMEM;
    memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    h->device = cfg->device;
    h->n_last_pose = -1;
    h->w = cfg->width; h->h = cfg->height; h->qw = qw; h->qh = qh;
    h->npix = (size_t)qw * qh;
    h->tiles_x = (qw + CK_TW - 1) / CK_TW; h->tiles_y = (qh + CK_TH - 1) / CK_TH;
    h->broot_cap = h->tiles_x * h->tiles_y * CK_RING_CAP;
    h->ring_len = (2 * ((size_t)h->tiles_y * qw + (size_t)h->tiles_x * qh) + 3) & ~(size_t)3; // frames stay 8-byte aligned
    h->frame_stride = round_up(cfg->width, 16);
    h->frame_pitch = (size_t)h->frame_stride * cfg->height;
    const size_t nb = (size_t)cfg->max_batch;
    int rc = CK_OK;
    auto fail = [&](int code) { ck_destroy(h); return code; };
#define CK_TRY(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            snprintf(ck_err_text, sizeof ck_err_text, "%s failed: %s", #call, hipGetErrorString(e_)); \
            (void)hipGetLastError(); /* (not left behind for a later call's launch check) */         \
            return fail(e_ == hipErrorOutOfMemory ? CK_ENOMEM : CK_EDEVICE);                          \
        }                                                                                             \
    } while (0)
    CK_TRY(hipSetDevice(h->device));
    CK_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CK_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (auto &e : h->ev) CK_TRY(hipEventCreate(&e));
    CK_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    CK_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    CK_TRY(hipEventCreateWithFlags(&h->ev_fit_fork, hipEventDisableTiming));
    for (auto &st : h->fit_stream) CK_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK_TRY(hipStreamCreateWithFlags(&h->seg_stream, hipStreamNonBlocking));
    for (auto &e : h->ev_seg) CK_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK_TRY(hipEventCreateWithFlags(&h->ev_seg_join, hipEventDisableTiming));
    for (auto &e : h->ev_fit_join) CK_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK_TRY(ck_malloc_dev(&h->d_frames, h->frame_pitch * nb));
    if (cfg->quad_decimate > 1) CK_TRY(ck_malloc_dev(&h->d_qframes, (size_t)round_up(qw, 16) * qh * nb));
    CK_TRY(ck_malloc_dev(&h->d_thresh, h->npix * nb));
    CK_TRY(ck_malloc_dev(&h->d_labels, h->npix * nb * sizeof(ck_label_t)));
    CK_TRY(ck_malloc_dev(&h->d_groot, (size_t)h->broot_cap * nb * sizeof(uint32_t)));
    CK_TRY(ck_malloc_dev(&h->d_gsize, (size_t)h->broot_cap * nb * sizeof(uint32_t)));
    CK_TRY(ck_malloc_dev(&h->d_gscratch, 2 * (size_t)h->broot_cap * nb * sizeof(uint32_t)));
    CK_TRY(ck_malloc_dev(&h->d_xband, 2 * (size_t)h->broot_cap * nb * sizeof(uint32_t)));
    CK_TRY(ck_malloc_dev(&h->d_broots, 2 * (size_t)h->broot_cap * nb * sizeof(ck_border_root)));
    CK_TRY(ck_malloc_dev(&h->d_tile_count, (size_t)h->tiles_x * h->tiles_y * nb * sizeof(uint32_t)));
    CK_TRY(ck_malloc_dev(&h->d_ring, h->ring_len * nb * sizeof(uint16_t)));
    rc = ck_stage_alloc(h);
    if (rc != CK_OK) return fail(rc);
#undef CK_TRY
    *out = h;
    return CK_OK;
}

extern "C" void ck_destroy(ck_handle_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    for (auto &st : h->fit_stream) if (st) (void)hipStreamSynchronize(st);
    if (h->seg_stream) (void)hipStreamSynchronize(h->seg_stream);
    ck_stage_free(h);
    (void)ck_free_dev(h->d_frames); (void)ck_free_dev(h->d_qframes); (void)ck_free_dev(h->d_thresh); (void)ck_free_dev(h->d_labels);
    (void)ck_free_dev(h->d_groot); (void)ck_free_dev(h->d_gsize); (void)ck_free_dev(h->d_gscratch); (void)ck_free_dev(h->d_xband); (void)ck_free_dev(h->d_broots); (void)ck_free_dev(h->d_tile_count); (void)ck_free_dev(h->d_ring);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_fit_fork) (void)hipEventDestroy(h->ev_fit_fork);
    for (auto &e : h->ev_fit_join) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->ev_seg) if (e) (void)hipEventDestroy(e);
    if (h->ev_seg_join) (void)hipEventDestroy(h->ev_seg_join);
    if (h->seg_stream) (void)hipStreamDestroy(h->seg_stream);
    for (auto &st : h->fit_stream) if (st) (void)hipStreamDestroy(st);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    (void)hipGetLastError(); // (what the clean-up calls above may have left behind — a handle that never got its device, say — is not the next call's error)
}

static int check_imgs(const ck_handle *h, const ck_image_u8_t *imgs, int n) {
    if (n < 0 || (n > 0 && !imgs)) return CK_EINVAL;
    if (n > h->cfg.max_batch) return CK_ECAPACITY;
    for (int i = 0; i < n; i++)
        if (!imgs[i].buf || imgs[i].width != h->w || imgs[i].height != h->h || imgs[i].stride < imgs[i].width) return CK_EINVAL;
    return CK_OK;
}

extern "C" int ck_upload_frames(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n) {
    if (!h) return CK_EINVAL;
    int rc = check_imgs(h, imgs, n);
    if (rc != CK_OK) return rc;
    CK_HIP(hipSetDevice(h->device));
    for (int i = 0; i < n; i++)
        CK_HIP(hipMemcpy2DAsync(h->d_frames + (size_t)i * h->frame_pitch, (size_t)h->frame_stride, imgs[i].buf, (size_t)imgs[i].stride,
                                (size_t)h->w, (size_t)h->h, hipMemcpyHostToDevice, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    h->n_staged = n;
    return CK_OK;
}

// Makes d_frames hold a 16-byte aligned copy of caller-resident device frames when their layout is not directly usable.
int ck_stage_device_frames(ck_handle *h, const uint8_t *d_frames, int n, int stride, int64_t frame_pitch, const uint8_t **use,
                           int *use_stride, size_t *use_pitch) {
    if (!d_frames || n < 0 || stride < h->w || frame_pitch < (int64_t)stride * h->h) return CK_EINVAL;
    if (n > h->cfg.max_batch) return CK_ECAPACITY;
    bool aligned = ((uintptr_t)d_frames % 16 == 0) && (stride % 16 == 0) && (frame_pitch % 16 == 0);
    if (aligned) { *use = d_frames; *use_stride = stride; *use_pitch = (size_t)frame_pitch; return CK_OK; }
    for (int i = 0; i < n; i++)
        CK_HIP(hipMemcpy2DAsync(h->d_frames + (size_t)i * h->frame_pitch, (size_t)h->frame_stride, d_frames + (size_t)i * frame_pitch,
                                (size_t)stride, (size_t)h->w, (size_t)h->h, hipMemcpyDeviceToDevice, h->stream));
    h->n_staged = n;
    *use = h->d_frames; *use_stride = h->frame_stride; *use_pitch = h->frame_pitch;
    return CK_OK;
}

// Runs decimate (if configured) + threshold + segment on n staged frames.
int ck_run_threshold_segment(ck_handle *h, const uint8_t *frames, int stride, size_t pitch, int n) {
    if (h->cfg.quad_decimate > 1) {
        int rc = ck_launch_decimate(h, frames, stride, pitch, n);
        if (rc != CK_OK) return rc;
        return ck_launch_threshold_segment(h, h->d_qframes, round_up(h->qw, 16), (size_t)round_up(h->qw, 16) * h->qh, n);
    }
    return ck_launch_threshold_segment(h, frames, stride, pitch, n);
}

static int stage_input(ck_handle *h, const ck_image_u8_t *imgs, int n) {
    if (imgs) return ck_upload_frames(h, imgs, n);
    if (n < 0 || n > h->n_staged) return CK_EINVAL;
    return CK_OK;
}

extern "C" int ck_threshold_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, uint8_t *thresh_out) {
    if (!h || !thresh_out) return CK_EINVAL;
    int rc = stage_input(h, imgs, n);
    if (rc != CK_OK) return rc;
    CK_HIP(hipSetDevice(h->device));
    rc = ck_run_threshold_segment(h, h->d_frames, h->frame_stride, h->frame_pitch, n);
    if (rc != CK_OK) return rc;
    CK_HIP(hipMemcpyAsync(thresh_out, h->d_thresh, h->npix * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    CK_HIP(hipStreamSynchronize(h->stream));
    return CK_OK;
}

extern "C" int ck_segment_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, uint32_t *labels_out, uint32_t *sizes_out) {
    if (!h || !labels_out) return CK_EINVAL;
    int rc = stage_input(h, imgs, n);
    if (rc != CK_OK) return rc;
    CK_HIP(hipSetDevice(h->device));
    rc = ck_run_threshold_segment(h, h->d_frames, h->frame_stride, h->frame_pitch, n);
    if (rc != CK_OK) return rc;
    size_t total = h->npix * (size_t)n;
    uint32_t *d_canon = nullptr, *d_sizes = nullptr;
    // (a failed allocation leaves the runtime's per-thread error behind: cleared, so that the next call's launch check does not meet it)
    if (ck_malloc_dev(&d_canon, total * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return CK_ENOMEM; }
    if (sizes_out && ck_malloc_dev(&d_sizes, total * sizeof(uint32_t)) != hipSuccess) { (void)ck_free_dev(d_canon); (void)hipGetLastError(); return CK_ENOMEM; }
    rc = ck_launch_canonical_labels(h, n, d_canon, d_sizes);
    if (rc == CK_OK) {
        hipError_t e = hipMemcpyAsync(labels_out, d_canon, total * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && sizes_out) e = hipMemcpyAsync(sizes_out, d_sizes, total * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { snprintf(ck_err_text, sizeof ck_err_text, "segment D2H failed: %s", hipGetErrorString(e)); rc = CK_EDEVICE; }
    }
    (void)ck_free_dev(d_canon); (void)ck_free_dev(d_sizes);
    return rc;
}

extern "C" int ck_time_threshold_segment(ck_handle_t *h, int32_t n, int32_t iters, float *ms_out) {
    if (!h || !ms_out || iters < 1 || n < 1 || n > h->n_staged) return CK_EINVAL;
    CK_HIP(hipSetDevice(h->device));
    int rc = ck_run_threshold_segment(h, h->d_frames, h->frame_stride, h->frame_pitch, n); // warm-up
    if (rc != CK_OK) return rc;
    CK_HIP(hipEventRecord(h->ev[0], h->stream));
    for (int i = 0; i < iters; i++) {
        rc = ck_run_threshold_segment(h, h->d_frames, h->frame_stride, h->frame_pitch, n);
        if (rc != CK_OK) return rc;
    }
    CK_HIP(hipEventRecord(h->ev[1], h->stream));
    CK_HIP(hipEventSynchronize(h->ev[1]));
    float ms = 0;
    CK_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    *ms_out = ms / (float)iters;
    return CK_OK;
}

extern "C" int ck_last_stage_ms(ck_handle_t *h, ck_stage_ms_t *out) {
    if (!h || !out) return CK_EINVAL;
    *out = h->last_ms;
    return CK_OK;
}

// fp64 conformance probe ------------------------------------------------------------------------------------------
__global__ void k_fp64_probe(int op, const double *a, const double *b, int n, double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r;
    switch (op) {
    case 0: r = x + y; break;
    case 1: r = x * y; break;
    case 2: r = x / y; break;
    case 3: r = sqrt(x); break;
    default: { double t = x * y; r = t + x; } break;
    }
    out[i] = r;
}
extern "C" int ck_selftest_fp64(ck_handle_t *h, int32_t op, const double *a, const double *b, int32_t n, double *out) {
    if (!h || !a || !out || n < 0) return CK_EINVAL;
    CK_HIP(hipSetDevice(h->device));
    if (n == 0) return CK_OK;
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    hipError_t e = ck_malloc_dev(&da, bytes);
    if (e == hipSuccess) e = ck_malloc_dev(&dout, bytes);
    if (e == hipSuccess && b) e = ck_malloc_dev(&db, bytes);
    if (e == hipSuccess) e = hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fp64_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, op, da, db, n, dout);
        e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    (void)ck_free_dev(da); (void)ck_free_dev(db); (void)ck_free_dev(dout); // one exit: nothing leaks on an error path
    if (e != hipSuccess) { snprintf(ck_err_text, sizeof ck_err_text, "fp64 probe failed: %s", hipGetErrorString(e)); return CK_EDEVICE; }
    return CK_OK;
}
