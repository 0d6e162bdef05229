// ck_comm.hip — the one collective of the path, owned by the C ABI: an RCCL all-gather of the 64-byte pose records.
//
// Frames shard over GPUs with no data-path collective (one camera stream, or one contiguous frame block, per GPU;
// DESIGN.md §6).  What the reference sends per frame is one VisionMeasurement datagram (crates/whacknet/src/lib.rs:43-66,
// 152-171); with N GPUs the records of a batch meet on every rank through ONE ncclAllGather of n x 64 bytes, issued on the
// communicator's own stream from a copy of the records the pose stage filled, so that the rendezvous of step i overlaps the
// kernels of step i + 1.  At 16 KiB per GPU the collective is latency-bound on xGMI: no bucketing, no ring tuning.
//
// librccl is opened on first use (dlopen), so hosts that never gather — and this container, which has no GPU — load the
// library without it.  A process that already holds an RCCL (PyTorch-ROCm ships one under the same soname) shares it.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <new>

#include "ck_internal.h"

namespace {

// the handful of RCCL declarations used here (rccl.h: ncclUniqueId is 128 opaque bytes; ncclChar = 0 in ncclDataType_t)
struct rccl_unique_id { char internal[CK_COMM_ID_BYTES]; };
typedef void *rccl_comm_t;
typedef int (*fn_get_unique_id)(rccl_unique_id *);
typedef int (*fn_comm_init_rank)(rccl_comm_t *, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(rccl_comm_t);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t);
typedef const char *(*fn_error_string)(int);

struct rccl_api {
    void *so = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
};

rccl_api *rccl_load(rccl_api &api, char *err, size_t err_len) {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        api.so = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (api.so) break;
    }
    if (!api.so) {
        snprintf(err, err_len, "librccl could not be loaded: %s", dlerror());
        return nullptr;
    }
    api.get_unique_id = (fn_get_unique_id)dlsym(api.so, "ncclGetUniqueId");
    api.comm_init_rank = (fn_comm_init_rank)dlsym(api.so, "ncclCommInitRank");
    api.comm_destroy = (fn_comm_destroy)dlsym(api.so, "ncclCommDestroy");
    api.all_gather = (fn_all_gather)dlsym(api.so, "ncclAllGather");
    api.error_string = (fn_error_string)dlsym(api.so, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_gather) {
        snprintf(err, err_len, "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllGather");
        dlclose(api.so);
        api.so = nullptr;
        return nullptr;
    }
    return &api;
}
// one host thread per GPU may arrive here at once: the table is filled exactly once (std::call_once), and every later caller
// sees it complete or not at all
rccl_api *rccl() {
    static rccl_api api;
    static std::once_flag once;
    static char load_err[256];
    std::call_once(once, [] { rccl_load(api, load_err, sizeof load_err); });
    if (!api.so) { snprintf(ck_err_text, sizeof ck_err_text, "%s", load_err); return nullptr; }
    return &api;
}

int rccl_fail(rccl_api *r, const char *what, int rc) {
    snprintf(ck_err_text, sizeof ck_err_text, "%s failed: %s (%d)", what, r->error_string ? r->error_string(rc) : "rccl error", rc);
    return CK_EDEVICE;
}

} // namespace

struct ck_comm {
    rccl_comm_t comm;
    ck_handle *h;       // compared with the caller's handle; never dereferenced by ck_comm_destroy / ck_comm_sync (the handle may be gone)
    int device;         // copy of the handle's device, taken at creation
    hipStream_t hstream; // the handle's stream: only ck_gather_poses (which has the live handle) enqueues on it
    hipStream_t stream;  // the communicator's OWN stream: the collective and the copy-out run here, beside the handle's next batch
    int world, rank;
    ck_vision_measurement_t *d_send[2]; // [max_batch] records each: the step's records, copied off the handle's buffer (which the next ck_process_* call rewrites)
    ck_vision_measurement_t *d_all;     // [world][max_batch] records, device
    hipEvent_t ev_ready[2], ev_done[2]; // send copy k filled (handle's stream) / consumed (communicator's stream)
    bool used[2];
    unsigned turn;
    size_t cap_records;
    char lib_path[256];                 // which librccl the communicator's calls resolve to (dladdr)
};

extern "C" int ck_backend(const ck_handle_t *h) { return h ? CK_BACKEND_HIP : CK_EINVAL; }

extern "C" int ck_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return CK_EINVAL;
    rccl_api *r = rccl();
    if (!r) return CK_EUNSUPPORTED;
    rccl_unique_id id;
    int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, CK_COMM_ID_BYTES);
    return CK_OK;
}

static void comm_release(ck_comm *c) {
    for (int k = 0; k < 2; k++) {
        if (c->d_send[k]) (void)hipFree(c->d_send[k]);
        if (c->ev_ready[k]) (void)hipEventDestroy(c->ev_ready[k]);
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
    }
    if (c->d_all) (void)hipFree(c->d_all);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int ck_comm_create(ck_handle_t *h, const uint8_t *id_in, int32_t world, int32_t rank, ck_comm_t **out) {
    if (!h || !id_in || !out || world < 1 || rank < 0 || rank >= world) return CK_EINVAL;
    *out = nullptr;
    rccl_api *r = rccl();
    if (!r) return CK_EUNSUPPORTED;
    CK_HIP(hipSetDevice(h->device));
    ck_comm *c = new (std::nothrow) ck_comm();
    if (!c) return CK_ENOMEM;
    c->h = h; c->device = h->device; c->hstream = h->stream; c->world = world; c->rank = rank;
    c->cap_records = (size_t)world * (size_t)h->cfg.max_batch;
    const size_t one = sizeof(ck_vision_measurement_t) * (size_t)h->cfg.max_batch;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&c->d_all, sizeof(ck_vision_measurement_t) * c->cap_records) == hipSuccess;
    for (int k = 0; k < 2 && ok; k++)
        ok = hipMalloc(&c->d_send[k], one) == hipSuccess && hipEventCreateWithFlags(&c->ev_ready[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); comm_release(c); return CK_ENOMEM; }
    Dl_info di;
    snprintf(c->lib_path, sizeof c->lib_path, "%s", (dladdr((void *)r->all_gather, &di) && di.dli_fname) ? di.dli_fname : "?");
    rccl_unique_id id;
    memcpy(id.internal, id_in, CK_COMM_ID_BYTES);
    int rc = r->comm_init_rank(&c->comm, world, id, rank);
    if (rc != 0) { comm_release(c); return rccl_fail(r, "ncclCommInitRank", rc); }
    *out = c;
    return CK_OK;
}

// path of the RCCL library this communicator's calls go to (a process that also holds PyTorch's process group can check that both
// resolved to one library); valid until ck_comm_destroy
extern "C" const char *ck_comm_library(const ck_comm_t *c) { return c ? c->lib_path : ""; }

extern "C" void ck_comm_destroy(ck_comm_t *c) {
    if (!c) return;
    rccl_api *r = rccl();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream); // (the communicator's own stream: whatever it still waits for on the handle's was enqueued before)
    if (r) (void)r->comm_destroy(c->comm);
    comm_release(c);
}

// All-gather of the records the last ck_process_* call of this handle left on the device (ws.d_meas), in rank order.  Every rank
// sends `rows` records: its own n_valid (which must be what that call produced) and, behind them, empty ones (all zero: tag_count 0),
// so that a ragged last shard needs no padding by the caller.
//
// The rendezvous is NOT on the handle's stream ("issue it on a side stream overlapped with the next batch"): the
// handle's stream only copies the step's records into one of two send buffers (16 KB at batch 256) and records an event; the
// collective and the copy to `out` run on the communicator's stream behind that event.  The next ck_process_* call can start at
// once — it rewrites ws.d_meas, not the send buffer — so step i's cross-rank wait hides under step i + 1's kernels; a send buffer
// is handed out again only after its previous collective has completed (ev_done: the handle's stream waits for it, which with
// two buffers means the collective of two steps ago).
//
// A rank whose arguments fail the LOCAL checks (a count that is not what its last call produced, e.g. because that call failed)
// still takes part with `rows` empty records when `rows` itself is usable, and then returns the error: its peers' collective
// completes instead of waiting for a rank that left.  Only a rank whose `rows` is unusable cannot take part; the other ranks then
// have to destroy the communicator (as after any error of a collective).
extern "C" int ck_gather_poses(ck_handle_t *h, ck_comm_t *c, int32_t n_valid, int32_t rows, ck_vision_measurement_t *out, int32_t sync) {
    if (!h || !c || c->h != h || !out) return CK_EINVAL;
    if (rows < 0) return CK_EINVAL;
    if (rows > h->cfg.max_batch) return CK_ECAPACITY;
    int status = CK_OK;
    if (n_valid < 0 || rows < n_valid) { status = CK_EINVAL; n_valid = 0; }
    else if (n_valid != (h->n_last_pose < 0 ? 0 : h->n_last_pose)) {
        snprintf(ck_err_text, sizeof ck_err_text, "ck_gather_poses: n_valid = %d, but the handle's last ck_process_* call left %d records (%d empty records sent)",
                 n_valid, h->n_last_pose, rows);
        status = CK_EINVAL; n_valid = 0;
    }
    if (rows == 0) return status;
    rccl_api *r = rccl();
    if (!r) return CK_EUNSUPPORTED;
    CK_HIP(hipSetDevice(c->device));
    const unsigned k = c->turn & 1u;
    const size_t rec = sizeof(ck_vision_measurement_t), bytes = rec * (size_t)rows;
    // the handle's stream: this step's records into send buffer k
    if (c->used[k]) CK_HIP(hipStreamWaitEvent(c->hstream, c->ev_done[k], 0));
    if (n_valid) CK_HIP(hipMemcpyAsync(c->d_send[k], h->ws.d_meas, rec * (size_t)n_valid, hipMemcpyDeviceToDevice, c->hstream));
    if (rows > n_valid) CK_HIP(hipMemsetAsync(c->d_send[k] + n_valid, 0, rec * (size_t)(rows - n_valid), c->hstream));
    CK_HIP(hipEventRecord(c->ev_ready[k], c->hstream));
    // the communicator's stream: rendezvous + copy-out
    CK_HIP(hipStreamWaitEvent(c->stream, c->ev_ready[k], 0));
    int rc = r->all_gather(c->d_send[k], c->d_all, bytes, /*ncclChar*/ 0, c->comm, c->stream);
    if (rc != 0) return rccl_fail(r, "ncclAllGather", rc);
    CK_HIP(hipMemcpyAsync(out, c->d_all, bytes * (size_t)c->world, hipMemcpyDefault, c->stream));
    CK_HIP(hipEventRecord(c->ev_done[k], c->stream));
    c->used[k] = true;
    c->turn++;
    if (sync) CK_HIP(hipStreamSynchronize(c->stream));
    return status;
}

extern "C" int ck_comm_sync(ck_comm_t *c) {
    if (!c) return CK_EINVAL;
    CK_HIP(hipSetDevice(c->device));
    CK_HIP(hipStreamSynchronize(c->stream));
    return CK_OK;
}
