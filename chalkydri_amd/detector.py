"""Host-side mirror of the reference's detector interface over the C ABI.

`AprilTagDetector` plays the role of `apriltag::Detector` as the reference uses it
(crates/apriltags/src/lib.rs:258-262 build, :301 detect, :306-314 id()/corners()), batched over frames.
`CatDetector` keeps the surface of crates/chalkydri-apriltags `Detector`
(src/lib.rs:158 new, :191 calc_otsu, :265 process_frame, :291 detect_corners, :319 thresh, :480 check_edges,
:501 connected_components).  Every call goes through libchalkydri_hip.so; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, default_config, lib

_P = C.POINTER


def _bind(L):
    if getattr(L, "_ck_bound", False):
        return L
    vp, i32, u32p = C.c_void_p, C.c_int32, _P(C.c_uint32)
    L.ck_last_error.restype = C.c_char_p
    L.ck_device_count.restype = C.c_int
    L.ck_create.argtypes = [_P(A.Config), _P(vp)]
    L.ck_destroy.argtypes = [vp]
    L.ck_destroy.restype = None
    L.ck_upload_frames.argtypes = [vp, _P(A.ImageU8), i32]
    L.ck_threshold_batch.argtypes = [vp, _P(A.ImageU8), i32, vp]
    L.ck_segment_batch.argtypes = [vp, _P(A.ImageU8), i32, vp, vp]
    L.ck_time_threshold_segment.argtypes = [vp, i32, i32, _P(C.c_float)]
    L.ck_detect_batch.argtypes = [vp, _P(A.ImageU8), i32, _P(A.Detection), i32, _P(i32), u32p]
    L.ck_detect_uploaded.argtypes = [vp, i32, _P(A.Detection), i32, _P(i32), u32p]
    L.ck_detect_batch_device.argtypes = [vp, vp, i32, i32, C.c_int64, _P(A.Detection), i32, _P(i32), u32p]
    L.ck_clusters_batch.argtypes = [vp, _P(A.ImageU8), i32, vp, i32, _P(i32), vp, i32, _P(i32)]
    L.ck_quads_batch.argtypes = [vp, _P(A.ImageU8), i32, _P(A.Quad), i32, _P(i32)]
    L.ck_last_stage_ms.argtypes = [vp, _P(A.StageMs)]
    L.ck_selftest_fp64.argtypes = [vp, i32, vp, vp, i32, vp]
    L.ck_cat_calc_otsu.argtypes = [vp, vp, i32, i32, vp]
    L.ck_cat_thresh.argtypes = [vp, vp, i32, i32, vp]
    L.ck_cat_detect_corners.argtypes = [vp, vp, i32, i32, vp, i32, _P(i32)]
    L.ck_cat_check_edges.argtypes = [vp, vp, i32, i32, vp, i32, vp, i32, _P(i32)]
    L.ck_cat_connected_components.argtypes = [vp, vp, i32, i32, vp, vp]
    L.ck_cat_process_frame.argtypes = [vp, vp, C.c_size_t, i32, i32, vp, vp, i32, _P(i32), vp, i32, _P(i32)]
    L.ck_sqpnp_solve_batch.argtypes = [vp, _P(A.SqpnpParams), _P(A.SqpnpProblem), i32, _P(A.Iso3), i32, vp, i32,
                                       _P(A.SqpnpResult)]
    L.ck_sqpnp_create_solver_camera_transform.argtypes = [C.c_double] * 6 + [_P(A.Iso3)]
    L.ck_sqpnp_create_solver_camera_transform.restype = None
    L.ck_process_uploaded.argtypes = [vp, i32, _P(A.ProcessParams), vp, vp, _P(A.VisionMeasurement), _P(i32)]
    L.ck_process_batch_device.argtypes = [vp, vp, i32, i32, C.c_int64, _P(A.ProcessParams), vp, vp,
                                          _P(A.VisionMeasurement), _P(i32)]
    L.ck_unproject_opencv5.argtypes = [_P(A.OpenCV5), vp, i32, vp, vp]
    L.ck_ingest_create.argtypes = [vp, i32, _P(vp)]
    L.ck_ingest_destroy.argtypes = [vp]
    L.ck_ingest_destroy.restype = None
    L.ck_ingest_stride.argtypes = [vp]
    L.ck_ingest_stride.restype = i32
    L.ck_ingest_frame.argtypes = [vp, i32, i32]
    L.ck_ingest_frame.restype = vp
    L.ck_ingest_write.argtypes = [vp, i32, i32, _P(A.ImageU8), C.c_uint32]
    L.ck_ingest_submit.argtypes = [vp, i32, i32]
    L.ck_detect_ingested.argtypes = [vp, i32, i32, _P(A.Detection), i32, _P(i32), u32p]
    L.ck_process_ingested.argtypes = [vp, i32, i32, _P(A.ProcessParams), vp, vp, _P(A.VisionMeasurement), _P(i32)]
    L._ck_bound = True
    return L


def device_count():
    return _bind(lib()).ck_device_count()


def _images(frames):
    """frames: uint8 array [n][h][stride>=w] (C-contiguous rows) -> (ImageU8 array, keepalive)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if frames.ndim == 2:
        frames = frames[None]
    n, h, w = frames.shape
    arr = (A.ImageU8 * n)()
    for i in range(n):
        arr[i].buf = frames[i].ctypes.data
        arr[i].width, arr[i].height, arr[i].stride = w, h, frames.strides[1]
    return arr, frames


class Detection:
    """Mirrors the accessors the reference uses on apriltag::Detection (crates/apriltags/src/lib.rs:306-314)."""
    __slots__ = ("_id", "_hamming", "_family", "_margin", "_c", "_p")

    def __init__(self, d):
        self._id, self._hamming, self._family, self._margin = d.id, d.hamming, d.family, d.decision_margin
        self._c = np.array([d.c[0], d.c[1]])
        self._p = np.array([[d.p[k][0], d.p[k][1]] for k in range(4)])

    def id(self):
        return self._id

    def hamming(self):
        return self._hamming

    def family(self):
        return self._family

    def decision_margin(self):
        return self._margin

    def center(self):
        return self._c

    def corners(self):
        return self._p

    def __repr__(self):
        return f"Detection(id={self._id}, hamming={self._hamming}, margin={self._margin:.1f})"


class AprilTagDetector:
    """One handle = one GPU + its stream; not thread-safe (mirrors `&mut self`)."""

    def __init__(self, width, height, max_batch=1, families=("tag36h11",), bits_corrected=3, device=0, **cfg):
        self._L = _bind(lib())
        self.cfg = default_config(width, height, max_batch, families, max_hamming=bits_corrected, device=device,
                                  **cfg)
        self.width, self.height, self.max_batch = width, height, max_batch
        self.qw, self.qh = width // self.cfg.quad_decimate, height // self.cfg.quad_decimate
        h = C.c_void_p()
        check(self._L.ck_create(C.byref(self.cfg), C.byref(h)), "ck_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.ck_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- staging ----------------------------------------------------------------------------------------
    def upload(self, frames):
        arr, keep = _images(frames)
        check(self._L.ck_upload_frames(self._h, arr, len(arr)), "ck_upload_frames")
        return len(arr)

    # -- stages -------------------------------------------------------------------------------------------
    def threshold(self, frames=None, n=None):
        arr, keep, n = self._in(frames, n)
        out = np.empty((n, self.qh, self.qw), np.uint8)
        check(self._L.ck_threshold_batch(self._h, arr, n, out.ctypes.data), "ck_threshold_batch")
        return out

    def segment(self, frames=None, n=None, sizes=True):
        arr, keep, n = self._in(frames, n)
        labels = np.empty((n, self.qh, self.qw), np.uint32)
        sz = np.empty((n, self.qh, self.qw), np.uint32) if sizes else None
        check(self._L.ck_segment_batch(self._h, arr, n, labels.ctypes.data, sz.ctypes.data if sizes else None),
              "ck_segment_batch")
        return labels, sz

    def time_threshold_segment(self, n, iters=10):
        ms = C.c_float(0)
        check(self._L.ck_time_threshold_segment(self._h, n, iters, C.byref(ms)), "ck_time_threshold_segment")
        return ms.value

    def clusters(self, frames=None, n=None, cluster_cap=None, point_cap=None):
        arr, keep, n = self._in(frames, n)
        ccap = cluster_cap or (self.qw * self.qh // 8 + 1024)
        pcap = point_cap or (2 * self.qw * self.qh)
        cl = np.zeros((n, ccap, 4), np.uint32)
        pts = np.zeros((n, pcap), np.dtype([("x", "<u2"), ("y", "<u2"), ("gx", "i1"), ("gy", "i1"), ("pad", "<u2")]))
        nc = (C.c_int32 * n)()
        npt = (C.c_int32 * n)()
        check(self._L.ck_clusters_batch(self._h, arr, n, cl.ctypes.data, ccap, nc, pts.ctypes.data, pcap, npt),
              "ck_clusters_batch")
        return [(cl[i, :nc[i]].copy(), pts[i, :npt[i]].copy()) for i in range(n)]

    def quads(self, frames=None, n=None, cap=1024):
        arr, keep, n = self._in(frames, n)
        q = (A.Quad * (n * cap))()
        nq = (C.c_int32 * n)()
        check(self._L.ck_quads_batch(self._h, arr, n, q, cap, nq), "ck_quads_batch")
        return [[q[i * cap + k] for k in range(nq[i])] for i in range(n)]

    # -- the call the reference makes: detector.detect(&image) -----------------------------------------------
    def detect(self, frame, cap=64):
        return self.detect_batch(frame[None] if np.ndim(frame) == 2 else frame, cap)[0]

    def detect_batch(self, frames=None, cap=64, n=None, return_status=False):
        arr, keep, n = self._in(frames, n)
        dets = (A.Detection * (n * cap))()
        counts = (C.c_int32 * n)()
        status = (C.c_uint32 * n)()
        if arr is None:
            check(self._L.ck_detect_uploaded(self._h, n, dets, cap, counts, status), "ck_detect_uploaded")
        else:
            check(self._L.ck_detect_batch(self._h, arr, n, dets, cap, counts, status), "ck_detect_batch")
        out = [[Detection(dets[i * cap + k]) for k in range(counts[i])] for i in range(n)]
        return (out, list(status)) if return_status else out

    def detect_device(self, ptr, n, stride, frame_pitch, cap=64):
        dets = (A.Detection * (n * cap))()
        counts = (C.c_int32 * n)()
        status = (C.c_uint32 * n)()
        check(self._L.ck_detect_batch_device(self._h, C.c_void_p(ptr), n, stride, frame_pitch, dets, cap, counts,
                                             status), "ck_detect_batch_device")
        return [[Detection(dets[i * cap + k]) for k in range(counts[i])] for i in range(n)], list(status)

    def stage_ms(self):
        ms = A.StageMs()
        check(self._L.ck_last_stage_ms(self._h, C.byref(ms)), "ck_last_stage_ms")
        return {k: getattr(ms, k) for k, _ in A.StageMs._fields_}

    def fp64_probe(self, op, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        out = np.empty_like(a)
        bb = np.ascontiguousarray(b, np.float64) if b is not None else None
        check(self._L.ck_selftest_fp64(self._h, op, a.ctypes.data, bb.ctypes.data if bb is not None else None,
                                       a.size, out.ctypes.data), "ck_selftest_fp64")
        return out

    def _in(self, frames, n):
        if frames is None:
            if n is None:
                raise ValueError("n is required when running on uploaded frames")
            return None, None, n
        arr, keep = _images(frames)
        return arr, keep, len(arr)


def fourcc(code):
    """'GREY' -> the little-endian u32 the C ABI takes (crates/chalkydri/src/cameras/gst_to_cu.rs:171-179)."""
    b = code.encode("ascii")
    if len(b) != 4:
        raise ValueError("fourcc must be exactly 4 characters")
    return int.from_bytes(b, "little")


class IngestRing:
    """Pinned host slots + asynchronous upload in front of a detector (the pooled host buffers of the reference's camera
    layer, gst_to_cu.rs:49-72,131-188).  slot_view(s) is a writable numpy view [max_batch][h][stride] of pinned memory."""

    def __init__(self, detector, n_slots=2):
        self.det, self._L = detector, detector._L
        g = C.c_void_p()
        check(self._L.ck_ingest_create(detector._h, n_slots, C.byref(g)), "ck_ingest_create")
        self._g, self.n_slots = g, n_slots
        self.stride = self._L.ck_ingest_stride(g)

    def close(self):
        if self._g:
            self._L.ck_ingest_destroy(self._g)
            self._g = None

    def slot_view(self, slot):
        cfg = self.det.cfg
        ptr = self._L.ck_ingest_frame(self._g, slot, 0)
        pitch = self._L.ck_ingest_frame(self._g, slot, 1) - ptr if cfg.max_batch > 1 else self.stride * cfg.height
        buf = (C.c_uint8 * (pitch * cfg.max_batch)).from_address(ptr)
        a = np.frombuffer(buf, np.uint8).reshape(cfg.max_batch, pitch)[:, :self.stride * cfg.height]
        return a.reshape(cfg.max_batch, cfg.height, self.stride)

    def write(self, slot, index, frame, code="GREY"):
        arr, keep = _images(frame)
        check(self._L.ck_ingest_write(self._g, slot, index, arr, fourcc(code)), "ck_ingest_write")

    def submit(self, slot, n):
        check(self._L.ck_ingest_submit(self._g, slot, n), "ck_ingest_submit")

    def detect(self, slot, n, cap=64):
        dets = (A.Detection * (cap * n))()
        counts = (C.c_int32 * n)()
        status = (C.c_uint32 * n)()
        check(self._L.ck_detect_ingested(self._g, slot, n, dets, cap, counts, status), "ck_detect_ingested")
        return [[Detection(dets[i * cap + k]) for k in range(min(counts[i], cap))] for i in range(n)], np.array(status[:])

    def process(self, slot, n, pp, gyro, has_gyro):
        out = (A.VisionMeasurement * n)()
        valid = (C.c_int32 * n)()
        g = np.ascontiguousarray(gyro, np.float64)
        hg = np.ascontiguousarray(has_gyro, np.uint8)
        if g.size != n or hg.size != n:
            raise ValueError("gyro / has_gyro must hold one entry per submitted frame")
        check(self._L.ck_process_ingested(self._g, slot, n, C.byref(pp), g.ctypes.data, hg.ctypes.data, out, valid), "ck_process_ingested")
        return out, np.array(valid[:])
