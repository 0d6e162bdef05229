"""The C ABI under misuse, on a device: null pointers, negative and oversized counts, zero capacities, calls before anything was
uploaded.  Every call must come back with an error code (or do nothing) — never abort, never touch memory it does not own — and
the handle must still work afterwards (include/chalkydri_hip.h: "int status, never throw/abort across the ABI")."""
import ctypes as C

import numpy as np
import pytest

from chalkydri_amd import _abi as A
from chalkydri_amd import synth

pytestmark = pytest.mark.gpu


def test_detector_entry_points_reject_bad_arguments(built):
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n = 320, 240, 2
    det = AprilTagDetector(w, h, max_batch=n)
    L, hd = det._L, det._h
    frames, _ = synth.render_batch(71, n, w, h, 2)
    want = [[(d.id(), d.corners().tobytes()) for d in fr] for fr in det.detect_batch(frames)]
    cap = 16
    dets = (A.Detection * (cap * 8))(); counts = (C.c_int32 * 8)(); status = (C.c_uint32 * 8)()
    imgs = (A.ImageU8 * 8)()
    for i in range(8):
        imgs[i].buf, imgs[i].width, imgs[i].height, imgs[i].stride = frames[i % n].ctypes.data, w, h, w
    bad = []
    def expect_error(rc, what):
        if rc == 0:
            bad.append(what)
    expect_error(L.ck_detect_batch(None, imgs, n, dets, cap, counts, status), "null handle")
    expect_error(L.ck_detect_batch(hd, imgs, -1, dets, cap, counts, status), "n < 0")
    expect_error(L.ck_detect_batch(hd, imgs, n + 1, dets, cap, counts, status), "n > max_batch")
    expect_error(L.ck_detect_batch(hd, imgs, n, None, cap, counts, status), "null detections")
    expect_error(L.ck_detect_batch(hd, imgs, n, dets, -1, counts, status), "cap < 0")
    expect_error(L.ck_detect_batch(hd, imgs, n, dets, cap, None, status), "null counts")
    wrong = (A.ImageU8 * n)()
    for i in range(n):
        wrong[i].buf, wrong[i].width, wrong[i].height, wrong[i].stride = frames[i].ctypes.data, w - 4, h, w
    expect_error(L.ck_detect_batch(hd, wrong, n, dets, cap, counts, status), "image of another width")
    for i in range(n):
        wrong[i].buf, wrong[i].width, wrong[i].height, wrong[i].stride = frames[i].ctypes.data, w, h, w - 1
    expect_error(L.ck_detect_batch(hd, wrong, n, dets, cap, counts, status), "stride < width")
    for i in range(n):
        wrong[i].buf, wrong[i].width, wrong[i].height, wrong[i].stride = None, w, h, w
    expect_error(L.ck_detect_batch(hd, wrong, n, dets, cap, counts, status), "null image buffer")
    expect_error(L.ck_detect_batch_device(hd, None, n, w, w * h, dets, cap, counts, status), "null device frames")
    expect_error(L.ck_detect_uploaded(hd, n + 5, dets, cap, counts, status), "more frames than were uploaded")
    expect_error(L.ck_threshold_batch(hd, imgs, n, None), "null threshold output")
    expect_error(L.ck_segment_batch(hd, imgs, n + 1, None, None), "segment: n > max_batch")
    # n = 0 is a call that does nothing
    assert L.ck_detect_batch(hd, imgs, 0, dets, cap, counts, status) == 0
    # a zero capacity is refused (or, were it served, would report the overflow and write nothing)
    if L.ck_detect_batch(hd, imgs, n, dets, 0, counts, status) == 0:
        assert all(counts[i] == 0 for i in range(n)) and all(status[i] & A.CK_FRAME_DETS_OVERFLOW for i in range(n) if want[i])
    assert not bad, f"accepted: {bad}"
    # the handle still works
    again = [[(d.id(), d.corners().tobytes()) for d in fr] for fr in det.detect_batch(frames)]
    assert again == want
    det.close()


def test_create_rejects_what_it_cannot_serve(built):
    from chalkydri_amd import default_config
    from chalkydri_amd._lib import lib
    L = lib()
    for kw in ({"width": 0}, {"height": -5}, {"width": 5000}, {"max_batch": 0}, {"quad_decimate": 0}, {"max_nmaxima": 3}, {"max_nmaxima": 13},
               {"device": 99}, {"n_families": 0}):
        cfg = default_config(320, 240)
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = L.ck_create(C.byref(cfg), C.byref(h))
        assert rc != 0 and not h.value, f"ck_create accepted {kw}"
    assert L.ck_create(None, C.byref(C.c_void_p())) != 0
    L.ck_destroy(None)   # a null handle is ignored


def test_ingest_and_pose_entry_points_reject_bad_arguments(built):
    import scenes
    from chalkydri_amd.apriltags import AprilTags
    from chalkydri_amd.detector import IngestRing, fourcc
    w, h, n = 320, 240, 2
    layout = scenes.wall_layout(4, cols=2)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.1, "y": 0.0, "z": 0.5}
    task = AprilTags(w, h, layout, scenes.pinhole_calib(300.0, w / 2.0, h / 2.0), r2c, cam_id=1, max_batch=n)
    det = task.detector
    L, hd = det._L, det._h
    frames, _ = synth.render_batch(72, n, w, h, 2)
    ring = IngestRing(det, n_slots=2)
    img = (A.ImageU8 * 1)()
    img[0].buf, img[0].width, img[0].height, img[0].stride = frames[0].ctypes.data, w, h, w
    code = fourcc("GREY")
    L.ck_ingest_frame.restype = C.c_void_p
    assert L.ck_ingest_write(ring._g, 5, 0, img, code) != 0          # no such slot
    assert L.ck_ingest_write(ring._g, -1, 0, img, code) != 0
    assert L.ck_ingest_write(ring._g, 0, n, img, code) != 0          # no such frame in the slot
    assert L.ck_ingest_write(ring._g, 0, 0, None, code) != 0
    assert L.ck_ingest_write(None, 0, 0, img, code) != 0
    assert not L.ck_ingest_frame(ring._g, 9, 0) and not L.ck_ingest_frame(ring._g, 0, n)
    assert L.ck_ingest_submit(ring._g, 0, n + 1) != 0 and L.ck_ingest_submit(ring._g, 7, 1) != 0 and L.ck_ingest_submit(ring._g, 0, -2) != 0
    g = C.c_void_p()
    assert L.ck_ingest_create(hd, 0, C.byref(g)) != 0 and L.ck_ingest_create(None, 2, C.byref(g)) != 0
    out = (A.VisionMeasurement * n)(); valid = (C.c_int32 * n)()
    gyro = (C.c_double * n)(); has = (C.c_uint8 * n)(1, 1)
    assert L.ck_process_uploaded(hd, n, None, gyro, has, out, valid) != 0            # no parameters
    assert L.ck_process_uploaded(hd, n, C.byref(task._pp), gyro, has, None, valid) != 0
    assert L.ck_process_uploaded(hd, n + 3, C.byref(task._pp), gyro, has, out, valid) != 0
    assert L.ck_process_uploaded(None, n, C.byref(task._pp), gyro, has, out, valid) != 0
    # everything still works afterwards, through the ring
    ring.slot_view(0)[:n, :, :w] = frames
    ring.submit(0, n)
    dets, status = ring.detect(0, n)
    assert [len(d) for d in dets] == [len(d) for d in det.detect_batch(frames)]
    ring.close()
    det.close()
