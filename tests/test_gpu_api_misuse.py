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


def test_cat_and_solver_entry_points_reject_bad_arguments(built):
    from chalkydri_amd.detector import AprilTagDetector
    w, h = 96, 64
    det = AprilTagDetector(w, h)
    L, hd = det._L, det._h
    rgb = np.random.default_rng(1).integers(0, 256, (h, w, 3), dtype=np.uint8)
    cls = np.zeros((h, w), np.uint8)
    pts = np.zeros(64, np.uint32); lines = np.zeros(64, np.uint32)
    n_pts = C.c_int32(0); n_lines = C.c_int32(0)
    R, K, P, Ln = rgb.ctypes.data, cls.ctypes.data, pts.ctypes.data, lines.ctypes.data
    bad = []
    def expect_error(rc, what):
        if rc == 0:
            bad.append(what)
    expect_error(L.ck_cat_calc_otsu(hd, None, w, h, K), "otsu: null frame")
    expect_error(L.ck_cat_calc_otsu(hd, R, w, h, None), "otsu: null output")
    expect_error(L.ck_cat_calc_otsu(hd, R, 0, h, K), "otsu: zero width")
    expect_error(L.ck_cat_calc_otsu(hd, R, w, -1, K), "otsu: negative height")
    expect_error(L.ck_cat_calc_otsu(None, R, w, h, K), "otsu: null handle")
    expect_error(L.ck_cat_thresh(None, R, w, h, K), "thresh: null handle")
    expect_error(L.ck_cat_thresh(hd, None, w, h, K), "thresh: null frame")
    expect_error(L.ck_cat_detect_corners(hd, K, w, h, P, -1, C.byref(n_pts)), "corners: negative capacity")
    expect_error(L.ck_cat_detect_corners(hd, K, w, h, P, 32, None), "corners: null count")
    expect_error(L.ck_cat_detect_corners(hd, None, w, h, P, 32, C.byref(n_pts)), "corners: null classes")
    expect_error(L.ck_cat_check_edges(hd, K, w, h, P, -3, Ln, 16, C.byref(n_lines)), "edges: negative point count")
    expect_error(L.ck_cat_check_edges(hd, K, w, h, None, 4, Ln, 16, C.byref(n_lines)), "edges: null points")
    expect_error(L.ck_cat_check_edges(hd, K, w, h, P, 4, Ln, 16, None), "edges: null count")
    expect_error(L.ck_cat_connected_components(hd, None, w, h, None, None), "components: nulls")
    # process_frame: the reference asserts the buffer length (lib.rs:267)
    rc = L.ck_cat_process_frame(hd, R, w * h * 3 - 1, w, h, K, P, 16, C.byref(n_pts), Ln, 16, C.byref(n_lines))
    assert rc == A.CK_EINVAL
    # the solver: counts and offsets that do not fit the arrays
    prm = A.SqpnpParams(); L.ck_sqpnp_params_default(C.byref(prm))
    prob = (A.SqpnpProblem * 1)(); res = (A.SqpnpResult * 1)()
    tags = (A.Iso3 * 2)(); bearings = np.zeros(3 * 8)
    for t in tags:
        t.q[0] = 1.0
    prob[0].robot_to_cam.q[0] = 1.0
    B = bearings.ctypes.data
    def solve(n=1, params=prm, out=res, n_tags_total=2, n_bearings_total=8, tg=tags, bp=B):
        return L.ck_sqpnp_solve_batch(hd, C.byref(params) if params is not None else None, prob, n, tg, n_tags_total, bp,
                                      n_bearings_total, out)
    prob[0].n_tags, prob[0].n_bearings, prob[0].tag_offset, prob[0].bearing_offset = 2, 8, 1, 0
    expect_error(solve(), "solver: tags 1..2 of 2")
    prob[0].tag_offset, prob[0].bearing_offset = 0, 4
    expect_error(solve(), "solver: bearings 4..11 of 8")
    prob[0].bearing_offset, prob[0].tag_offset = -1, 0
    expect_error(solve(), "solver: negative bearing offset")
    prob[0].bearing_offset, prob[0].tag_offset = 0, 2**31 - 1
    expect_error(solve(), "solver: tag offset + count wraps")
    prob[0].tag_offset, prob[0].n_tags = 0, -1
    expect_error(solve(), "solver: negative tag count")
    prob[0].n_tags = 2
    expect_error(solve(params=None), "solver: null parameters")
    expect_error(solve(n=-1), "solver: negative problem count")
    expect_error(solve(out=None), "solver: null results")
    expect_error(solve(tg=None), "solver: null tags")
    expect_error(solve(bp=None), "solver: null bearings")
    assert not bad, bad
    # a well-formed but unsolvable problem (all-zero bearings) is a result with valid = 0, not an error; no problems is a no-op
    assert solve() == 0 and res[0].valid == 0
    assert solve(n=0) == 0
    # and the handle still classifies a frame
    assert L.ck_cat_calc_otsu(hd, R, w, h, K) == 0 and set(np.unique(cls)) <= {0, 1, 2}
    det.close()
