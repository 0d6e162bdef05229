"""Ingest ring: frames written into pinned slots with a stride larger than the width, uploaded asynchronously, give exactly
the records of the plain upload path; formats without a leading luma plane are refused."""
import ctypes as C

import numpy as np
import pytest

import scenes
from chalkydri_amd import _abi as A
from chalkydri_amd.apriltags import AprilTags
from chalkydri_amd.detector import IngestRing, fourcc

pytestmark = pytest.mark.gpu


def test_ring_matches_upload_path(built):
    w, h, f, n = 640, 480, 600.0, 4
    layout = scenes.wall_layout(6, cols=3)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
    calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0)
    rng = np.random.default_rng(8)
    frames, gyros = [], []
    for i in range(2 * n):
        pose = (rng.uniform(1.8, 2.4), rng.uniform(-0.2, 0.2), rng.uniform(-0.1, 0.1))
        fr, _ = scenes.render_view(500 + i, w, h, f, layout, pose, r2c, noise_amp=2)
        frames.append(fr); gyros.append(pose[2])
    frames = np.stack(frames)
    task = AprilTags(w, h, layout, calib, r2c, cam_id=2, max_batch=n)
    want = [task.process_batch(frames[b * n:(b + 1) * n], gyros[b * n:(b + 1) * n]) for b in range(2)]
    ring = IngestRing(task.detector, n_slots=2)
    assert ring.stride >= w and ring.stride % 16 == 0
    # batch 0 through ck_ingest_write from camera buffers whose stride exceeds the width (gst_to_cu.rs:60-63)
    padded = np.full((n, h, w + 24), 0xAB, np.uint8)
    padded[:, :, :w] = frames[:n]
    for i in range(n):
        img = (A.ImageU8 * 1)()
        img[0].buf, img[0].width, img[0].height, img[0].stride = padded[i].ctypes.data, w, h, w + 24
        assert task.detector._L.ck_ingest_write(ring._g, 0, i, img, fourcc("GRAY")) == 0
    # batch 1 written straight into the pinned slot, as a camera layer that owns the ring would
    ring.slot_view(1)[:, :, :w] = frames[n:]
    ring.submit(0, n)
    ring.submit(1, n)                                   # uploads while batch 0 is being processed
    for b in range(2):
        out, valid = ring.process(b, n, task._pp, gyros[b * n:(b + 1) * n], np.ones(n, np.uint8))
        recs, v = want[b]
        assert np.array_equal(valid, v) and valid.all()
        for i in range(n):
            assert bytes(out[i]) == bytes(recs[i])
    # the caller's n must be the submitted count: the output arrays are sized from it
    from chalkydri_amd._lib import ChalkydriError
    with pytest.raises(ChalkydriError) as e:
        ring.detect(1, n - 1)
    assert e.value.code == A.CK_EINVAL
    with pytest.raises(ChalkydriError):
        ring.process(1, n + 1, task._pp, np.zeros(n + 1), np.ones(n + 1, np.uint8))
    dets, status = ring.detect(1, n)
    assert all(len(d) == 6 for d in dets) and not (status & ~np.uint32(A.CK_FRAME_UNVERIFIED_ID)).any()
    # a packed-colour format has no leading luma plane: refused, like the detector's 8-bit-luma assumption
    arr = (A.ImageU8 * 1)()
    arr[0].buf, arr[0].width, arr[0].height, arr[0].stride = frames[0].ctypes.data, w, h, w
    assert task.detector._L.ck_ingest_write(ring._g, 0, 0, arr, fourcc("YUYV")) != 0
    ring.close()
    task.detector.close()


def test_strided_host_and_device_frames(built, oracle):
    """image_u8_t {buf,width,height,stride} with stride > width (crates/apriltags/src/lib.rs:197-213; gst_to_cu.rs:60-63):
    host frames through ck_detect_batch, and device-resident frames whose rows are not 16-byte aligned through
    ck_detect_batch_device (the library restages them), give the detections of the tightly packed frames."""
    import torch
    from chalkydri_amd import default_config, synth
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n, pad = 640, 480, 2, 11                      # stride 651: neither a multiple of 16 nor of 4
    frames, _ = synth.render_batch(31, n, w, h, 4, noise_amp=2)
    padded = np.full((n, h, w + pad), 0x5A, np.uint8)
    padded[:, :, :w] = frames
    det = AprilTagDetector(w, h, max_batch=n)
    want = det.detect_batch(frames)
    # host path: hand-built image_u8_t views into the padded buffer
    imgs = (A.ImageU8 * n)()
    for i in range(n):
        imgs[i].buf, imgs[i].width, imgs[i].height, imgs[i].stride = padded[i].ctypes.data, w, h, w + pad
    cap = 64
    dets = (A.Detection * (cap * n))()
    counts = (C.c_int32 * n)()
    status = (C.c_uint32 * n)()
    assert det._L.ck_detect_batch(det._h, imgs, n, dets, cap, counts, status) == 0
    # device path: the same padded frames resident on the GPU
    d = torch.from_numpy(padded).cuda()
    dets2 = (A.Detection * (cap * n))()
    counts2 = (C.c_int32 * n)()
    status2 = (C.c_uint32 * n)()
    assert det._L.ck_detect_batch_device(det._h, C.c_void_p(d.data_ptr()), n, w + pad, (w + pad) * h, dets2, cap, counts2, status2) == 0
    cfg = default_config(w, h)
    for i in range(n):
        ref, _ = oracle.detect(frames[i], cfg)
        assert counts[i] == counts2[i] == len(want[i]) == len(ref) and status[i] == status2[i] and not status[i] & ~A.CK_FRAME_UNVERIFIED_ID
        for k in range(counts[i]):
            for got in (dets[i * cap + k], dets2[i * cap + k]):
                assert got.id == want[i][k].id() and got.hamming == want[i][k].hamming()
                assert [list(p) for p in got.p] == np.asarray(want[i][k].corners()).tolist()
    det.close()
