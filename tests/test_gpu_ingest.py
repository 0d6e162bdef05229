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
    dets, status = ring.detect(1, n)
    assert all(len(d) == 6 for d in dets) and not status.any()
    # a packed-colour format has no leading luma plane: refused, like the detector's 8-bit-luma assumption
    arr = (A.ImageU8 * 1)()
    arr[0].buf, arr[0].width, arr[0].height, arr[0].stride = frames[0].ctypes.data, w, h, w
    assert task.detector._L.ck_ingest_write(ring._g, 0, 0, arr, fourcc("YUYV")) != 0
    ring.close()
    task.detector.close()
