"""One handle per host thread (SURVEY.md §8b threading row: a handle mirrors `&mut self` and is not shared, but several handles
may be driven from separate host threads), and the argument checks of the batch entry points."""
import ctypes as C
import threading

import numpy as np
import pytest

from chalkydri_amd import _abi as A
from chalkydri_amd import synth

pytestmark = pytest.mark.gpu


def _key(frames_dets):
    return [[(d.id(), d.hamming(), np.asarray(d.corners()).tobytes()) for d in fr] for fr in frames_dets]


def test_two_handles_on_two_host_threads(built):
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n = 640, 480, 6
    ns = (n, 3)   # the second handle makes small calls: its quad-fit classes run on the handle's side streams
    stacks = [np.stack([synth.render(synth.frame_seed(7, 10 * t + i), w, h, 4, min_side=40, max_side=140, noise_amp=3)[0]
                        for i in range(ns[t])]) for t in range(2)]
    dets = [AprilTagDetector(w, h, max_batch=n) for _ in range(2)]
    want = [_key(dets[t].detect_batch(stacks[t])) for t in range(2)]
    assert all(sum(len(f) for f in k) >= len(k) for k in want)
    got = [[None] * 8 for _ in range(2)]
    errs = []

    def work(t):
        try:
            for r in range(8):                         # ctypes drops the GIL inside the call: the two handles really overlap
                got[t][r] = _key(dets[t].detect_batch(stacks[t]))
        except Exception as e:                         # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not errs, errs
    for t in range(2):
        assert all(g == want[t] for g in got[t])
    [d.close() for d in dets]


def test_batch_entry_points_reject_bad_arguments(built):
    from chalkydri_amd._abi import CK_ECAPACITY, CK_EINVAL
    from chalkydri_amd.detector import AprilTagDetector, _images
    w, h = 64, 48
    det = AprilTagDetector(w, h, max_batch=2)
    L, hd = det._L, det._h
    frames = np.zeros((3, h, w), np.uint8)
    arr, keep = _images(frames)
    out = (A.Detection * (3 * 4))()
    counts = (C.c_int32 * 3)()
    status = (C.c_uint32 * 3)()
    assert L.ck_detect_batch(hd, arr, 3, out, 4, counts, status) == CK_ECAPACITY      # n > max_batch
    assert L.ck_detect_batch(hd, arr, -1, out, 4, counts, status) == CK_EINVAL
    assert L.ck_detect_batch(hd, None, 1, out, 4, counts, status) == CK_EINVAL
    assert L.ck_detect_batch(hd, arr, 1, out, 0, counts, status) == CK_EINVAL         # no room for any detection
    assert L.ck_detect_batch(hd, arr, 1, None, 4, counts, status) == CK_EINVAL
    bad, keep2 = _images(np.zeros((1, h, w), np.uint8))
    bad[0].stride = w - 1
    assert L.ck_detect_batch(hd, bad, 1, out, 4, counts, status) == CK_EINVAL         # stride < width
    bad[0].stride = w
    bad[0].width = w + 4
    assert L.ck_detect_batch(hd, bad, 1, out, 4, counts, status) == CK_EINVAL         # not the geometry the handle was made for
    assert L.ck_detect_uploaded(hd, 1, out, 4, counts, status) == CK_EINVAL           # nothing staged yet
    assert L.ck_detect_batch(hd, arr, 0, out, 4, counts, status) == 0                 # empty batch is fine
    assert L.ck_detect_batch(hd, arr, 2, out, 4, counts, status) == 0 and list(counts)[:2] == [0, 0]
    det.close()
