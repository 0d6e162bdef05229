"""Randomised check of the input boundary: a frame must give the same detections however it reaches the device — tightly packed
host frames (ck_detect_batch), host image_u8_t views with stride > width, the pinned ingest ring (ck_ingest_write with a
padded source, or written straight into the slot), and device-resident frames with arbitrary stride and misaligned base
(ck_detect_batch_device: aligned ones are used in place, others restaged).  Any width and height, odd ones included.
usage: python tests/stress_ingest.py [cases] [seed]"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from chalkydri_amd import _abi as A, synth
from chalkydri_amd.detector import AprilTagDetector, IngestRing, fourcc


def _sig(dets):
    return [(d.id(), d.hamming(), d.family(), d.center().tobytes(), d.corners().tobytes()) for d in dets]


def _raw(dets, counts, cap, n):
    return [[(dets[i * cap + k].id, dets[i * cap + k].hamming, dets[i * cap + k].family, bytes(np.asarray(dets[i * cap + k].c[:], np.float64).tobytes()),
              np.asarray([list(p) for p in dets[i * cap + k].p], np.float64).tobytes()) for k in range(counts[i])] for i in range(n)]


def run(cases, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        w = int(rng.integers(200, 900)); h = int(rng.integers(150, 600))
        n = int(rng.integers(1, 5)); dec = int(rng.choice([1, 1, 2])); cap = 64
        frames, _ = synth.render_batch(60 + c, n, w, h, int(rng.integers(1, 5)), noise_amp=int(rng.choice([0, 2])))
        det = AprilTagDetector(w, h, max_batch=n, quad_decimate=dec)
        want = [_sig(d) for d in det.detect_batch(frames)]
        why = None
        pad = int(rng.integers(1, 40))
        padded = np.full((n, h, w + pad), 0x5A, np.uint8); padded[:, :, :w] = frames
        imgs = (A.ImageU8 * n)()
        for i in range(n):
            imgs[i].buf, imgs[i].width, imgs[i].height, imgs[i].stride = padded[i].ctypes.data, w, h, w + pad
        dets = (A.Detection * (cap * n))(); counts = (C.c_int32 * n)(); status = (C.c_uint32 * n)()
        if det._L.ck_detect_batch(det._h, imgs, n, dets, cap, counts, status) != 0 or _raw(dets, counts, cap, n) != want: why = "strided host views"
        # device-resident, base misaligned by `off` bytes, row stride w + pad
        off = int(rng.integers(0, 16))
        flat = torch.zeros(off + n * h * (w + pad) + 64, dtype=torch.uint8, device="cuda")
        flat[off:off + n * h * (w + pad)] = torch.from_numpy(padded.reshape(-1)).cuda()
        dets2 = (A.Detection * (cap * n))(); counts2 = (C.c_int32 * n)(); status2 = (C.c_uint32 * n)()
        rc = det._L.ck_detect_batch_device(det._h, C.c_void_p(flat.data_ptr() + off), n, w + pad, (w + pad) * h, dets2, cap, counts2, status2)
        if why is None and (rc != 0 or _raw(dets2, counts2, cap, n) != want): why = "device frames"
        ring = IngestRing(det, n_slots=2)
        for i in range(n):
            img = (A.ImageU8 * 1)()
            img[0].buf, img[0].width, img[0].height, img[0].stride = padded[i].ctypes.data, w, h, w + pad
            if det._L.ck_ingest_write(ring._g, 0, i, img, fourcc(str(rng.choice(["GREY", "GRAY", "Y800"])))) != 0 and why is None: why = "ck_ingest_write"
        ring.slot_view(1)[:n, :, :w] = frames
        ring.submit(0, n); ring.submit(1, n)
        for slot in (0, 1):
            got, _ = ring.detect(slot, n)
            if why is None and [_sig(d) for d in got] != want: why = f"ingest ring slot {slot}"
        ring.close()
        if why:
            bad += 1
            print(json.dumps({"case": c, "w": w, "h": h, "n": n, "dec": dec, "pad": pad, "off": off, "first_difference": why}))
        det.close()
    print(json.dumps({"cases": cases, "mismatching_cases": bad}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
