"""Randomised check that a frame's result does not depend on the call that carries it: whole batches (one stream, or split over
two with CK_STREAMS=2 / CK_PARTS), calls of at most four frames (calls of up to 16 run their fit classes side by side on three streams) and
single-frame calls must return the same bytes per frame — detections and 64-byte pose records.
usage: [CK_STREAMS=2 CK_PARTS=3] python tests/stress_batch.py [cases] [seed]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chalkydri_amd import scenes
from chalkydri_amd.apriltags import AprilTags


def _sig(dets):
    return [(d.id(), d.hamming(), d.family(), float(d.decision_margin()), d.center().tobytes(), d.corners().tobytes()) for d in dets]


def run(cases, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        w = int(rng.integers(320, 1040)); h = int(rng.integers(240, 720))   # any width and height, odd ones included
        n = int(rng.integers(2, 41))   # up to 16 frames run the fit classes side by side, more on one lane plus the tail stream
        dec = int(rng.choice([1, 2]))
        frames, gyro, layout, calib, r2c = scenes.bench_stream(3, n, w, h, int(rng.integers(1, 7)), stream=int(rng.integers(0, 50)), unique=n,
                                                               noise_amp=int(rng.choice([0, 1, 3])))
        task = AprilTags(w, h, layout, calib, r2c, cam_id=1, max_batch=n, quad_decimate=dec)
        det = task.detector
        whole = [_sig(d) for d in det.detect_batch(frames)]
        recs, valid = task.process_batch(frames, list(gyro))
        whole_recs = [bytes(r) for r in recs]
        k = int(rng.integers(1, 5))                            # a small call: the first k frames
        small = [_sig(d) for d in det.detect_batch(frames[:k])]
        srecs, _ = task.process_batch(frames[:k], list(gyro[:k]))
        i = int(rng.integers(0, n))                            # one frame alone
        single = _sig(det.detect_batch(frames[i:i + 1])[0])
        orec, _ = task.process_batch(frames[i:i + 1], [gyro[i]])
        ok = small == whole[:k] and [bytes(r) for r in srecs] == whole_recs[:k] and single == whole[i] and bytes(orec[0]) == whole_recs[i]
        if not ok:
            bad += 1
            print(json.dumps({"case": c, "w": w, "h": h, "n": n, "k": k, "i": i, "dec": dec}))
        det.close()
    print(json.dumps({"cases": cases, "mismatching_cases": bad, "CK_STREAMS": os.environ.get("CK_STREAMS", "1"), "CK_PARTS": os.environ.get("CK_PARTS", "-")}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
