"""Runs the bench batch repeatedly and checks that every pass returns the same bytes (records, validity, detections):
a cheap detector for races in the lock-free phases (union-find hooks, hash inserts, run hand-out)."""
import sys, os, hashlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chalkydri_amd import scenes
from chalkydri_amd.apriltags import AprilTags

w, h, n = 1280, 800, 256
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames, gyro, layout, calib, r2c = scenes.bench_stream(2, n, w, h, 6, stream=0, unique=64, noise_amp=3)
task = AprilTags(w, h, layout, calib, r2c, cam_id=0, max_batch=n)
task.detector.upload(frames)
seen = set()
for p in range(passes):
    recs, valid = task.process_batch(None, list(gyro), n=n)
    dets, status = task.detector.detect_batch(None, n=n, cap=64, return_status=True)
    hh = hashlib.sha256()
    for r in recs:
        hh.update(bytes(r))
    hh.update(np.asarray(valid, np.int32).tobytes()); hh.update(np.asarray(status, np.uint32).tobytes())
    for fr in dets:
        for d in fr:
            hh.update(np.asarray([d.id(), d.hamming()], np.int64).tobytes()); hh.update(np.asarray(d.corners(), np.float64).tobytes())
    seen.add(hh.hexdigest())
print(json.dumps({"passes": passes, "distinct_results": len(seen), "valid": int(np.sum(valid))}))
sys.exit(0 if len(seen) == 1 else 1)
