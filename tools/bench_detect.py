"""Dev loop: full detector on a resident batch; prints per-stage ms and FPS."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chalkydri_amd import _lib
if os.environ.get("LIB"):   # A/B against another build of the library in the same gpurun call
    _lib.LIB_PATH = os.environ["LIB"]
from chalkydri_amd import synth
from chalkydri_amd.detector import AprilTagDetector

w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
noise = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dec = int(sys.argv[5]) if len(sys.argv) > 5 else 1
uniq = min(n, 16)
base = np.stack([synth.render(synth.frame_seed(2, i), w, h, 6, noise_amp=noise)[0] for i in range(uniq)])
frames = np.concatenate([base] * ((n + uniq - 1) // uniq))[:n]
det = AprilTagDetector(w, h, max_batch=n, quad_decimate=dec)
det.upload(frames)
for it in range(3):
    t = time.time()
    dets, st = det.detect_batch(None, n=n, return_status=True)
    dt = time.time() - t
    ms = det.stage_ms()
    print(json.dumps({"n": n, "noise": noise, "decimate": dec, "wall_ms": round(dt * 1e3, 2), "fps": round(n / dt, 1),
                      "dets_per_frame": sum(len(d) for d in dets) / n, "status_or": int(np.bitwise_or.reduce(st)),
                      **{k: round(v, 3) for k, v in ms.items()}}))
