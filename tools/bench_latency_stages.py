"""Stage split of one frame per call (max_batch = 1).  LIB=<path> selects another build of the library."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chalkydri_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.environ["LIB"]
from chalkydri_amd import scenes
from chalkydri_amd.apriltags import AprilTags
w, h = 1280, 800
noise = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames, gyro, layout, calib, r2c = scenes.bench_stream(2, 8, w, h, 6, stream=0, unique=8, noise_amp=noise)
task = AprilTags(w, h, layout, calib, r2c, cam_id=0, max_batch=1, quad_decimate=dec)
acc = {}
for rep in range(6):
    for i in range(8):
        task.process_batch(frames[i:i + 1], [float(gyro[i])])
        if rep:
            for k, v in task.detector.stage_ms().items():
                acc.setdefault(k, []).append(v)
print(json.dumps({"noise": noise, "decimate": dec, **{k: round(float(np.median(v)), 3) for k, v in acc.items()}}))
