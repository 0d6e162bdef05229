import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as O
from chalkydri_amd import _abi as A, default_config, scenes
from chalkydri_amd.apriltags import AprilTags
w, h, n = 1280, 800, 32
frames, gyro, layout, calib, r2c = scenes.bench_stream(2, n, w, h, 6, stream=0, unique=32, noise_amp=3)
task = AprilTags(w, h, layout, calib, r2c, cam_id=0, max_batch=n)
recs, valid = task.process_batch(frames, list(gyro))
cfg = default_config(w, h)
L = O.lib()
for i in range(n):
    out = A.VisionMeasurement(); v = C.c_int(0)
    L.ora_process_frame(C.c_void_p(frames[i].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp), C.c_double(float(gyro[i])), 1, C.byref(out), C.byref(v))
    r = recs[i]
    ok = bool(v.value) == bool(valid[i]) and (not v.value or (abs(r.pose_x-out.pose_x) < 1e-6 and abs(r.pose_rot-out.pose_rot) < 1e-7))
    if not ok:
        print("frame", i, "gpu valid", bool(valid[i]), (r.pose_x, r.pose_y, r.pose_rot, r.tag_count), "oracle valid", bool(v.value), (out.pose_x, out.pose_y, out.pose_rot, out.tag_count))
print("checked", n)
