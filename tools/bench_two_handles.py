"""Experiment: two handles, each with a full 256-frame batch, driven by two host threads (their kernels interleave on the GPU: one
handle's low-occupancy tail — decode, pose — beside the other's threshold + segmentation).  usage: bench_two_handles.py [handles]"""
import sys, os, time, json, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chalkydri_amd import scenes
from chalkydri_amd.apriltags import AprilTags

w, h, n = 1280, 800, 256
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames, gyro, layout, calib, r2c = scenes.bench_stream(2, n, w, h, 6, stream=0, unique=32, noise_amp=3)
tasks = []
for p in range(parts):
    t = AprilTags(w, h, layout, calib, r2c, cam_id=p, max_batch=n)
    t.detector.upload(frames)
    tasks.append(t)

def run(p, steps):
    for _ in range(steps):
        tasks[p].process_batch(None, list(gyro), n=n)

for p in range(parts):
    run(p, 2)
steps = 10
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(p, steps)) for p in range(parts)]
[t.start() for t in th]
[t.join() for t in th]
dt = time.perf_counter() - t0
print(json.dumps({"handles": parts, "fps": round(parts * n * steps / dt, 1), "ms_per_256": round(dt * 1e3 / steps / parts, 3)}))
