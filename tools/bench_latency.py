"""Single-frame latency (max_batch = 1): what a Copper task that calls the detector once per frame would see."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chalkydri_amd import scenes
from chalkydri_amd.apriltags import AprilTags

w, h = 1280, 800
noise = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames, gyro, layout, calib, r2c = scenes.bench_stream(2, 8, w, h, 6, stream=0, unique=8, noise_amp=noise)
task = AprilTags(w, h, layout, calib, r2c, cam_id=0, max_batch=1, quad_decimate=dec)
for i in range(4):
    task.process_batch(frames[i:i + 1], [float(gyro[i])])
ts = []
for rep in range(5):
    for i in range(8):
        t0 = time.perf_counter()
        recs, valid = task.process_batch(frames[i:i + 1], [float(gyro[i])])   # host frame in, 64-byte record out
        ts.append(time.perf_counter() - t0)
        assert valid[0]
ts = np.array(ts) * 1e3
print(json.dumps({"workload": f"{w}x{h} one frame per call, noise+-{noise}, quad_decimate={dec}, host frame in (pageable), record out",
                  "median_ms": round(float(np.median(ts)), 3), "p10_ms": round(float(np.percentile(ts, 10)), 3), "p90_ms": round(float(np.percentile(ts, 90)), 3)}))
