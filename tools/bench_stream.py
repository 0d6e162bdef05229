"""PCIe-inclusive throughput: host frames -> pinned ring -> asynchronous upload -> detect+pose, double-buffered.
Reported next to the HBM-resident number of bench.py; it is never bench.py's `value`."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chalkydri_amd import scenes
from chalkydri_amd.apriltags import AprilTags
from chalkydri_amd.detector import IngestRing

w, h, n = 1280, 800, 256
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
noise = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dec = int(sys.argv[3]) if len(sys.argv) > 3 else 1
frames, gyro, layout, calib, r2c = scenes.bench_stream(2, n, w, h, 6, stream=0, unique=32, noise_amp=noise)
task = AprilTags(w, h, layout, calib, r2c, cam_id=0, max_batch=n, quad_decimate=dec)
ring = IngestRing(task.detector, 2)
has = np.ones(n, np.uint8)
views = [ring.slot_view(0), ring.slot_view(1)]
for v in views:
    v[:, :, :w] = frames                      # the camera layer's job; not timed (it overlaps on other host cores)
ring.submit(0, n)
ring.process(0, n, task._pp, gyro, has)       # warm-up
t0 = time.perf_counter()
ring.submit(0, n)
valid_total = 0
for b in range(batches):
    if b + 1 < batches:
        ring.submit((b + 1) & 1, n)           # next batch uploads while this one computes
    out, valid = ring.process(b & 1, n, task._pp, gyro, has)
    valid_total += int(valid.sum())
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"{w}x{h} batch={n} noise+-{noise} decimate={dec}", "batches": batches, "pcie_inclusive_fps": round(batches * n / dt, 1),
                  "ms_per_batch": round(dt * 1e3 / batches, 3), "valid": valid_total, "h2d_bytes_per_batch": int(n * h * ring.stride)}))
