"""Dev loop: time the threshold+segment kernels on a resident batch and report algorithmic GB/s (7 B/px)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chalkydri_amd import _lib
if os.environ.get("LIB"):   # A/B against another build of the library in the same gpurun call (boxes differ by a few per cent)
    _lib.LIB_PATH = os.environ["LIB"]
from chalkydri_amd import synth
from chalkydri_amd.detector import AprilTagDetector

w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "synth"
uniq = min(n, 16)
t = time.time()
if kind == "synth":
    base = np.stack([synth.render(synth.frame_seed(2, i), w, h, 6)[0] for i in range(uniq)])
elif kind == "clean":
    base = np.stack([synth.render(synth.frame_seed(2, i), w, h, 6, noise_amp=1)[0] for i in range(uniq)])
elif kind in ("bench", "bench_quiet"):   # the frames bench.py itself times: the headline batch / `also.threshold_segment_low_noise`
    from chalkydri_amd import scenes
    base = scenes.bench_stream(2, uniq, w, h, 6, stream=0, unique=uniq, noise_amp=3 if kind == "bench" else 1)[0]
else:
    base = np.random.default_rng(0).integers(0, 256, (uniq, h, w), dtype=np.uint8)
frames = np.concatenate([base] * ((n + uniq - 1) // uniq))[:n]
print("render", round(time.time() - t, 2), "s", file=sys.stderr)
det = AprilTagDetector(w, h, max_batch=n)
det.upload(frames)
for it in range(3):
    ms = det.time_threshold_segment(n, 10)
    gbs = 7.0 * w * h * n / (ms * 1e-3) / 1e9
    print(json.dumps({"w": w, "h": h, "n": n, "kind": kind, "ms_per_batch": round(ms, 4), "us_per_frame": round(ms * 1e3 / n, 3),
                      "alg_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4)}))
