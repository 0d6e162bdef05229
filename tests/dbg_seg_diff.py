"""Where do the device labels differ from the oracle's?  python tests/dbg_seg_diff.py <kind> <w> <h> [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle
import test_gpu_segment as T
from chalkydri_amd.detector import AprilTagDetector
kind, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 3
frames = T._frames(kind, w, h, 1, seed)
det = AprilTagDetector(w, h, max_batch=1)
th = det.threshold(frames)
labels, sizes = det.segment(frames)
oth = pyoracle.threshold(frames[0])
print("threshold equal:", np.array_equal(th[0], oth))
ol, osz = pyoracle.segment(oth)
bad = np.argwhere(labels[0] != ol)
print("label mismatches:", len(bad), "size mismatches:", np.count_nonzero(sizes[0] != osz))
for y, x in bad[:12]:
    g, o = int(labels[0][y, x]), int(ol[y, x])
    print(f"  px ({x},{y}) tile ({x//128},{y//32}) local ({x%128},{y%32}) v={oth[y,x]}  gpu={g} ({g%w},{g//w})  oracle={o} ({o%w},{o//w})")
if len(bad):
    ys, xs = bad[:, 0], bad[:, 1]
    print("rows mod 32 hist:", np.bincount(ys % 32, minlength=32).tolist())
    print("cols mod 128 hist (nonzero):", {int(k): int(v) for k, v in enumerate(np.bincount(xs % 128, minlength=128)) if v})
