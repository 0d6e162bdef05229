import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import pyoracle as O
from chalkydri_amd.detector import AprilTagDetector
w, h = 1280, 800
f = np.zeros((1, h, w), np.uint8)
for k in range(0, min(h, w) // 2 - 2, 4):
    f[:, k:h - k, k] = 255; f[:, k, k:w - k] = 255
    f[:, k + 2:h - k, w - 1 - k] = 255; f[:, h - 1 - k, k + 2:w - k] = 255
th = O.threshold(f[0]); ol, osz = O.segment(th)
det = AprilTagDetector(w, h, max_batch=1)
for it in range(12):
    lab, sz = det.segment(f)
    bad = np.argwhere(lab[0] != ol)
    if len(bad):
        for y, x in bad[:5]:
            print("iter", it, "pixel", (x, y), "tile", (x // 128, y // 64), "in-tile", (x % 128, y % 64), "thresh", th[y, x], "got", lab[0][y, x], divmod(int(lab[0][y, x]), w)[::-1], "want", ol[y, x], "size got/want", sz[0][y, x], osz[y, x], "n_bad", len(bad))
print("done")
