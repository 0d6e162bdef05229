"""debug: the split quad fit on one small frame: exit counters of k_tail (build with -DCK_FLAT_DEBUG) and the quads against the unsplit path"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chalkydri_amd import synth
from chalkydri_amd.detector import AprilTagDetector
from chalkydri_amd import _lib
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frames, _ = synth.render_batch(12, n, w, h, 6, ("tag36h11",), noise_amp=int(sys.argv[4]) if len(sys.argv) > 4 else 3)
det = AprilTagDetector(w, h, max_batch=n)
got = det.quads(frames)
print("quads per frame", [len(g) for g in got])
res = det.clusters(frames)
print("clusters per frame", [len(c) for c, _ in res], "points per frame", [len(p) for _, p in res])
lib = _lib.lib()
if hasattr(lib, "ck_flat_debug_read"):
    out = (ctypes.c_uint * 32)()
    lib.ck_flat_debug_read(out, 1)
    print("dbg", list(out))
