"""Diagnostic: per-phase clock shares of k_tile's first wave (thread 0's timeline, barrier waits included) from the -DCK_TILE_PROFILE
variant of the library (tools/build_variant.sh tileprof -DCK_TILE_PROFILE).  usage: prof_tile2.py [kind]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chalkydri_amd import _lib
_lib.LIB_PATH = os.environ.get("LIB") or os.path.join(ROOT, "chalkydri_amd", "lib", "ref", "libchalkydri_hip_tileprof.so")
import numpy as np
from chalkydri_amd import scenes
from chalkydri_amd.detector import AprilTagDetector
w, h, n = 1280, 800, 256
L = _lib.lib()
for noise in (3, 1):
    frames = scenes.bench_stream(2, n, w, h, 6, stream=0, unique=16, noise_amp=noise)[0]
    det = AprilTagDetector(w, h, max_batch=n)
    det.upload(frames)
    buf = (C.c_ulonglong * 16)()
    det.time_threshold_segment(n, 2)
    L.ck_tile_profile_read(buf, 1)
    iters = 4
    ms = det.time_threshold_segment(n, iters)
    L.ck_tile_profile_read(buf, 1)
    names = {0: "P0 load+stage", 1: "P1-2 minmax+dilate", 2: "P3 threshold+masks", 4: "P4 nodes+adopt+pool", 5: "P5b sweep", 6: "P5c pooled unions",
             7: "P6 flatten+sizes, P6b ring ids", 8: "ring out + label words + labels"}
    tiles = 250 * n * iters
    tot = sum(buf[k] for k in names) or 1
    print(f"noise {noise}: {ms:.3f} ms per batch; first-wave clocks per tile {tot / tiles:.0f}")
    for k, nm in names.items():
        print(f"  {nm:34s} {100.0 * buf[k] / tot:5.1f} %  {buf[k] / tiles:8.0f} clocks")
    print(f"  nodes per tile {buf[9] / tiles:.0f}, pooled links per tile {buf[10] / tiles:.0f}")
    det.close()
