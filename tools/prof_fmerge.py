"""Diagnostic: per-phase shader-clock shares of k_fmerge's first wave, from a -DCK_FM_PROFILE build of the library
(chalkydri_amd/lib/ref/libchalkydri_hip_fmprof.so: every .hip compiled with -DCK_FM_PROFILE).  usage: prof_fmerge.py w h n"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chalkydri_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "chalkydri_amd", "lib", "ref", "libchalkydri_hip_fmprof.so")
import numpy as np
from chalkydri_amd import synth
from chalkydri_amd.detector import AprilTagDetector
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = _lib.lib()
base = np.stack([synth.render(synth.frame_seed(2, i), w, h, 6)[0] for i in range(min(n, 8))])
frames = np.concatenate([base] * ((n + 7) // 8))[:n]
det = AprilTagDetector(w, h, max_batch=n)
det.upload(frames)
buf = (C.c_ulonglong * 16)()
det.time_threshold_segment(n, 2)
L.ck_fm_profile_read(buf, 1)
ms = det.time_threshold_segment(n, 5)
L.ck_fm_profile_read(buf, 1)
names = ["edge lists", "scan..pack (after edges)", "init", "sweep", "drain + barrier", "flatten", "sizes", "tables"]
tot = sum(buf[:8])
print(f"{w}x{h} x {n}: threshold+segment {ms:.3f} ms per batch; k_fmerge first-wave clock shares:")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {100.0 * buf[k] / max(tot, 1):5.1f} %   {buf[k] / (5 * 2 * n) / 1e3:9.1f} k cycles per workgroup")
