"""Diagnostic: builds a -DCK_TILE_PROFILE copy of the library into /tmp and prints per-phase cycle shares of k_tile."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "chalkydri_amd", "csrc")
out = "/tmp/cktprof"; os.makedirs(out, exist_ok=True)
objs = []
for f in sorted(os.listdir(src)):
    if f.endswith(".hip"):
        o = os.path.join(out, f + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
                               "-DCK_TILE_PROFILE", "-DCK_DIAG", "-I" + os.path.join(ROOT, "include"), "-I" + src, "-c", os.path.join(src, f), "-o", o])
        objs.append(o)
    elif f.endswith(".c"):
        o = os.path.join(out, f + ".o")
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + src, "-c", os.path.join(src, f), "-o", o])
        objs.append(o)
lib = os.path.join(out, "libchalkydri_hip.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
from chalkydri_amd import _lib
_lib.LIB_PATH = lib
import numpy as np
from chalkydri_amd import synth
from chalkydri_amd.detector import AprilTagDetector
w, h, n = 1280, 800, 64
if len(sys.argv) > 1 and sys.argv[1] == "single":   # one tile alone on the chip: the length of a workgroup's own dependency chain
    w, h, n = 128, 64, 1
L = _lib.lib()
for noise in (3, 1):
    if n == 1:
        frames = (128 + np.random.default_rng(noise).integers(-noise, noise + 1, (1, h, w))).astype(np.uint8)
    else:
        base = np.stack([synth.render(synth.frame_seed(2, i), w, h, 6, noise_amp=noise)[0] for i in range(16)])
        frames = np.concatenate([base] * 4)
    det = AprilTagDetector(w, h, max_batch=n)
    det.upload(frames)
    buf = (C.c_ulonglong * 16)()
    buf2 = (C.c_ulonglong * 8)()
    det.time_threshold_segment(n, 2)
    L.ck_tile_profile_read(buf, 1); L.ck_tile_profile2_read(buf2, 1)
    ms = det.time_threshold_segment(n, 4)
    L.ck_tile_profile_read(buf, 1); L.ck_tile_profile2_read(buf2, 1)
    names = ["P0 load", "P1-2 minmax", "P3 thresh+masks", "P4-5 unions", "P6 flatten+sizes", "P7 labels out", "P8 roots append"]
    tot = sum(buf[k] for k in range(7)) or 1
    tiles = (130 if n > 1 else 1) * n * 5
    print("noise", noise, "ms/pass", round(ms, 3), "cycles/tile", round(tot / tiles), {names[k]: round(100 * buf[k] / tot, 1) for k in range(7)})
    wv = tiles * 4   # waves
    print("   P5b: unions/tile", round(buf[8] / tiles), "find2 iterations/union", round(buf[9] / max(1, buf[8]), 2), "max lane iterations per wave", round(buf[10] / wv, 1))
    print("   P6 : runs/tile", round(buf[11] / tiles), "hops/run", round(buf[12] / max(1, buf[11]), 2), "max lane walk iterations per wave", round(buf[13] / wv, 1))
    print("   P4-5 split (cycles of wave 0 per tile): init+link masks", round(buf2[0] / tiles), "adoption", round(buf2[1] / tiles), "barrier", round(buf2[2] / tiles),
          "atomic unions", round(buf2[3] / tiles), "barrier", round(buf2[4] / tiles))
    print("   P6 split (cycles of wave 0 per tile): run extraction", round(buf[14] / tiles), "lock-step walk", round(buf[15] / tiles), "stores + size adds", round(buf[7] / tiles))
    det.close()
