"""Diagnostic: builds a -DCK_FIT_PROFILE copy of the library into /tmp and prints per-phase cycle shares of k_fit."""
import ctypes as C, os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "chalkydri_amd", "csrc")
out = "/tmp/ckprof"; os.makedirs(out, exist_ok=True)
objs = []
for f in sorted(os.listdir(src)):
    if f.endswith(".hip"):
        o = os.path.join(out, f + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
                               "-DCK_FIT_PROFILE", "-DCK_DIAG", "-I" + os.path.join(ROOT, "include"), "-I" + src, "-c", os.path.join(src, f), "-o", o])
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
base = np.stack([synth.render(synth.frame_seed(2, i), w, h, 6)[0] for i in range(16)])
frames = np.concatenate([base] * 4)
det = AprilTagDetector(w, h, max_batch=n)
det.upload(frames)
L = _lib.lib()
buf = (C.c_ulonglong * 48)()
det.detect_batch(None, n=n)
L.ck_fit_profile_read(buf, 1)
det.detect_batch(None, n=n)
L.ck_fit_profile_read(buf, 1)
names = ["bbox+keys", "sort", "dedupe", "weights", "chunks", "threshold", "survivors", "prefix@max", "pairs+combos", "final", "refine+out", "", "", "", "", "dequeue"]
for c, cn in enumerate(["S", "M", "L"]):
    row = [buf[c * 16 + k] for k in range(16)]
    tot = sum(row) or 1
    print(cn, "total Mcycles", round(tot / 1e6, 1), {names[k]: round(100 * row[k] / tot, 1) for k in range(16) if row[k]})
