"""Replays one case of tests/stress_segment.py (the frames are rebuilt from the case's seeds): python tests/dbg_segment_case.py <cases_seed> <case_index> [threshold|segment]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CHALKYDRI_HIP_LIB", os.path.join(ROOT, "chalkydri_amd", "lib", "diag", "libchalkydri_hip.so"))  # (CK_FMERGE_CAP: diagnostics build)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stress_segment as S
seed, want = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else "both"
rng = np.random.default_rng(seed)
kinds = ["synth", "noise", "flat", "stripes", "blobs", "spiral", "checker1", "vstripes1"]
for c in range(want + 1):
    w = int(rng.integers(40, 700)); h = int(rng.integers(40, 500))
    if rng.random() < 0.3: w = (w // 4) * 4
    n = int(rng.integers(1, 4))
    ka, kb = rng.choice(kinds, 2)
    sa, sb = int(rng.integers(1, 1000)), int(rng.integers(1, 1000))
    x0, y0 = int(rng.integers(0, w // 2)), int(rng.integers(0, h // 2))
    cap = int(rng.choice([16, 64, 300, 1200, 100000]))
fa, fb = S.tseg._frames(ka, w, h, n, sa), S.tseg._frames(kb, w, h, n, sb)
frames = fa.copy(); frames[:, y0:y0 + h // 2, x0:x0 + w // 2] = fb[:, y0:y0 + h // 2, x0:x0 + w // 2]
cap = int(os.environ.get("DBG_CAP", cap))
os.environ["CK_FMERGE_CAP"] = str(cap)
print(json.dumps({"w": w, "h": h, "n": n, "kinds": [str(ka), str(kb)], "cap": cap, "what": what}), flush=True)
det = S.AprilTagDetector(w, h, max_batch=n)
if what in ("threshold", "both"):
    th = det.threshold(frames); print("threshold ok", flush=True)
if what in ("segment", "both"):
    labels, sizes = det.segment(frames); print("segment ok", flush=True)
    ok = all(np.array_equal(labels[i], S.pyoracle.segment(S.pyoracle.threshold(frames[i]))[0]) for i in range(n))
    print("labels equal the oracle's:", ok, flush=True)
det.close()
