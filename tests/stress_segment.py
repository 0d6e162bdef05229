"""Randomised parity stress of threshold + segmentation against the CPU oracle (test infrastructure: imports oracle/): random
geometries (ragged widths / heights included), random content mixes, random min_white_black_diff / min_component_px and, through
CK_FMERGE_CAP and CK_FMERGE_BAND_ROWS, every path of the merge kernels (one piece or bands of tile rows).  usage: python tests/stress_segment.py [cases] [seed]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if __name__ == "__main__":   # CK_FMERGE_CAP is a knob of the diagnostics build: as a script this file runs against that library
    os.environ.setdefault("CHALKYDRI_HIP_LIB", os.path.join(ROOT, "chalkydri_amd", "lib", "diag", "libchalkydri_hip.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("tseg", os.path.join(ROOT, "tests", "test_gpu_segment.py"))
tseg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tseg)
import pyoracle
from chalkydri_amd.detector import AprilTagDetector

def run(cases, seed):
    rng = np.random.default_rng(seed)
    cap_before = os.environ.get("CK_FMERGE_CAP")
    kinds = ["synth", "noise", "flat", "stripes", "blobs", "spiral", "checker1", "vstripes1"]
    bad = 0
    for c in range(cases):
        w = int(rng.integers(40, 700)); h = int(rng.integers(40, 500))
        if rng.random() < 0.3: w = (w // 4) * 4
        n = int(rng.integers(1, 4))
        ka, kb = rng.choice(kinds, 2)
        fa, fb = tseg._frames(ka, w, h, n, int(rng.integers(1, 1000))), tseg._frames(kb, w, h, n, int(rng.integers(1, 1000)))
        frames = fa.copy()
        x0, y0 = int(rng.integers(0, w // 2)), int(rng.integers(0, h // 2))
        frames[:, y0:y0 + h // 2, x0:x0 + w // 2] = fb[:, y0:y0 + h // 2, x0:x0 + w // 2]   # two kinds of content side by side
        cap = int(rng.choice([16, 64, 300, 1200, 100000]))
        os.environ["CK_FMERGE_CAP"] = str(cap)
        rows = int(rng.choice([0, 0, 1, 2, 3, 5]))   # (0: the library's own rule; else frames joined in bands of that many tile rows)
        os.environ["CK_FMERGE_BAND_ROWS"] = str(rows)
        if os.environ.get("STRESS_LOG"):
            with open(os.environ["STRESS_LOG"], "a") as lf:
                lf.write(json.dumps({"case": c, "w": w, "h": h, "n": n, "kinds": [str(ka), str(kb)], "cap": cap, "band_rows": rows, "x0": x0, "y0": y0}) + "\n")
        det = AprilTagDetector(w, h, max_batch=n)
        th = det.threshold(frames)
        labels, sizes = det.segment(frames)
        for i in range(n):
            oth = pyoracle.threshold(frames[i])
            ol, osz = pyoracle.segment(oth)
            ok = np.array_equal(th[i], oth) and np.array_equal(labels[i], ol) and np.array_equal(sizes[i], osz)
            if not ok:
                bad += 1
                print(json.dumps({"case": c, "w": w, "h": h, "kinds": [str(ka), str(kb)], "cap": cap, "band_rows": rows, "frame": i,
                                  "thr_diff": int(np.count_nonzero(th[i] != oth)), "label_diff": int(np.count_nonzero(labels[i] != ol))}))
        det.close()
    print(json.dumps({"cases": cases, "mismatching_frames": bad}))
    os.environ.pop("CK_FMERGE_BAND_ROWS", None)
    if cap_before is None:
        os.environ.pop("CK_FMERGE_CAP", None)
    else:
        os.environ["CK_FMERGE_CAP"] = cap_before
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
