"""Randomised parity stress of the CAT path (crates/chalkydri-apriltags Detector: thresh, calc_otsu, detect_corners, check_edges,
connected_components) against the CPU oracle (test infrastructure: imports oracle/): random geometries and content mixes, every
output compared bit for bit.  usage: python tests/stress_cat.py [cases] [seed]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("tcat", os.path.join(ROOT, "tests", "test_gpu_cat.py"))
tcat = importlib.util.module_from_spec(spec); spec.loader.exec_module(tcat)
import pyoracle
from chalkydri_amd.cat import CatDetector

def run(cases, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        w = int(rng.integers(16, 400)); h = int(rng.integers(16, 300))
        ka, kb = rng.choice(["tags", "noise", "flat", "checker"], 2)
        if h < 100 or w < 100:
            ka = "noise" if ka == "tags" else ka; kb = "flat" if kb == "tags" else kb
        a, b = tcat._rgb(int(rng.integers(1, 999)), w, h, ka), tcat._rgb(int(rng.integers(1, 999)), w, h, kb)
        rgb = a.copy()
        x0, y0 = int(rng.integers(0, w // 2)), int(rng.integers(0, h // 2))
        rgb[y0:y0 + h // 2, x0:x0 + w // 2] = b[y0:y0 + h // 2, x0:x0 + w // 2]
        if os.environ.get("STRESS_LOG"):
            with open(os.environ["STRESS_LOG"], "a") as lf:
                lf.write(json.dumps({"case": c, "w": w, "h": h, "kinds": [str(ka), str(kb)]}) + "\n")
        det = CatDetector(w, h)
        why = None
        if not np.array_equal(det.thresh(rgb), pyoracle.cat_thresh(rgb)): why = "thresh"
        cls = det.calc_otsu(rgb).copy()
        ocls = pyoracle.cat_calc_otsu(rgb)
        if why is None and not np.array_equal(cls, ocls): why = "calc_otsu"
        pts = det.detect_corners()
        opts, n = pyoracle.cat_detect_corners(ocls)
        if why is None and not (n == len(opts) and np.array_equal(pts, opts)): why = "detect_corners"
        if why is None and len(opts) <= 400:
            lines = det.check_edges()
            olines, nl = pyoracle.cat_check_edges(ocls, opts)
            if not (nl == len(olines) and np.array_equal(lines, olines)): why = "check_edges"
        uf = det.connected_components()
        roots, sizes = pyoracle.cat_connected_components(ocls)
        if why is None and not (np.array_equal(uf._roots.reshape(h, w), roots) and np.array_equal(uf._sizes.reshape(h, w), sizes)): why = "connected_components"
        if why:
            bad += 1
            print(json.dumps({"case": c, "w": w, "h": h, "kinds": [str(ka), str(kb)], "first_difference": why}))
        det.close()
    print(json.dumps({"cases": cases, "mismatching_cases": bad}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
