"""Randomised parity stress of the whole detector against the CPU oracle (test infrastructure: imports oracle/): random
geometries, tag counts, noise levels, decimation, families and detector settings; detections must agree bit for bit (id,
hamming, margin, centre, corners) and so must the status word.  usage: python tests/stress_detect.py [cases] [seed]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
from chalkydri_amd import default_config, synth
from chalkydri_amd.detector import AprilTagDetector

def run(cases, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        dec = int(rng.choice([1, 1, 2]))
        scale = int(os.environ.get("STRESS_SCALE", "1"))   # 3: frames up to 2700 x 2100 (the largest fit class, two-colour merges)
        w = int(rng.integers(120, 900)) * scale; h = int(rng.integers(100, 700)) * scale
        if rng.random() < 0.5: w -= w % (4 * dec)   # half of the cases keep whatever width and height were drawn (odd ones included)
        n = int(rng.integers(1, 4))
        n_tags = int(rng.integers(0, 7))
        fams = ("tag36h11",) if rng.random() < 0.7 else ("tag16h5", "tag36h11")
        kw = {"noise_amp": int(rng.choice([0, 1, 3, 6]))}
        if min(w, h) < 300:
            kw.update(min_side=24, max_side=max(24, min(w, h) // 3))
        settings = {}
        if rng.random() < 0.3: settings["refine_edges"] = 0
        if rng.random() < 0.3: settings["max_nmaxima"] = int(rng.integers(4, 13))
        if rng.random() < 0.3: settings["min_component_px"] = int(rng.choice([5, 25, 60, 200]))
        if rng.random() < 0.2: settings["min_white_black_diff"] = int(rng.choice([2, 5, 12, 40]))
        if rng.random() < 0.2: settings["min_cluster_pixels"] = int(rng.choice([5, 24, 50, 120]))
        if rng.random() < 0.2: settings["max_line_fit_mse"] = float(rng.choice([1.0, 4.0, 10.0, 30.0]))
        if rng.random() < 0.2: settings["cos_critical_rad"] = float(rng.choice([0.5, 0.8, 0.984807753012208, 0.999]))
        if rng.random() < 0.2: settings["decode_sharpening"] = float(rng.choice([0.0, 0.25, 1.0]))
        if rng.random() < 0.2: settings["max_hamming"] = int(rng.choice([0, 1, 2, 3]))
        extra = int(rng.integers(0, 3))                       # the handle is sized for more frames than the call brings
        if os.environ.get("STRESS_CAPS"):                     # small capacities: overflow must be a status bit, the same one as the oracle's
            if rng.random() < 0.5: settings["max_points_per_frame"] = int(rng.choice([500, 5000, 50000]))
            if rng.random() < 0.5: settings["max_clusters_per_frame"] = int(rng.choice([1024, 2048]))
            if rng.random() < 0.5: settings["max_quads_per_frame"] = int(rng.choice([1, 4, 32]))
        if os.environ.get("STRESS_LOG"):
            with open(os.environ["STRESS_LOG"], "a") as lf:
                lf.write(json.dumps({"case": c, "w": w, "h": h, "n": n, "tags": n_tags, "fams": fams, "dec": dec, "kw": kw, "settings": settings}) + "\n")
        frames, _ = synth.render_batch(40 + c, n, w, h, n_tags, fams, **kw)
        bits = settings.pop("max_hamming", 3)
        det = AprilTagDetector(w, h, max_batch=n + extra, families=fams, quad_decimate=dec, bits_corrected=bits, **settings)
        got, status = det.detect_batch(frames, cap=256, return_status=True)   # the handle keeps at most 256 detections per frame (oracle: the same)
        cfg = default_config(w, h, families=fams, quad_decimate=dec, max_hamming=bits, **settings)
        for i in range(n):
            want, st = pyoracle.detect(frames[i], cfg)
            if (st & 15) or (status[i] & 15):
                # an overflow bit on either side: which points / clusters / quads were kept is the implementation's business, and so is
                # where exactly a capacity bites (the oracle counts every cluster against max_clusters_per_frame, the device the ones
                # it keeps, with twice as many table slots for the rest): nothing to compare, the call only has to come back
                ok = True
            else:
                ok = status[i] == st and len(got[i]) == len(want)
            if ok and not ((st | status[i]) & 15):   # with an overflow bit set, WHICH quads / detections were kept is not defined: only the flags and counts are

                for a, b in zip(got[i], want):
                    ok = ok and (a.id(), a.hamming(), a.family()) == (b["id"], b["hamming"], b["family"]) and \
                        np.float32(a.decision_margin()) == np.float32(b["margin"]) and np.array_equal(a.center(), b["c"]) and np.array_equal(a.corners(), b["p"])
            if not ok:
                bad += 1
                print(json.dumps({"case": c, "frame": i, "w": w, "h": h, "dec": dec, "got": len(got[i]), "want": len(want), "status": [int(status[i]), int(st)]}))
        det.close()
    print(json.dumps({"cases": cases, "mismatching_frames": bad}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
