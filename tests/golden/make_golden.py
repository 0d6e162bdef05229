#!/usr/bin/env python3
"""Generates tests/golden/detector_golden.json: synthetic frames (by renderer seed + parameters, with a CRC of the rendered
bytes) and the detections the CPU oracle returns for them.  The reference holds no fixtures for this path, so
these vectors pin OUR oracle + renderer against regressions; ground truth from the renderer rides along.

    python tests/golden/make_golden.py        (rewrites the JSON; commit the result)
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

CASES = [
    {"name": "c1_640x480_4tags", "seed_cfg": 1, "frame": 0, "w": 640, "h": 480, "n_tags": 4, "families": ["tag36h11"], "bits": 3, "decimate": 1, "params": {}},
    {"name": "c1_640x480_clean", "seed_cfg": 1, "frame": 1, "w": 640, "h": 480, "n_tags": 4, "families": ["tag36h11"], "bits": 3, "decimate": 1, "params": {"noise_amp": 0, "ramp_amp": 0}},
    {"name": "mixed_families", "seed_cfg": 5, "frame": 0, "w": 640, "h": 480, "n_tags": 4, "families": ["tag16h5", "tag36h11"], "bits": 1, "decimate": 1, "params": {"family_mode": 1}},
    {"name": "decimate2", "seed_cfg": 1, "frame": 2, "w": 640, "h": 480, "n_tags": 3, "families": ["tag36h11"], "bits": 3, "decimate": 2, "params": {"min_side": 64}},
    {"name": "c2_1280x800_6tags", "seed_cfg": 2, "frame": 0, "w": 1280, "h": 800, "n_tags": 6, "families": ["tag36h11"], "bits": 3, "decimate": 1, "params": {}},
    {"name": "ragged_tiles_272x200", "seed_cfg": 9, "frame": 0, "w": 272, "h": 200, "n_tags": 2, "families": ["tag36h11"], "bits": 3, "decimate": 1, "params": {"min_side": 28, "max_side": 70}},
]


def run_case(c):
    import pyoracle
    from chalkydri_amd import default_config, synth
    frame, truth = synth.render(synth.frame_seed(c["seed_cfg"], c["frame"]), c["w"], c["h"], c["n_tags"], tuple(c["families"]), **c["params"])
    cfg = default_config(c["w"], c["h"], families=tuple(c["families"]), max_hamming=c["bits"], quad_decimate=c["decimate"])
    dets, status = pyoracle.detect(frame, cfg)
    return {
        "case": c, "frame_crc32": zlib.crc32(frame.tobytes()), "status": status,
        "truth": [{"family": t["family"], "id": t["id"], "corners": t["corners"].tolist()} for t in truth],
        "detections": [{"family": d["family"], "id": d["id"], "hamming": d["hamming"], "margin": float(np.float32(d["margin"])),
                        "center": [float(v).hex() for v in d["c"]], "corners": [[float(v).hex() for v in p] for p in d["p"]]} for d in dets],
    }


def hexlist(a):
    return [float(v).hex() for v in np.asarray(a, np.float64).reshape(-1)]


def sqpnp_cases():
    """Solver problems (inputs as hex doubles) and what the oracle returns for them: random scenes, noisy scenes, and tags on
    one axis-aligned wall (the rank-deficient eigen-guess case)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import np_sqpnp as N
    import pyoracle
    from test_sqpnp_oracle import _wall_scene
    out = []
    rng = np.random.default_rng(2026)
    for k in range(12):
        if k < 6:
            tags, b, rtc, truth = N.make_scene(rng, int(rng.integers(1, 9)), noise_px=0.3 if k % 2 else 0.0)
        else:
            tags, b, rtc, truth = _wall_scene(rng, int(rng.integers(2, 9)), noise_px=0.2 if k % 2 else 0.0)
        gyro = float(truth["yaw"] + rng.uniform(-0.3, 0.3))
        r = pyoracle.sqpnp_solve(tags, b, rtc, gyro)
        out.append({"tags": [{"R": hexlist(R), "t": hexlist(t)} for R, t in tags], "bearings": hexlist(b), "rtc": {"R": hexlist(rtc[0]), "t": hexlist(rtc[1])},
                    "gyro": gyro.hex(), "valid": r is not None,
                    "result": None if r is None else {"rot": hexlist(r["rot"]), "pos": hexlist(r["pos"]), "std": hexlist(r["std"]), "yaw": float(r["yaw"]).hex(),
                                                      "energy": float(r["energy"]).hex()}})
    return out


def cat_cases():
    """CAT front-end on seeded RGB frames: CRC32 of every output of the oracle (all integer, so the pins are exact)."""
    import pyoracle
    from chalkydri_amd import synth
    out = []
    for seed, (w, h) in ((3, (160, 120)), (8, (320, 240))):
        g = synth.render(synth.frame_seed(5, seed), w, h, 3, min_side=40, max_side=min(150, h // 2), noise_amp=2)[0]
        rng = np.random.default_rng(seed)
        rgb = np.clip(np.stack([g, g, g], -1).astype(np.int16) + rng.integers(-3, 4, (h, w, 3)), 0, 255).astype(np.uint8)
        cls = pyoracle.cat_calc_otsu(rgb)
        pts, npn = pyoracle.cat_detect_corners(cls)
        lines, nl = pyoracle.cat_check_edges(cls, pts)
        roots, sizes = pyoracle.cat_connected_components(cls)
        crc = lambda a, dt: zlib.crc32(np.ascontiguousarray(a, dt).tobytes())
        out.append({"seed": seed, "w": w, "h": h, "rgb_crc32": crc(rgb, np.uint8), "thresh_crc32": crc(pyoracle.cat_thresh(rgb), np.uint8),
                    "classes_crc32": crc(cls, np.uint8), "n_points": int(npn), "points_crc32": crc(pts, np.uint32), "n_lines": int(nl),
                    "lines_crc32": crc(lines, np.uint32), "roots_crc32": crc(roots, np.uint32), "sizes_crc32": crc(sizes, np.uint32)})
    return out


def main():
    out = [run_case(c) for c in CASES]
    json.dump(out, open(os.path.join(HERE, "detector_golden.json"), "w"), indent=1)
    for o in out:
        print(o["case"]["name"], "detections", [d["id"] for d in o["detections"]], "truth", [t["id"] for t in o["truth"]])
    sq = sqpnp_cases()
    json.dump(sq, open(os.path.join(HERE, "sqpnp_golden.json"), "w"), indent=1)
    print("sqpnp cases", len(sq), "valid", sum(c["valid"] for c in sq))
    ct = cat_cases()
    json.dump(ct, open(os.path.join(HERE, "cat_golden.json"), "w"), indent=1)
    print("cat cases", [(c["n_points"], c["n_lines"]) for c in ct])


if __name__ == "__main__":
    main()
