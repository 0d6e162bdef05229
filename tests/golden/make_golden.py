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


def main():
    out = [run_case(c) for c in CASES]
    json.dump(out, open(os.path.join(HERE, "detector_golden.json"), "w"), indent=1)
    for o in out:
        print(o["case"]["name"], "detections", [d["id"] for d in o["detections"]], "truth", [t["id"] for t in o["truth"]])


if __name__ == "__main__":
    main()
