"""Readers for the committed golden vectors (tests/golden/*.json, written by tests/golden/make_golden.py)."""
import json
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


def unhex(lst, shape=None):
    a = np.array([float.fromhex(v) for v in lst], np.float64)
    return a if shape is None else a.reshape(shape)


def sqpnp_problem(c):
    tags = [(unhex(t["R"], (3, 3)), unhex(t["t"])) for t in c["tags"]]
    b = unhex(c["bearings"], (-1, 3))
    rtc = (unhex(c["rtc"]["R"], (3, 3)), unhex(c["rtc"]["t"]))
    return tags, b, rtc, float.fromhex(c["gyro"])


def sqpnp_result(c):
    r = c["result"]
    return None if r is None else {"rot": unhex(r["rot"], (3, 3)), "pos": unhex(r["pos"]), "std": unhex(r["std"]), "yaw": float.fromhex(r["yaw"]),
                                   "energy": float.fromhex(r["energy"])}


def cat_rgb(c):
    from chalkydri_amd import synth
    g = synth.render(synth.frame_seed(5, c["seed"]), c["w"], c["h"], 3, min_side=40, max_side=min(150, c["h"] // 2), noise_amp=2)[0]
    rng = np.random.default_rng(c["seed"])
    rgb = np.clip(np.stack([g, g, g], -1).astype(np.int16) + rng.integers(-3, 4, (c["h"], c["w"], 3)), 0, 255).astype(np.uint8)
    assert zlib.crc32(rgb.tobytes()) == c["rgb_crc32"], "renderer output changed"
    return rgb


def crc(a, dt):
    return zlib.crc32(np.ascontiguousarray(a, dt).tobytes())
