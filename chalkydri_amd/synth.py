"""Synthetic frame generator (deterministic renderer in csrc/synth.c) — inputs for tests and bench.py."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import family, lib

BASE_SEED = 0xC4A1D1


def frame_seed(config_idx, frame_idx, stream=0):
    return BASE_SEED + 1000 * config_idx + frame_idx + stream * 10**6


def render(seed, width, height, n_tags, families=("tag36h11",), stride=None, **params):
    """Returns (frame uint8 [H][stride], list of truth dicts)."""
    L = lib()
    sp = A.SynthParams()
    L.ck_synth_params_default(C.byref(sp), width, height, n_tags)
    for k, v in params.items():
        if not hasattr(sp, k):
            raise AttributeError(k)
        setattr(sp, k, v)
    stride = stride or width
    fams = (C.POINTER(A.Family) * len(families))(*[family(f) for f in families])
    out = np.zeros((height, stride), dtype=np.uint8)
    truth = (A.SynthTag * max(n_tags, 1))()
    n = C.c_int32(0)
    rc = L.ck_synth_render(seed, C.byref(sp), fams, len(families), out.ctypes.data, stride, truth,
                           max(n_tags, 1), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"ck_synth_render failed: {rc}")
    tags = []
    for i in range(n.value):
        t = truth[i]
        tags.append({"family": t.family, "id": t.id, "H": np.array(t.H[:]).reshape(3, 3),
                     "corners": np.array([[t.corners[k][0], t.corners[k][1]] for k in range(4)]),
                     "center": np.array(t.center[:])})
    return out, tags


def render_batch(config_idx, n_frames, width, height, n_tags, families=("tag36h11",), stream=0, first=0,
                 **params):
    frames = np.empty((n_frames, height, width), dtype=np.uint8)
    truths = []
    for i in range(n_frames):
        f, t = render(frame_seed(config_idx, first + i, stream), width, height, n_tags, families, **params)
        frames[i] = f
        truths.append(t)
    return frames, truths


def render_scene(seed, width, height, tags, families=("tag36h11",), **params):
    """tags: list of (family_idx, id, H 3x3) — draws them over the standard background."""
    L = lib()
    sp = A.SynthParams()
    L.ck_synth_params_default(C.byref(sp), width, height, len(tags))
    for k, v in params.items():
        setattr(sp, k, v)
    out = np.zeros((height, width), dtype=np.uint8)
    L.ck_synth_background(seed, C.byref(sp), out.ctypes.data, width)
    truth = []
    for fi, tid, H in tags:
        t = A.SynthTag()
        t.family, t.id = fi, tid
        for k, v in enumerate(np.asarray(H, dtype=np.float64).reshape(9)):
            t.H[k] = v
        L.ck_synth_fill_truth(C.byref(t))
        rc = L.ck_synth_draw_tag(C.byref(sp), family(families[fi]), C.byref(t), out.ctypes.data, width)
        if rc != 0:
            raise RuntimeError("ck_synth_draw_tag failed")
        truth.append({"family": fi, "id": tid,
                      "corners": np.array([[t.corners[k][0], t.corners[k][1]] for k in range(4)]),
                      "center": np.array(t.center[:])})
    return out, truth
