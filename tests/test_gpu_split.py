"""The post-segmentation stages can run as consecutive pieces of the batch on two streams (ck_stages.hip: run_pipeline,
CK_STREAMS=2).  Frames are independent, so every split — none (the default), two halves, three uneven pieces — must give
the same bytes."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import scenes
from chalkydri_amd.apriltags import AprilTags
import os
w, h, f, n = 640, 480, 600.0, int(os.environ.get("SPLIT_N", "7"))
layout = scenes.wall_layout(6, cols=3)
r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0)
rng = np.random.default_rng(5)
frames, gyros = [], []
for i in range(n):
    pose = (rng.uniform(1.8, 2.4), rng.uniform(-0.2, 0.2), rng.uniform(-0.1, 0.1))
    frames.append(scenes.render_view(900 + i, w, h, f, layout, pose, r2c, noise_amp=3)[0]); gyros.append(pose[2])
if n > 4: gyros[4] = None
task = AprilTags(w, h, layout, calib, r2c, cam_id=1, max_batch=n)
recs, valid = task.process_batch(np.stack(frames), gyros)
dets, status = task.detector.detect_batch(np.stack(frames), cap=32, return_status=True)
hh = hashlib.sha256()
for r in recs: hh.update(bytes(r))
hh.update(np.asarray(valid, np.int32).tobytes()); hh.update(np.asarray(status, np.uint32).tobytes())
for fr in dets:
    for d in fr:
        hh.update(np.asarray([d.id(), d.hamming()], np.int64).tobytes()); hh.update(np.asarray(d.corners(), np.float64).tobytes())
print("HASH", hh.hexdigest(), int(np.sum(valid)))
"""


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    if "CK_PARTS" in env_extra:   # the number of pieces is a knob of the diagnostics build (the product library cuts in two)
        from conftest import DIAG_LIB
        env["CHALKYDRI_HIP_LIB"] = DIAG_LIB
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0].split()
    return line[1], int(line[2])


def test_results_do_not_depend_on_the_split(built):
    single, nvalid = _run({"CK_STREAMS": "1"})
    assert nvalid == 6                       # 7 frames, one without gyro
    assert _run({})[0] == single                                        # default
    assert _run({"CK_STREAMS": "2"})[0] == single                       # two halves on two streams
    assert _run({"CK_STREAMS": "2", "CK_PARTS": "3"})[0] == single      # three uneven pieces


def test_small_calls_do_not_depend_on_the_split_either(built):
    """Three frames: the quad fit's size classes run side by side on their own streams (calls with at most 4 frames), with and
    without the two-stream split of the batch on top."""
    single, nvalid = _run({"CK_STREAMS": "1", "SPLIT_N": "3"})
    assert nvalid == 3
    assert _run({"CK_STREAMS": "2", "SPLIT_N": "3"})[0] == single
    assert _run({"CK_STREAMS": "2", "CK_PARTS": "3", "SPLIT_N": "3"})[0] == single
