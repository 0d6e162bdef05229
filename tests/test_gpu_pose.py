"""GPU parity for SQPnP and the AprilTags::process glue: HIP vs oracle within the stated float tolerance
(|dR|,|dt| <= 1e-9 for the solver; end-to-end pose: position 1e-6 m, yaw 1e-7 rad) and vs synthetic ground truth."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import np_sqpnp as N
import scenes
from chalkydri_amd import _abi as A
from chalkydri_amd import default_config

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _iso(R, t):
    from chalkydri_amd.sqpnp import iso3
    return iso3(t, N.mat_to_quat(np.asarray(R)))


def test_sqpnp_batch_matches_oracle(oracle):
    from chalkydri_amd.detector import AprilTagDetector
    from chalkydri_amd.sqpnp import SqPnP
    det = AprilTagDetector(64, 64)
    solver = SqPnP(det)
    rng = np.random.default_rng(7)
    probs, want = [], []
    for k in range(64):
        n_tags = int(rng.integers(1, 31))
        tags, b, rtc, truth = N.make_scene(rng, n_tags, noise_px=0.25 if k % 2 else 0.0)
        gyro = truth["yaw"] + rng.uniform(-0.6, 0.6)
        probs.append(([_iso(R, t) for R, t in tags], b, _iso(*rtc), gyro, 600.0))
        want.append(oracle.sqpnp_solve(tags, b, rtc, gyro))
    got = solver.solve_batch(probs)
    n_valid = 0
    for g, w in zip(got, want):
        assert (g is None) == (w is None)
        if g is None:
            continue
        n_valid += 1
        assert np.abs(g["rot"] - w["rot"]).max() < TOL and np.abs(g["pos"] - w["pos"]).max() < TOL
        assert abs(g["yaw"] - w["yaw"]) < TOL
        if w["energy"] > 1e-13:
            assert np.allclose(g["std_devs"], w["std"], rtol=1e-9, atol=0)
        R = g["rot"]
        assert np.abs(R.T @ R - np.eye(3)).max() < 1e-9 and abs(np.linalg.det(R) - 1) < 1e-9
    assert n_valid >= 60
    # guards (lib.rs:255)
    tags, b, rtc, truth = N.make_scene(rng, 2)
    bad = solver.solve_batch([([_iso(R, t) for R, t in tags], b[:-1], _iso(*rtc), 0.0, 600.0), ([], np.zeros((0, 3)), _iso(*rtc), 0.0, 600.0)])
    assert bad == [None, None]
    det.close()


def test_builders_and_transform(oracle):
    from chalkydri_amd.sqpnp import SqPnP
    s = SqPnP().max_iter(3).tolerance(1e-4)
    assert s._prm.max_iter == 3 and abs(s._prm.tol_sq - 1e-8) < 1e-20
    for args in [(0, 0, 0, 0, 0, 0), (0.3, -0.2, 0.5, 3.0, -10.0, 25.0), (0, 0, 0, 0, 0, 180.0)]:
        iso = SqPnP.create_solver_camera_transform(*args)
        t, q = oracle.create_solver_camera_transform(*args)
        assert np.abs(np.array(iso.t[:]) - t).max() < 1e-15 and np.abs(np.array(iso.q[:]) - q).max() < 1e-15


def test_unproject_matches_oracle(oracle):
    from chalkydri_amd.sqpnp import unproject_opencv5
    m = scenes.REF_CALIB["OpenCVModel5"]
    cam = tuple(m[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3"))
    px = np.random.default_rng(0).uniform(0, 1600, (500, 2))
    b, ok = unproject_opencv5(cam, px)
    ob, ook = oracle.unproject_opencv5(cam, px)
    assert np.array_equal(ok, ook) and np.abs(b - ob).max() < 1e-14


@pytest.mark.parametrize("calib_kind", ["pinhole", "reference"])
def test_process_end_to_end(oracle, calib_kind):
    """AprilTags::process on rendered 3-D scenes: device records == oracle records (tolerance) and ~ ground truth."""
    from chalkydri_amd.apriltags import AprilTags
    w, h, f = 1280, 800, 1100.0
    layout = scenes.wall_layout(12)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
    calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0) if calib_kind == "pinhole" else scenes.REF_CALIB
    rng = np.random.default_rng(3)
    frames, truths, gyros = [], [], []
    for i in range(4):
        pose = (rng.uniform(1.0, 2.5), rng.uniform(-0.6, 0.6), rng.uniform(-0.25, 0.25))
        fr, tr = scenes.render_view(1000 + i, w, h, f, layout, pose, r2c, noise_amp=1)
        frames.append(fr); truths.append(tr); gyros.append(pose[2] + rng.uniform(-0.02, 0.02))
    frames = np.stack(frames)
    gyros[3] = None   # "no gyro, no solve" gate (apriltags/src/lib.rs:330)
    task = AprilTags(w, h, layout, calib, r2c, cam_id=7, max_batch=4)
    recs, valid = task.process_batch(frames, gyros)
    cfg = default_config(w, h)
    for i in range(4):
        out = A.VisionMeasurement()
        v = C.c_int(0)
        oracle.lib().ora_process_frame(C.c_void_p(frames[i].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                       C.c_double(gyros[i] or 0.0), 0 if gyros[i] is None else 1, C.byref(out), C.byref(v))
        assert bool(v.value) == bool(valid[i])
        r = recs[i]
        assert (r.camera_id, r.tag_count) == (out.camera_id, out.tag_count)
        if not valid[i]:
            assert bytes(r) == bytes(out)  # the empty heartbeat record (lib.rs:365-376)
            continue
        assert abs(r.pose_x - out.pose_x) < 1e-6 and abs(r.pose_y - out.pose_y) < 1e-6 and abs(r.pose_rot - out.pose_rot) < 1e-7
        assert np.allclose([r.std_x, r.std_y, r.std_rot], [out.std_x, out.std_y, out.std_rot], rtol=1e-6)
        assert r.tag_count >= 6
        if calib_kind == "pinhole":   # the renderer is distortion-free, so only this calibration can recover truth
            assert abs(r.pose_x - truths[i]["twr"][0]) < 0.03 and abs(r.pose_y - truths[i]["twr"][1]) < 0.03
            assert abs((r.pose_rot - truths[i]["yaw"] + np.pi) % (2 * np.pi) - np.pi) < 0.03
    assert valid[:3].all() and not valid[3]
    task.detector.close()


def test_bench_stream_validity_matches_oracle(oracle):
    """The bench workload itself (one axis-aligned wall, 6 tags, noise +-3): every frame's record agrees with the
    oracle's AprilTags::process, and every frame yields a pose.  Regression for the rank-1 nearest_so3 guesses."""
    from chalkydri_amd.apriltags import AprilTags
    w, h, n = 1280, 800, 12
    frames, gyro, layout, calib, r2c = scenes.bench_stream(2, n, w, h, 6, stream=0, unique=n, noise_amp=3)
    task = AprilTags(w, h, layout, calib, r2c, cam_id=0, max_batch=n)
    recs, valid = task.process_batch(frames, list(gyro))
    cfg = default_config(w, h)
    for i in range(n):
        out = A.VisionMeasurement()
        v = C.c_int(0)
        oracle.lib().ora_process_frame(C.c_void_p(frames[i].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                       C.c_double(float(gyro[i])), 1, C.byref(out), C.byref(v))
        r = recs[i]
        assert bool(v.value) and bool(valid[i]), i
        assert r.tag_count == out.tag_count == 6
        assert abs(r.pose_x - out.pose_x) < 1e-6 and abs(r.pose_y - out.pose_y) < 1e-6 and abs(r.pose_rot - out.pose_rot) < 1e-7
    task.detector.close()


def test_dense_1080p_full_pipeline(oracle):
    """BASELINE config 3: 1920x1080, 30 tags per frame, detect + SQPnP: records equal the oracle's and recover the pose."""
    from chalkydri_amd.apriltags import AprilTags
    w, h, f, n = 1920, 1080, 1000.0, 2
    layout = scenes.wall_layout(30, cols=10)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
    calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0)
    rng = np.random.default_rng(30)
    frames, truths, gyros = [], [], []
    for i in range(n):
        pose = (rng.uniform(0.9, 1.3), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05))
        fr, tr = scenes.render_view(3000 + i, w, h, f, layout, pose, r2c, noise_amp=3)
        frames.append(fr); truths.append(tr); gyros.append(pose[2] + rng.uniform(-0.02, 0.02))
    frames = np.stack(frames)
    task = AprilTags(w, h, layout, calib, r2c, cam_id=3, max_batch=n)
    recs, valid = task.process_batch(frames, gyros)
    cfg = default_config(w, h)
    for i in range(n):
        assert len(truths[i]["tags"]) == 30
        out = A.VisionMeasurement()
        v = C.c_int(0)
        oracle.lib().ora_process_frame(C.c_void_p(frames[i].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                       C.c_double(gyros[i]), 1, C.byref(out), C.byref(v))
        r = recs[i]
        assert bool(v.value) and bool(valid[i])
        assert r.tag_count == out.tag_count == 30
        assert abs(r.pose_x - out.pose_x) < 1e-6 and abs(r.pose_y - out.pose_y) < 1e-6 and abs(r.pose_rot - out.pose_rot) < 1e-7
        assert abs(r.pose_x - truths[i]["twr"][0]) < 0.03 and abs(r.pose_y - truths[i]["twr"][1]) < 0.03
        assert abs((r.pose_rot - truths[i]["yaw"] + np.pi) % (2 * np.pi) - np.pi) < 0.03
    task.detector.close()


def test_config3_full_size_batch_properties(oracle):
    """BASELINE config 3 at full size: 1920x1080 x 512 frames, 30 tags per frame, the whole pipeline incl. SQPnP (a 50 GB handle,
    frame indices above 255 at 1080p, two merge workgroups per frame, every fit class at batch scale).  The oracle cannot run 512
    such frames in seconds, so the batch is 8 distinct frames in shuffled positions: every copy of a frame gives the same record
    wherever it sits, no frame reports an overflow, and the 8 distinct records equal the oracle's (and recover the rendered pose)."""
    from chalkydri_amd.apriltags import AprilTags
    w, h, f, n, uniq = 1920, 1080, 1000.0, 512, 8
    layout = scenes.wall_layout(30, cols=10)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
    calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0)
    rng = np.random.default_rng(33)
    frames8, truths, gyro8 = [], [], []
    for i in range(uniq):
        pose = (rng.uniform(0.9, 1.3), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05))
        fr, tr = scenes.render_view(3300 + i, w, h, f, layout, pose, r2c, noise_amp=3)
        frames8.append(fr); truths.append(tr); gyro8.append(pose[2] + rng.uniform(-0.02, 0.02))
    frames8 = np.stack(frames8)
    which = rng.permutation(np.repeat(np.arange(uniq), n // uniq))
    task = AprilTags(w, h, layout, calib, r2c, cam_id=3, max_batch=n)
    recs, valid = task.process_batch(frames8[which], [float(gyro8[k]) for k in which])
    assert valid.all()
    dets, status = task.detector.detect_batch(None, n=n, return_status=True)   # the frames are still staged: their status words
    status = np.asarray(status, np.uint32)
    assert not np.any(status & np.uint32(~A.CK_FRAME_UNVERIFIED_ID & 0xFFFFFFFF)), "a frame reported an overflow"
    first = {}
    for i in range(n):
        k = int(which[i])
        if k in first:
            assert bytes(recs[i]) == first[k], f"frame copy {i} of {k} differs"
        else:
            first[k] = bytes(recs[i])
    cfg = default_config(w, h)
    for k in range(uniq):
        out = A.VisionMeasurement()
        v = C.c_int(0)
        oracle.lib().ora_process_frame(C.c_void_p(frames8[k].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                       C.c_double(float(gyro8[k])), 1, C.byref(out), C.byref(v))
        r = A.VisionMeasurement.from_buffer_copy(first[k])
        assert v.value == 1 and r.tag_count == out.tag_count and 30 <= r.tag_count <= 31   # (the noise may decode to one more field tag: both sides then count it)
        assert abs(r.pose_x - out.pose_x) < 1e-6 and abs(r.pose_y - out.pose_y) < 1e-6 and abs(r.pose_rot - out.pose_rot) < 1e-7
        assert abs(r.pose_x - truths[k]["twr"][0]) < 0.03 and abs(r.pose_y - truths[k]["twr"][1]) < 0.03
    task.detector.close()


def test_sqpnp_golden_vectors(built):
    """The HIP solver against the committed vectors directly (no oracle in the loop), tolerance 1e-9."""
    import golden_util as G
    from chalkydri_amd.detector import AprilTagDetector
    from chalkydri_amd.sqpnp import SqPnP
    det = AprilTagDetector(64, 64)
    solver = SqPnP(det)
    cases = G.load("sqpnp_golden.json")
    probs = []
    for c in cases:
        tags, b, rtc, gyro = G.sqpnp_problem(c)
        probs.append(([_iso(R, t) for R, t in tags], b, _iso(*rtc), gyro, 600.0))
    got = solver.solve_batch(probs)
    for c, g in zip(cases, got):
        want = G.sqpnp_result(c)
        assert (want is None) == (g is None)
        if want is None:
            continue
        assert np.abs(g["rot"] - want["rot"]).max() < TOL and np.abs(g["pos"] - want["pos"]).max() < TOL and abs(g["yaw"] - want["yaw"]) < TOL
    det.close()


def test_rccl_gather_one_rank(built):
    """The multi-GPU exchange is one all_gather of 64-byte records from the device buffer the pose stage fills.  With one GPU
    the collective can still be rehearsed on a one-rank RCCL group: same call, same tensor, same layout.  Runs in a fresh
    process that initialises torch first, like bench.py (torch ships its own HIP runtime and wants to be the first user)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as tdist
torch.cuda.set_device(0)
import scenes
from chalkydri_amd import dist
from chalkydri_amd.apriltags import AprilTags
w, h, f, n = 640, 480, 600.0, 3
layout = scenes.wall_layout(6, cols=3)
r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0)
frames = np.stack([scenes.render_view(40 + i, w, h, f, layout, (2.0, 0.05 * i, 0.0), r2c, noise_amp=1)[0] for i in range(n)])
task = AprilTags(w, h, layout, calib, r2c, cam_id=9, max_batch=n)
want, valid = task.process_batch(frames, [0.0] * n)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
tdist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
d_gyro = torch.zeros(n, dtype=torch.float64, device=dev); d_has = torch.ones(n, dtype=torch.uint8, device=dev)
d_rec = torch.zeros((n, 64), dtype=torch.uint8, device=dev); d_valid = torch.zeros(n, dtype=torch.int32, device=dev)
task.detector.upload(frames)
task.process_uploaded_into(n, d_gyro.data_ptr(), d_has.data_ptr(), d_rec.data_ptr(), d_valid.data_ptr())
out = dist.gather_records(d_rec, 1, force=True)
torch.cuda.synchronize()
got = dist.records_to_numpy(out)
assert out.shape == (n, 64) and d_valid.cpu().numpy().tolist() == [int(v) for v in valid]
for i in range(n):
    assert bytes(out[i].cpu().numpy()) == bytes(want[i])
    assert got["camera_id"][i] == 9 and got["tag_count"][i] == 6
# the same exchange through the C ABI (ck_comm_create + ck_gather_poses: what a Rust/C++ host calls), to a host buffer and to
# a device buffer; the id of rank 0 travels through the torch group
comm = dist.PoseComm(task.detector, 0, 1, dev)
assert task.detector._L.ck_backend(task.detector._h) == 1
host = comm.gather(n)
d_out = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
comm.gather(n, out_ptr=d_out.data_ptr(), sync=False)
comm.sync()
for i in range(n):
    assert bytes(host[i]) == bytes(want[i]) == bytes(d_out[i].cpu().numpy())
assert task.detector._L.ck_gather_poses(task.detector._h, comm._c, n, n + 1, d_out.data_ptr(), 1) == -5      # CK_ECAPACITY: rows beyond max_batch
assert task.detector._L.ck_gather_poses(task.detector._h, comm._c, n - 1, n, d_out.data_ptr(), 1) == -1      # CK_EINVAL: not what the last call produced
# a ragged shard: the last process call produced n - 1 records, the collective's row count is n: the library pads with an empty record
task.process_uploaded_into(n - 1, d_gyro.data_ptr(), d_has.data_ptr(), d_rec.data_ptr(), d_valid.data_ptr())
ragged = comm.gather(n - 1, rows=n)
assert ragged.shape == (n, 64) and all(bytes(ragged[i]) == bytes(want[i]) for i in range(n - 1)) and not ragged[n - 1].any()
comm.close()
tdist.destroy_process_group()
print("GATHER OK")
"""
    r = subprocess.run([sys.executable, "-c", script % (root, os.path.join(root, "tests"))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GATHER OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_bench_loop_runs_the_c_abi_gather_on_one_rank(built):
    """bench.py under torch.distributed.run with ONE rank and CK_BENCH_FORCE_COMM=1: ck_comm_create, a ck_gather_poses(sync=0) per
    step and ck_comm_sync execute inside bench's own timed loop (the multi-GPU path of the round-end scaling run, on the one GPU
    a test box has), and the JSON line says so."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CK_BENCH_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29561",
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["config"]["gather"].startswith("ck_gather_poses") and out["config"]["gather_us"]["bytes_per_rank"] == 16 * 64
    assert out["config"]["frames_with_pose"] == 1.0 and out["n_gpus"] == 1


def test_full_size_batch_properties(oracle):
    """BASELINE config 2 at full size (1280x800 x 256 frames, the bench workload): the oracle cannot run 256 frames in seconds,
    so the batch is 8 distinct frames repeated 32 times in shuffled positions.  Properties: a frame's record does not depend
    on where it sits in the batch or on which stream processed it (idempotence over copies), a second run returns the same
    bytes, and the 8 distinct records equal the oracle's."""
    from chalkydri_amd.apriltags import AprilTags
    w, h, n, uniq = 1280, 800, 256, 8
    frames8, gyro8, layout, calib, r2c = scenes.bench_stream(2, uniq, w, h, 6, stream=0, unique=uniq, noise_amp=3)
    rng = np.random.default_rng(0)
    which = rng.permutation(np.repeat(np.arange(uniq), n // uniq))
    frames = frames8[which]
    gyro = [float(gyro8[k]) for k in which]
    task = AprilTags(w, h, layout, calib, r2c, cam_id=0, max_batch=n)
    recs, valid = task.process_batch(frames, gyro)
    again, valid2 = task.process_batch(frames, gyro)
    assert valid.all() and np.array_equal(valid, valid2)
    first = {}
    for i in range(n):
        assert bytes(recs[i]) == bytes(again[i])
        k = int(which[i])
        if k in first:
            assert bytes(recs[i]) == first[k], f"frame copy {i} of {k} differs"
        else:
            first[k] = bytes(recs[i])
    cfg = default_config(w, h)
    for k in range(uniq):
        out = A.VisionMeasurement()
        v = C.c_int(0)
        oracle.lib().ora_process_frame(C.c_void_p(frames8[k].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                       C.c_double(float(gyro8[k])), 1, C.byref(out), C.byref(v))
        r = A.VisionMeasurement.from_buffer_copy(first[k])
        assert v.value == 1 and r.tag_count == out.tag_count
        assert abs(r.pose_x - out.pose_x) < 1e-6 and abs(r.pose_y - out.pose_y) < 1e-6 and abs(r.pose_rot - out.pose_rot) < 1e-7
    task.detector.close()


def test_config4_stream_batch_properties(oracle):
    """BASELINE config 4, one GPU's share: one camera stream (seed offset s * 10^6) at batch 1024.  Eight distinct
    frames fill the batch in shuffled positions: every copy of a frame must give the same record wherever it sits (frame
    indices above 255 exercise the wide end of the frame << 20 | cluster work items), and the eight records equal the oracle's."""
    from chalkydri_amd.apriltags import AprilTags
    w, h, n, uniq, stream = 1280, 800, 1024, 8, 5
    frames8, gyro8, layout, calib, r2c = scenes.bench_stream(2, uniq, w, h, 6, stream=stream, unique=uniq, noise_amp=3)
    rng = np.random.default_rng(4)
    which = rng.permutation(np.repeat(np.arange(uniq), n // uniq))
    task = AprilTags(w, h, layout, calib, r2c, cam_id=stream, max_batch=n)
    recs, valid = task.process_batch(frames8[which], [float(gyro8[k]) for k in which])
    assert valid.all()
    first = {}
    for i in range(n):
        k = int(which[i])
        assert first.setdefault(k, bytes(recs[i])) == bytes(recs[i]), f"frame copy {i} of {k} differs"
    cfg = default_config(w, h)
    for k in range(uniq):
        out = A.VisionMeasurement()
        v = C.c_int(0)
        oracle.lib().ora_process_frame(C.c_void_p(frames8[k].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                       C.c_double(float(gyro8[k])), 1, C.byref(out), C.byref(v))
        r = A.VisionMeasurement.from_buffer_copy(first[k])
        assert v.value == 1 and r.tag_count == out.tag_count and r.camera_id == stream
        assert abs(r.pose_x - out.pose_x) < 1e-6 and abs(r.pose_y - out.pose_y) < 1e-6 and abs(r.pose_rot - out.pose_rot) < 1e-7
    task.detector.close()


def test_glue_filters(oracle):
    """AprilTags::process filters (crates/apriltags/src/lib.rs:306-330): tags missing from the field layout do not enter the
    solve but still count in tag_count; a frame whose tags are all unknown, a frame without tags, and a frame without gyro
    publish the empty record.  Device == oracle in every case."""
    from chalkydri_amd.apriltags import AprilTags
    w, h, f = 640, 480, 600.0
    full = scenes.wall_layout(6, cols=3)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
    calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0)
    pose = (2.1, 0.05, 0.02)
    seen, _ = scenes.render_view(61, w, h, f, full, pose, r2c, noise_amp=1)      # six tags, ids 1..6
    blank = np.full((h, w), 128, np.uint8)
    known3 = {"tags": full["tags"][:3], "field": full["field"]}                   # the layout knows ids 1..3 only
    other = scenes.wall_layout(6, cols=3, first_id=101)                           # a layout that knows none of them
    cfg = default_config(w, h)

    def both(layout, frames, gyros, allow_unverified=0):
        task = AprilTags(w, h, layout, calib, r2c, cam_id=4, max_batch=len(frames))
        task._pp.allow_unverified_ids = allow_unverified
        recs, valid = task.process_batch(np.stack(frames), gyros)
        outs = []
        for fr, g in zip(frames, gyros):
            out = A.VisionMeasurement()
            v = C.c_int(0)
            oracle.lib().ora_process_frame(C.c_void_p(fr.ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                           C.c_double(g or 0.0), 0 if g is None else 1, C.byref(out), C.byref(v))
            outs.append((out, v.value))
        task.detector.close()
        return recs, valid, outs

    recs, valid, outs = both(known3, [seen, blank, seen], [pose[2], pose[2], None])
    assert list(valid) == [1, 0, 0] == [o[1] for o in outs]
    assert recs[0].tag_count == outs[0][0].tag_count == 6                      # ALL detections are counted (lib.rs:354)
    assert abs(recs[0].pose_x - outs[0][0].pose_x) < 1e-6 and abs(recs[0].pose_rot - outs[0][0].pose_rot) < 1e-7
    assert abs(recs[0].pose_x - pose[0]) < 0.05                                # three known tags still locate the robot
    assert bytes(recs[1]) == bytes(outs[1][0]) and bytes(recs[2]) == bytes(outs[2][0]) and recs[1].tag_count == 0
    recs, valid, outs = both(other, [seen], [pose[2]])
    assert list(valid) == [0] == [outs[0][1]] and bytes(recs[0]) == bytes(outs[0][0])
    # ids past the built-in table's verified prefix (ck_family_t.n_upstream = 39) are not upstream ids: a layout that lists them
    # gets no pose from them unless the caller opts in
    far, _ = scenes.render_view(62, w, h, f, other, pose, r2c, noise_amp=1)       # six tags, ids 101..106
    recs, valid, outs = both(other, [far], [pose[2]])
    assert list(valid) == [0] == [outs[0][1]] and bytes(recs[0]) == bytes(outs[0][0])
    recs, valid, outs = both(other, [far], [pose[2]], allow_unverified=1)
    assert list(valid) == [1] == [outs[0][1]] and recs[0].tag_count == 6
    assert abs(recs[0].pose_x - outs[0][0].pose_x) < 1e-6 and abs(recs[0].pose_x - pose[0]) < 0.05
