"""The C++ host layer (include/chalkydri.hpp) over the C ABI: compiles and links with plain g++, refuses to run without a
GPU, and on a GPU returns byte-for-byte what the Python mirror of the same ABI returns (tests/cpp/host_demo.cpp)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "chalkydri_amd", "lib", "host_demo")


def _run(*args):
    return subprocess.run([DEMO, *map(str, args)], capture_output=True, text=True, timeout=600)


def test_host_demo_builds_and_selfchecks(built):
    assert os.path.exists(DEMO)
    r = _run("selfcheck")
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK")


def test_header_declares_reference_surface():
    """Names the Rust callers use (crates/chalkydri-apriltags, chalkydri_sqpnp, apriltags) exist in the C++ layer."""
    src = open(os.path.join(ROOT, "include", "chalkydri.hpp")).read()
    for name in ("class Detector", "class UnionFind", "calc_otsu", "process_frame", "detect_corners", "check_edges", "connected_components",
                 "thresh", "draw", "clone", "get_size", "class SqPnP", "max_iter", "tolerance", "solve_robot_pose",
                 "create_solver_camera_transform", "class AprilTags", "VisionMeasurement", "decision_margin", "corners"):
        assert name in src, name


def test_gather_ranks_harness_builds_and_skips_without_a_gpu(built):
    """tests/cpp/gather_ranks.cpp links against the C ABI alone; with no GPU in the KFD topology it says so and exits 0."""
    exe = os.path.join(ROOT, "chalkydri_amd", "lib", "gather_ranks")
    assert os.path.exists(exe)
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "GATHER_RANKS_SKIP" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_gather_ranks_one_process_per_gpu(built):
    """The pose gather the way a Rust / C++ host runs it — no Python, no torch in the ranks: the parent forks one child per visible
    GPU before any HIP call, the RCCL id travels through pipes, every rank processes two ragged batches, enqueues both gathers
    (sync = 0: step 1's rendezvous on the communicator's stream beside step 2's kernels) and the parent compares what every rank
    received with what every rank produced.  One rank on a one-GPU box, N on an N-GPU node (record layout:
    /root/reference/crates/whacknet/src/lib.rs:43-66)."""
    exe = os.path.join(ROOT, "chalkydri_amd", "lib", "gather_ranks")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "GATHER_RANKS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "librccl" in r.stderr


@pytest.mark.gpu
def test_cpp_host_matches_python_mirror(built, tmp_path):
    import scenes
    from chalkydri_amd import synth
    from chalkydri_amd.apriltags import AprilTags
    from chalkydri_amd.cat import CatDetector
    from chalkydri_amd.detector import AprilTagDetector
    w, h, f = 640, 480, 600.0
    layout = scenes.wall_layout(6, cols=3)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
    calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0)
    pose = (2.2, 0.1, 0.05)
    frame, truth = scenes.render_view(77, w, h, f, layout, pose, r2c, noise_amp=1)
    gyro = 0.06
    fpath, lpath = tmp_path / "frame.raw", tmp_path / "layout.txt"
    frame.tofile(fpath)
    with open(lpath, "w") as fo:
        for t in layout["tags"]:
            tr, q = t["pose"]["translation"], t["pose"]["rotation"]["quaternion"]
            fo.write(f"{t['ID']} {tr['x']!r} {tr['y']!r} {tr['z']!r} {q['W']!r} {q['X']!r} {q['Y']!r} {q['Z']!r}\n")
    m = calib["OpenCVModel5"]
    args = [m[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")] + [r2c[k] for k in ("roll", "pitch", "yaw", "x", "y", "z")] + [gyro]
    r = _run("run", w, h, fpath, lpath, *[repr(float(a)) for a in args])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    # the Python mirror on the same frame
    det = AprilTagDetector(w, h, max_batch=1)
    dets = det.detect_batch(frame[None])[0]
    det.close()
    cpp_dets = [ln.split() for ln in lines if ln.startswith("det ")]
    assert len(cpp_dets) == len(dets) == 6
    for c, d in zip(cpp_dets, dets):
        assert (int(c[1]), int(c[2])) == (d.id(), d.hamming())
        assert bytes.fromhex(c[3]) == np.float32(d.decision_margin()).tobytes()
        assert bytes.fromhex(c[4]) == np.asarray(d.center(), np.float64).tobytes()
        assert bytes.fromhex(c[5]) == np.asarray(d.corners(), np.float64).tobytes()
    task = AprilTags(w, h, layout, calib, r2c, cam_id=5, max_batch=1)
    recs, valid = task.process_batch(frame[None], [gyro])
    task.detector.close()
    ml = [ln.split() for ln in lines if ln.startswith("measurement ")][0]
    assert int(ml[1]) == int(valid[0]) == 1
    assert bytes.fromhex(ml[2]) == bytes(recs[0])
    assert [ln for ln in lines if ln.startswith("nogyro ")][0].split()[1:] == ["0", "0"]
    # the record is near the truth too
    rec = recs[0]
    assert abs(rec.pose_x - pose[0]) < 0.03 and abs(rec.pose_y - pose[1]) < 0.03

    # CAT front-end through C++ vs through Python
    rgb, _ = synth.render(synth.frame_seed(1, 3), 160, 120, 2, noise_amp=1, rgb=True) if "rgb" in synth.render.__code__.co_varnames else (None, None)
    if rgb is None:
        g, _ = synth.render(synth.frame_seed(1, 3), 160, 120, 2, noise_amp=1, min_side=30, max_side=60)
        rgb = np.repeat(g[:, :, None], 3, axis=2)
    rpath = tmp_path / "rgb.raw"
    np.ascontiguousarray(rgb, np.uint8).tofile(rpath)
    r = _run("cat", 160, 120, rpath)
    assert r.returncode == 0, r.stderr
    tok = r.stdout.split()
    cat = CatDetector(160, 120)
    pts, lns = cat.process_frame(rgb)
    uf = cat.connected_components()

    def fnv(a):
        hsh = 1469598103934665603
        for b in np.ascontiguousarray(a).tobytes():
            hsh = ((hsh ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return f"{hsh:016x}"
    assert tok[1] == fnv(cat.buf)
    assert int(tok[3]) == len(pts) and tok[4] == fnv(pts.astype(np.uint32))
    assert int(tok[6]) == len(lns) and tok[7] == fnv(lns.astype(np.uint32))
    assert tok[9] == fnv(uf._roots.astype(np.uint32)) and tok[11] == fnv(uf._sizes.astype(np.uint32))
    cat.close()
