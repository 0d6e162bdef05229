"""Randomised parity stress of AprilTags::process (detect -> known-tag / 4-corner / gyro filters -> unprojection -> SQPnP -> 64-byte
record) against the CPU oracle (test infrastructure: imports oracle/): random frame sizes, focal lengths, robot poses, camera
mounts, tag counts, noise, calibration kinds, gyro errors (incl. beyond MAX_GYRO_DELTA) and missing gyro; validity and the
record's integer fields must be equal, the pose within 1e-6 m / 1e-7 rad, the std-devs within 1e-6 relative.
usage: python tests/stress_pose.py [cases] [seed]"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
from chalkydri_amd import _abi as A, default_config, scenes
from chalkydri_amd.apriltags import AprilTags

def run(cases, seed):
    rng = np.random.default_rng(seed)
    bad, solved = 0, 0
    for c in range(cases):
        w = int(rng.integers(640, 1320)); h = int(rng.integers(400, 840))   # any width and height, odd ones included
        dec = int(rng.choice([1, 1, 2]))
        f = float(rng.uniform(0.6, 1.1)) * w
        n = int(rng.integers(1, 5))
        layout = scenes.wall_layout(int(rng.integers(1, 19)), spacing=float(rng.uniform(0.3, 0.6)), cols=int(rng.integers(2, 7)))
        r2c = {"roll": float(rng.uniform(-8, 8)), "pitch": float(rng.uniform(-12, 12)), "yaw": float(rng.uniform(-15, 15)),
               "x": float(rng.uniform(-0.3, 0.3)), "y": float(rng.uniform(-0.3, 0.3)), "z": float(rng.uniform(0.2, 0.9))}
        calib = scenes.pinhole_calib(f, w / 2.0, h / 2.0) if rng.random() < 0.6 else scenes.REF_CALIB
        frames, gyros = [], []
        for i in range(n):
            pose = (float(rng.uniform(0.5, 3.5)), float(rng.uniform(-1.0, 1.0)), float(rng.uniform(-0.5, 0.5)))
            fr, _ = scenes.render_view(5000 + 10 * c + i, w, h, f, layout, pose, r2c, noise_amp=int(rng.choice([0, 1, 3])))
            frames.append(fr)
            u = rng.random()
            gyros.append(None if u < 0.1 else pose[2] + (float(rng.uniform(-0.03, 0.03)) if u < 0.8 else float(rng.uniform(-1.2, 1.2))))
        frames = np.stack(frames)
        if os.environ.get("STRESS_LOG"):
            with open(os.environ["STRESS_LOG"], "a") as lf:
                lf.write(json.dumps({"case": c, "w": w, "h": h, "n": n, "tags": len(layout["tags"]), "r2c": r2c}) + "\n")
        task = AprilTags(w, h, layout, calib, r2c, cam_id=int(rng.integers(0, 200)), max_batch=n, quad_decimate=dec)
        recs, valid = task.process_batch(frames, gyros)
        cfg = default_config(w, h, quad_decimate=dec)
        for i in range(n):
            out = A.VisionMeasurement(); v = C.c_int(0)
            pyoracle.lib().ora_process_frame(C.c_void_p(frames[i].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp),
                                             C.c_double(gyros[i] or 0.0), 0 if gyros[i] is None else 1, C.byref(out), C.byref(v))
            r = recs[i]
            ok = bool(v.value) == bool(valid[i]) and (r.camera_id, r.tag_count, r.ts) == (out.camera_id, out.tag_count, out.ts)
            if ok and not valid[i]:
                ok = bytes(r) == bytes(out)
            elif ok:
                solved += 1
                ok = abs(r.pose_x - out.pose_x) < 1e-6 and abs(r.pose_y - out.pose_y) < 1e-6 and abs(r.pose_rot - out.pose_rot) < 1e-7 and \
                    bool(np.allclose([r.std_x, r.std_y, r.std_rot], [out.std_x, out.std_y, out.std_rot], rtol=1e-6))
            if not ok:
                bad += 1
                print(json.dumps({"case": c, "frame": i, "valid": [bool(valid[i]), bool(v.value)], "tags": [r.tag_count, out.tag_count],
                                  "dx": r.pose_x - out.pose_x, "dy": r.pose_y - out.pose_y, "drot": r.pose_rot - out.pose_rot}))
        task.detector.close()
    print(json.dumps({"cases": cases, "frames_with_pose": solved, "mismatching_frames": bad}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
