"""Randomised parity stress of the batched SQPnP solve (crates/chalkydri_sqpnp SqPnP::solve_robot_pose) against the CPU oracle,
away from the well-posed scenes of the unit test: large pixel noise, gyro headings far off, tags behind or beside the camera,
a single tag, coplanar walls, near-duplicate tags, mismatched bearing counts.  Validity must agree; where both solve, rotation
and position within 1e-9.  usage: python tests/stress_sqpnp.py [cases] [seed]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import np_sqpnp as N
import pyoracle
from chalkydri_amd.sqpnp import iso3
from chalkydri_amd.detector import AprilTagDetector
from chalkydri_amd.sqpnp import SqPnP

TOL = 1e-9


def _iso(R, t):
    return iso3(t, N.mat_to_quat(np.asarray(R)))


def run(cases, seed):
    rng = np.random.default_rng(seed)
    det = AprilTagDetector(64, 64)
    solver = SqPnP(det)
    bad = solved = 0
    for c0 in range(0, cases, 64):
        probs, want, kinds = [], [], []
        for k in range(min(64, cases - c0)):
            kind = str(rng.choice(["plain", "noisy", "gyro_off", "one_tag", "wall", "twins", "behind", "short", "many"]))
            n_tags = 1 if kind == "one_tag" else (int(rng.integers(100, 400)) if kind == "many" else int(rng.integers(1, 31)))   # many: far beyond a field's 30 tags
            tags, b, rtc, truth = N.make_scene(rng, n_tags, noise_px={"noisy": 8.0, "plain": 0.0}.get(kind, 0.3))
            gyro = truth["yaw"] + (rng.uniform(-3.1, 3.1) if kind == "gyro_off" else rng.uniform(-0.3, 0.3))
            if kind == "wall":                                   # all tags in one plane, same orientation
                R0, t0 = tags[0]
                tags = [(R0, t0 + R0 @ np.array([0.0, rng.uniform(-2, 2), rng.uniform(-1, 1)])) for _ in tags]
                world = np.concatenate([(R @ N.CORNERS.T).T + t for R, t in tags])
                Rrc, trc = rtc
                Rcw = Rrc @ truth["Rwr"].T; tcw = trc - Rcw @ truth["twr"]
                cam = world @ Rcw.T + tcw
                b = cam / np.linalg.norm(cam, axis=1, keepdims=True)
            if kind == "twins" and n_tags >= 2:                  # two tags at (almost) the same place
                tags[1] = (tags[0][0], tags[0][1] + 1e-9); b[4:8] = b[0:4]
            if kind == "behind":                                 # some bearings point backwards
                b = b.copy(); b[::3, 2] *= -1
            if kind == "short":                                  # bearing count does not match the tags (lib.rs:255 -> None)
                b = b[:-1]
            probs.append(([_iso(R, t) for R, t in tags], b, _iso(*rtc), gyro, 600.0))
            want.append(pyoracle.sqpnp_solve(tags, b, rtc, gyro))
            kinds.append(kind)
        got = solver.solve_batch(probs)
        for k, (g, w) in enumerate(zip(got, want)):
            ok = (g is None) == (w is None)
            if ok and g is not None:
                solved += 1
                ok = np.abs(g["rot"] - w["rot"]).max() < TOL and np.abs(g["pos"] - w["pos"]).max() < TOL and abs(g["yaw"] - w["yaw"]) < TOL
            if not ok:
                bad += 1
                print(json.dumps({"case": c0 + k, "kind": kinds[k], "device": g is not None, "oracle": w is not None,
                                  "drot": None if g is None or w is None else float(np.abs(g["rot"] - w["rot"]).max()),
                                  "dpos": None if g is None or w is None else float(np.abs(g["pos"] - w["pos"]).max())}))
    det.close()
    print(json.dumps({"cases": cases, "solved": solved, "mismatching_cases": bad}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
