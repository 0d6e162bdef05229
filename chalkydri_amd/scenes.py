"""3-D synthetic scenes: a camera mounted on a robot looks at field tags; frames are rendered with the deterministic
renderer (csrc/synth.c).  Used by bench.py (workload of BASELINE.json configs 2-4) and by the pose tests."""
import numpy as np

from . import synth

TAG_SIZE = 0.1651          # crates/chalkydri_sqpnp/src/lib.rs:38
S = TAG_SIZE / 2.0
NWU_TO_CV = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], float)  # lib.rs:449-453

REF_CALIB = {"OpenCVModel5": {"fx": 1368.3343056383071, "fy": 1368.513346806007, "cx": 784.1021700594862,
                              "cy": 655.1967162171935, "k1": -0.03428799012079279, "k2": -0.0021223103005884106,
                              "p1": -0.001, "p2": -0.00014085919680638913, "k3": 0.015316405591806586,
                              "width": 1600, "height": 1304}}  # the reference's deployed calibration, chalkydri.ron:29


def pinhole_calib(f, cx, cy):
    return {"OpenCVModel5": {"fx": f, "fy": f, "cx": cx, "cy": cy, "k1": 0.0, "k2": 0.0, "p1": 0.0, "p2": 0.0, "k3": 0.0}}


def quat_to_mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def euler_to_mat(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def solver_camera_transform(x, y, z, roll_deg, pitch_deg, yaw_deg):
    """(R, t) of cam_cv <- robot, the matrix form of SqPnP::create_solver_camera_transform (lib.rs:430-461)."""
    Rc = euler_to_mat(*np.radians([roll_deg, pitch_deg, yaw_deg])) @ NWU_TO_CV
    return Rc.T, -Rc.T @ np.array([x, y, z], float)


def wall_layout(n_tags, spacing=0.45, cols=6, x_wall=5.0, first_id=1):
    """field.json-shaped dict: n_tags tags on the plane x = x_wall facing -x (quaternion = rotation by pi about z)."""
    tags = []
    for k in range(n_tags):
        r, c = divmod(k, cols)
        tags.append({"ID": first_id + k,
                     "pose": {"translation": {"x": x_wall, "y": (c - (min(cols, n_tags) - 1) / 2.0) * spacing, "z": 1.0 + r * spacing},
                              "rotation": {"quaternion": {"W": 6.123233995736766e-17, "X": 0.0, "Y": 0.0, "Z": 1.0}}}})
    return {"tags": tags, "field": {"length": 16.518, "width": 8.043}}


def render_view(seed, w, h, f, layout, robot_xy_yaw, r2c, family="tag36h11", **params):
    """Renders `layout` as seen from the robot pose (x, y, yaw).  r2c: dict roll,pitch,yaw (deg), x,y,z (m)."""
    x, y, yaw = robot_xy_yaw
    Rwr = euler_to_mat(0, 0, yaw)
    twr = np.array([x, y, 0.0])
    Rrc, trc = solver_camera_transform(r2c["x"], r2c["y"], r2c["z"], r2c["roll"], r2c["pitch"], r2c["yaw"])
    Rcw = Rrc @ Rwr.T
    tcw = trc - Rcw @ twr
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]])
    corners_uv = np.array([[-1, 1, 1], [1, 1, 1], [1, -1, 1], [-1, -1, 1]], float)
    tags = []
    for t in layout["tags"]:
        tr, q = t["pose"]["translation"], t["pose"]["rotation"]["quaternion"]
        Rtw = quat_to_mat([q["W"], q["X"], q["Y"], q["Z"]])
        ttw = np.array([tr["x"], tr["y"], tr["z"]])
        # detection corners (-1,1),(1,1),(1,-1),(-1,-1) <-> tag-local (0,-S,-S),(0,S,-S),(0,S,S),(0,-S,S)  (lib.rs:383-388)
        M = np.stack([Rtw @ np.array([0, S, 0]), Rtw @ np.array([0, 0, -S]), ttw], 1)   # world = M @ (u, v, 1)
        Hc = K @ (Rcw @ M + np.outer(tcw, [0, 0, 1]))
        cam_pts = (Rcw @ (M @ corners_uv.T)).T + tcw
        if np.any(cam_pts[:, 2] < 0.3):
            continue
        px = (K @ cam_pts.T).T
        px = px[:, :2] / px[:, 2:3]
        if px[:, 0].min() < 8 or px[:, 1].min() < 8 or px[:, 0].max() > w - 8 or px[:, 1].max() > h - 8:
            continue
        tags.append((0, t["ID"], Hc))
    frame, truth = synth.render_scene(seed, w, h, tags, (family,), **params)
    return frame, {"tags": truth, "Rwr": Rwr, "twr": twr, "yaw": yaw}


def bench_stream(config_idx, n_frames, w, h, n_tags, stream=0, unique=None, **params):
    """Frames + gyro headings of one virtual camera stream: a robot wandering in front of a wall of `n_tags` field tags.
    `unique` frames are rendered and repeated cyclically to fill the batch (rendering is CPU work outside the timed path)."""
    unique = min(n_frames, unique or n_frames)
    f = float(w) * 0.9
    layout = wall_layout(n_tags, cols=max(3, (n_tags + 1) // 2) if n_tags <= 12 else 6)
    r2c = {"roll": 0.0, "pitch": 0.0, "yaw": 0.0, "x": 0.2, "y": 0.0, "z": 0.6}
    frames = np.empty((unique, h, w), np.uint8)
    gyro = np.empty(unique)
    for i in range(unique):
        seed = synth.frame_seed(config_idx, i, stream)
        rng = np.random.default_rng(seed)
        pose = (rng.uniform(0.6, 1.6), rng.uniform(-0.4, 0.4), rng.uniform(-0.1, 0.1))
        frames[i], _ = render_view(seed, w, h, f, layout, pose, r2c, **params)
        gyro[i] = pose[2] + rng.uniform(-0.035, 0.035)   # true yaw +- 2 degrees
    idx = np.arange(n_frames) % unique
    return frames[idx], gyro[idx], layout, pinhole_calib(f, w / 2.0, h / 2.0), r2c
