"""3-D synthetic scenes for the pose tests: a camera on a robot looking at a wall of field tags, rendered to pixels."""
import numpy as np

import np_sqpnp as N
from chalkydri_amd import synth

REF_CALIB = {"OpenCVModel5": {"fx": 1368.3343056383071, "fy": 1368.513346806007, "cx": 784.1021700594862, "cy": 655.1967162171935,
                              "k1": -0.03428799012079279, "k2": -0.0021223103005884106, "p1": -0.001,
                              "p2": -0.00014085919680638913, "k3": 0.015316405591806586, "width": 1600, "height": 1304}}  # chalkydri.ron:29


def pinhole_calib(f, cx, cy):
    return {"OpenCVModel5": {"fx": f, "fy": f, "cx": cx, "cy": cy, "k1": 0.0, "k2": 0.0, "p1": 0.0, "p2": 0.0, "k3": 0.0}}


def wall_layout(n_tags, spacing=0.45, cols=6, x_wall=5.0):
    """Field-layout dict with n_tags tags (ids 1..n) on the plane x = x_wall, facing -x (normal = tag-local +x rotated by pi about z)."""
    tags = []
    for k in range(n_tags):
        r, c = divmod(k, cols)
        y = (c - (cols - 1) / 2.0) * spacing
        z = 1.0 + r * spacing
        tags.append({"ID": k + 1, "pose": {"translation": {"x": x_wall, "y": y, "z": z},
                                           "rotation": {"quaternion": {"W": 6.123233995736766e-17, "X": 0.0, "Y": 0.0, "Z": 1.0}}}})
    return {"tags": tags, "field": {"length": 16.518, "width": 8.043}}


def render_view(seed, w, h, f, layout, robot_xy_yaw, r2c, family="tag36h11", **params):
    """Renders the layout seen from a robot pose.  Returns (frame, truth dict).  r2c: dict roll,pitch,yaw(deg),x,y,z."""
    x, y, yaw = robot_xy_yaw
    Rwr = N.euler_to_mat(0, 0, yaw)
    twr = np.array([x, y, 0.0])
    Rrc, trc = N.create_solver_camera_transform(r2c["x"], r2c["y"], r2c["z"], r2c["roll"], r2c["pitch"], r2c["yaw"])
    Rcw = Rrc @ Rwr.T
    tcw = trc - Rcw @ twr
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1]])
    tags = []
    for t in layout["tags"]:
        tr, q = t["pose"]["translation"], t["pose"]["rotation"]["quaternion"]
        Rtw = N.quat_to_mat([q["W"], q["X"], q["Y"], q["Z"]])
        ttw = np.array([tr["x"], tr["y"], tr["z"]])
        # tag plane coords (u,v in [-1,1] at the black border) -> tag-local (0, -u*S?, ...): the detector's corner order
        # (-1,1),(1,1),(1,-1),(-1,-1) must land on CORNERS[0..3] = (0,-S,-S),(0,S,-S),(0,S,S),(0,-S,S)
        # => local y = u*S, local z = -v*S
        S = N.S
        M = np.stack([Rtw @ np.array([0, S, 0]), Rtw @ np.array([0, 0, -S]), ttw], 1)   # world = M @ (u,v,1)
        Hc = K @ (Rcw @ M + np.outer(tcw, [0, 0, 1]))
        cam_pts = (Rcw @ (M @ np.array([[-1, 1, 1], [1, 1, 1], [1, -1, 1], [-1, -1, 1]]).T)).T + tcw
        if np.any(cam_pts[:, 2] < 0.3):
            continue
        px = (K @ cam_pts.T).T
        px = px[:, :2] / px[:, 2:3]
        if px[:, 0].min() < 8 or px[:, 1].min() < 8 or px[:, 0].max() > w - 8 or px[:, 1].max() > h - 8:
            continue
        tags.append((0, t["ID"], Hc))
    frame, truth = synth.render_scene(seed, w, h, tags, (family,), **params)
    return frame, {"tags": truth, "Rwr": Rwr, "twr": twr, "yaw": yaw}
