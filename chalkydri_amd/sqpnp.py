"""Host-side mirror of chalkydri_sqpnp::SqPnP (crates/chalkydri_sqpnp/src/lib.rs:183-222,297-304,430-437) over the C ABI.

Isometries are (translation[3], quaternion[w,x,y,z]) pairs, the content of nalgebra's Isometry3<f64>.
"""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, lib
from .detector import _bind


def iso3(t, q):
    iso = A.Iso3()
    for k in range(3):
        iso.t[k] = float(t[k])
    for k in range(4):
        iso.q[k] = float(q[k])
    return iso


class SqPnP:
    def __init__(self, handle=None):
        """`handle`: an AprilTagDetector whose device/stream the batched solve runs on."""
        self._L = _bind(lib())
        self._det = handle
        self._prm = A.SqpnpParams()
        self._L.ck_sqpnp_params_default(C.byref(self._prm))

    # builders, as in lib.rs:214-222
    def max_iter(self, n):
        self._prm.max_iter = int(n)
        return self

    def tolerance(self, tol):
        self._prm.tol_sq = float(tol) * float(tol)
        return self

    @staticmethod
    def create_solver_camera_transform(fwd_m, left_m, up_m, roll_deg, pitch_deg, yaw_deg):
        L = _bind(lib())
        out = A.Iso3()
        L.ck_sqpnp_create_solver_camera_transform(fwd_m, left_m, up_m, roll_deg, pitch_deg, yaw_deg, C.byref(out))
        return out

    def solve_robot_pose(self, points_isometry, points_2d, robot_to_cam, gyro, sign_change_error):
        """One problem, like the reference signature; returns (rot 3x3, pos, std_devs) or None."""
        r = self.solve_batch([(points_isometry, points_2d, robot_to_cam, gyro, sign_change_error)])[0]
        return None if r is None else (r["rot"], r["pos"], r["std_devs"])

    def solve_batch(self, problems):
        """problems: list of (tags [Iso3], bearings (m,3), robot_to_cam Iso3, gyro, sign_change_error)."""
        if self._det is None:
            raise RuntimeError("SqPnP needs a device handle (pass an AprilTagDetector)")
        n = len(problems)
        probs = (A.SqpnpProblem * max(n, 1))()
        tags, bear = [], []
        for i, (iso, p2, rtc, gyro, sce) in enumerate(problems):
            p2 = np.asarray(p2, np.float64).reshape(-1, 3)
            probs[i].n_tags, probs[i].n_bearings = len(iso), len(p2)
            probs[i].tag_offset, probs[i].bearing_offset = len(tags), sum(len(b) for b in bear)
            probs[i].robot_to_cam, probs[i].gyro, probs[i].sign_change_error = rtc, float(gyro), float(sce)
            tags.extend(iso)
            bear.append(p2)
        tarr = (A.Iso3 * max(len(tags), 1))(*tags)
        barr = np.ascontiguousarray(np.concatenate(bear) if bear else np.zeros((0, 3)), np.float64)
        res = (A.SqpnpResult * max(n, 1))()
        check(self._L.ck_sqpnp_solve_batch(self._det._h, C.byref(self._prm), probs, n, tarr, len(tags),
                                           barr.ctypes.data, len(barr), res), "ck_sqpnp_solve_batch")
        out = []
        for i in range(n):
            r = res[i]
            out.append(None if not r.valid else {"rot": np.array(r.rot[:]).reshape(3, 3), "pos": np.array(r.pos[:]),
                                                 "std_devs": np.array(r.std_devs[:]), "yaw": r.yaw, "energy": r.energy})
        return out


def unproject_opencv5(cam, px):
    L = _bind(lib())
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    out = np.empty((len(px), 3))
    ok = np.empty(len(px), np.uint8)
    c = A.OpenCV5(*[float(v) for v in cam])
    check(L.ck_unproject_opencv5(C.byref(c), px.ctypes.data, len(px), out.ctypes.data, ok.ctypes.data), "ck_unproject_opencv5")
    return out, ok.astype(bool)
