"""Host-side mirror of the `AprilTags` Copper sink task — the caller of the hot path
(crates/apriltags/src/lib.rs:166-182 struct, :221-291 new, :293-379 process) — batched over frames.

Configuration follows the reference: `family` (default tag36h11), `bits_corrected` (default 3), `cam_id`,
`robot_to_cam` JSON {roll,pitch,yaw,x,y,z} and `calib` JSON {"OpenCVModel5": {...}} (chalkydri.ron:28-29), and the
field layout JSON (field.json; crates/apriltags/src/field_layout.rs:18-44).
"""
import ctypes as C
import json

import numpy as np

from . import _abi as A
from ._lib import check
from .detector import AprilTagDetector
from .sqpnp import SqPnP, iso3

SIGN_FLIP_CONST = 600.0  # crates/apriltags/src/lib.rs:6


def load_field_layout(path_or_dict):
    """field_layout.rs:18-44: id -> Isometry(translation, UnitQuaternion::from_quaternion(W,X,Y,Z)) (normalised)."""
    d = path_or_dict if isinstance(path_or_dict, dict) else json.load(open(path_or_dict))
    tags = {}
    for t in d["tags"]:
        tr, q = t["pose"]["translation"], t["pose"]["rotation"]["quaternion"]
        qq = np.array([q["W"], q["X"], q["Y"], q["Z"]], float)
        qq /= np.linalg.norm(qq)
        tags[int(t["ID"])] = iso3([tr["x"], tr["y"], tr["z"]], qq)
    return tags


class AprilTags:
    def __init__(self, width, height, field, calib, robot_to_cam, cam_id=0, family="tag36h11", bits_corrected=3,
                 max_batch=1, device=0, **cfg):
        calib = json.loads(calib) if isinstance(calib, str) else calib
        r2c = json.loads(robot_to_cam) if isinstance(robot_to_cam, str) else robot_to_cam
        m = calib["OpenCVModel5"]
        self.cam = A.OpenCV5(*[float(m[k]) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")])
        # argument order of the reference call (lib.rs:247-254): x, y, z, roll, pitch, yaw
        self.robot_to_cam = SqPnP.create_solver_camera_transform(r2c["x"], r2c["y"], r2c["z"], r2c["roll"], r2c["pitch"], r2c["yaw"])
        self.detector = AprilTagDetector(width, height, max_batch=max_batch, families=(family,), bits_corrected=bits_corrected,
                                         device=device, **cfg)
        self.solver = SqPnP(self.detector)
        self.tags = field if isinstance(field, dict) and all(isinstance(v, A.Iso3) for v in field.values()) else load_field_layout(field)
        self.cam_id = cam_id
        self._field = (A.FieldTag * max(len(self.tags), 1))()
        for i, (tid, pose) in enumerate(sorted(self.tags.items())):
            self._field[i].id, self._field[i].pose = tid, pose
        self._pp = A.ProcessParams()
        self._pp.cam, self._pp.robot_to_cam = self.cam, self.robot_to_cam
        self._pp.field, self._pp.n_field = self._field, len(self.tags)
        self._pp.camera_id, self._pp.sign_change_error = cam_id, SIGN_FLIP_CONST
        self._pp.sqpnp = self.solver._prm

    def process_batch(self, frames=None, gyro=None, n=None):
        """frames [n][h][w] (or None to reuse uploaded frames); gyro: per-frame heading or None entries ("no gyro").
        Returns (records as a structured array of 64-byte VisionMeasurement, valid flags)."""
        det = self.detector
        if frames is not None:
            n = det.upload(frames)
        g = np.zeros(n, np.float64)
        has = np.zeros(n, np.uint8)
        for i in range(n):
            gi = None if gyro is None else (gyro if np.isscalar(gyro) else gyro[i])
            if gi is not None:
                g[i], has[i] = gi, 1
        out = (A.VisionMeasurement * n)()
        valid = (C.c_int32 * n)()
        check(det._L.ck_process_uploaded(det._h, n, C.byref(self._pp), g.ctypes.data, has.ctypes.data, out, valid), "ck_process_uploaded")
        return out, np.array(valid[:], bool)

    def process_uploaded_into(self, n, gyro_ptr, has_gyro_ptr, out_ptr, valid_ptr):
        """Runs the whole path on the uploaded frames; every pointer may be host or device memory (no host round trip when
        the caller hands over device buffers, e.g. torch tensors that feed the RCCL pose gather)."""
        det = self.detector
        check(det._L.ck_process_uploaded(det._h, n, C.byref(self._pp), C.c_void_p(gyro_ptr), C.c_void_p(has_gyro_ptr),
                                         C.cast(C.c_void_p(out_ptr), C.POINTER(A.VisionMeasurement)),
                                         C.cast(C.c_void_p(valid_ptr), C.POINTER(C.c_int32))), "ck_process_uploaded")

    def process_device(self, ptr, n, stride, frame_pitch, gyro):
        det = self.detector
        g = np.ascontiguousarray(gyro, np.float64)
        has = np.ones(n, np.uint8)
        out = (A.VisionMeasurement * n)()
        valid = (C.c_int32 * n)()
        check(det._L.ck_process_batch_device(det._h, C.c_void_p(ptr), n, stride, frame_pitch, C.byref(self._pp), g.ctypes.data,
                                             has.ctypes.data, out, valid), "ck_process_batch_device")
        return out, np.array(valid[:], bool)


def measurements_to_bytes(records):
    """The exact datagram bytes whacknet would put on the wire (crates/whacknet/src/lib.rs:43-66,84-86)."""
    return [bytes(r) for r in records]
