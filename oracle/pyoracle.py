"""ctypes loader for the CPU oracle (oracle/libck_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from chalkydri_amd import _abi as A  # noqa: E402  (POD struct mirrors only)

# CK_ORACLE_LIB: another build of the same sources (the -O3 -march=native one of the CPU baseline, checked against the goldens)
LIB_PATH = os.environ.get("CK_ORACLE_LIB") or os.path.join(_HERE, "libck_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data


def threshold(img, min_white_black_diff=5):
    img, p = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_threshold(C.c_void_p(p), w, h, w, min_white_black_diff, C.c_void_p(out.ctypes.data))
    return out


def segment(thresh):
    t, p = _u8(thresh)
    h, w = t.shape
    labels = np.empty((h, w), np.uint32)
    sizes = np.empty((h, w), np.uint32)
    lib().ora_segment(C.c_void_p(p), w, h, C.c_void_p(labels.ctypes.data), C.c_void_p(sizes.ctypes.data))
    return labels, sizes


def clusters(thresh, labels, sizes, min_component_px=25):
    t, p = _u8(thresh)
    h, w = t.shape
    pcap, ccap = 4 * w * h, w * h // 4 + 1024
    cl = (A.Cluster * ccap)()
    pts = (A.ClusterPoint * pcap)()
    nc, npn = C.c_int(0), C.c_int(0)
    ov = lib().ora_clusters(C.c_void_p(p), C.c_void_p(labels.ctypes.data), C.c_void_p(sizes.ctypes.data), w, h,
                            min_component_px, cl, ccap, C.byref(nc), pts, pcap, C.byref(npn))
    cl_np = np.frombuffer(cl, dtype=np.uint32, count=nc.value * 4).reshape(-1, 4).copy()
    pt_np = np.frombuffer(pts, dtype=np.dtype([("x", "<u2"), ("y", "<u2"), ("gx", "i1"), ("gy", "i1"), ("pad", "<u2")]),
                          count=npn.value).copy()
    return cl_np, pt_np, ov


def fit_quads(img, cfg, cl_np, pt_np, quad_img=None):
    img, p = _u8(img)
    h, w = img.shape
    q = img if quad_img is None else np.ascontiguousarray(quad_img, np.uint8)
    qh, qw = q.shape
    cl = np.ascontiguousarray(cl_np, np.uint32)
    pts = np.ascontiguousarray(pt_np)
    cap = 4096
    quads = (A.Quad * cap)()
    nq = C.c_int(0)
    ov = lib().ora_fit_quads(C.c_void_p(q.ctypes.data), qw, qh, qw, C.c_void_p(p), w, h, w, C.byref(cfg),
                             C.c_void_p(cl.ctypes.data), len(cl), C.c_void_p(pts.ctypes.data), quads, cap,
                             C.byref(nq))
    return [quads[i] for i in range(nq.value)], ov


def quads_to_np(quads):
    out = np.zeros((len(quads), 11), np.float64)
    for i, q in enumerate(quads):
        out[i, :8] = [q.p[k][j] for k in range(4) for j in range(2)]
        out[i, 8], out[i, 9], out[i, 10] = q.reversed_border, q.rep0, q.rep1
    return out


def detect(img, cfg, cap=256):
    img, p = _u8(img)
    h, w = img.shape
    dets = (A.Detection * cap)()
    n = C.c_int(0)
    st = C.c_uint32(0)
    rc = lib().ora_detect(C.c_void_p(p), w, h, w, C.byref(cfg), dets, cap, C.byref(n), C.byref(st))
    assert rc == 0
    return dets_to_list(dets, n.value), st.value


def dets_to_list(dets, n):
    out = []
    for i in range(n):
        d = dets[i]
        out.append({"id": d.id, "hamming": d.hamming, "family": d.family, "margin": d.decision_margin,
                    "c": np.array([d.c[0], d.c[1]]),
                    "p": np.array([[d.p[k][0], d.p[k][1]] for k in range(4)])})
    return out


# ---- SQPnP ---------------------------------------------------------------------------------------------
def _iso(R, t):
    from np_sqpnp import mat_to_quat
    iso = A.Iso3()
    q = mat_to_quat(np.asarray(R, float))
    for k in range(3):
        iso.t[k] = float(t[k])
    for k in range(4):
        iso.q[k] = float(q[k])
    return iso


def sqpnp_solve(tags, bearings, robot_to_cam, gyro, sign_change_error=600.0, max_iter=15, tol_sq=1e-16):
    L = lib()
    prm = A.SqpnpParams(max_iter, tol_sq)
    isos = (A.Iso3 * max(len(tags), 1))(*[_iso(R, t) for R, t in tags])
    b = np.ascontiguousarray(bearings, np.float64)
    rtc = _iso(*robot_to_cam)
    res = A.SqpnpResult()
    L.ora_sqpnp_solve_robot_pose.restype = C.c_int
    L.ora_sqpnp_solve_robot_pose.argtypes = [C.POINTER(A.SqpnpParams), C.POINTER(A.Iso3), C.c_int, C.c_void_p, C.c_int,
                                             C.POINTER(A.Iso3), C.c_double, C.c_double, C.POINTER(A.SqpnpResult)]
    ok = L.ora_sqpnp_solve_robot_pose(C.byref(prm), isos, len(tags), b.ctypes.data, len(b), C.byref(rtc), gyro,
                                      sign_change_error, C.byref(res))
    if not ok:
        return None
    return {"rot": np.array(res.rot[:]).reshape(3, 3), "pos": np.array(res.pos[:]), "std": np.array(res.std_devs[:]),
            "yaw": res.yaw, "energy": res.energy}


def create_solver_camera_transform(*args):
    L = lib()
    out = A.Iso3()
    L.ora_sqpnp_create_solver_camera_transform.restype = None
    L.ora_sqpnp_create_solver_camera_transform.argtypes = [C.c_double] * 6 + [C.POINTER(A.Iso3)]
    L.ora_sqpnp_create_solver_camera_transform(*[float(a) for a in args], C.byref(out))
    return np.array(out.t[:]), np.array(out.q[:])


def unproject_opencv5(cam, px):
    L = lib()
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    out = np.empty((len(px), 3))
    ok = np.empty(len(px), np.uint8)
    c = A.OpenCV5(*[float(v) for v in cam])
    L.ora_unproject_opencv5(C.byref(c), C.c_void_p(px.ctypes.data), len(px), C.c_void_p(out.ctypes.data), C.c_void_p(ok.ctypes.data))
    return out, ok.astype(bool)


# ---- CAT -------------------------------------------------------------------------------------------------
def cat_calc_otsu(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().ora_cat_calc_otsu(C.c_void_p(rgb.ctypes.data), w, h, C.c_void_p(out.ctypes.data))
    return out


def cat_thresh(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().ora_cat_thresh(C.c_void_p(rgb.ctypes.data), w, h, C.c_void_p(out.ctypes.data))
    return out


def cat_detect_corners(classes, cap=None):
    c = np.ascontiguousarray(classes, np.uint8)
    h, w = c.shape
    cap = cap or w * h
    pts = np.zeros((cap, 2), np.uint32)
    n = lib().ora_cat_detect_corners(C.c_void_p(c.ctypes.data), w, h, C.c_void_p(pts.ctypes.data), cap)
    return pts[:min(n, cap)].copy(), n


def cat_check_edges(classes, pts, cap=None):
    c = np.ascontiguousarray(classes, np.uint8)
    h, w = c.shape
    pts = np.ascontiguousarray(pts, np.uint32)
    cap = cap or max(16, 2 * len(pts) ** 2)
    lines = np.zeros((cap, 4), np.uint32)
    n = lib().ora_cat_check_edges(C.c_void_p(c.ctypes.data), w, h, C.c_void_p(pts.ctypes.data), len(pts), C.c_void_p(lines.ctypes.data), cap)
    return lines[:min(n, cap)].copy(), n


def cat_connected_components(classes):
    c = np.ascontiguousarray(classes, np.uint8)
    h, w = c.shape
    roots = np.empty((h, w), np.uint32)
    sizes = np.empty((h, w), np.uint32)
    lib().ora_cat_connected_components_canonical(C.c_void_p(c.ctypes.data), w, h, C.c_void_p(roots.ctypes.data), C.c_void_p(sizes.ctypes.data))
    return roots, sizes


def cat_connected_components_reference(classes):
    """The reference-faithful sequential UnionFind (union by size): returns (parent, sizes) as u64 arrays."""
    c = np.ascontiguousarray(classes, np.uint8)
    h, w = c.shape
    parent = np.empty(h * w, np.uint64)
    sizes = np.empty(h * w, np.uint64)
    lib().ora_cat_connected_components(C.c_void_p(c.ctypes.data), w, h, C.c_void_p(parent.ctypes.data), C.c_void_p(sizes.ctypes.data))
    return parent, sizes
