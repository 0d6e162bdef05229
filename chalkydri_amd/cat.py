"""Host-side mirror of crates/chalkydri-apriltags `Detector` ("CAT") and `UnionFind` over the C ABI.

Keeps the reference's surface: new(width, height, valid_tags), calc_otsu, thresh, process_frame, detect_corners,
check_edges, connected_components -> UnionFind{find, get_size} (src/lib.rs:42-113,158-181,191,265,291,319,480,501).
State lives where the reference keeps it: the class buffer, the corner points and the checked lines.
"""
import ctypes as C

import numpy as np

from ._lib import check, default_config, lib
from .detector import AprilTagDetector, _bind

BLACK, WHITE, OTHER = 0, 1, 2  # utils.rs:2-6


class UnionFind:
    """Result of connected_components: canonical root (smallest index of the set) and set size per pixel."""

    def __init__(self, roots, sizes):
        self._roots, self._sizes = roots.reshape(-1), sizes.reshape(-1)

    def find(self, i):
        return int(self._roots[i])

    def get_size(self, i):
        return int(self._sizes[i])


class CatDetector:
    def __init__(self, width, height, valid_tags=(), device=0):
        self.width, self.height, self.valid_tags = width, height, tuple(valid_tags)
        self._det = AprilTagDetector(width, height, max_batch=1, device=device)
        self._L = _bind(lib())
        self.buf = np.zeros((height, width), np.uint8)          # alloc_zeroed: all Black (lib.rs:166-167)
        self.points = np.zeros((0, 2), np.uint32)
        self.lines = np.zeros((0, 4), np.uint32)

    def close(self):
        self._det.close()

    def calc_otsu(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        check(self._L.ck_cat_calc_otsu(self._det._h, rgb.ctypes.data, self.width, self.height, self.buf.ctypes.data), "ck_cat_calc_otsu")
        return self.buf

    def thresh(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        check(self._L.ck_cat_thresh(self._det._h, rgb.ctypes.data, self.width, self.height, self.buf.ctypes.data), "ck_cat_thresh")
        return self.buf

    def detect_corners(self, cap=None):
        cap = cap or self.width * self.height
        pts = np.zeros((cap, 2), np.uint32)
        n = C.c_int32(0)
        check(self._L.ck_cat_detect_corners(self._det._h, self.buf.ctypes.data, self.width, self.height, pts.ctypes.data, cap, C.byref(n)),
              "ck_cat_detect_corners")
        self.points = pts[:min(n.value, cap)].copy()
        return self.points

    def check_edges(self, cap=None):
        cap = cap or max(16, 2 * len(self.points) ** 2)
        lines = np.zeros((cap, 4), np.uint32)
        n = C.c_int32(0)
        pts = np.ascontiguousarray(self.points, np.uint32)
        check(self._L.ck_cat_check_edges(self._det._h, self.buf.ctypes.data, self.width, self.height, pts.ctypes.data, len(pts),
                                         lines.ctypes.data, cap, C.byref(n)), "ck_cat_check_edges")
        self.lines = lines[:min(n.value, cap)].copy()
        return self.lines

    def process_frame(self, rgb, point_cap=None, line_cap=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        if rgb.size != self.width * self.height * 3:
            raise AssertionError("input is not width*height*3 bytes")  # assert_eq! in the reference (lib.rs:267)
        point_cap = point_cap or 65536
        line_cap = line_cap or 1 << 20
        pts = np.zeros((point_cap, 2), np.uint32)
        lines = np.zeros((line_cap, 4), np.uint32)
        npn, nl = C.c_int32(0), C.c_int32(0)
        check(self._L.ck_cat_process_frame(self._det._h, rgb.ctypes.data, rgb.size, self.width, self.height, self.buf.ctypes.data,
                                           pts.ctypes.data, point_cap, C.byref(npn), lines.ctypes.data, line_cap, C.byref(nl)),
              "ck_cat_process_frame")
        self.points = pts[:min(npn.value, point_cap)].copy()
        self.lines = lines[:min(nl.value, line_cap)].copy()
        return self.points, self.lines

    def connected_components(self):
        roots = np.zeros((self.height, self.width), np.uint32)
        sizes = np.zeros((self.height, self.width), np.uint32)
        check(self._L.ck_cat_connected_components(self._det._h, self.buf.ctypes.data, self.width, self.height, roots.ctypes.data,
                                                  sizes.ctypes.data), "ck_cat_connected_components")
        return UnionFind(roots, sizes)
