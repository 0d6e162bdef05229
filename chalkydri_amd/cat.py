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


# ---- src/utils.rs helpers.  None of them is on the device path (the reference never calls the last three either);
# ---- they are kept so that code written against the crate's `utils` module finds the same names and results.
def grayscale(r, g, b):
    """utils.rs:33-46: trunc(fma(r, 0.33f, fma(g, 0.33f, b * 0.33f))) in f32, saturating cast to u8."""
    k = np.float32(0.33)
    inner = np.float32(np.float64(np.float32(g)) * np.float64(k) + np.float64(np.float32(b) * k))   # single rounding = fmaf
    outer = np.float32(np.float64(np.float32(r)) * np.float64(k) + np.float64(inner))
    return int(min(255.0, max(0.0, np.trunc(outer))))


def fast_angle(p):
    """utils.rs:51-72: FAST ring position 1..16 -> degrees in steps of 22.5."""
    if not 1 <= p <= 16:
        raise ValueError("invalid FAST point")
    return (p - 1) * 22.5


COLLINEAR, CLOCKWISE, COUNTERCLOCKWISE = 0, 1, 2  # utils.rs:74-79 (derive order)


def orientation(p, q, r):
    """utils.rs:82-101: sign of (qy-py)(rx-qx) - (qx-px)(ry-qy) in i32."""
    v = (int(q[1]) - int(p[1])) * (int(r[0]) - int(q[0])) - (int(q[0]) - int(p[0])) * (int(r[1]) - int(q[1]))
    return COLLINEAR if v == 0 else (CLOCKWISE if v > 0 else COUNTERCLOCKWISE)


def find_convex_hull(points):
    """utils.rs:113-152 (gift wrapping from the left-most point, first one on ties; the last counter-clockwise candidate wins)."""
    n = len(points)
    if n == 0:
        raise IndexError("find_convex_hull of no points")   # the reference indexes points[0] and panics
    l = 0
    for i in range(n):
        if points[i][0] < points[l][0]:
            l = i
    hull, p = [], l
    while p != l or not hull:
        hull.append(tuple(points[p]))
        q = (p + 1) % n
        for i in range(n):
            if orientation(points[p], points[i], points[q]) == COUNTERCLOCKWISE:
                q = i
        p = q
        if len(hull) > n:   # collinear duplicates can make the reference loop forever; stop where it would start repeating
            break
    return hull


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
        self.width, self.height, self.valid_tags, self.device = width, height, tuple(valid_tags), device
        self._det = AprilTagDetector(width, height, max_batch=1, device=device)
        self._L = _bind(lib())
        self.buf = np.zeros((height, width), np.uint8)          # alloc_zeroed: all Black (lib.rs:166-167)
        self.points = np.zeros((0, 2), np.uint32)
        self.lines = np.zeros((0, 4), np.uint32)

    def close(self):
        self._det.close()

    def clone(self):
        """lib.rs:663-667: a fresh detector of the same size with no valid tags."""
        return CatDetector(self.width, self.height, (), device=self.device)

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

    def draw(self, path="lines.ppm"):
        """lib.rs:615-661: classes as black / white / 0x777777 and, in green, every checked line whose end points lie in
        one component.  The reference writes lines.png through `ril`; this writes a binary PPM (no PNG encoder here).
        Returns the drawn lines."""
        img = np.zeros((self.height, self.width, 3), np.uint8)
        img[self.buf == WHITE] = 255
        img[self.buf == OTHER] = 0x77
        uf = self.connected_components()
        drawn = []
        for x1, y1, x2, y2 in self.lines.tolist():
            if uf.find(y1 * self.width + x1) != uf.find(y2 * self.width + x2):
                continue
            drawn.append((x1, y1, x2, y2))
            n = max(abs(x2 - x1), abs(y2 - y1), 1)
            for k in range(n + 1):   # integer DDA
                x = x1 + ((x2 - x1) * k * 2 + n) // (2 * n) if x2 >= x1 else x1 - ((x1 - x2) * k * 2 + n) // (2 * n)
                y = y1 + ((y2 - y1) * k * 2 + n) // (2 * n) if y2 >= y1 else y1 - ((y1 - y2) * k * 2 + n) // (2 * n)
                img[y, x] = (0, 255, 0)
        if path:
            with open(path, "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (self.width, self.height))
                f.write(img.tobytes())
        return drawn
