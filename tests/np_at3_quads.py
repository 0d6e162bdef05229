"""An AprilTag-3-STYLE quad fit in floating point (numpy / python floats), written independently of oracle/detector.c.

Why it exists: oracle/detector.c departs from AprilTag-3 in ways chosen so that a GPU can be bit-identical:
integer weights isqrt(gx^2+gy^2)+1 and int64 moments, an exact 60-bit angular sort key, one reciprocal instead of five
divisions in the line fit, a closed-form line normal instead of atan2f/cosf/sinf in edge refinement.  Nothing in the
reference pins how far those choices move corners (the detector is an external C library).  This file restates
the SAME stages the way AprilTag-3 does them — atan2-ordered points, double-precision moments with W = sqrt(gx^2+gy^2)+1,
divisions, float32 trigonometry for the refinement normal — so that tests/test_at3_float_delta.py can measure the distance
between the two on the golden scenes and on noisy frames.  It consumes the oracle's clusters (threshold, segmentation and
clustering are integer stages with nothing to approximate) and produces quads in the oracle's format.

Not a port of apriltag's source (which is not in this image): written from the published algorithm, same stage structure
and defaults as SURVEY.md Appendix B.
"""
import math

import numpy as np

K_SMOOTH = [math.exp(-(j * j) / 2.0) for j in (-3, -2, -1, 0, 1, 2, 3)]   # sigma = 1, cut at 0.05


def _fit_line(pre, sz, i0, i1, want_line):
    """pre: (sz, 6) prefix sums Mx, My, Mxx, Mxy, Myy, W in pixel units.  Returns (line | None, err, mse)."""
    if i0 < i1:
        m = pre[i1].copy()
        if i0 > 0:
            m -= pre[i0 - 1]
        n = i1 - i0 + 1
    else:
        m = pre[sz - 1] - pre[i0 - 1] + pre[i1]
        n = sz - i0 + i1 + 1
    mx, my, mxx, mxy, myy, w = m
    ex, ey = mx / w, my / w
    cxx, cxy, cyy = mxx / w - ex * ex, mxy / w - ex * ey, myy / w - ey * ey
    disc = math.sqrt((cxx - cyy) * (cxx - cyy) + 4.0 * cxy * cxy)
    eig_small = 0.5 * (cxx + cyy - disc)
    line = None
    if want_line:
        eig = 0.5 * (cxx + cyy + disc)
        nx1, ny1 = cxx - eig, cxy
        nx2, ny2 = cxy, cyy - eig
        m1, m2 = nx1 * nx1 + ny1 * ny1, nx2 * nx2 + ny2 * ny2
        nx, ny, mm = (nx1, ny1, m1) if m1 > m2 else (nx2, ny2, m2)
        ln = math.sqrt(mm)
        line = (ex, ey, 0.0, 0.0) if ln < 1e-12 else (ex, ey, nx / ln, ny / ln)
    return line, n * eig_small, eig_small


def fit_quad(pts, qim, cfg, min_tag_width, normal_ok=True, reversed_ok=False):
    """pts: structured array (x, y half-pixel, gx, gy) of one cluster.  Returns (4x2 corners, reversed) or None."""
    sz0 = len(pts)
    qh, qw = qim.shape
    if sz0 < max(cfg.min_cluster_pixels, 24) or sz0 > 3 * (2 * qw + 2 * qh):
        return None
    x = pts["x"].astype(np.float64)
    y = pts["y"].astype(np.float64)
    xmin, xmax, ymin, ymax = x.min(), x.max(), y.min(), y.max()
    if (xmax - xmin) * (ymax - ymin) < min_tag_width:
        return None
    # AprilTag-3: centre of the bounding box, nudged off the lattice
    cx, cy = (xmin + xmax) * 0.5 + 0.05118, (ymin + ymax) * 0.5 - 0.028581
    dot = float(np.sum((x - cx) * pts["gx"] + (y - cy) * pts["gy"]))
    reversed_border = dot < 0
    if (reversed_border and not reversed_ok) or (not reversed_border and not normal_ok):
        return None
    # order by angle about the centre (AprilTag-3 uses a quadrant + slope key: any monotone function of the angle gives the
    # same order); start where the oracle's key starts (dx < 0, dy < 0, angle measured the same way round)
    ang = np.arctan2(-(y - cy), -(x - cx))
    order = np.lexsort((y, x, ang))
    xs, ys = x[order], y[order]
    keep = np.ones(sz0, bool)
    keep[1:] = (xs[1:] != xs[:-1]) | (ys[1:] != ys[:-1])
    xs, ys = xs[keep], ys[keep]
    sz = len(xs)
    if sz < 24:
        return None
    # weights from the image gradient at the boundary point, pixel coordinates with the +0.5 pixel-centre offset
    ix = ((xs + 1) // 2).astype(int)
    iy = ((ys + 1) // 2).astype(int)
    w = np.ones(sz)
    inside = (ix > 0) & (ix + 1 < qw) & (iy > 0) & (iy + 1 < qh)
    q = qim.astype(np.float64)
    gx = q[iy[inside], ix[inside] + 1] - q[iy[inside], ix[inside] - 1]
    gy = q[iy[inside] + 1, ix[inside]] - q[iy[inside] - 1, ix[inside]]
    w[inside] = np.sqrt(gx * gx + gy * gy) + 1.0
    px, py = xs * 0.5 + 0.5, ys * 0.5 + 0.5
    pre = np.cumsum(np.stack([w * px, w * py, w * px * px, w * px * py, w * py * py, w], 1), 0)
    ksz = min(20, sz // 12)
    if ksz < 2:
        return None
    errs = np.array([_fit_line(pre, sz, (i + sz - ksz) % sz, (i + ksz) % sz, False)[1] for i in range(sz)])
    sm = np.zeros(sz)
    for j in range(7):
        sm += np.roll(errs, 3 - j) * K_SMOOTH[j]
    maxima = [i for i in range(sz) if sm[i] > sm[(i + 1) % sz] and sm[i] > sm[(i - 1) % sz]]
    if len(maxima) < 4:
        return None
    if len(maxima) > cfg.max_nmaxima:
        thr = sorted((sm[i] for i in maxima), reverse=True)[cfg.max_nmaxima]
        maxima = [i for i in maxima if sm[i] > thr]
    best, best_err = None, math.inf
    nm = len(maxima)
    for a in range(nm - 3):
        for b in range(a + 1, nm - 2):
            l01, e01, s01 = _fit_line(pre, sz, maxima[a], maxima[b], True)
            if s01 > cfg.max_line_fit_mse:
                continue
            for c in range(b + 1, nm - 1):
                l12, e12, s12 = _fit_line(pre, sz, maxima[b], maxima[c], True)
                if s12 > cfg.max_line_fit_mse or abs(l01[2] * l12[2] + l01[3] * l12[3]) > cfg.cos_critical_rad:
                    continue
                for d in range(c + 1, nm):
                    _, e23, s23 = _fit_line(pre, sz, maxima[c], maxima[d], False)
                    if s23 > cfg.max_line_fit_mse:
                        continue
                    _, e30, s30 = _fit_line(pre, sz, maxima[d], maxima[a], False)
                    if s30 > cfg.max_line_fit_mse:
                        continue
                    e = e01 + e12 + e23 + e30
                    if e < best_err:
                        best_err, best = e, (maxima[a], maxima[b], maxima[c], maxima[d])
    if best is None or best_err / sz >= cfg.max_line_fit_mse:
        return None
    lines = []
    for i in range(4):
        ln, _, mse = _fit_line(pre, sz, best[i], best[(i + 1) & 3], True)
        if mse > cfg.max_line_fit_mse:
            return None
        lines.append(ln)
    corners = np.zeros((4, 2))
    for i in range(4):
        li, lj = lines[i], lines[(i + 1) & 3]
        a00, a01, a10, a11 = li[3], -lj[3], -li[2], lj[2]
        b0, b1 = -li[0] + lj[0], -li[1] + lj[1]
        det = a00 * a11 - a10 * a01
        if abs(det) < 0.001:
            return None
        l0 = (a11 / det) * b0 + (-a01 / det) * b1
        corners[i] = (li[0] + l0 * a00, li[1] + l0 * a10)
    area = 0.0
    for t in ((0, 1, 2), (2, 3, 0)):
        ln = [np.linalg.norm(corners[t[(i + 1) % 3]] - corners[t[i]]) for i in range(3)]
        p = sum(ln) / 2.0
        area += math.sqrt(max(0.0, p * (p - ln[0]) * (p - ln[1]) * (p - ln[2])))
    if area < 0.95 * min_tag_width * min_tag_width:
        return None
    for i in range(4):
        d1 = corners[(i + 1) & 3] - corners[i]
        d2 = corners[(i + 2) & 3] - corners[(i + 1) & 3]
        cs = float(d1 @ d2) / math.sqrt(float(d1 @ d1) * float(d2 @ d2))
        if abs(cs) > cfg.cos_critical_rad or d1[0] * d2[1] < d1[1] * d2[0]:
            return None
    return corners, reversed_border


def refine_edges(im, corners, reversed_border, decimate=1):
    """AprilTag-3's edge refinement with its float32 trigonometry for the fitted line's normal."""
    h, w = im.shape
    lines = []
    for edge in range(4):
        a, b = edge, (edge + 1) & 3
        nx, ny = corners[b][1] - corners[a][1], -corners[b][0] + corners[a][0]
        mag = math.hypot(nx, ny)
        nx, ny = nx / mag, ny / mag
        if reversed_border:
            nx, ny = -nx, -ny
        nsamples = max(16, int(mag / 8.0))
        mx = my = mxx = mxy = myy = n = 0.0
        for s in range(nsamples):
            alpha = (1.0 + s) / (nsamples + 1.0)
            x0 = alpha * corners[a][0] + (1 - alpha) * corners[b][0]
            y0 = alpha * corners[a][1] + (1 - alpha) * corners[b][1]
            mn = mcount = 0.0
            rng = decimate + 1
            for k in range(-rng, rng + 1):
                x1, y1 = int(x0 + (k + 1.0) * nx), int(y0 + (k + 1.0) * ny)
                x2, y2 = int(x0 + (k - 1.0) * nx), int(y0 + (k - 1.0) * ny)
                if not (0 <= x1 < w and 0 <= y1 < h and 0 <= x2 < w and 0 <= y2 < h):
                    continue
                g1, g2 = int(im[y1, x1]), int(im[y2, x2])
                if g1 < g2:
                    continue
                wt = float((g2 - g1) * (g2 - g1))
                mn += wt * k
                mcount += wt
            if mcount == 0:
                continue
            n0 = mn / mcount
            bx, by = x0 + n0 * nx, y0 + n0 * ny
            mx += bx; my += by; mxx += bx * bx; mxy += bx * by; myy += by * by; n += 1.0
        if n < 2:
            lines.append((0.5 * (corners[a][0] + corners[b][0]), 0.5 * (corners[a][1] + corners[b][1]), nx, ny))
            continue
        ex, ey = mx / n, my / n
        cxx, cxy, cyy = mxx / n - ex * ex, mxy / n - ex * ey, myy / n - ey * ey
        theta = np.float32(0.5) * np.arctan2(np.float32(-2.0 * cxy), np.float32(cyy - cxx))   # atan2f
        lines.append((ex, ey, float(np.cos(np.float32(theta))), float(np.sin(np.float32(theta)))))   # cosf / sinf
    out = np.array(corners, float)
    for i in range(4):
        li, lj = lines[i], lines[(i + 1) & 3]
        a00, a01, a10, a11 = li[3], -lj[3], -li[2], lj[2]
        b0, b1 = -li[0] + lj[0], -li[1] + lj[1]
        det = a00 * a11 - a10 * a01
        if abs(det) > 0.001:
            l0 = (a11 / det) * b0 + (-a01 / det) * b1
            out[(i + 1) & 3] = (li[0] + l0 * a00, li[1] + l0 * a10)
    return out
