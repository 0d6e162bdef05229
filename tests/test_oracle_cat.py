"""CPU: oracle/cat.c against an independent numpy restatement of crates/chalkydri-apriltags (small images)."""
import numpy as np
import pytest


def _gray(rgb):
    # fmaf(r, .33f, fmaf(g, .33f, b*.33f)) evaluated exactly: products/sums of these magnitudes are exact in f64
    k = np.float32(0.33).astype(np.float64)
    r, g, b = [rgb[..., i].astype(np.float64) for i in range(3)]
    t = (b * k).astype(np.float32).astype(np.float64)
    t = (g * k + t).astype(np.float32).astype(np.float64)
    t = (r * k + t).astype(np.float32)
    return np.clip(np.trunc(t), 0, 255).astype(np.uint8)


def _otsu(rgb):
    g = _gray(rgb).astype(np.float64)
    h, w = g.shape
    out = np.zeros((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            win = np.sort(g[max(0, y - 2):min(h, y + 3), max(0, x - 2):min(w, x + 3)].ravel())
            n = len(win)
            p = g[y, x]
            if y > 0 and x > 0 and win[-1] - win[0] < 5.0:
                m = win[n // 2] if n % 2 else (win[n // 2 - 1] + win[n // 2]) / 2.0
                out[y, x] = 0 if m < 60 else (1 if m > 160 else 2)
            else:
                def q(tau):
                    hh = (n + 1.0 / 3.0) * tau + 1.0 / 3.0
                    hf = int(hh)
                    if hf <= 0:
                        return win[0]
                    if hf >= n:
                        return win[-1]
                    return win[hf - 1] + (hh - hf) * (win[hf] - win[hf - 1])
                uq, lq = int(np.clip(np.trunc(q(0.75)), 0, 255)), int(np.clip(np.trunc(q(0.25)), 0, 255))
                out[y, x] = 1 if p >= uq else (0 if p <= lq else 2)
    return out


@pytest.mark.parametrize("seed", range(3))
def test_calc_otsu_and_thresh(oracle, seed):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    rgb[5:15, 8:20] = rng.integers(100, 104, (10, 12, 3))     # a low-contrast patch -> median branch
    assert np.array_equal(oracle.cat_calc_otsu(rgb), _otsu(rgb))
    g = _gray(rgb)
    assert np.array_equal(oracle.cat_thresh(rgb), np.where(g < 60, 0, np.where(g > 160, 1, 2)).astype(np.uint8))


def test_quartiles_match_numpy_r8(oracle):
    # statrs' quantile is the R-8 ("median_unbiased") estimator: cross-check the restated formula on window sizes the detector sees
    rng = np.random.default_rng(2)
    for n in (9, 12, 15, 16, 20, 25):
        x = np.sort(rng.integers(0, 256, n).astype(float))
        for tau in (0.25, 0.75):
            hh = (n + 1 / 3) * tau + 1 / 3
            hf = int(hh)
            mine = x[0] if hf <= 0 else x[-1] if hf >= n else x[hf - 1] + (hh - hf) * (x[hf] - x[hf - 1])
            assert abs(mine - np.quantile(x, tau, method="median_unbiased")) < 1e-9


def test_corners_edges_components(oracle):
    rng = np.random.default_rng(4)
    cls = rng.choice(np.array([0, 1, 2], np.uint8), size=(40, 56), p=[0.45, 0.45, 0.1])
    cls[10:30, 12:40] = 1
    cls[14:26, 16:36] = 0                                        # a black square on white: its corners are CAT corners
    pts, n = oracle.cat_detect_corners(cls)
    assert n == len(pts)
    # x-major order, inside the domain, and each one satisfies the diagonal parity rule
    keys = pts[:, 0].astype(int) * 1000 + pts[:, 1]
    assert (np.diff(keys) > 0).all()
    for x, y in pts:
        assert cls[y, x] == 0
        d = [cls[y - 1, x - 1] == 0, cls[y - 1, x + 1] == 0, cls[y + 1, x - 1] == 0, cls[y + 1, x + 1] == 0]
        assert sum(d) % 2 == 1
    lines, nl = oracle.cat_check_edges(cls, pts[:40])
    assert nl == len(lines) and all((l[:2] == pts[:40]).all(1).any() for l in lines)
    roots, sizes = oracle.cat_connected_components(cls)
    parent, psz = oracle.cat_connected_components_reference(cls)
    # the canonical view and the reference's union-by-size forest describe the same partition with the same sizes
    def find(i):
        while parent[i] != i:
            i = parent[i]
        return i
    rep = np.array([int(find(i)) for i in range(cls.size)], dtype=np.int64)
    r = roots.ravel()
    assert len(set(zip(rep.tolist(), r.tolist()))) == len(set(rep.tolist())) == len(set(r.tolist()))
    assert (sizes.ravel() == psz[rep]).all()
    assert (roots.ravel()[cls.ravel() == 2] == np.arange(cls.size)[cls.ravel() == 2]).all()


def test_golden_vectors(oracle):
    """oracle/cat.c reproduces the committed CRCs of every CAT output (tests/golden/cat_golden.json) — all integer, exact."""
    import golden_util as G
    for c in G.load("cat_golden.json"):
        rgb = G.cat_rgb(c)
        cls = oracle.cat_calc_otsu(rgb)
        pts, npn = oracle.cat_detect_corners(cls)
        lines, nl = oracle.cat_check_edges(cls, pts)
        roots, sizes = oracle.cat_connected_components(cls)
        assert G.crc(oracle.cat_thresh(rgb), np.uint8) == c["thresh_crc32"] and G.crc(cls, np.uint8) == c["classes_crc32"]
        assert (npn, G.crc(pts, np.uint32)) == (c["n_points"], c["points_crc32"])
        assert (nl, G.crc(lines, np.uint32)) == (c["n_lines"], c["lines_crc32"])
        assert G.crc(roots, np.uint32) == c["roots_crc32"] and G.crc(sizes, np.uint32) == c["sizes_crc32"]
