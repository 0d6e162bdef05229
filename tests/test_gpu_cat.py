"""GPU parity for CAT (crates/chalkydri-apriltags Detector): every method bit-exact vs oracle/cat.c."""
import numpy as np
import pytest

from chalkydri_amd import synth

pytestmark = pytest.mark.gpu


def _rgb(seed, w, h, kind):
    rng = np.random.default_rng(seed)
    if kind == "tags":
        g = synth.render(synth.frame_seed(5, seed), w, h, 3, min_side=40, max_side=min(150, h // 2), noise_amp=2)[0]
        rgb = np.stack([g, g, g], -1).astype(np.int16) + rng.integers(-3, 4, (h, w, 3))
        return np.clip(rgb, 0, 255).astype(np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == "flat":
        return np.full((h, w, 3), 200, np.uint8)
    if kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        g = (((yy // 9) + (xx // 9)) % 2 * 220 + 10).astype(np.uint8)
        return np.stack([g, g, g], -1)
    raise ValueError(kind)


@pytest.mark.parametrize("w,h", [(64, 48), (160, 120), (320, 240)])
@pytest.mark.parametrize("kind", ["tags", "noise", "flat", "checker"])
def test_cat_stages_bit_exact(oracle, w, h, kind):
    from chalkydri_amd.cat import CatDetector
    if kind == "tags" and h < 100:
        pytest.skip("tags need room")
    rgb = _rgb(3, w, h, kind)
    det = CatDetector(w, h)
    assert np.array_equal(det.thresh(rgb), oracle.cat_thresh(rgb))
    cls = det.calc_otsu(rgb).copy()
    ocls = oracle.cat_calc_otsu(rgb)
    assert np.array_equal(cls, ocls), f"{np.count_nonzero(cls != ocls)} classes differ"
    pts = det.detect_corners()
    opts, n = oracle.cat_detect_corners(ocls)
    assert n == len(opts) and np.array_equal(pts, opts)
    if len(opts) <= 600:                      # check_edges is O(P^2) in the reference
        lines = det.check_edges()
        olines, nl = oracle.cat_check_edges(ocls, opts)
        assert nl == len(olines) and np.array_equal(lines, olines)
    uf = det.connected_components()
    roots, sizes = oracle.cat_connected_components(ocls)
    assert np.array_equal(uf._roots.reshape(h, w), roots) and np.array_equal(uf._sizes.reshape(h, w), sizes)
    det.close()


def test_process_frame_and_assert(oracle):
    from chalkydri_amd.cat import CatDetector
    w, h = 320, 240
    rgb = _rgb(8, w, h, "tags")
    det = CatDetector(w, h)
    pts, lines = det.process_frame(rgb)
    ocls = oracle.cat_calc_otsu(rgb)
    opts, _ = oracle.cat_detect_corners(ocls)
    olines, _ = oracle.cat_check_edges(ocls, opts)
    assert np.array_equal(det.buf, ocls) and np.array_equal(pts, opts) and np.array_equal(lines, olines)
    with pytest.raises(AssertionError):
        det.process_frame(rgb[:, :-1])
    det.close()


def test_golden_vectors(built):
    """The HIP CAT path against the committed vectors directly (no oracle in the loop)."""
    import golden_util as G
    from chalkydri_amd.cat import CatDetector
    for c in G.load("cat_golden.json"):
        rgb = G.cat_rgb(c)
        det = CatDetector(c["w"], c["h"])
        assert G.crc(det.thresh(rgb), np.uint8) == c["thresh_crc32"]
        assert G.crc(det.calc_otsu(rgb), np.uint8) == c["classes_crc32"]
        pts = det.detect_corners()
        assert (len(pts), G.crc(pts, np.uint32)) == (c["n_points"], c["points_crc32"])
        lines = det.check_edges()
        assert (len(lines), G.crc(lines, np.uint32)) == (c["n_lines"], c["lines_crc32"])
        uf = det.connected_components()
        assert G.crc(uf._roots, np.uint32) == c["roots_crc32"] and G.crc(uf._sizes, np.uint32) == c["sizes_crc32"]
        det.close()


def test_draw_and_clone(oracle, tmp_path):
    """draw (lib.rs:615-661) keeps exactly the lines whose end points share a component; clone is a fresh detector."""
    from chalkydri_amd.cat import CatDetector
    w, h = 320, 240
    rgb = _rgb(8, w, h, "tags")
    det = CatDetector(w, h, valid_tags=(1, 2))
    det.process_frame(rgb)
    out = tmp_path / "lines.ppm"
    drawn = det.draw(str(out))
    roots, _ = oracle.cat_connected_components(oracle.cat_calc_otsu(rgb))
    want = [tuple(l) for l in det.lines.tolist() if roots[l[1], l[0]] == roots[l[3], l[2]]]
    assert drawn == want
    raw = out.read_bytes()
    head = b"P6\n%d %d\n255\n" % (w, h)
    assert raw.startswith(head) and len(raw) == len(head) + 3 * w * h
    px = np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3)
    for x1, y1, x2, y2 in drawn:
        assert tuple(px[y1, x1]) == (0, 255, 0) and tuple(px[y2, x2]) == (0, 255, 0)
    other = det.clone()
    assert (other.width, other.height, other.valid_tags) == (w, h, ()) and not other.buf.any() and len(other.points) == 0
    other.close()
    det.close()
