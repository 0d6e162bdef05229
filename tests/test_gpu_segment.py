"""GPU parity: adaptive threshold + union-find segmentation (HIP) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from chalkydri_amd import synth

pytestmark = pytest.mark.gpu


def _frames(kind, w, h, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "synth":
        return np.stack([synth.render(synth.frame_seed(9, seed * 100 + i), w, h, 3, min_side=24,
                                      max_side=min(120, h // 3))[0] for i in range(n)])
    if kind == "noise":      # every tile has contrast: dense binary noise, worst case for the union-find
        return rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    if kind == "flat":       # contrast below min_white_black_diff everywhere -> all 127
        return np.full((n, h, w), 90, np.uint8) + rng.integers(0, 3, (n, h, w), dtype=np.uint8)
    if kind == "stripes":    # long runs crossing many tiles, plus vertical bars
        f = np.zeros((n, h, w), np.uint8)
        f[:, ::7, :] = 255
        f[:, :, ::13] = 255
        f[:, h // 2:, :] = 255 - f[:, h // 2:, :]
        return f
    if kind == "blobs":      # smooth random field -> large irregular components spanning tiles
        base = rng.random((n, h // 8 + 2, w // 8 + 2))
        up = np.kron(base, np.ones((8, 8)))[:, :h, :w]
        return (up * 255).astype(np.uint8)
    if kind == "spiral":
        f = np.zeros((n, h, w), np.uint8)
        for k in range(0, min(h, w) // 2 - 2, 4):
            f[:, k:h - k, k] = 255; f[:, k, k:w - k] = 255
            f[:, k + 2:h - k, w - 1 - k] = 255; f[:, h - 1 - k, k + 2:w - k] = 255
        return f
    if kind == "checker1":   # one-pixel checkerboard: every black pixel its own component, all white pixels one (8-connected)
        yy, xx = np.mgrid[0:h, 0:w]
        return np.broadcast_to((((yy + xx) & 1) * 255).astype(np.uint8), (n, h, w)).copy()
    if kind == "vstripes1":  # one-pixel vertical stripes: the most runs a row can have, every one spanning all tiles of its column
        f = np.zeros((n, h, w), np.uint8)
        f[:, :, ::2] = 255
        return f
    raise ValueError(kind)


# ragged sizes: widths that are not multiples of 4 / 16 / 32, a last tile column of 1..3 pixels (129, 131), a last tile row of
# 1..3 rows (33, 35, 450 = 14 * 32 + 2): any image_u8_t geometry the reference accepts (crates/apriltags/src/lib.rs:204-209)
@pytest.mark.parametrize("w,h", [(640, 480), (272, 200), (128, 64), (132, 68), (1280, 800), (129, 33), (131, 35), (641, 450), (258, 98), (61, 47)])
@pytest.mark.parametrize("kind", ["synth", "noise", "flat", "stripes", "blobs", "spiral", "checker1", "vstripes1"])
def test_threshold_segment_bit_exact(oracle, w, h, kind):
    from chalkydri_amd.detector import AprilTagDetector
    n = 2
    frames = _frames(kind, w, h, n, 3)
    det = AprilTagDetector(w, h, max_batch=n)
    th = det.threshold(frames)
    labels, sizes = det.segment(frames)
    for i in range(n):
        oth = oracle.threshold(frames[i])
        assert np.array_equal(th[i], oth), f"threshold differs in {np.count_nonzero(th[i] != oth)} px"
        ol, osz = oracle.segment(oth)
        bad = np.count_nonzero(labels[i] != ol)
        assert bad == 0, f"{bad} label words differ"
        assert np.array_equal(sizes[i], osz)
    det.close()


MERGE_PATH_CASES = [("noise", 64), ("blobs", 64), ("spiral", 64), ("checker1", 64),
                    ("noise", 1200), ("noise", 2000), ("noise", 3000), ("checker1", 2000), ("spiral", 300)]


def merge_path_cases():
    """Runs in a child process against the diagnostics build (the knob does not exist in the product library)."""
    import os
    import pyoracle
    from chalkydri_amd import _lib
    from chalkydri_amd.detector import AprilTagDetector
    assert b"CK_FMERGE_CAP" in open(_lib.LIB_PATH, "rb").read(), "not the diagnostics build: " + _lib.LIB_PATH
    w, h, n = 640, 480, 2
    for kind, cap in MERGE_PATH_CASES:
        os.environ["CK_FMERGE_CAP"] = str(cap)   # (read per call)
        frames = _frames(kind, w, h, n, 5)
        det = AprilTagDetector(w, h, max_batch=n)
        labels, sizes = det.segment(frames)
        for i in range(n):
            ol, osz = pyoracle.segment(pyoracle.threshold(frames[i]))
            assert np.array_equal(labels[i], ol) and np.array_equal(sizes[i], osz), (kind, cap, i)
        det.close()
    print("MERGE_PATHS_OK", len(MERGE_PATH_CASES))


# (kind, width, height, CK_FMERGE_BAND_ROWS, CK_FMERGE_CAP): frames joined in bands of tile rows (k_fmerge per band, k_fseam / k_fapply across
# them), every band height from one tile row up, ragged last bands and last tile rows, and the bands' own paths forced small
BAND_CASES = [("noise", 640, 480, 1, 0), ("noise", 640, 480, 2, 0), ("noise", 640, 480, 7, 0), ("blobs", 640, 480, 3, 0), ("spiral", 640, 480, 1, 0),
              ("spiral", 640, 480, 4, 0), ("checker1", 640, 480, 2, 0), ("vstripes1", 641, 450, 1, 0), ("synth", 641, 450, 5, 0), ("noise", 129, 33, 1, 0),
              ("noise", 131, 99, 2, 0), ("noise", 640, 480, 2, 64), ("noise", 640, 480, 3, 600), ("spiral", 640, 480, 2, 64), ("stripes", 272, 200, 1, 0),
              ("noise", 1280, 800, 6, 0), ("blobs", 1280, 800, 9, 0)]


def band_cases():
    """Runs in a child process against the diagnostics build (the knobs do not exist in the product library)."""
    import os
    import pyoracle
    from chalkydri_amd import _lib, default_config, scenes
    from chalkydri_amd.detector import AprilTagDetector
    assert b"CK_FMERGE_BAND_ROWS" in open(_lib.LIB_PATH, "rb").read(), "not the diagnostics build: " + _lib.LIB_PATH
    n = 2
    for kind, w, h, rows, cap in BAND_CASES:
        os.environ["CK_FMERGE_BAND_ROWS"] = str(rows)   # (both read per call)
        os.environ["CK_FMERGE_CAP"] = str(cap)
        frames = _frames(kind, w, h, n, 5)
        det = AprilTagDetector(w, h, max_batch=n)
        labels, sizes = det.segment(frames)
        for i in range(n):
            ol, osz = pyoracle.segment(pyoracle.threshold(frames[i]))
            assert np.array_equal(labels[i], ol) and np.array_equal(sizes[i], osz), (kind, w, h, rows, cap, i)
        det.close()
    # the slot tables' sizes decide which components the later stages see: whole detections on frames joined in bands
    os.environ["CK_FMERGE_CAP"] = "0"
    for rows in (1, 3, 8):
        os.environ["CK_FMERGE_BAND_ROWS"] = str(rows)
        w, h = 1280, 800
        frames = scenes.bench_stream(3, 2, w, h, 6)[0]
        det = AprilTagDetector(w, h, max_batch=2)
        dets = det.detect_batch(frames)
        cfg = default_config(w, h)
        for i in range(2):
            want, _ = pyoracle.detect(frames[i], cfg)
            assert [(d.id(), d.corners().tobytes()) for d in dets[i]] == [(d["id"], d["p"].tobytes()) for d in want], ("detections", rows, i)
            assert len(want) > 0
        det.close()
    print("BAND_CASES_OK", len(BAND_CASES))


def test_merge_paths_beyond_one_workgroups_lds(oracle):
    """A frame with more ring-touching roots than one merge workgroup's LDS holds (parents + keys) is joined by two workgroups,
    one per colour; a workgroup with still more roots keeps only the parents in LDS (up to twice the capacity), and beyond
    that runs in global memory.  Forced with small capacities (CK_FMERGE_CAP, a knob of the diagnostics build: hence the child
    process) on maps whose components span many tiles: 640x480 noise has about 3 400 roots per colour, so 64 means global memory,
    2000 and 3000 parents-only, 1200 a mix."""
    import os, subprocess, sys
    from conftest import ROOT, diag_env
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import importlib.util as u; "
            "sp = u.spec_from_file_location('tseg', %r); m = u.module_from_spec(sp); sp.loader.exec_module(m); m.merge_path_cases()"
            % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=diag_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MERGE_PATHS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_frames_joined_in_bands_of_tile_rows(oracle):
    """A frame with more than 600 tiles is joined in bands of tile rows (k_fmerge per band, then k_fseam / k_fapply across the bands:
    2448x2048 in three, 4092x2200 in five — test_large_frames_merge_per_colour runs those).  Here the band height is forced
    (CK_FMERGE_BAND_ROWS, diagnostics build: hence the child process) on small and ragged frames, down to one tile row per band, with
    the bands' own merge paths forced too, and whole detections are compared on frames joined that way."""
    import os, subprocess, sys
    from conftest import ROOT, diag_env
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import importlib.util as u; "
            "sp = u.spec_from_file_location('tseg', %r); m = u.module_from_spec(sp); sp.loader.exec_module(m); m.band_cases()"
            % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=diag_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BAND_CASES_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("w,h", [(1920, 1080), (2448, 2048), (4092, 2200)])
def test_large_frames_merge_per_colour(oracle, w, h):
    """Dense noise at 1920x1080 has about 45 000 ring-touching roots per frame: two workgroups, one per colour, each in LDS;
    at 2448x2048 about 55 000 per colour: parents-only LDS; 4092x2200 has 2 208 tiles, whose per-tile arrays take LDS from the
    roots (the merge kernel's request once exceeded the 160 KB of a CU there) and whose roots run in global memory."""
    from chalkydri_amd.detector import AprilTagDetector
    frames = _frames("noise", w, h, 1, 7)
    frames[0, 200:900, 300:1500] = _frames("blobs", 1200, 700, 1, 8)[0]
    det = AprilTagDetector(w, h, max_batch=1)
    labels, sizes = det.segment(frames)
    ol, osz = oracle.segment(oracle.threshold(frames[0]))
    assert np.array_equal(labels[0], ol) and np.array_equal(sizes[0], osz)
    det.close()


def test_strided_input_and_determinism(oracle):
    from chalkydri_amd.detector import AprilTagDetector
    w, h = 640, 480
    padded = np.zeros((3, h, w + 37), np.uint8)
    frames = _frames("synth", w, h, 3, 5)
    padded[:, :, :w] = frames
    padded[:, :, w:] = 0xAB
    det = AprilTagDetector(w, h, max_batch=3)
    a = det.segment(padded[:, :, :w])[0]      # non-contiguous view: stride 677
    b = det.segment(frames)[0]
    c = det.segment(frames)[0]
    assert np.array_equal(a, b) and np.array_equal(b, c)
    det.close()


def test_fp64_ops_match_host():
    """sqrt / div / mul / add on the device are IEEE-exact and unfused (the bit-parity contract of the fp64 stages)."""
    from chalkydri_amd.detector import AprilTagDetector
    det = AprilTagDetector(64, 64)
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.random(200000) * 1e6, rng.random(200000), 10.0 ** rng.uniform(-300, 300, 100000)])
    b = np.concatenate([rng.random(200000) + 1e-3, rng.random(200000) * 1e3 + 1e-9, 10.0 ** rng.uniform(-150, 150, 100000)])
    assert np.array_equal(det.fp64_probe(3, a), np.sqrt(a))
    assert np.array_equal(det.fp64_probe(2, a, b), a / b)
    assert np.array_equal(det.fp64_probe(1, a, b), a * b)
    assert np.array_equal(det.fp64_probe(0, a, b), a + b)
    assert np.array_equal(det.fp64_probe(4, a, b), (a * b) + a)
    det.close()
