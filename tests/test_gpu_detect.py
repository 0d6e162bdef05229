"""GPU parity for the irregular stages: gradient clusters, quad fit (+edge refinement), decode — bit-exact vs the oracle."""
import ctypes as C

import numpy as np
import pytest

from chalkydri_amd import _abi as A
from chalkydri_amd import default_config, synth

pytestmark = pytest.mark.gpu


def _synth(cfg_idx, w, h, n, n_tags, **kw):
    frames, truths = synth.render_batch(cfg_idx, n, w, h, n_tags, **kw)
    return frames, truths


def _cluster_dict(cl, pts, lo=24, hi=1 << 30):
    out = {}
    for rep0, rep1, start, count in cl:
        if count < lo or count > hi:
            continue
        p = pts[start:start + count]
        arr = np.stack([p["x"].astype(np.int64), p["y"].astype(np.int64), p["gx"].astype(np.int64), p["gy"].astype(np.int64)], 1)
        order = np.lexsort((arr[:, 3], arr[:, 2], arr[:, 1], arr[:, 0]))
        out[(int(rep0), int(rep1))] = arr[order]
    return out


@pytest.mark.parametrize("w,h,n_tags,kw", [
    (640, 480, 4, {}),
    (640, 480, 3, {"noise_amp": 1}),
    (272, 200, 2, {"min_side": 24, "max_side": 60}),
    (1280, 800, 6, {}),
])
def test_clusters_match_oracle(oracle, w, h, n_tags, kw):
    from chalkydri_amd.detector import AprilTagDetector
    n = 2
    frames, _ = _synth(11, w, h, n, n_tags, **kw)
    det = AprilTagDetector(w, h, max_batch=n)
    got = det.clusters(frames)
    maxpts = 3 * (2 * w + 2 * h)
    for i in range(n):
        th = oracle.threshold(frames[i])
        lab, sz = oracle.segment(th)
        ocl, opts, ov = oracle.clusters(th, lab, sz)
        assert not ov
        want = _cluster_dict(ocl, opts, 24, maxpts)
        have = _cluster_dict(*got[i])
        assert set(want) == set(have), f"cluster keys differ: {len(want)} vs {len(have)}"
        for k in want:
            assert np.array_equal(want[k], have[k]), f"points of cluster {k} differ"
    det.close()


def _quads_np(quads):
    arr = np.zeros((len(quads), 11))
    for i, q in enumerate(quads):
        arr[i, :8] = [q.p[k][j] for k in range(4) for j in range(2)]
        arr[i, 8:] = [q.reversed_border, q.rep0, q.rep1]
    if len(arr):
        arr = arr[np.lexsort((arr[:, 10], arr[:, 9]))]
    return arr


@pytest.mark.parametrize("w,h,n_tags,kw", [
    (640, 480, 4, {}),
    (640, 480, 4, {"noise_amp": 1}),
    (272, 200, 2, {"min_side": 24, "max_side": 60}),
    (1280, 800, 6, {}),
])
def test_quads_match_oracle(oracle, w, h, n_tags, kw):
    from chalkydri_amd.detector import AprilTagDetector
    n = 2
    frames, _ = _synth(12, w, h, n, n_tags, **kw)
    det = AprilTagDetector(w, h, max_batch=n)
    got = det.quads(frames)
    cfg = default_config(w, h)
    for i in range(n):
        th = oracle.threshold(frames[i])
        lab, sz = oracle.segment(th)
        ocl, opts, _ = oracle.clusters(th, lab, sz)
        oq, ov = oracle.fit_quads(frames[i], cfg, ocl, opts)
        want, have = _quads_np(oq), _quads_np(got[i])
        assert want.shape == have.shape, f"{len(want)} oracle quads vs {len(have)} device quads"
        assert np.array_equal(want, have), f"max |diff| = {np.abs(want - have).max()}"
    det.close()


def _shapes(w, h, seed):
    """Hand-drawn frame that drives the sort of every size class through both of its paths: thin bars and spokes put
    hundreds of boundary points into one angle bucket (bitonic fallback), blobs and rings spread them out (bucket path);
    the outline rectangles are large-class clusters."""
    rng = np.random.default_rng(seed)
    im = np.full((h, w), 40, np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    im[20:23, 30:w - 40] = 220                                  # long thin bars
    im[40:h - 30, 14:17] = 220
    im[(np.abs((yy - 60) - (xx - 40) * 0.31) < 1.6) & (xx > 40) & (xx < w - 60)] = 220   # thin slanted line
    im[60:h - 20, 60:w - 20] = 215                              # big plate ...
    im[70:h - 30, 70:w - 30] = 35                               # ... hollowed: two long outlines
    cx, cy = w // 2, h // 2 + 10
    r = np.hypot(xx - cx, yy - cy)
    im[(r < 0.28 * h) & (r > 0.2 * h)] = 225                    # ring
    for k in range(12):                                         # spokes inside the ring
        a = k * np.pi / 6 + 0.1
        d = np.abs((xx - cx) * np.sin(a) - (yy - cy) * np.cos(a))
        im[(d < 1.3) & (r < 0.18 * h) & (r > 6)] = 225
    for _ in range(10):                                         # filled quadrilaterals of assorted sizes
        x0, y0 = rng.integers(80, w - 160), rng.integers(80, h - 120)
        sw, sh = rng.integers(14, 70), rng.integers(14, 70)
        sk = rng.uniform(-0.4, 0.4)
        m = (np.abs((xx - x0) - sk * (yy - y0)) < sw / 2) & (np.abs(yy - y0) < sh / 2)
        im[m] = 228 if rng.random() < 0.5 else 20
    noise = rng.integers(-1, 2, im.shape)
    return np.clip(im.astype(np.int64) + noise, 0, 255).astype(np.uint8)


# the two large frames have outlines of more than 16384 points (at most 3 * (2w + 2h), AprilTag-3's bound): the largest size class,
# whose per-point arrays live in global memory
@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (1280, 800, 2), (1920, 1080, 4), (2448, 2048, 3)])
def test_quads_match_oracle_on_drawn_shapes(oracle, w, h, seed):
    from chalkydri_amd.detector import AprilTagDetector
    frames = np.stack([_shapes(w, h, seed), _shapes(w, h, seed + 10)[::-1].copy()])
    det = AprilTagDetector(w, h, max_batch=2)
    got = det.quads(frames)
    cfg = default_config(w, h)
    sizes, nq = [], 0
    for i in range(2):
        th = oracle.threshold(frames[i])
        lab, sz = oracle.segment(th)
        ocl, opts, _ = oracle.clusters(th, lab, sz)
        sizes += [int(c[3]) for c in ocl]
        oq, ov = oracle.fit_quads(frames[i], cfg, ocl, opts)
        want, have = _quads_np(oq), _quads_np(got[i])
        assert want.shape == have.shape, f"{len(want)} oracle quads vs {len(have)} device quads"
        assert np.array_equal(want, have), f"max |diff| = {np.abs(want - have).max()}"
        nq += len(want)
    assert nq >= 3
    # all three size classes of the fit kernel were exercised
    assert any(s <= 512 for s in sizes) and any(512 < s <= 4096 for s in sizes) and any(4096 < s <= 16384 for s in sizes)
    if 3 * (2 * w + 2 * h) > 16384:
        assert any(16384 < s <= 3 * (2 * w + 2 * h) for s in sizes)
    det.close()


def _same_dets(have, want):
    assert len(have) == len(want), f"{len(have)} vs {len(want)} detections"
    for a, b in zip(have, want):
        assert (a.id(), a.hamming(), a.family()) == (b["id"], b["hamming"], b["family"])
        assert np.float32(a.decision_margin()) == np.float32(b["margin"])
        assert np.array_equal(a.center(), b["c"]) and np.array_equal(a.corners(), b["p"])


@pytest.mark.parametrize("w,h,n_tags,fams,bits,kw", [
    (640, 480, 4, ("tag36h11",), 3, {}),
    (640, 480, 4, ("tag36h11",), 3, {"noise_amp": 0, "ramp_amp": 0}),
    (640, 480, 4, ("tag16h5", "tag36h11"), 1, {"family_mode": 1}),   # 16h5 (distance 5) cannot take 3 corrected bits
    (1280, 800, 6, ("tag36h11",), 3, {}),
    # BASELINE config 5: 5 MP, mixed families (16h5 at one corrected bit needs larger tags to survive the noise)
    (2448, 2048, 20, ("tag16h5", "tag36h11"), 1, {"family_mode": 1, "truth_min_side": 40}),
])
def test_detect_matches_oracle_and_truth(oracle, w, h, n_tags, fams, bits, kw):
    from chalkydri_amd.detector import AprilTagDetector
    n = 3
    kw = dict(kw)
    truth_min_side = kw.pop("truth_min_side", 28)
    frames, truths = synth.render_batch(13, n, w, h, n_tags, fams, **kw)
    det = AprilTagDetector(w, h, max_batch=n, families=fams, bits_corrected=bits)
    got, status = det.detect_batch(frames, cap=64, return_status=True)
    cfg = default_config(w, h, families=fams, max_hamming=bits)
    for i in range(n):
        want, st = oracle.detect(frames[i], cfg)
        assert status[i] == st and not (st & ~A.CK_FRAME_UNVERIFIED_ID)   # random ids may lie past the verified prefix
        assert bool(st & A.CK_FRAME_UNVERIFIED_ID) == any(d["id"] >= cfg.families[d["family"]].contents.n_upstream for d in want)
        _same_dets(got[i], want)
        # and both agree with the renderer's ground truth: every rendered tag found, corners within 1.5 px
        for t in truths[i]:
            sides = [np.linalg.norm(t["corners"][k] - t["corners"][(k + 1) % 4]) for k in range(4)]
            xs, ys = np.asarray(t["corners"])[:, 0], np.asarray(t["corners"])[:, 1]
            area = 0.5 * abs(np.dot(xs, np.roll(ys, -1)) - np.dot(ys, np.roll(xs, -1)))
            if min(sides) < truth_min_side or area < 0.35 * max(sides) ** 2:
                continue   # strongly foreshortened / tiny / near-degenerate tags may be missed or loose
            cand = [d for d in got[i] if (d.family(), d.id()) == (t["family"], t["id"])]
            assert cand, f"tag {t['id']} missed"
            d = min(cand, key=lambda d: np.abs(d.center() - t["center"]).max())
            assert np.abs(d.corners() - t["corners"]).max() < 1.5
    det.close()


def test_config5_full_size_batch_properties(oracle):
    """BASELINE config 5 at full size: 2448x2048 x 256 frames, 20 tags of mixed families (a 60 GB handle, the per-colour split of
    the merge kernel and the largest fit class at batch scale).  8 distinct frames in shuffled positions: every copy of a frame
    gives the same detections (bytes) wherever it sits, no frame reports an overflow, and the 8 distinct results equal the oracle's."""
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n, uniq, fams, bits = 2448, 2048, 256, 8, ("tag16h5", "tag36h11"), 1
    frames8, truths = synth.render_batch(55, uniq, w, h, 20, fams, family_mode=1)
    rng = np.random.default_rng(5)
    which = rng.permutation(np.repeat(np.arange(uniq), n // uniq))
    det = AprilTagDetector(w, h, max_batch=n, families=fams, bits_corrected=bits)
    got, status = det.detect_batch(frames8[which], cap=64, return_status=True)
    status = np.asarray(status, np.uint32)
    assert not np.any(status & np.uint32(~A.CK_FRAME_UNVERIFIED_ID & 0xFFFFFFFF)), "a frame reported an overflow"
    key = lambda ds: [(d.family(), d.id(), d.hamming(), np.float32(d.decision_margin()).tobytes(), d.corners().tobytes()) for d in ds]
    first = {}
    for i in range(n):
        k = int(which[i])
        if k in first:
            assert key(got[i]) == key(got[first[k]]) and status[i] == status[first[k]], f"frame copy {i} of {k} differs"
        else:
            first[k] = i
    cfg = default_config(w, h, families=fams, max_hamming=bits)
    found = 0
    for k in range(uniq):
        want, st = oracle.detect(frames8[k], cfg)
        assert status[first[k]] == st
        _same_dets(got[first[k]], want)
        found += len(want)
    assert found >= 8 * 10   # most of the 20 rendered tags per frame decode
    det.close()


def test_detect_is_deterministic_and_batch_invariant(oracle):
    from chalkydri_amd.detector import AprilTagDetector
    w, h = 640, 480
    frames, _ = synth.render_batch(14, 4, w, h, 4)
    det = AprilTagDetector(w, h, max_batch=4)
    a = det.detect_batch(frames)
    b = det.detect_batch(frames)
    single = [det.detect(frames[i]) for i in range(4)]
    for x, y, z in zip(a, b, single):
        assert [(d.id(), d.corners().tobytes()) for d in x] == [(d.id(), d.corners().tobytes()) for d in y]
        assert [(d.id(), d.corners().tobytes()) for d in x] == [(d.id(), d.corners().tobytes()) for d in z]
    det.close()


def test_detector_golden_vectors(built):
    """The HIP detector against the committed vectors directly (tests/golden/detector_golden.json; no oracle in the loop):
    ids, hamming, f32 margin and f64 corners bit for bit, including the decimate-2 and mixed-family cases."""
    import json
    import os
    import zlib
    from chalkydri_amd.detector import AprilTagDetector
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detector_golden.json")))
    for g in golden:
        c = g["case"]
        frame, _ = synth.render(synth.frame_seed(c["seed_cfg"], c["frame"]), c["w"], c["h"], c["n_tags"], tuple(c["families"]), **c["params"])
        assert zlib.crc32(frame.tobytes()) == g["frame_crc32"], "renderer output changed"
        det = AprilTagDetector(c["w"], c["h"], max_batch=1, families=tuple(c["families"]), bits_corrected=c["bits"], quad_decimate=c["decimate"])
        got, status = det.detect_batch(frame[None], cap=64, return_status=True)
        det.close()
        assert int(status[0]) == g["status"] and len(got[0]) == len(g["detections"]), c["name"]
        for d, e in zip(got[0], g["detections"]):
            assert (d.family(), d.id(), d.hamming()) == (e["family"], e["id"], e["hamming"])
            assert np.float32(d.decision_margin()) == np.float32(e["margin"])
            assert [float.fromhex(v) for v in e["center"]] == list(d.center())
            assert [[float.fromhex(v) for v in p] for p in e["corners"]] == np.asarray(d.corners()).tolist()


@pytest.mark.parametrize("w,h", [(16, 16), (32, 20), (68, 52), (132, 68), (260, 72), (516, 36),
                                 (64, 18), (132, 70), (260, 135), (640, 483),    # these four: height not a multiple of 4
                                 (261, 135), (322, 94), (643, 481)])            # width not a multiple of 4: the weight image's last, partial group
def test_small_and_ragged_frames(oracle, w, h):
    """Frames much smaller than a 64x128 tile and not multiples of it: binary noise, blobs and a frame-filling square —
    labels, clusters, quads and detections still equal the oracle's (nothing reads or writes outside the frame).  Heights
    that are not multiples of 4 leave rows below the last whole 4x4 tile: they take that tile's threshold, as in the oracle."""
    from chalkydri_amd.detector import AprilTagDetector
    rng = np.random.default_rng(w * 1000 + h)
    f0 = rng.integers(0, 256, (h, w), dtype=np.uint8)                                   # full-contrast noise
    f1 = np.full((h, w), 30, np.uint8); f1[2:h - 2, 2:w - 2] = 220; f1[4:h - 4, 4:w - 4] = 25   # nested frame-sized squares
    f2 = np.clip(128 + 100 * np.sin(np.add.outer(np.arange(h) * 0.9, np.arange(w) * 0.7)) + rng.integers(-2, 3, (h, w)), 0, 255).astype(np.uint8)
    frames = np.stack([f0, f1, f2])
    det = AprilTagDetector(w, h, max_batch=3)
    cfg = default_config(w, h)
    labels, sizes = det.segment(frames)
    quads = det.quads(frames)
    dets, status = det.detect_batch(frames, cap=16, return_status=True)
    for i in range(3):
        th = oracle.threshold(frames[i])
        lab, sz = oracle.segment(th)
        assert np.array_equal(labels[i], lab) and np.array_equal(sizes[i], sz)
        ocl, opts, _ = oracle.clusters(th, lab, sz)
        oq, _ = oracle.fit_quads(frames[i], cfg, ocl, opts)
        assert np.array_equal(_quads_np(oq), _quads_np(quads[i]))
        want, st = oracle.detect(frames[i], cfg)
        assert status[i] == st
        _same_dets(dets[i], want)
    det.close()


def test_capacity_overflow_is_a_status_bit(oracle):
    """Undersized workspaces never abort or touch memory they do not own: the frame that overflows is flagged
    (CK_FRAME_*_OVERFLOW), its batch neighbours still equal the oracle."""
    from chalkydri_amd.detector import AprilTagDetector
    w, h = 640, 480
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    tags, _ = synth.render(synth.frame_seed(21, 0), w, h, 4, noise_amp=1)
    many, _ = synth.render(synth.frame_seed(21, 1), w, h, 12, noise_amp=0, ramp_amp=0, min_side=40, max_side=70)
    cfg = default_config(w, h)
    want_tags, _ = oracle.detect(tags, cfg)
    # points / clusters: a noise frame needs far more than 20 000 points and 2 048 hash slots
    det = AprilTagDetector(w, h, max_batch=3, max_points_per_frame=20000, max_clusters_per_frame=1024)
    dets, status = det.detect_batch(np.stack([tags, noise, tags]), cap=64, return_status=True)
    assert status[1] & (A.CK_FRAME_POINTS_OVERFLOW | A.CK_FRAME_CLUSTERS_OVERFLOW)
    assert not (status[0] | status[2]) & ~A.CK_FRAME_UNVERIFIED_ID
    _same_dets(dets[0], want_tags); _same_dets(dets[2], want_tags)
    det.close()
    # points only: the clusters that k_scan has no room for are holes in the cluster table; they must read as empty, not as what
    # an earlier call (or the allocator) left there — the fit once followed such a record out of its buffers
    # (tests/stress_detect.py with STRESS_CAPS=1 found it).  Calls alternate so that every hole has a stale record under it.
    det = AprilTagDetector(w, h, max_batch=2, max_points_per_frame=50000, max_clusters_per_frame=1024)
    for frames in ([tags, tags], [noise, tags], [tags, noise], [noise, noise], [tags, tags]):
        dets, status = det.detect_batch(np.stack(frames), cap=64, return_status=True)
        for f, d, st in zip(frames, dets, status):
            if f is tags:
                assert not st & ~A.CK_FRAME_UNVERIFIED_ID
                _same_dets(d, want_tags)
            else:
                assert st & (A.CK_FRAME_POINTS_OVERFLOW | A.CK_FRAME_CLUSTERS_OVERFLOW)
    det.close()
    # quads: twelve clean tags against room for four candidate quads
    det = AprilTagDetector(w, h, max_batch=2, max_quads_per_frame=4)
    dets, status = det.detect_batch(np.stack([many, tags]), cap=64, return_status=True)
    assert status[0] & A.CK_FRAME_QUADS_OVERFLOW and len(dets[0]) <= 4
    det.close()
    # detections: the caller's per-frame capacity
    det = AprilTagDetector(w, h, max_batch=1)
    dets, status = det.detect_batch(many[None], cap=2, return_status=True)
    assert status[0] & A.CK_FRAME_DETS_OVERFLOW and len(dets[0]) == 2
    det.close()


@pytest.mark.parametrize("kw", [{"max_nmaxima": 12}, {"max_nmaxima": 4}, {"refine_edges": 0}, {"max_line_fit_mse": 4.0, "cos_critical_rad": 0.9}])
def test_quads_match_oracle_under_other_settings(oracle, kw):
    """Quad-fit parameters away from their defaults (12 maxima = 66 forward pairs, more than a wave; 4 = a single subset;
    no edge refinement; tighter thresholds): quads and detections still equal the oracle's."""
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n = 640, 480, 2
    frames, _ = _synth(15, w, h, n, 4, noise_amp=3)
    det = AprilTagDetector(w, h, max_batch=n, **kw)
    got = det.quads(frames)
    dets, status = det.detect_batch(frames, cap=64, return_status=True)
    cfg = default_config(w, h, **kw)
    for i in range(n):
        th = oracle.threshold(frames[i])
        lab, sz = oracle.segment(th)
        ocl, opts, _ = oracle.clusters(th, lab, sz)
        oq, _ = oracle.fit_quads(frames[i], cfg, ocl, opts)
        assert np.array_equal(_quads_np(oq), _quads_np(got[i]))
        want, st = oracle.detect(frames[i], cfg)
        assert status[i] == st
        _same_dets(dets[i], want)
    det.close()


@pytest.mark.parametrize("kw", [{"min_white_black_diff": 20}, {"min_component_px": 60, "min_cluster_pixels": 80}, {"decode_sharpening": 0.0}])
def test_detect_matches_oracle_under_other_front_end_settings(oracle, kw):
    """Threshold contrast gate, component / cluster size gates and decode sharpening away from their defaults."""
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n = 640, 480, 2
    frames, _ = _synth(16, w, h, n, 4, noise_amp=3)
    det = AprilTagDetector(w, h, max_batch=n, **kw)
    cfg = default_config(w, h, **kw)
    thr = det.threshold(frames)
    labels, sizes = det.segment(frames)
    dets, status = det.detect_batch(frames, cap=64, return_status=True)
    for i in range(n):
        oth = oracle.threshold(frames[i], cfg.min_white_black_diff)
        assert np.array_equal(thr[i], oth)
        lab, sz = oracle.segment(oth)
        assert np.array_equal(labels[i], lab) and np.array_equal(sizes[i], sz)
        want, st = oracle.detect(frames[i], cfg)
        assert status[i] == st
        _same_dets(dets[i], want)
    det.close()


@pytest.mark.parametrize("kind", ["checker1", "vstripes1", "random", "saturated", "tag_in_random"])
def test_adversarial_frames_match_oracle(oracle, kind):
    """Frames built to stress the lock-free phases rather than to look like a camera image: the whole pipeline must return what
    the oracle returns (detections, status bits), not hang and not fault."""
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n = 640, 480, 2
    rng = np.random.default_rng(77)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "checker1":
        frames = np.broadcast_to((((yy + xx) & 1) * 255).astype(np.uint8), (n, h, w)).copy()
    elif kind == "vstripes1":
        frames = np.zeros((n, h, w), np.uint8); frames[:, :, ::2] = 255
    elif kind == "random":
        frames = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    elif kind == "saturated":
        frames = np.full((n, h, w), 255, np.uint8); frames[1] = 0
    else:   # a real tag scene with every second row replaced by full-range noise
        frames, _ = synth.render_batch(15, n, w, h, 4)
        frames = frames.copy(); frames[:, ::2, :] = rng.integers(0, 256, (n, h // 2, w), dtype=np.uint8)
    det = AprilTagDetector(w, h, max_batch=n)
    got, status = det.detect_batch(frames, cap=64, return_status=True)
    cfg = default_config(w, h)
    for i in range(n):
        want, st = oracle.detect(frames[i], cfg)
        full = 1 | 2   # CK_FRAME_POINTS_OVERFLOW | CK_FRAME_CLUSTERS_OVERFLOW: the oracle reports both kinds as the first
        assert bool(status[i] & full) == bool(st & full)
        if not (st & full):   # which clusters survive a full buffer is not part of the contract; that it is reported is
            assert status[i] == st
            _same_dets(got[i], want)
    det.close()


# 652 / 2 = 326: two rows below the last whole 4x4 tile; 600 / 2 = 300 and 472 / 2 = 236 are not multiples of 16: the decimated
# copy's rows are padded (a width that was a multiple of 16 hid a row-pitch mismatch until tools/stress_detect.py found it)
@pytest.mark.parametrize("w,h", [(640, 480), (1280, 800), (800, 652), (600, 404), (472, 232)])
def test_detect_matches_oracle_at_the_default_decimation(oracle, w, h):
    """quad_decimate = 2 is the detector's library default and therefore what the reference runs (it never changes detector
    settings, crates/apriltags/src/lib.rs:258-262): threshold / segmentation / clusters / fit on the half-size image, edge
    refinement and decoding on the full one — detections and status bits equal the oracle's."""
    from chalkydri_amd.detector import AprilTagDetector
    n = 2
    frames, _ = _synth(17, w, h, n, 4, noise_amp=3)
    det = AprilTagDetector(w, h, max_batch=n, quad_decimate=2)
    cfg = default_config(w, h, quad_decimate=2)
    dets, status = det.detect_batch(frames, cap=64, return_status=True)
    for i in range(n):
        want, st = oracle.detect(frames[i], cfg)
        assert status[i] == st and len(want) >= 2
        _same_dets(dets[i], want)
    det.close()


def test_caller_supplied_family_table(oracle):
    """The path INTEGRATION.md §3 gives integrators for upstream's tag36h11 table: a ck_family_t built by the caller.  Here the
    caller's table is the built-in one with its ids permuted (and marked wholly upstream): the frames still show the built-in
    codes, so every tag must come back under the PERMUTED id, corners unchanged, and without CK_FRAME_UNVERIFIED_ID — while
    the built-in table flags every frame that holds an id past its verified prefix."""
    from chalkydri_amd import family
    from chalkydri_amd.detector import AprilTagDetector
    w, h, n = 640, 480, 3
    frames, truths = synth.render_batch(29, n, w, h, 4)
    src = family("tag36h11").contents
    nc = src.ncodes
    perm = np.random.default_rng(5).permutation(nc)                 # new id i holds the code of built-in id perm[i]
    inv = np.argsort(perm)
    codes = (C.c_uint64 * nc)(*[src.codes[int(perm[i])] for i in range(nc)])
    mine = A.Family()
    C.memmove(C.byref(mine), C.byref(src), C.sizeof(A.Family))
    mine.name = b"tag36h11"
    mine.codes = C.cast(codes, C.POINTER(C.c_uint64))
    mine.n_upstream = nc
    mine_p = C.pointer(mine)
    det0 = AprilTagDetector(w, h, max_batch=n)
    base, st0 = det0.detect_batch(frames, return_status=True)
    det0.close()
    det = AprilTagDetector(w, h, max_batch=n, families=(mine_p,))
    got, st = det.detect_batch(frames, return_status=True)
    cfg = default_config(w, h, families=(mine_p,))
    for i in range(n):
        assert len(base[i]) >= 3 and len(got[i]) == len(base[i])
        assert sorted((int(inv[d.id()]), d.corners().tobytes()) for d in base[i]) == sorted((d.id(), d.corners().tobytes()) for d in got[i])
        assert st[i] == 0
        assert bool(st0[i] & A.CK_FRAME_UNVERIFIED_ID) == any(d.id() >= src.n_upstream for d in base[i])
        want, ost = oracle.detect(frames[i], cfg)
        assert ost == 0
        _same_dets(got[i], want)
    assert any(s & A.CK_FRAME_UNVERIFIED_ID for s in st0)           # the scene holds ids >= 39 (random ids out of 587)
    det.close()
