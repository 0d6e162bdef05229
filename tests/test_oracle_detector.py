"""CPU: the detector oracle against the committed golden vectors, the renderer's ground truth and independent references."""
import json
import os
import zlib

import numpy as np
import pytest

from chalkydri_amd import default_config, synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "detector_golden.json")))


@pytest.mark.parametrize("g", GOLDEN, ids=[g["case"]["name"] for g in GOLDEN])
def test_golden_vectors(oracle, g):
    c = g["case"]
    frame, truth = synth.render(synth.frame_seed(c["seed_cfg"], c["frame"]), c["w"], c["h"], c["n_tags"], tuple(c["families"]), **c["params"])
    assert zlib.crc32(frame.tobytes()) == g["frame_crc32"], "renderer output changed"
    cfg = default_config(c["w"], c["h"], families=tuple(c["families"]), max_hamming=c["bits"], quad_decimate=c["decimate"])
    dets, status = oracle.detect(frame, cfg)
    assert status == g["status"] and len(dets) == len(g["detections"])
    for d, e in zip(dets, g["detections"]):
        assert (d["family"], d["id"], d["hamming"]) == (e["family"], e["id"], e["hamming"])
        assert np.float32(d["margin"]) == np.float32(e["margin"])
        assert [float.fromhex(v) for v in e["center"]] == d["c"].tolist()
        assert [[float.fromhex(v) for v in p] for p in e["corners"]] == d["p"].tolist()
    # ground truth: every sufficiently large rendered tag is found with the right id and corner order
    for t in g["truth"]:
        tc = np.array(t["corners"])
        if min(np.linalg.norm(tc[k] - tc[(k + 1) % 4]) for k in range(4)) < (28 if c["decimate"] == 1 else 60):
            continue   # strongly foreshortened / tiny tags are allowed to be missed or loose
        cand = [d for d in dets if (d["family"], d["id"]) == (t["family"], t["id"])]
        assert cand, f"tag {t['id']} missed"
        assert min(np.abs(d["p"] - tc).max() for d in cand) < 1.5


def _bfs_labels(t):
    """Independent reference for the segmentation rule (CAT lib.rs:506-545): explicit edge list + flood fill."""
    h, w = t.shape
    n = h * w
    adj = [[] for _ in range(n)]
    for y in range(h):
        for x in range(1, w - 1):
            v = t[y, x]
            if v == 127:
                continue
            i = y * w + x
            nb = [(x - 1, y)]
            if y > 0:
                nb.append((x, y - 1))
                if v == 255:
                    nb += [(x - 1, y - 1), (x + 1, y - 1)]
            for xx, yy in nb:
                if t[yy, xx] == v:
                    j = yy * w + xx
                    adj[i].append(j); adj[j].append(i)
    lab = np.full(n, 0xFFFFFFFF, np.uint32)
    for s in range(n):
        if t.flat[s] == 127 or lab[s] != 0xFFFFFFFF:
            continue
        stack, comp = [s], [s]
        lab[s] = s
        while stack:
            a = stack.pop()
            for b in adj[a]:
                if lab[b] == 0xFFFFFFFF:
                    lab[b] = s; stack.append(b); comp.append(b)
        lab[comp] = min(comp)
    return lab.reshape(h, w)


@pytest.mark.parametrize("seed", range(4))
def test_segment_against_flood_fill(oracle, seed):
    rng = np.random.default_rng(seed)
    t = rng.choice(np.array([0, 127, 255], np.uint8), size=(37, 53), p=[0.42, 0.1, 0.48])
    lab, sz = oracle.segment(t)
    ref = _bfs_labels(t)
    assert np.array_equal(lab, ref)
    for r in np.unique(ref[ref != 0xFFFFFFFF]):
        assert (sz[ref == r] == np.count_nonzero(ref == r)).all()
    assert (sz[ref == 0xFFFFFFFF] == 0).all()


def test_threshold_rules(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    th = oracle.threshold(img)
    # numpy restatement: 4x4 tile min/max, 3x3 dilation, tri-state
    tmin = img.reshape(12, 4, 16, 4).min((1, 3)).astype(int)
    tmax = img.reshape(12, 4, 16, 4).max((1, 3)).astype(int)
    pmin = np.pad(tmin, 1, constant_values=255)
    pmax = np.pad(tmax, 1, constant_values=0)
    dmin = np.min([pmin[i:i + 12, j:j + 16] for i in range(3) for j in range(3)], 0)
    dmax = np.max([pmax[i:i + 12, j:j + 16] for i in range(3) for j in range(3)], 0)
    mn, mx = np.kron(dmin, np.ones((4, 4), int)), np.kron(dmax, np.ones((4, 4), int))
    want = np.where(mx - mn < 5, 127, np.where(img.astype(int) > mn + (mx - mn) // 2, 255, 0)).astype(np.uint8)
    assert np.array_equal(th, want)
    assert (oracle.threshold(np.full((32, 32), 77, np.uint8)) == 127).all()      # no contrast anywhere


def test_empty_and_cluttered_frames(oracle):
    w, h = 320, 240
    cfg = default_config(w, h)
    assert oracle.detect(np.full((h, w), 128, np.uint8), cfg)[0] == []
    noise = np.random.default_rng(1).integers(0, 256, (h, w), dtype=np.uint8)
    dets, st = oracle.detect(noise, cfg)          # pure noise: the 11-bit code distance keeps false positives away
    assert all(d["hamming"] <= 3 for d in dets) and len(dets) <= 1


def test_decimate2_corners_are_full_resolution(oracle):
    w, h = 640, 480
    frame, truth = synth.render(synth.frame_seed(1, 7), w, h, 3, min_side=80, max_side=200)
    cfg = default_config(w, h, quad_decimate=2)
    dets, _ = oracle.detect(frame, cfg)
    ids = {d["id"]: d for d in dets}
    for t in truth:
        assert t["id"] in ids
        assert np.abs(ids[t["id"]]["p"] - t["corners"]).max() < 2.0


def test_native_build_reproduces_the_goldens(oracle):
    """bench.py times the oracle as `-O3 -march=native -ffp-contract=off` (oracle/Makefile: native).  That build must be
    the same function as the -O2 checker: this file's golden-vector test and the SQPnP / CAT ones, re-run in a child process
    against it (CK_ORACLE_LIB)."""
    import subprocess, sys
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "native"])
    env = dict(os.environ, CK_ORACLE_LIB=os.path.join(root, "oracle", "libck_oracle_native.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-k", "golden_vectors",
                        os.path.join(HERE, "test_oracle_detector.py"), os.path.join(HERE, "test_sqpnp_oracle.py"), os.path.join(HERE, "test_oracle_cat.py")],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
