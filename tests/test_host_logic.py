"""CPU: host-side mirrors (field layout loader, wire record, sharding helpers) and a 2-rank gloo rehearsal of the pose gather."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_field_layout_loader_on_reference_field_json(built):
    from chalkydri_amd.apriltags import load_field_layout
    tags = load_field_layout(os.path.join(HERE, "golden", "field.json"))   # copy of the reference's field.json (data fixture)
    assert len(tags) == 32 and set(tags) == set(range(1, 33))
    t1 = tags[1]
    assert abs(t1.t[0] - 11.863959) < 1e-12 and abs(t1.t[2] - 0.889) < 1e-12
    q = np.array(t1.q[:])
    assert abs(np.linalg.norm(q) - 1) < 1e-15 and abs(q[3] - 1.0) < 1e-12   # W,X,Y,Z -> (w,x,y,z), normalised (field_layout.rs:36-38)


def test_shard_helpers():
    from chalkydri_amd import dist
    for n, world in [(256, 8), (257, 8), (5, 8), (1024, 3)]:
        blocks = [dist.shard_frames(n, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1
    assert dist.stream_of_rank(3, 8, 8) == [3] and dist.stream_of_rank(1, 8, 2) == [1, 3, 5, 7]


def test_record_view_roundtrip():
    from chalkydri_amd import _abi as A
    from chalkydri_amd import dist
    r = A.VisionMeasurement()
    r.pose_x, r.pose_y, r.pose_rot, r.std_x, r.ts, r.camera_id, r.tag_count = 1.5, -2.25, 0.75, 0.01, 123456, 7, 5
    raw = np.frombuffer(bytes(r), np.uint8).reshape(1, 64)
    v = dist.records_to_numpy(raw)[0]
    assert (v["pose_x"], v["pose_y"], v["pose_rot"], v["std_x"], v["ts"], v["camera_id"], v["tag_count"]) == (1.5, -2.25, 0.75, 0.01, 123456, 7, 5)
    # little-endian f64 at offset 0, ts at 48, ids at 56/57 — crates/whacknet/src/lib.rs:43-66
    assert raw[0, 56] == 7 and raw[0, 57] == 5 and np.frombuffer(raw[0, 48:56].tobytes(), "<u8")[0] == 123456


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from chalkydri_amd import dist
rank, local, world = dist.init("gloo")
lo, hi = dist.shard_frames(10, rank, world)
rec = torch.zeros((hi - lo, 64), dtype=torch.uint8)
for i in range(lo, hi):
    rec[i - lo, 0] = i          # frame index in byte 0
    rec[i - lo, 56] = rank      # camera_id
out = dist.gather_records(rec, world)
assert out.shape == (10, 64)
assert out[:, 0].tolist() == list(range(10)), out[:, 0].tolist()
assert out[:, 56].tolist() == [0] * 5 + [1] * 5
# ragged shards (11 frames over 2 ranks: 6 + 5): padded to the common row count, trimmed back to frame order
lo, hi = dist.shard_frames(11, rank, world)
rec = torch.zeros((hi - lo, 64), dtype=torch.uint8)
for i in range(lo, hi):
    rec[i - lo, 0] = i + 1
    rec[i - lo, 57] = 3         # tag_count
rows = dist.shard_rows(11, world)
out = dist.gather_records(rec, world, rows=rows)
assert rows == 6 and out.shape == (12, 64)
assert out[:, 57].tolist() == [3] * 6 + [3] * 5 + [0]          # the padding is the empty record (tag_count 0)
trimmed = dist.trim_gathered(out, 11, world)
assert trimmed.shape == (11, 64) and trimmed[:, 0].tolist() == list(range(1, 12))
import torch.distributed as td
td.barrier(); td.destroy_process_group()
print("ok", rank)
'''


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "ok 0" in outs[0] and "ok 1" in outs[1]


def test_cat_utils_helpers(oracle):
    """src/utils.rs helpers mirrored on the host: grayscale against the oracle's fmaf version over a value sweep,
    fast_angle, orientation and the gift-wrapping hull (utils.rs:33-46,51-72,82-101,113-152)."""
    from chalkydri_amd import cat
    L = oracle.lib()
    rng = np.random.default_rng(4)
    for r, g, b in rng.integers(0, 256, (4000, 3)).tolist() + [[v, v, v] for v in range(256)]:
        assert cat.grayscale(r, g, b) == L.ora_cat_grayscale(r, g, b), (r, g, b)
    assert [cat.fast_angle(p) for p in (1, 5, 9, 16)] == [0.0, 90.0, 180.0, 337.5]
    with pytest.raises(ValueError):
        cat.fast_angle(0)
    assert cat.orientation((0, 0), (4, 4), (8, 8)) == cat.COLLINEAR
    assert cat.orientation((0, 0), (4, 4), (8, 0)) == cat.CLOCKWISE
    assert cat.orientation((0, 0), (4, 4), (0, 8)) == cat.COUNTERCLOCKWISE
    pts = [(5, 5), (0, 0), (10, 0), (10, 10), (0, 10), (3, 7), (6, 2)]
    hull = cat.find_convex_hull(pts)
    assert sorted(hull) == [(0, 0), (0, 10), (10, 0), (10, 10)] and hull[0] == (0, 0)
    # walking the hull never turns clockwise-positive: every interior point is on one side of every edge
    for i in range(len(hull)):
        a, b = hull[i], hull[(i + 1) % len(hull)]
        assert all(cat.orientation(a, p, b) != cat.COUNTERCLOCKWISE for p in pts)


def test_whacknet_comm_loopback():
    """crates/whacknet/src/lib.rs:99-185: the listener thread keeps the last heading sent to its port (0.0 before the first one,
    short datagrams zero-extended), publish() puts exactly the 64 bytes of the record on the wire through the sender thread."""
    import socket, struct, time
    from chalkydri_amd import whacknet
    from chalkydri_amd._abi import VisionMeasurement
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.bind(("127.0.0.1", 0))
    rx.settimeout(2.0)
    comm = whacknet.Comm(gyro_port=0, remote=("127.0.0.1", rx.getsockname()[1]))
    try:
        assert comm.gyro_angle() == 0.0
        tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        heading = -2.356194490192345
        for _ in range(200):
            tx.sendto(struct.pack("<d", heading), ("127.0.0.1", comm.gyro_port))
            time.sleep(0.005)
            if comm.gyro_angle() == heading:
                break
        assert comm.gyro_angle() == heading
        comm.publish(4, 3, 0x0102030405060708, (1.0, -2.0, 0.5), (0.01, 0.02, 0.03))
        data = rx.recv(128)
        want = VisionMeasurement()
        want.pose_x, want.pose_y, want.pose_rot, want.std_x, want.std_y, want.std_rot = 1.0, -2.0, 0.5, 0.01, 0.02, 0.03
        want.ts, want.camera_id, want.tag_count = 0x0102030405060708, 4, 3
        assert len(data) == 64 and data == bytes(want)
        assert data[:8] == struct.pack("<d", 1.0) and data[48:56] == struct.pack("<Q", 0x0102030405060708) and data[56] == 4 and data[57] == 3
        assert whacknet.decode_gyro(struct.pack("<d", 1.25)) == 1.25 and whacknet.decode_gyro(b"\x00" * 4) is None
    finally:
        comm.close()
        rx.close()
