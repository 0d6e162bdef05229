"""Randomised parity stress, a small helping of it in the GPU suite (the scripts beside this file run thousands of cases:
`python tests/stress_detect.py 1500 23`).  Every case draws its own geometry, content and settings; the device path must agree
with the CPU oracle bit for bit (poses: within the stated tolerance).  These runs found the padded-row bug of the decimated
copy (widths whose half is not a multiple of 16) that the fixed-size tests had missed."""
import pytest

pytestmark = pytest.mark.gpu


def test_stress_segment(oracle):
    """(a child process against the diagnostics build: the cases draw CK_FMERGE_CAP, which the product library does not read)"""
    import os, subprocess, sys
    from conftest import diag_env
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "stress_segment.py"), "60", "101"], env=diag_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '"mismatching_frames": 0' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_stress_segment_product_library(oracle):
    """the same kind of cases on the product library (every frame takes the merge path its size asks for)"""
    import stress_segment
    assert stress_segment.run(25, 112) == 0


def test_stress_detect(oracle):
    import stress_detect
    assert stress_detect.run(40, 102) == 0


def test_stress_detect_small_capacities(oracle, monkeypatch):
    """Undersized point / cluster / quad capacities drawn at random: an overflow is a status bit on both sides, never a fault."""
    import stress_detect
    monkeypatch.setenv("STRESS_CAPS", "1")
    monkeypatch.setenv("CK_POISON", "1")   # the handles' buffers start as 0xA5 bytes: an entry read without having been written shows
    assert stress_detect.run(60, 107) == 0


def test_stress_pose(oracle):
    import stress_pose
    assert stress_pose.run(25, 103) == 0


def test_stress_cat(oracle):
    import stress_cat
    assert stress_cat.run(40, 104) == 0


def test_stress_batch(oracle):
    import stress_batch
    assert stress_batch.run(15, 105) == 0


def test_stress_ingest(built):
    import stress_ingest
    assert stress_ingest.run(15, 106) == 0


def test_stress_sqpnp(oracle):
    import stress_sqpnp
    assert stress_sqpnp.run(256, 108) == 0


@pytest.mark.parametrize("caps", [False, True])
def test_stress_detect_split_fit_on_small_calls(oracle, caps):
    """A call of a few frames runs the unsplit quad fit (its size classes side by side); a batch runs the split one (k_seq -> k_chunk ->
    k_tail).  CK_FIT_FLAT=2 (read once per process, hence the child) sends the small calls of the stress cases through the split
    path too — with undersized, poisoned buffers in the second run."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    from conftest import diag_env
    env = diag_env(CK_FIT_FLAT="2")   # (a knob of the diagnostics build)
    if caps: env.update(STRESS_CAPS="1", CK_POISON="1")
    r = subprocess.run([sys.executable, os.path.join(here, "stress_detect.py"), "40", "111" if caps else "109"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '"mismatching_frames": 0' in r.stdout, r.stdout[-2000:]
