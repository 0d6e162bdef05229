"""CPU: the segmentation stage's connectivity rule as k_tile executes it (chalkydri_amd/csrc/ck_links.h: runs of 32-pixel row
words and their links to earlier runs) replayed through a sequential union-find and compared with the oracle's ora_segment on
thousands of random and adversarial tri-state maps (tests/cpp/links_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_link_rule_matches_oracle(oracle, tmp_path):
    exe = str(tmp_path / "links_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "cpp", "links_check.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "oracle"), "-lck_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    r = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
