"""CPU: the Rust FFI layer cannot drift from the C ABI.  rust/chalkydri_hip_sys/src/lib.rs is generated from
include/chalkydri_hip.h (tools/gen_rust_sys.py); the two safe crates keep the public surface of the reference crates.  No Rust
toolchain exists in this image, so nothing here compiles Rust: the checks are textual."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sys_crate_matches_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_abi_function_is_declared_once_with_the_right_arity():
    header = open(os.path.join(ROOT, "include", "chalkydri_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    rust = open(os.path.join(ROOT, "rust", "chalkydri_hip_sys", "src", "lib.rs")).read()
    protos = re.findall(r"\b(ck_\w+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)
    assert len(protos) >= 45
    for name, args in protos:
        n_c = 0 if args.strip() in ("", "void") else len(args.split(","))
        m = re.findall(r"pub fn %s\((.*?)\)" % re.escape(name), rust)
        assert len(m) == 1, name
        n_r = 0 if not m[0].strip() else len(m[0].split(","))
        assert n_c == n_r, (name, n_c, n_r)


def test_safe_crates_keep_the_reference_surface():
    cat = open(os.path.join(ROOT, "rust", "chalkydri-apriltags", "src", "lib.rs")).read()
    for sig in ("pub fn new(width: usize, height: usize, valid_tags: &'static [usize]) -> Self", "pub fn calc_otsu(&mut self, input: &mut [u8])",
                "pub fn process_frame(&mut self, input: &[u8])", "pub fn detect_corners(&mut self)", "pub unsafe fn thresh(", "pub fn check_edges(&mut self)",
                "pub fn connected_components(&self) -> UnionFind", "pub fn draw(&self)", "impl Clone for Detector", "impl Drop for Detector",
                "pub fn find(&mut self, id: usize) -> usize", "pub fn union(&mut self, id1: usize, id2: usize)", "pub fn get_size(&self, id: usize) -> usize",
                "pub mod utils", "pub fn detect("):
        assert sig in cat, sig
    sq = open(os.path.join(ROOT, "rust", "chalkydri_sqpnp", "src", "lib.rs")).read()
    for sig in ("pub fn new() -> Self", "pub const fn max_iter(mut self, max_iter: usize) -> Self", "pub const fn tolerance(mut self, tol: f64) -> Self",
                "pub fn solve_robot_pose(", "sign_change_error: f64", "-> Option<(Rot3, Vec3, Vec3)>", "pub fn create_solver_camera_transform(fwd_m: f64, left_m: f64, up_m: f64, roll_deg: f64, pitch_deg: f64, yaw_deg: f64) -> Iso3",
                "impl Default for SqPnP", "pub const TAG_SIZE: f64 = 0.1651"):
        assert sig in sq, sig
    utils = open(os.path.join(ROOT, "rust", "chalkydri-apriltags", "src", "utils.rs")).read()
    for name in ("fn grayscale", "fn fast_angle", "fn orientation", "fn find_convex_hull", "enum Color"):
        assert name in utils, name
