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


def _consts(text):
    """name -> value text of every `const NAME: f64 = ...;` (an expression over earlier names is evaluated)"""
    out = {}
    for name, expr in re.findall(r"^\s*(?:pub )?const (\w+): f64 = ([^;]+);", text, flags=re.M):
        out[name] = float(eval(expr, {"__builtins__": {}}, dict(out)))
    return out


def test_shim_constants_equal_the_references():
    """every constant the shim re-declares has the reference's value (crates/chalkydri_sqpnp/src/lib.rs:29-39); the reference
    is read as text, and only here in the container — the GPU box has no /root/reference, the test is CPU-only"""
    ref_path = "/root/reference/crates/chalkydri_sqpnp/src/lib.rs"
    want = {"XY_STD_DEV_SCALAR": 5.0, "THETA_STD_DEV_SCALAR": 2.0, "MAX_TRUSTABLE_RMS": 0.1, "MAX_GYRO_DELTA": 30.0, "TAG_SIZE": 0.1651,
            "CORNER_DISTANCE": 0.08255}
    if os.path.exists(ref_path):
        ref = _consts(open(ref_path).read())
        assert {k: ref[k] for k in want} == want, "the reference's constants have changed: update this test and the shim"
    mine = _consts(open(os.path.join(ROOT, "rust", "chalkydri_sqpnp", "src", "lib.rs")).read())
    assert {k: mine.get(k) for k in want} == want


def test_nalgebra_feature_gates_the_typed_signatures():
    sq = open(os.path.join(ROOT, "rust", "chalkydri_sqpnp", "src", "lib.rs")).read()
    toml = open(os.path.join(ROOT, "rust", "chalkydri_sqpnp", "Cargo.toml")).read()
    assert 'nalgebra = ["dep:nalgebra"]' in toml and "optional = true" in toml
    gated = sq[sq.index('#[cfg(feature = "nalgebra")]'):]
    flat = re.sub(r"\s+", " ", gated)
    assert ("pub fn solve_robot_pose( &mut self, points_isometry: &[Isometry3<f64>], points_2d: &[Vec3], robot_to_cam: &Isometry3<f64>, "
            "gyro: f64, sign_change_error: f64, ) -> Option<(Rot3, Vec3, Vec3)>") in flat
    assert "pub fn create_solver_camera_transform( fwd_m: f64, left_m: f64, up_m: f64, roll_deg: f64, pitch_deg: f64, yaw_deg: f64, ) -> Iso3" in flat
