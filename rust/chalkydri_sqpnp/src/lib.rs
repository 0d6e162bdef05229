//! `chalkydri_sqpnp` over the MI355X library: the public surface of the reference crate
//! (crates/chalkydri_sqpnp/src/lib.rs — line numbers below refer to it) on plain arrays, forwarding to
//! `ck_sqpnp_solve_batch` / `ck_sqpnp_create_solver_camera_transform` (include/chalkydri_hip.h).
//!
//! The reference's types are nalgebra's (`Isometry3<f64>`, `Rotation3<f64>`, `SVector<f64, 3>`; lib.rs:15-26).  nalgebra is not
//! available offline, so the core API takes the same data as arrays: an isometry is a translation `[f64; 3]` plus a unit
//! quaternion `[f64; 4]` in (w, x, y, z) order — exactly what `Isometry3` holds — and a rotation is a row-major 3x3.
//! With the `nalgebra` feature (a maintainer who has the crate: `nalgebra = "0.34.1"`, crates/chalkydri_sqpnp/Cargo.toml:7) the
//! module `na` below offers the reference's own signatures (lib.rs:297-304, 430-437) as thin conversions onto this core.
use chalkydri_hip_sys as sys;

pub type Mat3 = [[f64; 3]; 3];
pub type Vec3 = [f64; 3];
/// translation + unit quaternion (w, x, y, z): the content of nalgebra's `Isometry3<f64>` (lib.rs:24)
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct Iso3 { pub translation: Vec3, pub rotation_wxyz: [f64; 4] }
/// row-major rotation matrix: the content of `Rotation3<f64>` (lib.rs:26)
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct Rot3(pub Mat3);
impl Rot3 {
    /// `.euler_angles().2` — the yaw the caller publishes (crates/apriltags/src/lib.rs:343)
    pub fn yaw(&self) -> f64 { self.0[1][0].atan2(self.0[0][0]) }
}
impl Iso3 {
    fn raw(&self) -> sys::ck_iso3_t { sys::ck_iso3_t { t: self.translation, q: self.rotation_wxyz } }
}

// lib.rs:29-39
pub const XY_STD_DEV_SCALAR: f64 = 5.0;
pub const THETA_STD_DEV_SCALAR: f64 = 2.0;
pub const MAX_TRUSTABLE_RMS: f64 = 0.1;
pub const MAX_GYRO_DELTA: f64 = 30.0; // degrees, as in the reference (lib.rs:35)
pub const TAG_SIZE: f64 = 0.1651;
pub const CORNER_DISTANCE: f64 = TAG_SIZE / 2.0;

/// lib.rs:183-192.  The buffers of the reference struct live on the device; what stays here is the configuration and the
/// handle the solves run on.
#[derive(Clone, Debug)]
pub struct SqPnP {
    max_iter: usize,
    tol_sq: f64,
    handle: *mut sys::ck_handle_t,
}
impl Default for SqPnP {
    fn default() -> Self { Self::new() }
}
impl SqPnP {
    /// lib.rs:201-212
    pub fn new() -> Self { Self { max_iter: 15, tol_sq: 1e-16, handle: std::ptr::null_mut() } }
    /// lib.rs:214-217
    pub const fn max_iter(mut self, max_iter: usize) -> Self { self.max_iter = max_iter; self }
    /// lib.rs:219-222
    pub const fn tolerance(mut self, tol: f64) -> Self { self.tol_sq = tol * tol; self }
    /// the device the solves run on: any live handle (a detector's `raw()`), which must outlive this solver
    pub fn on(mut self, handle: *mut sys::ck_handle_t) -> Self { self.handle = handle; self }

    /// lib.rs:297-377.  `points_isometry`: field poses of the seen tags; `points_2d`: four bearings per tag in the detector's
    /// corner order; returns (robot rotation, robot position, std devs) or None exactly where the reference does.
    pub fn solve_robot_pose(&mut self, points_isometry: &[Iso3], points_2d: &[Vec3], robot_to_cam: &Iso3, gyro: f64,
                            sign_change_error: f64) -> Option<(Rot3, Vec3, Vec3)> {
        assert!(!self.handle.is_null(), "SqPnP::on(handle) first: the solver runs on the GPU");
        let tags: Vec<sys::ck_iso3_t> = points_isometry.iter().map(Iso3::raw).collect();
        let bearings: Vec<f64> = points_2d.iter().flat_map(|v| v.iter().copied()).collect();
        let problem = sys::ck_sqpnp_problem_t {
            n_tags: tags.len() as i32, n_bearings: points_2d.len() as i32, tag_offset: 0, bearing_offset: 0,
            robot_to_cam: robot_to_cam.raw(), gyro, sign_change_error,
        };
        let params = sys::ck_sqpnp_params_t { max_iter: self.max_iter as i32, tol_sq: self.tol_sq };
        let mut out = unsafe { std::mem::zeroed::<sys::ck_sqpnp_result_t>() };
        let rc = unsafe {
            sys::ck_sqpnp_solve_batch(self.handle, &params, &problem, 1, tags.as_ptr(), tags.len() as i32, bearings.as_ptr(),
                                      points_2d.len() as i32, &mut out)
        };
        assert_eq!(rc, sys::CK_OK, "ck_sqpnp_solve_batch");
        if out.valid == 0 { return None; }
        let r = out.rot;
        Some((Rot3([[r[0], r[1], r[2]], [r[3], r[4], r[5]], [r[6], r[7], r[8]]]), out.pos, out.std_devs))
    }

    /// lib.rs:430-461: cam_cv <- robot from NWU offsets (metres) and roll / pitch / yaw (degrees); pure host arithmetic
    pub fn create_solver_camera_transform(fwd_m: f64, left_m: f64, up_m: f64, roll_deg: f64, pitch_deg: f64, yaw_deg: f64) -> Iso3 {
        let mut out = unsafe { std::mem::zeroed::<sys::ck_iso3_t>() };
        unsafe { sys::ck_sqpnp_create_solver_camera_transform(fwd_m, left_m, up_m, roll_deg, pitch_deg, yaw_deg, &mut out) };
        Iso3 { translation: out.t, rotation_wxyz: out.q }
    }
}


/// The reference's typed surface (lib.rs:15-26, 297-304, 430-437) for builds that have nalgebra: the same type aliases and the
/// two entry points with the reference's exact signatures, converting to and from the array core above.
#[cfg(feature = "nalgebra")]
pub mod na {
    use nalgebra::{Isometry3, Matrix3, Quaternion, Rotation3, SVector, Translation3, UnitQuaternion};

    pub type Mat3 = Matrix3<f64>;
    pub type Vec3 = SVector<f64, 3>;
    pub type Iso3 = Isometry3<f64>;
    pub type Rot3 = Rotation3<f64>;

    fn to_core(iso: &Isometry3<f64>) -> super::Iso3 {
        let q = iso.rotation.quaternion(); // coords are (i, j, k, w)
        super::Iso3 { translation: [iso.translation.x, iso.translation.y, iso.translation.z], rotation_wxyz: [q.w, q.i, q.j, q.k] }
    }
    fn from_core(iso: &super::Iso3) -> Isometry3<f64> {
        let [w, x, y, z] = iso.rotation_wxyz;
        Isometry3::from_parts(Translation3::new(iso.translation[0], iso.translation[1], iso.translation[2]),
                              UnitQuaternion::from_quaternion(Quaternion::new(w, x, y, z)))
    }

    /// `SqPnP` with the reference's method signatures; `Clone + Debug + Default` like the reference's
    #[derive(Clone, Debug, Default)]
    pub struct SqPnP(pub super::SqPnP);
    impl SqPnP {
        pub fn new() -> Self { Self(super::SqPnP::new()) }
        pub const fn max_iter(self, max_iter: usize) -> Self { Self(self.0.max_iter(max_iter)) }
        pub const fn tolerance(self, tol: f64) -> Self { Self(self.0.tolerance(tol)) }
        pub fn on(self, handle: *mut chalkydri_hip_sys::ck_handle_t) -> Self { Self(self.0.on(handle)) }

        /// lib.rs:297-304
        pub fn solve_robot_pose(
            &mut self,
            points_isometry: &[Isometry3<f64>],
            points_2d: &[Vec3],
            robot_to_cam: &Isometry3<f64>,
            gyro: f64,
            sign_change_error: f64,
        ) -> Option<(Rot3, Vec3, Vec3)> {
            let tags: Vec<super::Iso3> = points_isometry.iter().map(to_core).collect();
            let bearings: Vec<super::Vec3> = points_2d.iter().map(|v| [v[0], v[1], v[2]]).collect();
            let (rot, pos, dev) = self.0.solve_robot_pose(&tags, &bearings, &to_core(robot_to_cam), gyro, sign_change_error)?;
            let m = rot.0;
            let r = Matrix3::new(m[0][0], m[0][1], m[0][2], m[1][0], m[1][1], m[1][2], m[2][0], m[2][1], m[2][2]);
            Some((Rotation3::from_matrix_unchecked(r), Vec3::new(pos[0], pos[1], pos[2]), Vec3::new(dev[0], dev[1], dev[2])))
        }

        /// lib.rs:430-437
        pub fn create_solver_camera_transform(
            fwd_m: f64,
            left_m: f64,
            up_m: f64,
            roll_deg: f64,
            pitch_deg: f64,
            yaw_deg: f64,
        ) -> Iso3 {
            from_core(&super::SqPnP::create_solver_camera_transform(fwd_m, left_m, up_m, roll_deg, pitch_deg, yaw_deg))
        }
    }
}
