//! `chalkydri-apriltags` over the MI355X library: the public surface of the reference crate
//! (crates/chalkydri-apriltags/src/lib.rs — line numbers below refer to it), forwarding to the C ABI of
//! `libchalkydri_hip.so` (include/chalkydri_hip.h).  No arithmetic happens here: every result is what the HIP kernels
//! return.  Zero crates.io dependencies; not compiled in the build image (it has no Rust toolchain) — the C ABI is the
//! tested contract and tests/test_rust_shim.py keeps the FFI declarations in step with the header.
//!
//! Additions the reference crate does not have (it stops at corners / edges / components and produces no tag ids):
//! `Detector::detect` / `detect_batch` with `Detection::{id, corners, center, hamming, decision_margin}`, mirroring the
//! `apriltag` crate calls the production path makes (crates/apriltags/src/lib.rs:301-314).
#![allow(clippy::missing_safety_doc)]

pub mod utils;

use chalkydri_hip_sys as sys;
use std::ffi::CStr;

fn check(rc: i32, what: &str) {
    if rc != sys::CK_OK {
        let msg = unsafe { CStr::from_ptr(sys::ck_strerror(rc)) }.to_string_lossy().into_owned();
        let detail = if rc == sys::CK_EDEVICE { unsafe { CStr::from_ptr(sys::ck_last_error()) }.to_string_lossy().into_owned() } else { String::new() };
        panic!("{what}: {msg} ({rc}) {detail}"); // the reference unwraps / expects at the same places (apriltags/src/lib.rs:228-262)
    }
}

/// Union-Find over pixel indices (lib.rs:42-113).  The device returns the finished partition (canonical root = smallest
/// index of the set, and the set's size); `union` on it is supported on the host for callers that keep merging.
#[derive(Debug, Clone)]
pub struct UnionFind {
    parent: Vec<usize>,
    cluster_sizes: Vec<usize>,
}
impl UnionFind {
    /// lib.rs:49-66: every element its own set of size 1
    pub fn new(len: usize) -> Self {
        Self { parent: (0..len).collect(), cluster_sizes: vec![1; len] }
    }
    fn from_device(roots: Vec<u32>, sizes: Vec<u32>) -> Self {
        Self { parent: roots.into_iter().map(|r| r as usize).collect(), cluster_sizes: sizes.into_iter().map(|s| s as usize).collect() }
    }
    /// lib.rs:67-77 (path compression)
    pub fn find(&mut self, id: usize) -> usize {
        let mut root = id;
        while self.parent[root] != root { root = self.parent[root]; }
        let mut cur = id;
        while self.parent[cur] != root { let next = self.parent[cur]; self.parent[cur] = root; cur = next; }
        root
    }
    /// lib.rs:78-95 (union by size, ties keep root1)
    pub fn union(&mut self, id1: usize, id2: usize) {
        let (r1, r2) = (self.find(id1), self.find(id2));
        if r1 == r2 { return; }
        let (s1, s2) = (self.cluster_sizes[r1], self.cluster_sizes[r2]);
        if s1 >= s2 { self.parent[r2] = r1; self.cluster_sizes[r1] = s1 + s2; } else { self.parent[r1] = r2; self.cluster_sizes[r2] = s1 + s2; }
    }
    /// lib.rs:96-98
    pub fn get_size(&self, id: usize) -> usize {
        let mut root = id;
        while self.parent[root] != root { root = self.parent[root]; }
        self.cluster_sizes[root]
    }
}

/// One decoded tag (what `apriltag::Detection` gives the reference, crates/apriltags/src/lib.rs:306-314).
#[derive(Clone, Copy)]
pub struct Detection(sys::ck_detection_t);
impl Detection {
    pub fn id(&self) -> usize { self.0.id as usize }
    pub fn corners(&self) -> [[f64; 2]; 4] { self.0.p }
    pub fn center(&self) -> [f64; 2] { self.0.c }
    pub fn hamming(&self) -> usize { self.0.hamming as usize }
    pub fn decision_margin(&self) -> f32 { self.0.decision_margin }
    /// false when the id lies past the verified prefix of the built-in table (ck_family_t.n_upstream)
    pub fn family(&self) -> usize { self.0.family as usize }
}

/// lib.rs:142-153.  `classes` replaces the raw `buf: *mut Color`, `points` / `lines` the raw point buffer and line list.
pub struct Detector {
    h: *mut sys::ck_handle_t,
    valid_tags: &'static [usize],
    classes: Vec<u8>,                         // Color as u8: 0 Black, 1 White, 2 Other (utils.rs:1-6)
    points: Vec<(usize, usize)>,
    lines: Vec<(usize, usize, usize, usize)>,
    width: usize,
    height: usize,
}
unsafe impl Send for Detector {} // lib.rs:136-137; one handle = one GPU + stream, `&mut self` keeps it single-threaded

impl Detector {
    /// lib.rs:158-181
    pub fn new(width: usize, height: usize, valid_tags: &'static [usize]) -> Self {
        let mut cfg = unsafe { std::mem::zeroed::<sys::ck_config_t>() };
        unsafe { sys::ck_config_default(&mut cfg, width as i32, height as i32, 1) };
        let mut h = std::ptr::null_mut();
        check(unsafe { sys::ck_create(&cfg, &mut h) }, "ck_create");
        Self { h, valid_tags, classes: vec![0; width * height], points: Vec::new(), lines: Vec::new(), width, height }
    }
    /// `DetectorBuilder::default().add_family_bits(family, bits).build()` of the production path (crates/apriltags/src/lib.rs:258-262)
    pub fn with_family(width: usize, height: usize, family: &str, bits_corrected: usize, max_batch: usize) -> Self {
        let mut cfg = unsafe { std::mem::zeroed::<sys::ck_config_t>() };
        unsafe { sys::ck_config_default(&mut cfg, width as i32, height as i32, max_batch as i32) };
        let name = std::ffi::CString::new(family).expect("family name");
        let fam = unsafe { sys::ck_family_builtin(name.as_ptr()) };
        assert!(!fam.is_null(), "unknown tag family {family}");
        cfg.families[0] = fam;
        cfg.n_families = 1;
        cfg.max_hamming = bits_corrected as i32;
        let mut h = std::ptr::null_mut();
        check(unsafe { sys::ck_create(&cfg, &mut h) }, "ck_create");
        Self { h, valid_tags: &[], classes: vec![0; width * height], points: Vec::new(), lines: Vec::new(), width, height }
    }
    /// lib.rs:191-259: 5x5 local-statistics tri-state classes of an RGB frame
    pub fn calc_otsu(&mut self, input: &mut [u8]) {
        check(unsafe { sys::ck_cat_calc_otsu(self.h, input.as_ptr(), self.width as i32, self.height as i32, self.classes.as_mut_ptr()) }, "ck_cat_calc_otsu");
    }
    /// lib.rs:265-287 (asserts the buffer length like the reference's `assert_eq!`, lib.rs:267)
    pub fn process_frame(&mut self, input: &[u8]) {
        assert_eq!(input.len(), self.width * self.height * 3);
        let cap = self.width * self.height;
        let (mut pts, mut lines) = (vec![0u32; 2 * cap.min(1 << 20)], vec![0u32; 4 * (1 << 20)]);
        let (mut np, mut nl) = (0i32, 0i32);
        check(unsafe {
            sys::ck_cat_process_frame(self.h, input.as_ptr(), input.len(), self.width as i32, self.height as i32, self.classes.as_mut_ptr(),
                                      pts.as_mut_ptr(), (pts.len() / 2) as i32, &mut np, lines.as_mut_ptr(), (lines.len() / 4) as i32, &mut nl)
        }, "ck_cat_process_frame");
        self.points = (0..np as usize).map(|i| (pts[2 * i] as usize, pts[2 * i + 1] as usize)).collect();
        self.lines = (0..nl as usize).map(|i| (lines[4 * i] as usize, lines[4 * i + 1] as usize, lines[4 * i + 2] as usize, lines[4 * i + 3] as usize)).collect();
    }
    /// lib.rs:291-309
    pub fn detect_corners(&mut self) {
        let cap = (self.width * self.height).min(1 << 20);
        let mut pts = vec![0u32; 2 * cap];
        let mut np = 0i32;
        check(unsafe { sys::ck_cat_detect_corners(self.h, self.classes.as_ptr(), self.width as i32, self.height as i32, pts.as_mut_ptr(), cap as i32, &mut np) }, "ck_cat_detect_corners");
        self.points = (0..np as usize).map(|i| (pts[2 * i] as usize, pts[2 * i + 1] as usize)).collect();
    }
    /// lib.rs:319-334 (fixed <60 / >160 split; `unsafe` kept from the reference signature)
    pub unsafe fn thresh(&mut self, input: &[u8]) {
        check(sys::ck_cat_thresh(self.h, input.as_ptr(), self.width as i32, self.height as i32, self.classes.as_mut_ptr()), "ck_cat_thresh");
    }
    /// lib.rs:480-499
    pub fn check_edges(&mut self) {
        let flat: Vec<u32> = self.points.iter().flat_map(|&(x, y)| [x as u32, y as u32]).collect();
        let cap = 1usize << 20;
        let mut lines = vec![0u32; 4 * cap];
        let mut nl = 0i32;
        check(unsafe { sys::ck_cat_check_edges(self.h, self.classes.as_ptr(), self.width as i32, self.height as i32, flat.as_ptr(), self.points.len() as i32, lines.as_mut_ptr(), cap as i32, &mut nl) }, "ck_cat_check_edges");
        self.lines = (0..nl as usize).map(|i| (lines[4 * i] as usize, lines[4 * i + 1] as usize, lines[4 * i + 2] as usize, lines[4 * i + 3] as usize)).collect();
    }
    /// lib.rs:501-549
    pub fn connected_components(&self) -> UnionFind {
        let n = self.width * self.height;
        let (mut roots, mut sizes) = (vec![0u32; n], vec![0u32; n]);
        check(unsafe { sys::ck_cat_connected_components(self.h, self.classes.as_ptr(), self.width as i32, self.height as i32, roots.as_mut_ptr(), sizes.as_mut_ptr()) }, "ck_cat_connected_components");
        UnionFind::from_device(roots, sizes)
    }
    /// lib.rs:615-661 wrote `lines.png` through `ril`; without an image crate this writes the same picture as `lines.ppm`
    pub fn draw(&self) {
        let mut uf = self.connected_components();
        let mut img = vec![0u8; self.width * self.height * 3];
        for &(x1, y1, x2, y2) in &self.lines {
            if uf.find(y1 * self.width + x1) != uf.find(y2 * self.width + x2) { continue; }
            let (dx, dy) = (x2 as i64 - x1 as i64, y2 as i64 - y1 as i64);
            let steps = dx.abs().max(dy.abs()).max(1);
            for s in 0..=steps {
                let (x, y) = ((x1 as i64 + dx * s / steps) as usize, (y1 as i64 + dy * s / steps) as usize);
                if x < self.width && y < self.height { img[(y * self.width + x) * 3 + 1] = 255; }
            }
        }
        let mut out = format!("P6\n{} {}\n255\n", self.width, self.height).into_bytes();
        out.extend_from_slice(&img);
        let _ = std::fs::write("lines.ppm", out);
    }
    /// `self.detector.detect(&image)` of the production path (crates/apriltags/src/lib.rs:301): one mono8 frame
    pub fn detect(&mut self, buf: &[u8], stride: usize) -> Vec<Detection> {
        self.detect_batch(&[(buf, stride)]).pop().unwrap_or_default()
    }
    /// the batched form the GPU wants: frames as (pixels, stride) pairs, `image_u8_t` fields as in lib.rs:204-209
    pub fn detect_batch(&mut self, frames: &[(&[u8], usize)]) -> Vec<Vec<Detection>> {
        const CAP: usize = 64;
        let imgs: Vec<sys::ck_image_u8_t> = frames.iter().map(|(b, s)| {
            assert!(b.len() >= s * (self.height - 1) + self.width);
            sys::ck_image_u8_t { buf: b.as_ptr() as *mut u8, width: self.width as i32, height: self.height as i32, stride: *s as i32 }
        }).collect();
        let mut dets = vec![unsafe { std::mem::zeroed::<sys::ck_detection_t>() }; CAP * frames.len()];
        let mut counts = vec![0i32; frames.len()];
        let mut status = vec![0u32; frames.len()];
        check(unsafe { sys::ck_detect_batch(self.h, imgs.as_ptr(), frames.len() as i32, dets.as_mut_ptr(), CAP as i32, counts.as_mut_ptr(), status.as_mut_ptr()) }, "ck_detect_batch");
        (0..frames.len()).map(|i| dets[i * CAP..i * CAP + counts[i] as usize].iter().map(|d| Detection(*d)).collect()).collect()
    }
    pub fn valid_tags(&self) -> &'static [usize] { self.valid_tags }
    pub fn points(&self) -> &[(usize, usize)] { &self.points }
    pub fn lines(&self) -> &[(usize, usize, usize, usize)] { &self.lines }
    /// raw handle for the crates that share it (chalkydri_sqpnp solves on the same device)
    pub fn raw(&self) -> *mut sys::ck_handle_t { self.h }
}
/// lib.rs:663-667: a clone is a fresh detector of the same size
impl Clone for Detector {
    fn clone(&self) -> Self { Self::new(self.width, self.height, &[]) }
}
/// lib.rs:668-681
impl Drop for Detector {
    fn drop(&mut self) { unsafe { sys::ck_destroy(self.h) } }
}
