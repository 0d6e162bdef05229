//! crates/chalkydri-apriltags/src/utils.rs on the host (these helpers carry no per-pixel work; the per-pixel users run on
//! the device).  Same names and results as the reference's `pub(crate)` items.

/// utils.rs:1-20
#[derive(Clone, Copy, PartialEq, Eq, PartialOrd, Ord, Debug)]
pub enum Color { Black, White, Other }
impl Color {
    #[inline(always)] pub fn is_black(&self) -> bool { *self == Color::Black }
    #[inline(always)] pub fn is_white(&self) -> bool { *self == Color::White }
    #[inline(always)] pub fn is_good(&self) -> bool { *self != Color::Other }
}
/// utils.rs:27-29
#[inline(always)]
pub const fn px(x: usize, y: usize, width: usize) -> usize { y * width + x }
/// utils.rs:33-46: trunc(fma(r, 0.33, fma(g, 0.33, b * 0.33))) in f32, saturating
pub fn grayscale(data: &[u8]) -> u8 {
    let (r, g, b) = (data[0] as f32, data[1] as f32, data[2] as f32);
    r.mul_add(0.33, g.mul_add(0.33, b * 0.33)) as u8
}
/// utils.rs:51-71: angle of position p (1..=16) on the 16-point circle
pub fn fast_angle(p: u8) -> f32 {
    assert!((1..=16).contains(&p), "fast_angle: position out of range");
    (p as f32 - 1.0) * 22.5
}
/// utils.rs:74-80
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum Orientation { Collinear, Clockwise, Counterclockwise }
/// utils.rs:82-101 (i32 cross product sign)
pub fn orientation((px, py): (usize, usize), (qx, qy): (usize, usize), (rx, ry): (usize, usize)) -> Orientation {
    match ((qy as i32 - py as i32) * (rx as i32 - qx as i32)) - ((qx as i32 - px as i32) * (ry as i32 - qy as i32)) {
        0 => Orientation::Collinear,
        i if i > 0 => Orientation::Clockwise,
        _ => Orientation::Counterclockwise,
    }
}
/// utils.rs:104-153: gift wrapping
pub struct PresentWrapper {}
impl PresentWrapper {
    pub fn find_convex_hull(points: &[(usize, usize)]) -> Vec<(usize, usize)> {
        if points.len() < 3 { return points.to_vec(); }
        let mut hull = Vec::new();
        let leftmost = (0..points.len()).min_by_key(|&i| points[i].0).unwrap();
        let mut p = leftmost;
        loop {
            hull.push(points[p]);
            let mut q = (p + 1) % points.len();
            for i in 0..points.len() {
                if orientation(points[p], points[i], points[q]) == Orientation::Counterclockwise { q = i; }
            }
            p = q;
            if p == leftmost || hull.len() > points.len() { break; }
        }
        hull
    }
}
