// chalkydri.hpp — C++17 host layer over the C ABI of libchalkydri_hip.so (include/chalkydri_hip.h).
//
// The reference's host side is Rust; this image has no Rust toolchain, so the host side above the C ABI is written in
// C++ and mirrors the reference's public surface for the hot path: same type and method names, argument meaning and
// error behaviour, so that a test written against the Rust crates reads the same here.  (INTEGRATION.md holds the
// Rust `extern "C"` shim a maintainer would drop into the reference itself.)
//
//   chalkydri::apriltags::{Detector, UnionFind}   crates/chalkydri-apriltags/src/lib.rs:42-113,142-181,191,265,291,319,480,501,663
//   chalkydri::sqpnp::SqPnP                       crates/chalkydri_sqpnp/src/lib.rs:183-222,297-304,430-437
//   chalkydri::AprilTags (+ Detection)            crates/apriltags/src/lib.rs:166-183,217-379 and the `apriltag` crate calls at :301-314
//   chalkydri::whacknet::VisionMeasurement        crates/whacknet/src/lib.rs:19-66 (64-byte wire record)
//
// Error behaviour: where the Rust code panics (`assert_eq!`, `unwrap`, `expect`) these wrappers throw
// chalkydri::Panic; where it returns `None`/skips, they return std::nullopt.  There is no CPU fallback: without a HIP
// device every constructor throws (CK_ENODEVICE).
#ifndef CHALKYDRI_HPP
#define CHALKYDRI_HPP

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "chalkydri_hip.h"

namespace chalkydri {

// A Rust panic on this path (failed assertion, unwrap on an error).  `code` is the C ABI status when there is one.
struct Panic : std::runtime_error {
    int code;
    explicit Panic(const std::string &what, int code_ = 0) : std::runtime_error(what), code(code_) {}
};
inline void check(int rc, const char *what) {
    if (rc != CK_OK) throw Panic(std::string(what) + ": " + ck_strerror(rc), rc);
}

// RAII handle: one handle = one GPU + its stream; not thread-safe, like `&mut self`.
class Handle {
  public:
    Handle(int width, int height, int max_batch, const std::vector<std::string> &families, int bits_corrected, int quad_decimate, int device) {
        ck_config_default(&cfg_, width, height, max_batch);
        cfg_.device = device;
        cfg_.quad_decimate = quad_decimate;
        cfg_.max_hamming = bits_corrected;
        cfg_.n_families = (int)families.size();
        for (size_t i = 0; i < families.size(); i++) {
            cfg_.families[i] = ck_family_builtin(families[i].c_str());
            if (!cfg_.families[i]) throw Panic("unknown tag family " + families[i]); // DetectorBuilder::add_family_bits on a bad name
        }
        check(ck_create(&cfg_, &h_), "ck_create");
    }
    ~Handle() { ck_destroy(h_); }
    Handle(const Handle &) = delete;
    Handle &operator=(const Handle &) = delete;
    ck_handle_t *get() const { return h_; }
    const ck_config_t &config() const { return cfg_; }

  private:
    ck_config_t cfg_{};
    ck_handle_t *h_ = nullptr;
};

// What the reference reads from an `apriltag::Detection` (crates/apriltags/src/lib.rs:306-314).
class Detection {
  public:
    explicit Detection(const ck_detection_t &d) : d_(d) {}
    size_t id() const { return (size_t)d_.id; }
    size_t hamming() const { return (size_t)d_.hamming; }
    float decision_margin() const { return d_.decision_margin; }
    std::array<double, 2> center() const { return {d_.c[0], d_.c[1]}; }
    std::array<std::array<double, 2>, 4> corners() const {
        return {{{d_.p[0][0], d_.p[0][1]}, {d_.p[1][0], d_.p[1][1]}, {d_.p[2][0], d_.p[2][1]}, {d_.p[3][0], d_.p[3][1]}}};
    }
    const ck_detection_t &raw() const { return d_; }

  private:
    ck_detection_t d_;
};

namespace apriltags {

enum class Color : uint8_t { Black = 0, White = 1, Other = 2 }; // src/utils.rs:2-6

// src/utils.rs helpers (crate-private in the reference and, except grayscale, unused by its pipeline; kept under the same names).
namespace utils {
// utils.rs:33-46: trunc(fma(r, .33f, fma(g, .33f, b * .33f))), saturating cast
inline uint8_t grayscale(uint8_t r, uint8_t g, uint8_t b) {
    float v = std::fmaf((float)r, 0.33f, std::fmaf((float)g, 0.33f, (float)b * 0.33f));
    return v >= 255.0f ? 255 : (v <= 0.0f ? 0 : (uint8_t)v);
}
// utils.rs:51-72: FAST ring position 1..16 -> degrees
inline float fast_angle(uint8_t p) {
    if (p < 1 || p > 16) throw Panic("invalid FAST point", CK_EINVAL);
    return (float)(p - 1) * 22.5f;
}
enum class Orientation { Collinear, Clockwise, Counterclockwise }; // utils.rs:74-79
using Point = std::pair<size_t, size_t>;
// utils.rs:82-101
inline Orientation orientation(Point p, Point q, Point r) {
    int32_t v = ((int32_t)q.second - (int32_t)p.second) * ((int32_t)r.first - (int32_t)q.first) -
                ((int32_t)q.first - (int32_t)p.first) * ((int32_t)r.second - (int32_t)q.second);
    return v == 0 ? Orientation::Collinear : (v > 0 ? Orientation::Clockwise : Orientation::Counterclockwise);
}
// utils.rs:113-152: gift wrapping from the left-most point
struct PresentWrapper {
    static std::vector<Point> find_convex_hull(const std::vector<Point> &points) {
        if (points.empty()) throw Panic("index out of bounds: find_convex_hull of no points", CK_EINVAL);
        size_t l = 0, n = points.size();
        for (size_t i = 0; i < n; i++)
            if (points[i].first < points[l].first) l = i;
        std::vector<Point> hull;
        size_t p = l;
        while (p != l || hull.empty()) {
            hull.push_back(points[p]);
            size_t q = (p + 1) % n;
            for (size_t i = 0; i < n; i++)
                if (orientation(points[p], points[i], points[q]) == Orientation::Counterclockwise) q = i;
            p = q;
            if (hull.size() > n) break; // degenerate (collinear duplicate) input: stop where the walk starts repeating
        }
        return hull;
    }
};
} // namespace utils

// Result of Detector::connected_components (lib.rs:42-113).  The device returns the forest already flattened:
// find() is the canonical root (smallest index of the set), get_size() the size of the set.
class UnionFind {
  public:
    UnionFind(std::vector<uint32_t> roots, std::vector<uint32_t> sizes) : parent_(std::move(roots)), size_(std::move(sizes)) {}
    explicit UnionFind(size_t len) : parent_(len), size_(len, 1) { // UnionFind::new: singletons (lib.rs:49-65)
        for (size_t i = 0; i < len; i++) parent_[i] = (uint32_t)i;
    }
    size_t find(size_t id) { // lib.rs:67-76 (path compression)
        size_t r = id;
        while (parent_[r] != r) r = parent_[r];
        while (parent_[id] != r) { size_t n = parent_[id]; parent_[id] = (uint32_t)r; id = n; }
        return r;
    }
    void union_(size_t id1, size_t id2) { // lib.rs:78-94: by size, ties keep root1
        size_t r1 = find(id1), r2 = find(id2);
        if (r1 == r2) return;
        if (size_[r1] < size_[r2]) std::swap(r1, r2);
        parent_[r2] = (uint32_t)r1;
        size_[r1] += size_[r2];
    }
    size_t get_size(size_t id) const { // lib.rs:96-98: size stored at `id` (meaningful at roots, as in the reference)
        return size_[id];
    }
    size_t len() const { return parent_.size(); }

  private:
    std::vector<uint32_t> parent_, size_;
};

// chalkydri_apriltags::Detector — the experimental front-end ("CAT"), plus `detect`/`detect_batch` which the Rust shim adds
// because CAT itself exposes no IDs or corners.
class Detector {
  public:
    Detector(size_t width, size_t height, const std::vector<size_t> &valid_tags, int device = 0, int max_batch = 1)
        : width_(width), height_(height), valid_tags_(valid_tags), device_(device), max_batch_(max_batch),
          h_(std::make_shared<Handle>((int)width, (int)height, max_batch, std::vector<std::string>{"tag36h11"}, 3, 1, device)),
          buf_(width * height, (uint8_t)Color::Black) {} // alloc_zeroed: all Black (lib.rs:166-167)

    // Clone = a fresh detector of the same size with no valid tags (lib.rs:663-667)
    Detector clone() const { return Detector(width_, height_, {}, device_, max_batch_); }

    size_t width() const { return width_; }
    size_t height() const { return height_; }
    const std::vector<uint8_t> &buf() const { return buf_; }                             // Color per pixel
    const std::vector<std::pair<size_t, size_t>> &points() const { return points_; }      // (x, y), x-major order
    const std::vector<std::array<size_t, 4>> &lines() const { return lines_; }            // (x1, y1, x2, y2)

    // lib.rs:191-259.  `input` is RGB8 [h][w][3].
    void calc_otsu(std::vector<uint8_t> &input) {
        need_rgb(input.size());
        check(ck_cat_calc_otsu(h_->get(), input.data(), (int)width_, (int)height_, buf_.data()), "ck_cat_calc_otsu");
    }
    // lib.rs:319-334
    void thresh(const std::vector<uint8_t> &input) {
        need_rgb(input.size());
        check(ck_cat_thresh(h_->get(), input.data(), (int)width_, (int)height_, buf_.data()), "ck_cat_thresh");
    }
    // lib.rs:265-287: panics unless input.len() == width*height*3
    void process_frame(const std::vector<uint8_t> &input) {
        if (input.size() != width_ * height_ * 3) throw Panic("assertion `left == right` failed: input.len() == width * height * 3", CK_EINVAL);
        std::vector<uint32_t> pts(2 * point_cap()), lines(4 * line_cap_);
        int32_t np = 0, nl = 0;
        check(ck_cat_process_frame(h_->get(), input.data(), input.size(), (int)width_, (int)height_, buf_.data(), pts.data(), (int)point_cap(), &np,
                                   lines.data(), (int)line_cap_, &nl),
              "ck_cat_process_frame");
        store_points(pts, np);
        store_lines(lines, nl);
    }
    // lib.rs:291-309
    void detect_corners() {
        std::vector<uint32_t> pts(2 * point_cap());
        int32_t np = 0;
        check(ck_cat_detect_corners(h_->get(), buf_.data(), (int)width_, (int)height_, pts.data(), (int)point_cap(), &np), "ck_cat_detect_corners");
        store_points(pts, np);
    }
    // lib.rs:480-499
    void check_edges() {
        std::vector<uint32_t> pts(2 * points_.size() + 2), lines(4 * line_cap_);
        for (size_t i = 0; i < points_.size(); i++) { pts[2 * i] = (uint32_t)points_[i].first; pts[2 * i + 1] = (uint32_t)points_[i].second; }
        int32_t nl = 0;
        check(ck_cat_check_edges(h_->get(), buf_.data(), (int)width_, (int)height_, pts.data(), (int)points_.size(), lines.data(), (int)line_cap_, &nl),
              "ck_cat_check_edges");
        store_lines(lines, nl);
    }
    // lib.rs:501-549
    UnionFind connected_components() const {
        std::vector<uint32_t> roots(width_ * height_), sizes(width_ * height_);
        check(ck_cat_connected_components(h_->get(), buf_.data(), (int)width_, (int)height_, roots.data(), sizes.data()), "ck_cat_connected_components");
        return UnionFind(std::move(roots), std::move(sizes));
    }
    // lib.rs:615-661 writes lines.png for debugging; here: the lines whose end points share a component, returned instead of drawn
    std::vector<std::array<size_t, 4>> draw() const {
        UnionFind uf = connected_components();
        std::vector<std::array<size_t, 4>> out;
        for (const auto &l : lines_)
            if (uf.find(l[1] * width_ + l[0]) == uf.find(l[3] * width_ + l[2])) out.push_back(l);
        return out;
    }

    // Added by the shim: the production detector on a mono8 frame (stride in bytes, >= width).
    std::vector<Detection> detect(const uint8_t *mono8, size_t stride) {
        ck_image_u8_t img{const_cast<uint8_t *>(mono8), (int32_t)width_, (int32_t)height_, (int32_t)stride};
        std::vector<ck_detection_t> dets(det_cap_);
        int32_t n = 0;
        uint32_t st = 0;
        check(ck_detect_batch(h_->get(), &img, 1, dets.data(), (int)det_cap_, &n, &st), "ck_detect_batch");
        std::vector<Detection> out;
        for (int i = 0; i < n && i < (int)det_cap_; i++) out.emplace_back(dets[i]);
        return out;
    }
    std::vector<std::vector<Detection>> detect_batch(const std::vector<ck_image_u8_t> &imgs) {
        if ((int)imgs.size() > max_batch_) throw Panic("detect_batch: more frames than max_batch", CK_EINVAL);
        std::vector<ck_detection_t> dets(det_cap_ * imgs.size());
        std::vector<int32_t> counts(imgs.size());
        std::vector<uint32_t> st(imgs.size());
        check(ck_detect_batch(h_->get(), imgs.data(), (int)imgs.size(), dets.data(), (int)det_cap_, counts.data(), st.data()), "ck_detect_batch");
        std::vector<std::vector<Detection>> out(imgs.size());
        for (size_t f = 0; f < imgs.size(); f++)
            for (int i = 0; i < counts[f] && i < (int)det_cap_; i++) out[f].emplace_back(dets[f * det_cap_ + i]);
        return out;
    }

  private:
    void need_rgb(size_t len) const {
        if (len != width_ * height_ * 3) throw Panic("input is not width*height*3 bytes", CK_EINVAL);
    }
    size_t point_cap() const { return width_ * height_; } // the reference's points buffer holds one entry per pixel (lib.rs:168-169)
    void store_points(const std::vector<uint32_t> &pts, int32_t n) {
        points_.clear();
        for (int32_t i = 0; i < n; i++) points_.emplace_back(pts[2 * i], pts[2 * i + 1]);
    }
    void store_lines(const std::vector<uint32_t> &l, int32_t n) {
        lines_.clear();
        for (int32_t i = 0; i < n && (size_t)i < line_cap_; i++) lines_.push_back({l[4 * i], l[4 * i + 1], l[4 * i + 2], l[4 * i + 3]});
    }
    size_t width_, height_;
    std::vector<size_t> valid_tags_; // stored, never read — as in the reference (lib.rs:145)
    int device_, max_batch_;
    std::shared_ptr<Handle> h_;
    std::vector<uint8_t> buf_;
    std::vector<std::pair<size_t, size_t>> points_;
    std::vector<std::array<size_t, 4>> lines_;
    size_t line_cap_ = 1 << 20, det_cap_ = 256;
};

} // namespace apriltags

namespace sqpnp {

using Vec3 = std::array<double, 3>;
using Rot3 = std::array<double, 9>; // row-major 3x3
struct Iso3 {                       // nalgebra Isometry3<f64>: translation + unit quaternion (w, x, y, z)
    Vec3 translation{0, 0, 0};
    std::array<double, 4> rotation{1, 0, 0, 0};
    ck_iso3_t raw() const {
        ck_iso3_t r;
        std::memcpy(r.t, translation.data(), sizeof r.t);
        std::memcpy(r.q, rotation.data(), sizeof r.q);
        return r;
    }
    static Iso3 from_raw(const ck_iso3_t &r) {
        Iso3 o;
        std::memcpy(o.translation.data(), r.t, sizeof r.t);
        std::memcpy(o.rotation.data(), r.q, sizeof r.q);
        return o;
    }
};

// chalkydri_sqpnp::SqPnP (lib.rs:183-222): builder-style `max_iter` / `tolerance`, `solve_robot_pose`,
// `create_solver_camera_transform`.  The solve runs on the device (one wave per problem).
class SqPnP {
  public:
    explicit SqPnP(int device = 0) : h_(std::make_shared<Handle>(64, 64, 1, std::vector<std::string>{"tag36h11"}, 3, 1, device)) {
        ck_sqpnp_params_default(&prm_); // max_iter 15, tol_sq 1e-16 (lib.rs:201-212)
    }
    SqPnP &max_iter(size_t n) { prm_.max_iter = (int32_t)n; return *this; }          // lib.rs:214-217
    SqPnP &tolerance(double tol) { prm_.tol_sq = tol * tol; return *this; }          // lib.rs:219-222

    // lib.rs:297-377.  Returns (pivoted rotation, pivoted position, std devs) or nullopt where the reference returns None.
    std::optional<std::tuple<Rot3, Vec3, Vec3>> solve_robot_pose(const std::vector<Iso3> &points_isometry, const std::vector<Vec3> &points_2d,
                                                                const Iso3 &robot_to_cam, double gyro, double sign_change_error) {
        std::vector<ck_iso3_t> tags;
        for (const auto &t : points_isometry) tags.push_back(t.raw());
        std::vector<double> b;
        for (const auto &v : points_2d) b.insert(b.end(), v.begin(), v.end());
        ck_sqpnp_problem_t pb{};
        pb.n_tags = (int32_t)tags.size(); pb.n_bearings = (int32_t)points_2d.size();
        pb.tag_offset = 0; pb.bearing_offset = 0;
        pb.robot_to_cam = robot_to_cam.raw(); pb.gyro = gyro; pb.sign_change_error = sign_change_error;
        ck_sqpnp_result_t res{};
        ck_iso3_t dummy_tag{};
        double dummy_b[3] = {0, 0, 1};
        check(ck_sqpnp_solve_batch(h_->get(), &prm_, &pb, 1, tags.empty() ? &dummy_tag : tags.data(), (int32_t)tags.size(),
                                   b.empty() ? dummy_b : b.data(), (int32_t)points_2d.size(), &res),
              "ck_sqpnp_solve_batch");
        if (!res.valid) return std::nullopt;
        Rot3 R; Vec3 p, s;
        std::memcpy(R.data(), res.rot, sizeof res.rot);
        std::memcpy(p.data(), res.pos, sizeof res.pos);
        std::memcpy(s.data(), res.std_devs, sizeof res.std_devs);
        last_yaw_ = res.yaw;
        return std::make_tuple(R, p, s);
    }
    // euler_angles().2 of the last returned rotation — what the caller publishes (crates/apriltags/src/lib.rs:343)
    double last_yaw() const { return last_yaw_; }

    // lib.rs:430-461
    static Iso3 create_solver_camera_transform(double fwd_m, double left_m, double up_m, double roll_deg, double pitch_deg, double yaw_deg) {
        ck_iso3_t o;
        ck_sqpnp_create_solver_camera_transform(fwd_m, left_m, up_m, roll_deg, pitch_deg, yaw_deg, &o);
        return Iso3::from_raw(o);
    }

  private:
    std::shared_ptr<Handle> h_;
    ck_sqpnp_params_t prm_{};
    double last_yaw_ = 0.0;
};

} // namespace sqpnp

namespace whacknet {
// crates/whacknet/src/lib.rs:19-66 — the 64-byte record sent as one datagram
using VisionMeasurement = ck_vision_measurement_t;
static_assert(sizeof(VisionMeasurement) == 64, "wire record is 64 bytes (crates/whacknet/src/lib.rs:92-95)");

// WhacknetClient (lib.rs:68-89): a UDP socket bound to 0.0.0.0:0 and connected to the roboRIO; send() puts the 64 raw
// bytes of one measurement on the wire as one datagram, so a consumer cannot tell which backend produced it.
class WhacknetClient {
  public:
    explicit WhacknetClient(const std::string &remote_ip = "10.45.33.2", uint16_t remote_port = 7001) { // REMOTE_ADDR, lib.rs:14
        fd_ = ::socket(AF_INET, SOCK_DGRAM, 0);
        if (fd_ < 0) throw Panic("whacknet: socket() failed");
        sockaddr_in local{};
        local.sin_family = AF_INET; local.sin_addr.s_addr = htonl(INADDR_ANY); local.sin_port = 0;            // BIND_ADDR, lib.rs:13
        sockaddr_in remote{};
        remote.sin_family = AF_INET; remote.sin_port = htons(remote_port);
        if (::bind(fd_, reinterpret_cast<sockaddr *>(&local), sizeof local) != 0 || ::inet_pton(AF_INET, remote_ip.c_str(), &remote.sin_addr) != 1 ||
            ::connect(fd_, reinterpret_cast<sockaddr *>(&remote), sizeof remote) != 0) {
            ::close(fd_);
            throw Panic("whacknet: bind/connect failed");
        }
    }
    ~WhacknetClient() { if (fd_ >= 0) ::close(fd_); }
    WhacknetClient(const WhacknetClient &) = delete;
    WhacknetClient &operator=(const WhacknetClient &) = delete;
    // lib.rs:83-89; false where the reference returns the io::Error
    bool send(const VisionMeasurement &m) const { return ::send(fd_, &m, sizeof m, 0) == (ssize_t)sizeof m; }

  private:
    int fd_ = -1;
};

// The gyro heading arrives as one little-endian f64 per datagram on port 7002 (lib.rs:112-130)
inline std::optional<double> decode_gyro(const uint8_t *buf, size_t len) {
    if (len < 8) return std::nullopt;
    uint64_t bits = 0;
    for (int i = 7; i >= 0; i--) bits = (bits << 8) | buf[i];
    double v;
    std::memcpy(&v, &bits, sizeof v);
    return v;
}

// Comm (lib.rs:99-185): the gyro listener and the measurement publisher, each on a thread of its own.  The listener binds
// 0.0.0.0:<gyro_port> (7002 in the reference) and stores every 8-byte datagram as the current heading; publish() queues a
// measurement for the sender thread, which puts it on the wire through a WhacknetClient.  gyro_angle() starts at 0.0 like
// the reference's `Some(0f64)`; dropping the Comm ends both threads (the reference's listener only notices on its next
// datagram; here the socket has a receive timeout so that the destructor returns).
class Comm {
  public:
    explicit Comm(uint16_t gyro_port = 7002, const std::string &remote_ip = "10.45.33.2", uint16_t remote_port = 7001)
        : client_(remote_ip, remote_port) {
        gyro_fd_ = ::socket(AF_INET, SOCK_DGRAM, 0);
        sockaddr_in local{};
        local.sin_family = AF_INET; local.sin_addr.s_addr = htonl(INADDR_ANY); local.sin_port = htons(gyro_port);
        if (gyro_fd_ < 0 || ::bind(gyro_fd_, reinterpret_cast<sockaddr *>(&local), sizeof local) != 0) {
            if (gyro_fd_ >= 0) ::close(gyro_fd_);
            throw Panic("whacknet: gyro socket bind failed"); // `.unwrap()` at lib.rs:113
        }
        socklen_t ll = sizeof local;
        ::getsockname(gyro_fd_, reinterpret_cast<sockaddr *>(&local), &ll);
        gyro_port_ = ntohs(local.sin_port);
        timeval tv{0, 100000};
        ::setsockopt(gyro_fd_, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        listener_ = std::thread([this] {
            uint8_t buf[8];
            while (!stop_.load(std::memory_order_acquire)) {
                std::memset(buf, 0, sizeof buf);                                  // `buf = [0u8; 8]` per datagram (lib.rs:128)
                const ssize_t got = ::recv(gyro_fd_, buf, sizeof buf, 0);
                if (got < 0) continue;                                            // Err(_) => {} (lib.rs:125)
                uint64_t bits = 0;                                                // f64::from_le_bytes(buf): a short datagram leaves zero bytes
                for (int i = 7; i >= 0; i--) bits = (bits << 8) | buf[i];
                gyro_bits_.store(bits, std::memory_order_release);
            }
        });
        sender_ = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu_);
            for (;;) {
                cv_.wait(lk, [this] { return !queue_.empty() || stop_.load(std::memory_order_acquire); });
                if (queue_.empty()) return;                                       // stopping and drained
                const VisionMeasurement m = queue_.front();
                queue_.pop_front();
                lk.unlock();
                client_.send(m);                                                  // `.ok()`: a failed send is dropped (lib.rs:142)
                lk.lock();
            }
        });
    }
    ~Comm() {
        stop_.store(true, std::memory_order_release);
        cv_.notify_all();
        if (sender_.joinable()) sender_.join();
        if (listener_.joinable()) listener_.join();
        ::close(gyro_fd_);
    }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    // lib.rs:154-172
    void publish(uint8_t cam_id, uint8_t tag_count, uint64_t ts, double x, double y, double rot, double std_x, double std_y, double std_rot) {
        VisionMeasurement m{};
        m.pose_x = x; m.pose_y = y; m.pose_rot = rot; m.std_x = std_x; m.std_y = std_y; m.std_rot = std_rot;
        m.ts = ts; m.camera_id = cam_id; m.tag_count = tag_count;
        publish(m);
    }
    void publish(const VisionMeasurement &m) {
        { std::lock_guard<std::mutex> lk(mu_); queue_.push_back(m); }
        cv_.notify_one();
    }
    // lib.rs:174-179: the last heading received (0.0 before the first datagram)
    std::optional<double> gyro_angle() const {
        const uint64_t bits = gyro_bits_.load(std::memory_order_acquire);
        double v;
        std::memcpy(&v, &bits, sizeof v);
        return v;
    }
    uint16_t gyro_port() const { return gyro_port_; } // the bound port (pass 0 to let the system pick one: tests)

  private:
    WhacknetClient client_;
    int gyro_fd_ = -1;
    uint16_t gyro_port_ = 0;
    std::atomic<uint64_t> gyro_bits_{0};
    std::atomic<bool> stop_{false};
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<VisionMeasurement> queue_;
    std::thread listener_, sender_;
};
} // namespace whacknet

// Pinned host slots + asynchronous upload (the pooled host buffers of the camera layer,
// crates/chalkydri/src/cameras/gst_to_cu.rs:49-72,131-188): write frames, submit the slot, process it; submitting slot
// k+1 before processing slot k overlaps its upload with the compute.
class IngestRing {
  public:
    IngestRing(const std::shared_ptr<Handle> &h, int n_slots = 2) : h_(h) { check(ck_ingest_create(h_->get(), n_slots, &g_), "ck_ingest_create"); }
    ~IngestRing() { ck_ingest_destroy(g_); }
    IngestRing(const IngestRing &) = delete;
    IngestRing &operator=(const IngestRing &) = delete;
    int stride() const { return ck_ingest_stride(g_); }
    uint8_t *frame(int slot, int index) { return ck_ingest_frame(g_, slot, index); }
    static uint32_t fourcc(const char (&c)[5]) { return (uint32_t)(uint8_t)c[0] | ((uint32_t)(uint8_t)c[1] << 8) | ((uint32_t)(uint8_t)c[2] << 16) | ((uint32_t)(uint8_t)c[3] << 24); }
    void write(int slot, int index, const ck_image_u8_t &img, uint32_t code) { check(ck_ingest_write(g_, slot, index, &img, code), "ck_ingest_write"); }
    void submit(int slot, int n) { check(ck_ingest_submit(g_, slot, n), "ck_ingest_submit"); }
    ck_ingest_t *get() const { return g_; }

  private:
    std::shared_ptr<Handle> h_;
    ck_ingest_t *g_ = nullptr;
};

// crates/apriltags/src/lib.rs:185-192
struct RobotToCamOffset { double roll = 0, pitch = 0, yaw = 0, x = 0, y = 0, z = 0; };

// The AprilTags sink task (crates/apriltags/src/lib.rs:166-183,217-379): built from the task's config values, `process`
// turns one frame (+ the gyro heading, if any) into the measurement `Comm::publish` would send.
class AprilTags {
  public:
    struct Config {
        size_t width = 1280, height = 800;
        std::string family = "tag36h11";       // lib.rs:229
        size_t bits_corrected = 3;              // lib.rs:230
        ck_opencv5_t calib{};                   // "calib" JSON -> OpenCVModel5 (lib.rs:232-238)
        RobotToCamOffset robot_to_cam{};        // "robot_to_cam" JSON (lib.rs:240-254)
        std::map<size_t, sqpnp::Iso3> layout;   // AprilTagFieldLayout::load (field_layout.rs:18-44)
        uint8_t cam_id = 0;                     // lib.rs:256
        int device = 0, max_batch = 1, quad_decimate = 1;
    };
    explicit AprilTags(const Config &c)
        : cfg_(c), h_(std::make_shared<Handle>((int)c.width, (int)c.height, c.max_batch, std::vector<std::string>{c.family}, (int)c.bits_corrected,
                                               c.quad_decimate, c.device)) {
        for (const auto &kv : c.layout) {
            ck_field_tag_t t{};
            t.id = (int32_t)kv.first;
            t.pose = kv.second.raw();
            field_.push_back(t);
        }
        pp_.cam = c.calib;
        pp_.robot_to_cam = sqpnp::SqPnP::create_solver_camera_transform(c.robot_to_cam.x, c.robot_to_cam.y, c.robot_to_cam.z, c.robot_to_cam.roll,
                                                                        c.robot_to_cam.pitch, c.robot_to_cam.yaw).raw(); // lib.rs:247-254
        pp_.field = field_.data();
        pp_.n_field = (int32_t)field_.size();
        pp_.camera_id = c.cam_id;
        pp_.sign_change_error = 600.0; // SIGN_FLIP_CONST (lib.rs:6)
        ck_sqpnp_params_default(&pp_.sqpnp);
    }

    // lib.rs:293-379 for a batch of frames: measurement i is what `comm.publish` would send for frame i; valid[i] == false
    // reproduces the paths that publish nothing but the heartbeat (no detections, unknown tags only, no gyro, solver None).
    std::vector<std::pair<whacknet::VisionMeasurement, bool>> process(const std::vector<ck_image_u8_t> &imgs, const std::vector<std::optional<double>> &gyro) {
        const int n = (int)imgs.size();
        if (n > cfg_.max_batch || gyro.size() != imgs.size()) throw Panic("process: batch larger than max_batch or gyro size mismatch", CK_EINVAL);
        check(ck_upload_frames(h_->get(), imgs.data(), n), "ck_upload_frames");
        std::vector<double> g(n);
        std::vector<uint8_t> has(n);
        for (int i = 0; i < n; i++) { has[i] = gyro[i].has_value(); g[i] = gyro[i].value_or(0.0); }
        std::vector<whacknet::VisionMeasurement> out(n);
        std::vector<int32_t> valid(n);
        check(ck_process_uploaded(h_->get(), n, &pp_, g.data(), has.data(), out.data(), valid.data()), "ck_process_uploaded");
        std::vector<std::pair<whacknet::VisionMeasurement, bool>> r;
        for (int i = 0; i < n; i++) r.emplace_back(out[i], valid[i] != 0);
        return r;
    }
    // `ts` of a record is the processing latency in microseconds, clock.now() - tov (lib.rs:351,366): the device leaves it
    // zero, the host stamps it when the batch comes back.  tov_us[i] = time of validity of frame i on the same clock.
    static void stamp(std::vector<std::pair<whacknet::VisionMeasurement, bool>> &recs, const std::vector<uint64_t> &tov_us, uint64_t now_us) {
        for (size_t i = 0; i < recs.size() && i < tov_us.size(); i++) recs[i].first.ts = now_us - tov_us[i];
    }
    std::pair<whacknet::VisionMeasurement, bool> process(const ck_image_u8_t &img, std::optional<double> gyro) {
        return process(std::vector<ck_image_u8_t>{img}, std::vector<std::optional<double>>{gyro})[0];
    }
    const ck_process_params_t &params() const { return pp_; }
    const std::shared_ptr<Handle> &handle() const { return h_; }
    // the same for a batch that was written into an IngestRing slot and submitted
    std::vector<std::pair<whacknet::VisionMeasurement, bool>> process(IngestRing &ring, int slot, const std::vector<std::optional<double>> &gyro) {
        const int n = (int)gyro.size();
        std::vector<double> g(n);
        std::vector<uint8_t> has(n);
        for (int i = 0; i < n; i++) { has[i] = gyro[i].has_value(); g[i] = gyro[i].value_or(0.0); }
        std::vector<whacknet::VisionMeasurement> out(n);
        std::vector<int32_t> valid(n);
        check(ck_process_ingested(ring.get(), slot, n, &pp_, g.data(), has.data(), out.data(), valid.data()), "ck_process_ingested"); // CK_EINVAL unless n frames were submitted
        std::vector<std::pair<whacknet::VisionMeasurement, bool>> r;
        for (int i = 0; i < n; i++) r.emplace_back(out[i], valid[i] != 0);
        return r;
    }

  private:
    Config cfg_;
    std::shared_ptr<Handle> h_;
    std::vector<ck_field_tag_t> field_;
    ck_process_params_t pp_{};
};

} // namespace chalkydri
#endif // CHALKYDRI_HPP
