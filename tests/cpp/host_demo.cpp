// host_demo — exercises the C++ host layer (include/chalkydri.hpp) the way the reference's Rust callers use their crates.
//
//   host_demo selfcheck
//       no GPU needed: header compiles and links; pure-host entry points work; constructing a detector without a HIP
//       device fails loudly (Panic carrying CK_ENODEVICE) instead of falling back to a CPU path.
//   host_demo run <w> <h> <frame.raw> <layout.txt> <fx fy cx cy k1 k2 p1 p2 k3> <roll pitch yaw x y z> <gyro>
//       GPU: detect + AprilTags::process on one mono8 frame; prints detections and the 64-byte measurement as hex so the
//       caller (tests/test_cpp_host.py) can compare them bit for bit with the Python mirror of the same ABI.
//   host_demo cat <w> <h> <rgb.raw>
//       GPU: CAT front-end (process_frame, connected_components) — prints point/line counts and an FNV hash of each output.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "chalkydri.hpp"

using namespace chalkydri;

static std::vector<uint8_t> slurp(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Panic(std::string("cannot open ") + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static unsigned long long fnv(const void *p, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static void hex(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; i++) std::printf("%02x", b[i]);
}

static int selfcheck() {
    // create_solver_camera_transform is pure host arithmetic: with zero offsets camera +z maps to robot +x
    sqpnp::Iso3 t = sqpnp::SqPnP::create_solver_camera_transform(0, 0, 0, 0, 0, 0);
    double n2 = 0;
    for (double q : t.rotation) n2 += q * q;
    if (std::fabs(n2 - 1.0) > 1e-12) { std::puts("FAIL quaternion not unit"); return 1; }
    // UnionFind::new / union / find / get_size on the host
    apriltags::UnionFind uf(8);
    uf.union_(1, 2); uf.union_(2, 5);
    if (uf.find(5) != uf.find(1) || uf.get_size(uf.find(1)) != 3 || uf.find(7) != 7) { std::puts("FAIL union-find"); return 1; }
    {   // src/utils.rs helpers
        namespace u = apriltags::utils;
        std::vector<u::Point> pts = {{5, 5}, {0, 0}, {10, 0}, {10, 10}, {0, 10}, {3, 7}};
        auto hull = u::PresentWrapper::find_convex_hull(pts);
        if (hull.size() != 4 || hull[0] != u::Point{0, 0}) { std::puts("FAIL convex hull"); return 1; }
        if (u::grayscale(200, 200, 200) != 198 || u::grayscale(255, 255, 255) != 252 || u::fast_angle(9) != 180.0f ||
            u::orientation({0, 0}, {4, 4}, {8, 0}) != u::Orientation::Clockwise) { std::puts("FAIL utils"); return 1; }
    }
    {   // whacknet: one measurement = one 64-byte datagram with exactly the struct's bytes; the gyro is a little-endian f64
        int rx = ::socket(AF_INET, SOCK_DGRAM, 0);
        sockaddr_in a{};
        a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
        socklen_t al = sizeof a;
        if (rx < 0 || ::bind(rx, reinterpret_cast<sockaddr *>(&a), sizeof a) != 0 || ::getsockname(rx, reinterpret_cast<sockaddr *>(&a), &al) != 0) {
            std::puts("FAIL loopback socket"); return 1;
        }
        timeval tv{2, 0};
        ::setsockopt(rx, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        whacknet::WhacknetClient tx("127.0.0.1", ntohs(a.sin_port));
        whacknet::VisionMeasurement m{};
        m.pose_x = 1.25; m.pose_y = -3.5; m.pose_rot = 0.125; m.std_x = 0.01; m.std_y = 0.02; m.std_rot = 0.05; m.ts = 0x1122334455667788ull;
        m.camera_id = 3; m.tag_count = 6;
        unsigned char buf[128];
        if (!tx.send(m) || ::recv(rx, buf, sizeof buf, 0) != 64 || std::memcmp(buf, &m, 64) != 0) { std::puts("FAIL whacknet datagram"); return 1; }
        ::close(rx);
        const double g = -1.5707963267948966;
        unsigned char gb[8];
        std::memcpy(gb, &g, 8);
        auto back = whacknet::decode_gyro(gb, 8);
        if (!back || *back != g || whacknet::decode_gyro(gb, 4)) { std::puts("FAIL gyro decode"); return 1; }
    }
    {   // whacknet::Comm: the gyro listener thread picks up a heading sent to its port, publish() reaches the wire through
        // the sender thread (crates/whacknet/src/lib.rs:99-185)
        int rx = ::socket(AF_INET, SOCK_DGRAM, 0);
        sockaddr_in a{};
        a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
        socklen_t al = sizeof a;
        if (rx < 0 || ::bind(rx, reinterpret_cast<sockaddr *>(&a), sizeof a) != 0 || ::getsockname(rx, reinterpret_cast<sockaddr *>(&a), &al) != 0) {
            std::puts("FAIL loopback socket"); return 1;
        }
        timeval tv{2, 0};
        ::setsockopt(rx, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        {
            whacknet::Comm comm(0, "127.0.0.1", ntohs(a.sin_port));
            auto g0 = comm.gyro_angle();
            if (!g0 || *g0 != 0.0) { std::puts("FAIL comm initial gyro"); return 1; }
            int tx = ::socket(AF_INET, SOCK_DGRAM, 0);
            sockaddr_in to{};
            to.sin_family = AF_INET; to.sin_addr.s_addr = htonl(INADDR_LOOPBACK); to.sin_port = htons(comm.gyro_port());
            const double heading = 0.7853981633974483;
            bool seen = false;
            for (int tries = 0; tries < 200 && !seen; tries++) {
                ::sendto(tx, &heading, 8, 0, reinterpret_cast<sockaddr *>(&to), sizeof to);
                ::usleep(5000);
                seen = *comm.gyro_angle() == heading;
            }
            ::close(tx);
            if (!seen) { std::puts("FAIL comm gyro listener"); return 1; }
            comm.publish(2, 5, 0xABCDEF0123456789ull, 1.5, 2.5, -0.25, 0.1, 0.2, 0.3);
            unsigned char buf[128];
            whacknet::VisionMeasurement want{};
            want.pose_x = 1.5; want.pose_y = 2.5; want.pose_rot = -0.25; want.std_x = 0.1; want.std_y = 0.2; want.std_rot = 0.3;
            want.ts = 0xABCDEF0123456789ull; want.camera_id = 2; want.tag_count = 5;
            if (::recv(rx, buf, sizeof buf, 0) != 64 || std::memcmp(buf, &want, 64) != 0) { std::puts("FAIL comm publish"); return 1; }
        } // ~Comm joins both threads
        ::close(rx);
    }
    int devs = ck_device_count();
    if (devs <= 0) {
        try {
            apriltags::Detector d(64, 64, {});
            std::puts("FAIL detector constructed without a device");
            return 1;
        } catch (const Panic &p) {
            if (p.code != CK_ENODEVICE) { std::printf("FAIL wrong error %d\n", p.code); return 1; }
            std::puts("OK no device: Panic(CK_ENODEVICE), no CPU fallback");
            return 0;
        }
    }
    apriltags::Detector d(64, 64, {});
    try {
        d.process_frame(std::vector<uint8_t>(10));
        std::puts("FAIL wrong-length frame accepted");
        return 1;
    } catch (const Panic &) {}
    std::puts("OK device present");
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc >= 2 && std::string(argv[1]) == "selfcheck") return selfcheck();
        if (argc >= 5 && std::string(argv[1]) == "cat") {
            size_t w = std::atoi(argv[2]), h = std::atoi(argv[3]);
            std::vector<uint8_t> rgb = slurp(argv[4]);
            apriltags::Detector det(w, h, {1, 2, 3});
            det.process_frame(rgb);
            apriltags::UnionFind uf = det.connected_components();
            std::vector<uint32_t> roots(w * h), sizes(w * h), pts, lines;
            for (size_t i = 0; i < w * h; i++) { roots[i] = (uint32_t)uf.find(i); sizes[i] = (uint32_t)uf.get_size(i); }
            for (auto &p : det.points()) { pts.push_back((uint32_t)p.first); pts.push_back((uint32_t)p.second); }
            for (auto &l : det.lines()) for (size_t v : l) lines.push_back((uint32_t)v);
            std::printf("classes %016llx points %zu %016llx lines %zu %016llx roots %016llx sizes %016llx\n", fnv(det.buf().data(), det.buf().size()),
                        det.points().size(), fnv(pts.data(), pts.size() * 4), det.lines().size(), fnv(lines.data(), lines.size() * 4),
                        fnv(roots.data(), roots.size() * 4), fnv(sizes.data(), sizes.size() * 4));
            return 0;
        }
        if (argc >= 22 && std::string(argv[1]) == "run") {
            size_t w = std::atoi(argv[2]), h = std::atoi(argv[3]);
            std::vector<uint8_t> frame = slurp(argv[4]);
            if (frame.size() != w * h) throw Panic("frame size mismatch");
            AprilTags::Config c;
            c.width = w; c.height = h;
            std::ifstream lf(argv[5]);
            size_t id;
            sqpnp::Iso3 iso;
            while (lf >> id >> iso.translation[0] >> iso.translation[1] >> iso.translation[2] >> iso.rotation[0] >> iso.rotation[1] >> iso.rotation[2] >>
                   iso.rotation[3])
                c.layout[id] = iso;
            double v[16];
            for (int i = 0; i < 16; i++) v[i] = std::atof(argv[6 + i]);
            c.calib = ck_opencv5_t{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
            c.robot_to_cam = RobotToCamOffset{v[9], v[10], v[11], v[12], v[13], v[14]};
            c.cam_id = 5;
            const double gyro = v[15];
            apriltags::Detector det(w, h, {});
            for (const Detection &d : det.detect(frame.data(), w)) {
                std::printf("det %zu %zu ", d.id(), d.hamming());
                hex(&d.raw().decision_margin, 4); std::printf(" ");
                hex(d.raw().c, 16); std::printf(" ");
                hex(d.raw().p, 64); std::printf("\n");
            }
            AprilTags task(c);
            ck_image_u8_t img{frame.data(), (int32_t)w, (int32_t)h, (int32_t)w};
            auto r = task.process(img, gyro);
            std::printf("measurement %d ", r.second ? 1 : 0);
            hex(&r.first, sizeof r.first);
            std::printf("\n");
            {   // the same frame through the pinned ingest ring must give the same 64 bytes
                IngestRing ring(task.handle(), 2);
                ring.write(1, 0, img, IngestRing::fourcc("GREY"));
                ring.submit(1, 1);
                auto rr = task.process(ring, 1, {gyro});
                if (rr[0].second != r.second || std::memcmp(&rr[0].first, &r.first, sizeof r.first) != 0) throw Panic("ingest ring result differs");
            }
            auto none = task.process(img, std::nullopt); // "no gyro, no solve" (crates/apriltags/src/lib.rs:330)
            std::printf("nogyro %d %u\n", none.second ? 1 : 0, (unsigned)none.first.tag_count);
            return 0;
        }
        std::fprintf(stderr, "usage: host_demo selfcheck | run ... | cat ...\n");
        return 2;
    } catch (const Panic &p) {
        std::fprintf(stderr, "panic: %s (code %d)\n", p.what(), p.code);
        return 101; // Rust's panic exit code
    }
}
