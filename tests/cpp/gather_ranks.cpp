// gather_ranks — the multi-GPU path of the C ABI without Python or torch: one process per GPU, the way a Rust / C++ host
// would run it (record layout crates/whacknet/src/lib.rs:43-66; one camera chain per process as in
// chalkydri.ron:3-104).
//
//   gather_ranks [N]     N ranks (default and at most: the GPUs the KFD topology lists; 1 on a one-GPU box)
//
// The parent forks its N children BEFORE any HIP call (a forked copy of an initialised runtime is unusable) and never touches
// the GPU itself: it counts devices from sysfs and relays the 128-byte RCCL id of rank 0 through pipes.  Every child:
// ck_create(device = rank) -> ck_upload_frames + ck_process_uploaded on its own small batch (ragged: odd ranks one frame
// short) -> ck_comm_create -> ck_gather_poses(sync = 0) -> AT ONCE the next batch's ck_process_uploaded and a second
// ck_gather_poses(sync = 0) (the two send buffers of the communicator: step 1's rendezvous runs beside step 2's kernels) ->
// ck_comm_sync, then ships its local records and both gathered arrays to the parent, which checks that every rank received
// every rank's records, in rank order, padded with empty records.  Exit code 0 and a line "GATHER_RANKS_OK ..." on success.
#include <dirent.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "chalkydri_hip.h"
#include "../../chalkydri_amd/csrc/synth.h"

namespace {

constexpr int W = 640, H = 480, B = 4, STEPS = 2;

int gpus_from_sysfs() { // KFD topology nodes with SIMDs are GPUs (CPUs have simd_count 0)
    int n = 0;
    DIR *d = opendir("/sys/class/kfd/kfd/topology/nodes");
    if (!d) return 0;
    while (dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        std::string p = std::string("/sys/class/kfd/kfd/topology/nodes/") + e->d_name + "/properties";
        FILE *f = fopen(p.c_str(), "r");
        if (!f) continue;
        char key[64];
        unsigned long long val;
        while (fscanf(f, "%63s %llu", key, &val) == 2)
            if (!strcmp(key, "simd_count") && val > 0) { n++; break; }
        fclose(f);
    }
    closedir(d);
    return n;
}

bool write_all(int fd, const void *p, size_t n) {
    const char *b = static_cast<const char *>(p);
    while (n) { ssize_t k = write(fd, b, n); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}
bool read_all(int fd, void *p, size_t n) {
    char *b = static_cast<char *>(p);
    while (n) { ssize_t k = read(fd, b, n); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}

#define CHECK(call)                                                                                         \
    do {                                                                                                    \
        int rc_ = (call);                                                                                   \
        if (rc_ != CK_OK) {                                                                                 \
            fprintf(stderr, "[rank %d] %s -> %s (%d) %s\n", rank, #call, ck_strerror(rc_), rc_, ck_last_error()); \
            return 10;                                                                                      \
        }                                                                                                   \
    } while (0)

int child(int rank, int world, int id_in, int id_out, int res_out) {
    uint8_t id[CK_COMM_ID_BYTES] = {0};
    if (rank == 0) {
        CHECK(ck_comm_unique_id(id));
        if (!write_all(id_out, id, sizeof id)) return 11;
    }
    if (!read_all(id_in, id, sizeof id)) return 12;
    ck_config_t cfg;
    ck_config_default(&cfg, W, H, B);
    cfg.device = rank;
    const ck_family_t *fam = ck_family_builtin("tag36h11");
    cfg.n_families = 1; cfg.families[0] = fam;
    ck_handle_t *h = nullptr;
    CHECK(ck_create(&cfg, &h));
    ck_comm_t *comm = nullptr;
    CHECK(ck_comm_create(h, id, world, rank, &comm));
    fprintf(stderr, "[rank %d] device %d, librccl = %s\n", rank, cfg.device, ck_comm_library(comm));
    // a field of 30 tags at arbitrary places: whether a frame yields a pose does not matter here, its record's bytes do
    std::vector<ck_field_tag_t> field(30);
    for (int i = 0; i < 30; i++) {
        memset(&field[i], 0, sizeof field[i]);
        field[i].id = i;
        field[i].pose.t[0] = 2.0 + 0.1 * i; field[i].pose.t[1] = 0.3 * (i % 5); field[i].pose.t[2] = 0.5;
        field[i].pose.q[0] = 1.0; // (w, x, y, z)
    }
    ck_process_params_t pp;
    memset(&pp, 0, sizeof pp);
    pp.cam.fx = pp.cam.fy = 600.0; pp.cam.cx = W / 2.0; pp.cam.cy = H / 2.0;
    ck_sqpnp_create_solver_camera_transform(0.2, 0.0, 0.6, 0, 0, 0, &pp.robot_to_cam);
    pp.field = field.data(); pp.n_field = 30; pp.camera_id = (uint8_t)rank; pp.sign_change_error = 600.0;
    ck_sqpnp_params_default(&pp.sqpnp);
    ck_synth_params_t sp;
    ck_synth_params_default(&sp, W, H, 3);
    sp.max_id = 29;
    const int n_valid = B - (rank & 1); // a ragged shard on the odd ranks: the library pads it to the common row count
    std::vector<uint8_t> frames((size_t)W * H * B);
    std::vector<ck_vision_measurement_t> local((size_t)STEPS * B), all((size_t)STEPS * world * B);
    memset(local.data(), 0, local.size() * sizeof local[0]);
    double gyro[B] = {0.1, -0.2, 0.3, 0.0};
    uint8_t has_gyro[B] = {1, 1, 1, 1};
    int32_t valid[B];
    for (int s = 0; s < STEPS; s++) {
        ck_image_u8_t imgs[B];
        for (int i = 0; i < n_valid; i++) {
            uint8_t *fr = frames.data() + (size_t)i * W * H;
            ck_synth_tag_t truth[8];
            int32_t nt = 0;
            ck_synth_render(0xC4A1D1ull + 100000ull * (unsigned)rank + 1000ull * (unsigned)s + (unsigned)i, &sp, &fam, 1, fr, W, truth, 8, &nt);
            imgs[i].buf = fr; imgs[i].width = W; imgs[i].height = H; imgs[i].stride = W;
        }
        CHECK(ck_upload_frames(h, imgs, n_valid));
        CHECK(ck_process_uploaded(h, n_valid, &pp, gyro, has_gyro, local.data() + (size_t)s * B, valid));
        // sync = 0: only enqueued; the next step's kernels follow at once on the handle's stream
        CHECK(ck_gather_poses(h, comm, n_valid, B, all.data() + (size_t)s * world * B, 0));
    }
    CHECK(ck_comm_sync(comm));
    int tags = 0;
    for (int s = 0; s < STEPS; s++)
        for (int i = 0; i < n_valid; i++) tags += local[(size_t)s * B + i].tag_count;
    fprintf(stderr, "[rank %d] %d frames per step, %d detections in all\n", rank, n_valid, tags);
    ck_comm_destroy(comm);
    ck_destroy(h);
    if (!write_all(res_out, &tags, sizeof tags) || !write_all(res_out, local.data(), local.size() * sizeof local[0]) ||
        !write_all(res_out, all.data(), all.size() * sizeof all[0]))
        return 13;
    return 0;
}

} // namespace

int main(int argc, char **argv) {
    const int gpus = gpus_from_sysfs();
    int world = argc > 1 ? atoi(argv[1]) : gpus;
    if (gpus < 1) { puts("GATHER_RANKS_SKIP no GPU in the KFD topology"); return 0; }
    if (world < 1 || world > gpus) world = gpus; // one rank per GPU: RCCL refuses two ranks on one device
    if (world > 8) world = 8;
    std::vector<int> down(2 * world), res(2 * world);
    int up[2];
    if (pipe(up) != 0) return 2;
    for (int r = 0; r < world; r++)
        if (pipe(&down[2 * r]) != 0 || pipe(&res[2 * r]) != 0) return 2;
    std::vector<pid_t> pids(world);
    for (int r = 0; r < world; r++) { // (no HIP call has happened in this process)
        pids[r] = fork();
        if (pids[r] < 0) return 3;
        if (pids[r] == 0) {
            for (int q = 0; q < world; q++) { close(down[2 * q + 1]); close(res[2 * q]); if (q != r) { close(down[2 * q]); close(res[2 * q + 1]); } }
            close(up[0]);
            if (r != 0) close(up[1]); // (only rank 0 answers: the parent's read must see end-of-file if that rank dies first)
            _exit(child(r, world, down[2 * r], up[1], res[2 * r + 1]));
        }
    }
    close(up[1]);
    for (int r = 0; r < world; r++) { close(down[2 * r]); close(res[2 * r + 1]); }
    uint8_t id[CK_COMM_ID_BYTES];
    bool ok = read_all(up[0], id, sizeof id);
    for (int r = 0; r < world && ok; r++) ok = write_all(down[2 * r + 1], id, sizeof id);
    for (int r = 0; r < world; r++) close(down[2 * r + 1]); // (a rank still waiting for the id sees end-of-file and leaves)
    const size_t nl = (size_t)STEPS * B, na = (size_t)STEPS * world * B;
    std::vector<std::vector<ck_vision_measurement_t>> local(world, std::vector<ck_vision_measurement_t>(nl)), all(world, std::vector<ck_vision_measurement_t>(na));
    int tags_total = 0;
    for (int r = 0; r < world && ok; r++) {
        int tags = 0;
        ok = read_all(res[2 * r], &tags, sizeof tags) && read_all(res[2 * r], local[r].data(), nl * sizeof(ck_vision_measurement_t)) &&
             read_all(res[2 * r], all[r].data(), na * sizeof(ck_vision_measurement_t));
        tags_total += tags;
    }
    int bad_exit = 0;
    for (int r = 0; r < world; r++) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "rank %d ended with status 0x%x\n", r, st); bad_exit++; }
    }
    if (!ok || bad_exit) { puts("GATHER_RANKS_FAIL a rank did not deliver"); return 4;}
    int mism = 0;
    for (int c = 0; c < world; c++)          // what rank c received ...
        for (int s = 0; s < STEPS; s++)
            for (int r = 0; r < world; r++)  // ... from rank r: r's own records, then empty ones up to the common row count
                if (memcmp(&all[c][((size_t)s * world + r) * B], &local[r][(size_t)s * B], B * sizeof(ck_vision_measurement_t)) != 0) mism++;
    if (mism || tags_total == 0) { printf("GATHER_RANKS_FAIL %d blocks differ, %d detections\n", mism, tags_total); return 5; }
    printf("GATHER_RANKS_OK world=%d gpus=%d steps=%d rows=%d detections=%d\n", world, gpus, STEPS, B, tags_total);
    return 0;
}
