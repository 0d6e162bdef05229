/*
 * chalkydri_hip.h — C ABI of the MI355X-native AprilTag detect + SQPnP pose hot path.
 *
 * This is the drop-in boundary.  Every entry point is plain C: pointers, sizes and
 * POD structs, no C++/torch types.  Reference citations are relative to /root/reference.
 *
 * What each group replaces in the reference:
 *   ck_image_u8_t ............ the hand-built apriltag `image_u8_t {buf,width,height,stride}` that
 *                              `image_from_cuimage` hands to the C detector  (crates/apriltags/src/lib.rs:197-213)
 *   ck_create/ck_destroy ..... `DetectorBuilder::default().add_family_bits(family,bits).build()` and Drop
 *                              (crates/apriltags/src/lib.rs:258-262,279-282)
 *   ck_detect_batch* ......... `self.detector.detect(&image)` → Vec<Detection> with id()/corners()
 *                              (crates/apriltags/src/lib.rs:301-314), batched over frames
 *   ck_threshold/segment ..... the stages `north_star` scores against the HBM roofline
 *   ck_cat_* ................. chalkydri-apriltags "CAT" `Detector::{calc_otsu,thresh,process_frame,
 *                              detect_corners,check_edges,connected_components}`
 *                              (crates/chalkydri-apriltags/src/lib.rs:191,265,291,319,480,501)
 *   ck_sqpnp_* ............... `SqPnP::{new,max_iter,tolerance,solve_robot_pose,
 *                              create_solver_camera_transform}` (crates/chalkydri_sqpnp/src/lib.rs:201-222,297-377,430-461)
 *   ck_unproject_* ........... `cam_model.unproject(corners)` for OpenCVModel5 (crates/apriltags/src/lib.rs:316-322)
 *   ck_process_batch ......... `AprilTags::process` glue: filter → unproject → solve → VisionMeasurement
 *                              (crates/apriltags/src/lib.rs:293-379; wire struct crates/whacknet/src/lib.rs:43-66)
 *
 * Error convention: every function returning int returns CK_OK (0) or a negative CK_E* code; nothing
 * throws or aborts across the ABI.  Per-frame capacity overflows are reported in status words, not as
 * failures.  A handle is bound to one HIP device and is NOT thread-safe (mirrors `&mut self`).
 */
#ifndef CHALKYDRI_HIP_H
#define CHALKYDRI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CK_ABI_VERSION 3 /* 3: ck_gather_poses takes (n_valid, rows) — a changed prototype is a new version, like a changed struct */

/* ---- status codes ------------------------------------------------------------------------- */
enum {
    CK_OK = 0,
    CK_EINVAL = -1,      /* bad argument (null pointer, size mismatch, unsupported geometry) */
    CK_ENOMEM = -2,      /* host or device allocation failed */
    CK_EDEVICE = -3,     /* HIP runtime error; ck_last_error() has the text */
    CK_ENODEVICE = -4,   /* no HIP device visible: the product path has NO CPU fallback */
    CK_ECAPACITY = -5,   /* batch larger than the handle was created for */
    CK_EUNSUPPORTED = -6 /* valid request this build does not implement */
};

/* per-frame status bits (ck_detect_batch `status[]`) */
enum {
    CK_FRAME_OK = 0,
    CK_FRAME_POINTS_OVERFLOW = 1,   /* boundary-point buffer full: clusters may be missing */
    CK_FRAME_CLUSTERS_OVERFLOW = 2, /* cluster table full */
    CK_FRAME_QUADS_OVERFLOW = 4,    /* more candidate quads than capacity */
    CK_FRAME_DETS_OVERFLOW = 8,     /* more detections than `cap_per_frame` */
    CK_FRAME_UNVERIFIED_ID = 16     /* a detection's id is >= its family's n_upstream (see ck_family_t) */
};

/* ---- images --------------------------------------------------------------------------------- */
/* Layout-identical to apriltag's image_u8_t (crates/apriltags/src/lib.rs:204-209). stride >= width. */
typedef struct ck_image_u8 {
    uint8_t *buf;
    int32_t width;
    int32_t height;
    int32_t stride;
} ck_image_u8_t;

/* ---- tag families (runtime data; AprilTag-3 layout convention) -------------------------------- */
typedef struct ck_family {
    char name[32];
    uint32_t nbits;           /* 36 for tag36h11, 16 for tag16h5 */
    uint32_t ncodes;
    const uint64_t *codes;    /* bit (nbits-1-i) of a code is the cell at (bit_x[i], bit_y[i]) */
    const uint32_t *bit_x;    /* cell coordinates, origin = outer corner of the black border */
    const uint32_t *bit_y;
    int32_t width_at_border;  /* 8 for tag36h11, 6 for tag16h5 */
    int32_t total_width;      /* 10 / 8 (adds the white quiet ring) */
    int32_t reversed_border;  /* 0 for both classic families */
    uint32_t min_hamming;     /* 11 / 5 */
    uint32_t n_upstream;      /* IDs 0..n_upstream-1 are known to equal the upstream AprilTag table of this name.  A caller
                               * that passes upstream's own table sets n_upstream = ncodes.  Detections with a larger id
                               * raise CK_FRAME_UNVERIFIED_ID and are ignored by the pose glue unless
                               * ck_process_params_t.allow_unverified_ids is set.  0 (a zero-initialised table, or one
                               * built against ABI version 1, which had no such field) means ncodes: the caller vouches
                               * for its table; a value above ncodes is refused by ck_create (CK_EINVAL). */
} ck_family_t;

/* Built-in tables.  "tag16h5": all 30 upstream codes (n_upstream = 30).  "tag36h11": upstream layout, 587 codes of which
 * IDs 0..38 are upstream codes (n_upstream = 39: every tag of the reference's field.json, IDs 1..32) and IDs 39..586 are a
 * stand-in lexicode from tools/gen_family36.c that keeps the codebook search at its real size — NOT upstream IDs; an
 * integrator who needs them passes upstream's tag36h11.c table through ck_config_t.families.
 * Returns NULL for unknown names. */
const ck_family_t *ck_family_builtin(const char *name);

/* ---- detector configuration -------------------------------------------------------------------- */
#define CK_MAX_FAMILIES 4

typedef struct ck_config {
    int32_t width, height;        /* frame geometry, fixed per handle (cf. Detector::new(width,height,..)) */
    int32_t max_batch;            /* frames per ck_detect_batch call the workspace is sized for.  Every stage keeps worst-case capacity
                                   * resident, about 86 bytes per pixel of quad-stage image and frame at the default capacities
                                   * (1280 x 800: 88 MB per frame of max_batch; max_points_per_frame / max_clusters_per_frame
                                   * shrink it); ck_create fails with CK_ENOMEM when the device has no room */
    int32_t device;               /* HIP device ordinal */
    /* AprilTag-3 detector defaults the reference inherits unchanged */
    int32_t quad_decimate;        /* 1 (full resolution) or 2 (AT3 default) */
    int32_t min_white_black_diff; /* 5 */
    int32_t min_component_px;     /* 25: components smaller than this emit no boundary points (>= 1) */
    int32_t min_cluster_pixels;   /* 24: smallest cluster handed to the quad fitter */
    int32_t max_nmaxima;          /* 10 */
    double cos_critical_rad;      /* cos(10 deg) */
    double max_line_fit_mse;      /* 10.0 */
    int32_t refine_edges;         /* 1 */
    double decode_sharpening;     /* 0.25 */
    int32_t max_hamming;          /* bits_corrected: 3 with a config, 1 without (lib.rs:230,280) */
    int32_t n_families;
    const ck_family_t *families[CK_MAX_FAMILIES];
    /* capacities of the irregular stages (0 = derive from geometry: 4 points per pixel — the most a frame can produce,
     * one per forward neighbour —, one cluster per 32 pixels, 1024 quads) */
    int32_t max_points_per_frame;
    int32_t max_clusters_per_frame;
    int32_t max_quads_per_frame;
} ck_config_t;

void ck_config_default(ck_config_t *cfg, int32_t width, int32_t height, int32_t max_batch);

/* One decoded tag.  Corner order = apriltag's: bottom-left, bottom-right, top-right, top-left in the
 * tag's own frame, which is what corner_points_from_center assumes (chalkydri_sqpnp/src/lib.rs:383-388). */
typedef struct ck_detection {
    int32_t id;
    int32_t hamming;
    int32_t family;          /* index into ck_config_t.families */
    float decision_margin;
    double c[2];             /* centre, pixels */
    double p[4][2];          /* corners, pixels */
} ck_detection_t;

typedef struct ck_handle ck_handle_t;

int ck_abi_version(void);
const char *ck_strerror(int code);
const char *ck_last_error(void); /* text of the most recent CK_EDEVICE on this thread */
int ck_device_count(void);       /* 0 when no HIP device is visible */

/* Frame geometry accepted by ck_create: any width and height of 16..4095 pixels (at least 8 after quad_decimate) — what an
 * image_u8_t can describe within the 13-bit half-pixel coordinates of the boundary points; quad_decimate 1 or 2
 * (CK_EUNSUPPORTED otherwise); min_component_px >= 1; 1..CK_MAX_FAMILIES families, each with n_upstream <= ncodes.
 * CK_EINVAL for everything else.  Arguments are validated before a device is looked for (CK_ENODEVICE). */
int ck_create(const ck_config_t *cfg, ck_handle_t **out);
void ck_destroy(ck_handle_t *h);

/* ---- full pipeline ------------------------------------------------------------------------------ */
/* Host frames in, host detections out (H2D copy inside).  dets is [n][cap_per_frame]; counts[n] gets the
 * number written per frame (sorted by id, then hamming, then margin desc); status[n] may be NULL. */
int ck_detect_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, ck_detection_t *dets,
                    int32_t cap_per_frame, int32_t *counts, uint32_t *status);

/* Frames already resident in HBM: d_frames is a device pointer to n frames of height rows, row pitch
 * `stride` bytes, frame pitch `frame_pitch` bytes.  Results land in host arrays as above. */
int ck_detect_batch_device(ck_handle_t *h, const uint8_t *d_frames, int32_t n, int32_t stride,
                           int64_t frame_pitch, ck_detection_t *dets, int32_t cap_per_frame,
                           int32_t *counts, uint32_t *status);

/* Device-resident staging owned by the handle (used by benchmarks / streaming callers). */
int ck_upload_frames(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n);
int ck_detect_uploaded(ck_handle_t *h, int32_t n, ck_detection_t *dets, int32_t cap_per_frame,
                       int32_t *counts, uint32_t *status);

/* ---- stage entry points (parity tests + roofline measurement) ------------------------------------ */
/* thresh_out: [n][height][width] bytes in {0,127,255}.  Runs on frames staged by ck_upload_frames when
 * imgs == NULL. */
int ck_threshold_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, uint8_t *thresh_out);
/* labels_out: [n][height][width] u32, canonical label = smallest pixel index (y*width+x) of the
 * component, 0xFFFFFFFF for 127-pixels.  sizes_out (optional): component size at every pixel. */
int ck_segment_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, uint32_t *labels_out,
                     uint32_t *sizes_out);
/* Times only the threshold+segment kernels on the already-uploaded frames: runs them `iters` times on the
 * handle's stream between HIP events and returns the mean milliseconds per pass in *ms_out. */
int ck_time_threshold_segment(ck_handle_t *h, int32_t n, int32_t iters, float *ms_out);

/* Boundary points grouped in clusters.  A point packs x,y in half-pixel units and the gradient sign. */
typedef struct ck_cluster_point {
    uint16_t x, y;  /* half-pixel coordinates 2*px+dx, 2*py+dy */
    int8_t gx, gy;  /* sign of the black→white step along x / y, in {-1,0,1} */
    uint16_t pad;
} ck_cluster_point_t;
typedef struct ck_cluster {
    uint32_t rep0, rep1;  /* canonical labels of the two components, rep0 < rep1 */
    uint32_t start, count; /* range in the frame's point array */
} ck_cluster_t;
/* Emits clusters sorted by (rep0,rep1) and points sorted by emission order inside each cluster. */
int ck_clusters_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, ck_cluster_t *clusters,
                      int32_t cluster_cap, int32_t *n_clusters, ck_cluster_point_t *points,
                      int32_t point_cap, int32_t *n_points);

typedef struct ck_quad {
    double p[4][2];          /* corners in pixels, winding as fitted */
    int32_t reversed_border;
    uint32_t rep0, rep1;     /* cluster the quad came from */
} ck_quad_t;
/* Candidate quads after fit (+ edge refinement when enabled), sorted by (rep0,rep1). */
int ck_quads_batch(ck_handle_t *h, const ck_image_u8_t *imgs, int32_t n, ck_quad_t *quads,
                   int32_t quad_cap, int32_t *n_quads);

/* per-stage timings of the last ck_detect_* call, milliseconds (HIP events on the handle's stream) */
typedef struct ck_stage_ms {
    float h2d, threshold, segment, clusters, quads, decode, d2h, total;
} ck_stage_ms_t;
int ck_last_stage_ms(ck_handle_t *h, ck_stage_ms_t *out);

/* ---- CAT: chalkydri-apriltags experimental detector front-end --------------------------------------- */
/* Classes are the reference's `Color` enum: 0 Black, 1 White, 2 Other (src/utils.rs:1-6). */
/* calc_otsu on an RGB8 frame [h][w][3] (lib.rs:191-259). classes_out: [h][w]. */
int ck_cat_calc_otsu(ck_handle_t *h, const uint8_t *rgb, int32_t width, int32_t height,
                     uint8_t *classes_out);
/* thresh(): fixed <60 / >160 split (lib.rs:319-334). */
int ck_cat_thresh(ck_handle_t *h, const uint8_t *rgb, int32_t width, int32_t height,
                  uint8_t *classes_out);
/* detect_corners over a class map (lib.rs:291-309,345-400); points in the reference's x-major order. */
int ck_cat_detect_corners(ck_handle_t *h, const uint8_t *classes, int32_t width, int32_t height,
                          uint32_t *points_xy, int32_t cap, int32_t *n_points);
/* check_edges (lib.rs:409-499): lines (x1,y1,x2,y2) in the reference's push order. */
int ck_cat_check_edges(ck_handle_t *h, const uint8_t *classes, int32_t width, int32_t height,
                       const uint32_t *points_xy, int32_t n_points, uint32_t *lines_xyxy, int32_t cap,
                       int32_t *n_lines);
/* connected_components (lib.rs:501-549): canonical root (min index) and component size per pixel;
 * pixels of class Other and never-visited pixels are their own singleton sets, as in UnionFind::new. */
int ck_cat_connected_components(ck_handle_t *h, const uint8_t *classes, int32_t width, int32_t height,
                                uint32_t *roots_out, uint32_t *sizes_out);
/* process_frame = calc_otsu → detect_corners → check_edges (lib.rs:265-287). Returns CK_EINVAL when
 * rgb_len != width*height*3 (the reference asserts). */
int ck_cat_process_frame(ck_handle_t *h, const uint8_t *rgb, size_t rgb_len, int32_t width,
                         int32_t height, uint8_t *classes_out, uint32_t *points_xy, int32_t point_cap,
                         int32_t *n_points, uint32_t *lines_xyxy, int32_t line_cap, int32_t *n_lines);

/* ---- SQPnP ------------------------------------------------------------------------------------------ */
/* Isometry = translation + unit quaternion (w,x,y,z), matching nalgebra's Isometry3<f64> content. */
typedef struct ck_iso3 {
    double t[3];
    double q[4]; /* w, x, y, z */
} ck_iso3_t;

typedef struct ck_sqpnp_params {
    int32_t max_iter;  /* 15  (lib.rs:203) */
    double tol_sq;     /* 1e-16 (lib.rs:204); SqPnP::tolerance(t) sets t*t (lib.rs:219-222) */
} ck_sqpnp_params_t;

/* One solve_robot_pose problem (lib.rs:297-304). tags[] / bearings[] live in caller-provided arrays. */
typedef struct ck_sqpnp_problem {
    int32_t n_tags;            /* points_isometry.len() */
    int32_t n_bearings;        /* points_2d.len(); must equal 4*n_tags for a solve (lib.rs:255) */
    int32_t tag_offset;        /* first tag of this problem in the tags[] array */
    int32_t bearing_offset;    /* first bearing in the bearings[] array (3 doubles each) */
    ck_iso3_t robot_to_cam;
    double gyro;
    double sign_change_error;  /* 600.0 at the reference call site (apriltags/src/lib.rs:6,337) */
} ck_sqpnp_problem_t;

typedef struct ck_sqpnp_result {
    int32_t valid;         /* 0 = the reference would return None */
    int32_t pad;
    double rot[9];         /* pivoted robot rotation, row-major 3x3 */
    double pos[3];         /* pivoted robot position */
    double std_devs[3];
    double yaw;            /* euler_angles().2 of rot — what the caller publishes (apriltags/src/lib.rs:343) */
    double energy;         /* pure geometric energy r^T Omega r of the chosen candidate */
} ck_sqpnp_result_t;

void ck_sqpnp_params_default(ck_sqpnp_params_t *p);
int ck_sqpnp_solve_batch(ck_handle_t *h, const ck_sqpnp_params_t *params,
                         const ck_sqpnp_problem_t *problems, int32_t n, const ck_iso3_t *tags,
                         int32_t n_tags_total, const double *bearings, int32_t n_bearings_total,
                         ck_sqpnp_result_t *out);
/* SqPnP::create_solver_camera_transform (lib.rs:430-461); pure host arithmetic, no device needed. */
void ck_sqpnp_create_solver_camera_transform(double fwd_m, double left_m, double up_m, double roll_deg,
                                             double pitch_deg, double yaw_deg, ck_iso3_t *out);

/* ---- glue: AprilTags::process ------------------------------------------------------------------------ */
typedef struct ck_opencv5 {
    double fx, fy, cx, cy, k1, k2, p1, p2, k3;
} ck_opencv5_t;

/* The 64-byte record whacknet puts on the wire (crates/whacknet/src/lib.rs:43-66). */
typedef struct ck_vision_measurement {
    double pose_x, pose_y, pose_rot;
    double std_x, std_y, std_rot;
    uint64_t ts;
    uint8_t camera_id;
    uint8_t tag_count;
    uint8_t reserved[6];
} ck_vision_measurement_t;

typedef struct ck_field_tag {
    int32_t id;
    int32_t pad;
    ck_iso3_t pose;
} ck_field_tag_t;

typedef struct ck_process_params {
    ck_opencv5_t cam;
    ck_iso3_t robot_to_cam;
    const ck_field_tag_t *field;   /* known field tags (field.json) */
    int32_t n_field;
    uint8_t camera_id;
    double sign_change_error;
    ck_sqpnp_params_t sqpnp;
    int32_t allow_unverified_ids;  /* 0 (default): detections with id >= their family's n_upstream never reach the solver */
} ck_process_params_t;

/* detect → known-tag filter → unproject → solve_robot_pose → measurement, per frame.
 * gyro[n]: heading per frame; has_gyro[n] == 0 reproduces the "no gyro, no solve" gate (lib.rs:330).
 * out[i].tag_count = number of ALL detections in the frame (lib.rs:354); frames without a pose get a
 * zeroed record with tag_count 0 (lib.rs:365-376). */
int ck_process_batch_device(ck_handle_t *h, const uint8_t *d_frames, int32_t n, int32_t stride,
                            int64_t frame_pitch, const ck_process_params_t *pp, const double *gyro,
                            const uint8_t *has_gyro, ck_vision_measurement_t *out, int32_t *valid);
int ck_process_uploaded(ck_handle_t *h, int32_t n, const ck_process_params_t *pp, const double *gyro,
                        const uint8_t *has_gyro, ck_vision_measurement_t *out, int32_t *valid);

/* ---- ingest ring: pinned host slots + asynchronous upload ---------------------------------------------------------------
 * What the reference's camera layer hands the detector is a pooled host buffer per frame {buf,width,height,stride} in one
 * of the 8-bit-luma formats (crates/chalkydri/src/cameras/gst_to_cu.rs:49-72,131-188; fourcc GREY/GRAY/Y800, or the Y
 * plane that leads NV12/NV21/I420/YV12).  A ring owns `n_slots` pinned host buffers and as many device buffers of
 * max_batch frames each: the caller writes frames into a slot, submits it (one asynchronous copy on the ring's copy
 * stream) and processes it; submitting slot k+1 before processing slot k overlaps the upload with the compute. */
typedef struct ck_ingest ck_ingest_t;
int ck_ingest_create(ck_handle_t *h, int32_t n_slots, ck_ingest_t **out);
void ck_ingest_destroy(ck_ingest_t *ing);
int32_t ck_ingest_stride(const ck_ingest_t *ing);                                  /* row stride of a slot frame, bytes */
uint8_t *ck_ingest_frame(ck_ingest_t *ing, int32_t slot, int32_t index);           /* pinned host memory of one frame */
/* stride-aware copy of a caller frame into the slot; fourcc as four ASCII bytes, little-endian ("GREY" = 0x59455247) */
int ck_ingest_write(ck_ingest_t *ing, int32_t slot, int32_t index, const ck_image_u8_t *img, uint32_t fourcc);
int ck_ingest_submit(ck_ingest_t *ing, int32_t slot, int32_t n);
/* n = frames the output (and gyro) arrays hold; CK_EINVAL unless it is the count the slot was submitted with */
int ck_detect_ingested(ck_ingest_t *ing, int32_t slot, int32_t n, ck_detection_t *dets, int32_t cap_per_frame,
                       int32_t *counts, uint32_t *status);
int ck_process_ingested(ck_ingest_t *ing, int32_t slot, int32_t n, const ck_process_params_t *pp, const double *gyro,
                        const uint8_t *has_gyro, ck_vision_measurement_t *out, int32_t *valid);

/* ---- multi-GPU: the final pose gather ----------------------------------------------------------------------------------
 * Frames shard over GPUs without any data-path collective (one handle, one process or host thread per GPU).  The only
 * exchange is the gather of the 64-byte records (the wire struct of crates/whacknet/src/lib.rs:43-66): ONE ncclAllGather
 * (RCCL over xGMI) of n x 64 bytes per batch, on the communicator's own stream beside the handle's next batch.  The host distributes the 128-byte id that rank 0
 * obtains from ck_comm_unique_id over whatever channel it has (the reference has UDP; the Python mirror uses
 * torch.distributed's store).  librccl is opened on first use: CK_EUNSUPPORTED when it cannot be loaded. */
#define CK_COMM_ID_BYTES 128
#define CK_BACKEND_HIP 1
typedef struct ck_comm ck_comm_t;
int ck_backend(const ck_handle_t *h);                      /* CK_BACKEND_HIP: the library has no CPU backend */
int ck_comm_unique_id(uint8_t *id_out);                    /* rank 0: id_out[CK_COMM_ID_BYTES] */
int ck_comm_create(ck_handle_t *h, const uint8_t *id, int32_t world, int32_t rank, ck_comm_t **out); /* collective: every rank calls it */
void ck_comm_destroy(ck_comm_t *comm);
/* Gathers the records the handle's last ck_process_* call produced (they are still on the device) from every rank into
 * out[world*rows] in rank order.  n_valid = the frames of that call (CK_EINVAL when it is not: the send buffer is the handle's
 * own, so a wrong count would ship stale records); rows = the common row count of the collective, the same on every rank,
 * n_valid <= rows <= max_batch: the library pads a ragged last shard with empty records (all zero, tag_count = 0 — what a
 * frame without a pose publishes anyway: crates/apriltags/src/lib.rs:365-376).  `out` may be a host or a device pointer.
 * sync = 0 only enqueues: the handle's stream copies the records aside (the next ck_process_* call may follow at once) and the
 * collective runs on the communicator's stream; `out` is complete after ck_comm_sync (or a later call with sync = 1) — NOT after
 * a synchronisation of the handle alone.  A rank whose n_valid fails the local check still takes part with `rows` empty records
 * and then returns CK_EINVAL, so its peers complete; after any other error of a collective destroy the communicator on every rank.
 * Destroy a communicator before the handle it was made for. */
int ck_gather_poses(ck_handle_t *h, ck_comm_t *comm, int32_t n_valid, int32_t rows, ck_vision_measurement_t *out, int32_t sync);
int ck_comm_sync(ck_comm_t *comm);
const char *ck_comm_library(const ck_comm_t *comm);      /* path of the librccl the communicator's calls resolved to */

/* OpenCVModel5 unprojection of pixel points to bearings (x,y,1)/norm; ok[i]=0 when it does not converge. */
int ck_unproject_opencv5(const ck_opencv5_t *cam, const double *px, int32_t n, double *bearings,
                         uint8_t *ok);

/* fp64 conformance probe used by the parity tests: out[i] = op(a[i], b[i]) computed on the device with
 * the same flags as the kernels. op: 0 add, 1 mul, 2 div, 3 sqrt(a), 4 a*b+c style unfused (a*b)+a. */
int ck_selftest_fp64(ck_handle_t *h, int32_t op, const double *a, const double *b, int32_t n,
                     double *out);

#ifdef __cplusplus
}
#endif
#endif /* CHALKYDRI_HIP_H */
