/*
 * ck_oracle.h — CPU oracle for the AprilTag detect + SQPnP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it may be imported, linked or
 * executed only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and only as the
 * checker.  The product (chalkydri_amd/) never calls into it and has no CPU fallback.
 *
 * PARITY UNPINNED for the detector: the arithmetic that produces tag IDs/corners in the reference lives in
 * the external AprilTag-3 C library reached through the `apriltag`/`apriltag-sys` git crates
 * (crates/apriltags/Cargo.toml:10-11, branch master, no Cargo.lock) — not under /root/reference, and no
 * Rust toolchain exists here.  The reference holds no golden vectors, KATs or image fixtures for this path
 * (its only test is a struct-size check, crates/whacknet/src/lib.rs:92-95).  detector.c therefore restates
 * the published AprilTag-3 algorithm (stage structure and defaults per SURVEY.md Appendix B) with
 * integer-exact choices documented at each function, and is pinned by synthetic ground truth instead.
 * cat.c and sqpnp.c restate code that IS under /root/reference line by line (citations inline); their
 * third-party arithmetic (statrs 0.18.0, nalgebra 0.34.1) is restated from the published algorithms.
 */
#ifndef CK_ORACLE_H
#define CK_ORACLE_H

#include "../include/chalkydri_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- detector stages (detector.c) ---------------------------------------------------------------- */
void ora_decimate(const uint8_t *img, int w, int h, int stride, int f, uint8_t *out /* [h/f][w/f] */);
void ora_threshold(const uint8_t *img, int w, int h, int stride, int min_white_black_diff,
                   uint8_t *out /* [h][w] */);
void ora_segment(const uint8_t *thresh, int w, int h, uint32_t *labels, uint32_t *sizes);
int ora_clusters(const uint8_t *thresh, const uint32_t *labels, const uint32_t *sizes, int w, int h,
                 int min_component_px, ck_cluster_t *clusters, int cluster_cap, int *n_clusters,
                 ck_cluster_point_t *points, int point_cap, int *n_points);
int ora_fit_quads(const uint8_t *quad_img, int qw, int qh, int qstride, const uint8_t *orig_img, int w, int h,
                  int stride, const ck_config_t *cfg, const ck_cluster_t *clusters, int n_clusters,
                  const ck_cluster_point_t *points, ck_quad_t *quads, int quad_cap, int *n_quads);
int ora_decode_quads(const uint8_t *img, int w, int h, int stride, const ck_config_t *cfg,
                     const ck_quad_t *quads, int n_quads, ck_detection_t *dets, int det_cap, int *n_dets);
/* whole pipeline on one frame; returns CK_OK and a status word like the product */
int ora_detect(const uint8_t *img, int w, int h, int stride, const ck_config_t *cfg, ck_detection_t *dets,
               int det_cap, int *n_dets, uint32_t *status);
/* threshold + segment only: the part timed as cpu_baseline for the roofline stage */
void ora_threshold_segment(const uint8_t *img, int w, int h, int stride, int min_white_black_diff,
                           uint8_t *thresh, uint32_t *labels, uint32_t *sizes);

/* ---- CAT (cat.c) ------------------------------------------------------------------------------------ */
uint8_t ora_cat_grayscale(uint8_t r, uint8_t g, uint8_t b);
void ora_cat_calc_otsu(const uint8_t *rgb, int w, int h, uint8_t *classes);
void ora_cat_thresh(const uint8_t *rgb, int w, int h, uint8_t *classes);
int ora_cat_detect_corners(const uint8_t *classes, int w, int h, uint32_t *points_xy, int cap);
int ora_cat_check_edges(const uint8_t *classes, int w, int h, const uint32_t *points_xy, int n_points,
                        uint32_t *lines_xyxy, int cap);
/* reference-faithful sequential UnionFind (union by size, ties -> root1) */
void ora_cat_connected_components(const uint8_t *classes, int w, int h, uint64_t *parent, uint64_t *sizes);
/* the same partition, canonicalised: root = min index of the set, size per pixel */
void ora_cat_connected_components_canonical(const uint8_t *classes, int w, int h, uint32_t *roots,
                                            uint32_t *sizes);

/* ---- SQPnP (sqpnp.c) ---------------------------------------------------------------------------------- */
int ora_sqpnp_solve_robot_pose(const ck_sqpnp_params_t *params, const ck_iso3_t *tags, int n_tags,
                               const double *bearings, int n_bearings, const ck_iso3_t *robot_to_cam,
                               double gyro, double sign_change_error, ck_sqpnp_result_t *out);
void ora_sqpnp_create_solver_camera_transform(double fwd_m, double left_m, double up_m, double roll_deg,
                                              double pitch_deg, double yaw_deg, ck_iso3_t *out);
int ora_unproject_opencv5(const ck_opencv5_t *cam, const double *px, int n, double *bearings, uint8_t *ok);
int ora_process_frame(const uint8_t *img, int w, int h, int stride, const ck_config_t *cfg,
                      const ck_process_params_t *pp, double gyro, int has_gyro, ck_vision_measurement_t *out,
                      int *valid);

/* bench_threads.c: n_work frames (cycling through n_frames) through ora_process_frame on `threads` POSIX threads; returns the
 * wall seconds of the threaded region (negative on failure), *n_valid = frames that produced a pose.  bench.py's cpu_baseline leg. */
double ora_bench_process(const uint8_t *frames, int w, int h, int stride, size_t frame_pitch, int n_frames, int n_work,
                         const ck_config_t *cfg, const ck_process_params_t *pp, const double *gyro, int threads, int *n_valid);

#ifdef __cplusplus
}
#endif
#endif
