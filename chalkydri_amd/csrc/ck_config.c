// Host-side defaults and error strings of the C ABI (no device code).
#include "chalkydri_hip.h"
#include <string.h>

int ck_abi_version(void) { return CK_ABI_VERSION; }

const char *ck_strerror(int code) {
    switch (code) {
    case CK_OK: return "ok";
    case CK_EINVAL: return "invalid argument";
    case CK_ENOMEM: return "out of memory";
    case CK_EDEVICE: return "HIP runtime error";
    case CK_ENODEVICE: return "no HIP device (this library has no CPU fallback)";
    case CK_ECAPACITY: return "batch exceeds handle capacity";
    case CK_EUNSUPPORTED: return "unsupported in this build";
    default: return "unknown error";
    }
}

// Defaults = what the reference gets from DetectorBuilder::default() + add_family_bits(tag36h11, 3)
// (crates/apriltags/src/lib.rs:45,228-233,258-262) with AprilTag-3's stock parameters, with ONE
// deliberate difference: quad_decimate is 1 here, 2.0 upstream.  The roofline contract of the threshold+segment stage is
// quoted at full resolution ("f = 1 for the headline"), and bench.py / the parity tests run there; a caller
// that wants the reference's exact front end sets cfg.quad_decimate = 2 (bench.py reports that run under "also",
// tests/test_gpu_detect.py::test_detect_matches_oracle_at_the_default_decimation covers it).  DESIGN.md §2 lists it.
void ck_config_default(ck_config_t *cfg, int32_t width, int32_t height, int32_t max_batch) {
    memset(cfg, 0, sizeof *cfg);
    cfg->width = width; cfg->height = height; cfg->max_batch = max_batch > 0 ? max_batch : 1;
    cfg->device = 0;
    cfg->quad_decimate = 1;
    cfg->min_white_black_diff = 5;
    cfg->min_component_px = 25;
    cfg->min_cluster_pixels = 24;
    cfg->max_nmaxima = 10;
    cfg->cos_critical_rad = 0.984807753012208; /* cos(10 deg) */
    cfg->max_line_fit_mse = 10.0;
    cfg->refine_edges = 1;
    cfg->decode_sharpening = 0.25;
    cfg->max_hamming = 3;
    cfg->n_families = 1;
    cfg->families[0] = ck_family_builtin("tag36h11");
}

void ck_sqpnp_params_default(ck_sqpnp_params_t *p) {
    p->max_iter = 15;   /* chalkydri_sqpnp/src/lib.rs:203 */
    p->tol_sq = 1e-16;  /* chalkydri_sqpnp/src/lib.rs:204 */
}
