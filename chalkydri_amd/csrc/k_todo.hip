// Entry points whose kernels are not built yet: they exist so the ABI is complete and fail loudly.
#include "ck_internal.h"


extern "C" {
int ck_cat_calc_otsu(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint8_t *) { return CK_EUNSUPPORTED; }
int ck_cat_thresh(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint8_t *) { return CK_EUNSUPPORTED; }
int ck_cat_detect_corners(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint32_t *, int32_t, int32_t *) { return CK_EUNSUPPORTED; }
int ck_cat_check_edges(ck_handle_t *, const uint8_t *, int32_t, int32_t, const uint32_t *, int32_t, uint32_t *, int32_t, int32_t *) { return CK_EUNSUPPORTED; }
int ck_cat_connected_components(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint32_t *, uint32_t *) { return CK_EUNSUPPORTED; }
int ck_cat_process_frame(ck_handle_t *, const uint8_t *, size_t, int32_t, int32_t, uint8_t *, uint32_t *, int32_t, int32_t *, uint32_t *, int32_t, int32_t *) { return CK_EUNSUPPORTED; }
int ck_sqpnp_solve_batch(ck_handle_t *, const ck_sqpnp_params_t *, const ck_sqpnp_problem_t *, int32_t, const ck_iso3_t *, int32_t, const double *, int32_t, ck_sqpnp_result_t *) { return CK_EUNSUPPORTED; }
void ck_sqpnp_create_solver_camera_transform(double, double, double, double, double, double, ck_iso3_t *) {}
int ck_process_batch_device(ck_handle_t *, const uint8_t *, int32_t, int32_t, int64_t, const ck_process_params_t *, const double *, const uint8_t *, ck_vision_measurement_t *, int32_t *) { return CK_EUNSUPPORTED; }
int ck_process_uploaded(ck_handle_t *, int32_t, const ck_process_params_t *, const double *, const uint8_t *, ck_vision_measurement_t *, int32_t *) { return CK_EUNSUPPORTED; }
int ck_unproject_opencv5(const ck_opencv5_t *, const double *, int32_t, double *, uint8_t *) { return CK_EUNSUPPORTED; }
}
