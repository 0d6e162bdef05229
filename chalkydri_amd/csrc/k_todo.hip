// Entry points whose kernels are not built yet: they exist so the ABI is complete and fail loudly.
#include "ck_internal.h"


extern "C" {
int ck_cat_calc_otsu(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint8_t *) { return CK_EUNSUPPORTED; }
int ck_cat_thresh(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint8_t *) { return CK_EUNSUPPORTED; }
int ck_cat_detect_corners(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint32_t *, int32_t, int32_t *) { return CK_EUNSUPPORTED; }
int ck_cat_check_edges(ck_handle_t *, const uint8_t *, int32_t, int32_t, const uint32_t *, int32_t, uint32_t *, int32_t, int32_t *) { return CK_EUNSUPPORTED; }
int ck_cat_connected_components(ck_handle_t *, const uint8_t *, int32_t, int32_t, uint32_t *, uint32_t *) { return CK_EUNSUPPORTED; }
int ck_cat_process_frame(ck_handle_t *, const uint8_t *, size_t, int32_t, int32_t, uint8_t *, uint32_t *, int32_t, int32_t *, uint32_t *, int32_t, int32_t *) { return CK_EUNSUPPORTED; }
}
