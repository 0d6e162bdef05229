// Synthetic frame renderer: deterministic inputs + ground truth for tests and bench.py.
#ifndef CK_SYNTH_H
#define CK_SYNTH_H
#include "chalkydri_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ck_synth_tag {
    int32_t family;       /* index into the family array passed to the renderer */
    int32_t id;
    double H[9];          /* row-major homography: tag coords (border edge at +-1) -> pixels */
    double corners[4][2]; /* ground truth in detection order (-1,1),(1,1),(1,-1),(-1,-1) */
    double center[2];
} ck_synth_tag_t;

typedef struct ck_synth_params {
    int32_t width, height, n_tags;
    int32_t min_side, max_side;  /* apparent tag side in pixels */
    int32_t max_tilt_t64;        /* tilt = 2*atan(t/64), |t| <= this (33 -> 55 degrees) */
    int32_t noise_amp;           /* uniform noise in [-amp, amp] */
    int32_t ramp_amp;            /* low-frequency linear ramp amplitude */
    int32_t black, white, bg;
    int32_t family_mode;         /* 0: every tag from fams[0]; 1: tag k from fams[k % n_fams] */
    int32_t max_id;              /* ids drawn from [0, max_id]; -1 = whole family */
} ck_synth_params_t;

void ck_synth_params_default(ck_synth_params_t *p, int32_t width, int32_t height, int32_t n_tags);
void ck_synth_background(uint64_t seed, const ck_synth_params_t *p, uint8_t *out, int32_t stride);
void ck_synth_fill_truth(ck_synth_tag_t *t);
int ck_synth_draw_tag(const ck_synth_params_t *p, const ck_family_t *fam, const ck_synth_tag_t *t,
                      uint8_t *out, int32_t stride);
int ck_synth_render(uint64_t seed, const ck_synth_params_t *p, const ck_family_t *const *fams, int32_t n_fams,
                    uint8_t *out, int32_t stride, ck_synth_tag_t *truth, int32_t truth_cap, int32_t *n_truth);

#ifdef __cplusplus
}
#endif
#endif
