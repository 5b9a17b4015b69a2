/*
 * nl_oracle.h -- CPU oracle for Nightlight's per-pixel stacking hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * This is a plain-C restatement of the reference's Go code (scalar loops,
 * fp32 arithmetic evaluated left to right, no FMA contraction); every
 * function cites the reference file:line it follows (paths relative to the
 * reference tree).  The reference is Go and no Go toolchain exists in the
 * build image, so the reference itself cannot be compiled (no oracle/_ref).
 *
 * Pinning status: QSelectMedian is pinned by the reference's only test on
 * this path (internal/qsort/qsort_test.go:25-53, restated in
 * tests/test_oracle_qsort.py).  The reference has no test, golden vector or
 * fixture for any Stack* function, MeanStdDev, LinearRegression,
 * EstimateNoise or MedianFilter3x3: for those, parity is UNPINNED by the
 * reference and is pinned only by hand-derived known answers
 * (tests/golden/kat.json) and by an independent second restatement
 * (oracle/pyref.py) that must agree bit for bit.
 */
#ifndef NL_ORACLE_H
#define NL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stack modes, numbered as internal/ops/stack/stack.go:33-42 */
enum {
    NLO_ST_MEDIAN = 0,
    NLO_ST_MEAN = 1,
    NLO_ST_SIGMA = 2,
    NLO_ST_WINSOR_SIGMA = 3,
    NLO_ST_MAD_SIGMA = 4,
    NLO_ST_LINEAR_FIT = 5,
    NLO_ST_AUTO = 6
};

/* weighting modes, internal/ops/stack/stack.go:57-63 */
enum {
    NLO_WEIGHT_NONE = 0,
    NLO_WEIGHT_EXPOSURE = 1,
    NLO_WEIGHT_INVERSE_NOISE = 2,
    NLO_WEIGHT_INVERSE_HFR = 3
};

/* error codes of nlo_stack_apply / nlo_get_weights */
enum {
    NLO_OK = 0,
    NLO_ERR_INVALID_MODE = -1,        /* "invalid stacking mode"            stack.go:119 */
    NLO_ERR_MISSING_EXPOSURE = -2,    /* "... Missing exposure information" stack.go:238 */
    NLO_ERR_INVALID_WEIGHTING = -3,   /* "Invalid weighting mode %d"        stack.go:267 */
    NLO_ERR_WEIGHTED_MAD = -4,        /* reference panics here              stack.go:185 */
    NLO_ERR_NO_INPUTS = -5            /* "stack operator needs inputs"      stack.go:103 */
};

/* ---- internal/qsort/qsort.go ---- */
void  nlo_qsort_f32(float *a, int n);                       /* :26-32  */
int   nlo_qpartition_f32(float *a, int n);                  /* :38-56  */
float nlo_qselect_first_quartile_f32(float *a, int n);      /* :61-63  */
float nlo_qselect_median_f32(float *a, int n);              /* :68-82  */
float nlo_qselect_f32(float *a, int n, int k);              /* :94-126 */

/* ---- internal/stats/stats.go ---- */
void   nlo_mean_stddev(const float *xs, int n, float *mean, float *stddev);   /* :246-261 */
void   nlo_linear_regression(const float *xs, const float *ys, int n,
                             float *slope, float *intercept, float *xmean,
                             float *xstddev, float *ymean, float *ystddev);   /* :569-586 */
void   nlo_min_mean_max(const float *data, int64_t n,
                        float *mn, float *mean, float *mx);                   /* :264-277 */
double nlo_variance(const float *data, int64_t n, float mean);                /* :280-287 */
/* the same two in the 4-lane order of the AVX2 assembly
 * (internal/stats/stats_amd64.s:28-92, :102-143); n is rounded up to a
 * multiple of 4 by the assembly (it over-reads), callers pass n%4==0 */
void   nlo_min_mean_max_lanes4(const float *data, int64_t n,
                               float *mn, float *mean, float *mx);
double nlo_variance_lanes4(const float *data, int64_t n, float mean);

/* ---- internal/stats/noise.go:32-55 ---- */
float  nlo_estimate_noise(const float *data, int64_t n, int32_t width);

/* ---- internal/median/median3x3.go ---- */
float  nlo_median9(float *a);                                                 /* :85-110 */
float  nlo_median_f32(float *a, int n);                                       /* :115-119 */
/* internal/median/gather.go:26-38, ops/pre/badpixels.go:54-77, star/findstars.go:187-200 */
float  nlo_gather_and_median(const float *data, int64_t n, int32_t index, const int32_t *mask, int mask_len,
                             float *buffer);
void   nlo_median_filter_mask(float *out, const float *data, int64_t n, const int32_t *mask, int mask_len,
                              unsigned char *full);
int    nlo_create_mask(int32_t width, float radius, int32_t *mask, int cap);
void   nlo_median_filter_3x3(float *out, const float *data, int64_t n, int32_t width); /* :26-77 */

/* ---- internal/ops/stack/stack.go : the nine per-pixel stackers ----
 * lights = n_frames pointers, each to npix floats (one slice per frame, as
 * fits.Image.Data is); res = npix floats.  clip counters are int32 in the
 * reference (stack.go:140); here int64, equal while below 2^31. */
void nlo_stack_median(const float *const *lights, int n_frames, int64_t npix,
                      float ref_loc, float *res);                                         /* :274-303 */
void nlo_stack_mean(const float *const *lights, int n_frames, int64_t npix,
                    float ref_loc, float *res);                                           /* :307-333 */
void nlo_stack_mean_weighted(const float *const *lights, const float *weights, int n_frames,
                             int64_t npix, float ref_loc, float *res);                    /* :337-366 */
void nlo_stack_sigma(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                     float sigma_low, float sigma_high, float *res,
                     int64_t *clip_low, int64_t *clip_high);                              /* :372-436 */
void nlo_stack_sigma_weighted(const float *const *lights, const float *weights, int n_frames,
                              int64_t npix, float ref_loc, float sigma_low, float sigma_high,
                              float *res, int64_t *clip_low, int64_t *clip_high);         /* :442-531 */
void nlo_stack_mad_sigma(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                         float sigma_low, float sigma_high, float *res,
                         int64_t *clip_low, int64_t *clip_high);                          /* :536-605 */
void nlo_stack_winsor_sigma(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                            float sigma_low, float sigma_high, float *res,
                            int64_t *clip_low, int64_t *clip_high);                       /* :611-705 */
void nlo_stack_winsor_sigma_weighted(const float *const *lights, const float *weights,
                                     int n_frames, int64_t npix, float ref_loc,
                                     float sigma_low, float sigma_high, float *res,
                                     int64_t *clip_low, int64_t *clip_high);              /* :710-829 */
void nlo_stack_linear_fit(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                          float sigma_low, float sigma_high, float *res,
                          int64_t *clip_low, int64_t *clip_high);                         /* :834-918 */

/* ---- stack.go:45-55, :231-270, :115-227, :924-944 ---- */
int  nlo_auto_select_mode(int n_frames);
/* per_frame = exposure (mode 1), noise (mode 2) or HFR (mode 3) of each frame.
 * weights_out must hold n_frames floats; *has_weights is 0 for mode none. */
int  nlo_get_weights(int weighting, const float *per_frame, int n_frames,
                     float *weights_out, int *has_weights, int *bad_index);
/* timing aid: pin worker t of nlo_stack_apply's pool to the t-th allowed CPU (0 = off, the default) */
void nlo_set_pin_workers(int on);
/* OpStack.Apply: mode (0..6), optional weights (NULL = none), batching rule
 * numBatches=max(4*N*P/8MiB, 8*num_cpu), num_cpu worker threads. */
int  nlo_stack_apply(int mode, const float *const *lights, const float *weights,
                     int n_frames, int64_t npix, float ref_loc,
                     float sigma_low, float sigma_high, int num_cpu,
                     float *res, int64_t *clip_low, int64_t *clip_high, int *mode_used);
void nlo_stack_incremental(float *stack, const float *light, int64_t npix,
                           float weight, int first);                                      /* :924-937 */
void nlo_stack_incremental_finalize(float *stack, int64_t npix, float weight_sum);        /* :940-943 */

/* ---- goal-seek spec (dead code in the reference),
 *      internal/ops/stack/stackfindsigma.go:48-98 (bisection) ----
 * Runs repeated nlo_stack_apply; returns number of stack passes made. */
int  nlo_find_sigmas_bisect(int mode, const float *const *lights, const float *weights,
                            int n_frames, int64_t npix, float ref_loc,
                            float clip_perc_low, float clip_perc_high, int num_cpu,
                            float *res, int64_t *clip_low, int64_t *clip_high,
                            float *sigma_low, float *sigma_high);
/* stackfindsigma.go:101-170 (Newton's method, the linear-fit branch of :40-41), quirks kept */
int  nlo_find_sigmas_newton(int mode, const float *const *lights, const float *weights,
                            int n_frames, int64_t npix, float ref_loc,
                            float clip_perc_low, float clip_perc_high, int num_cpu,
                            float *res, int64_t *clip_low, int64_t *clip_high,
                            float *sigma_low, float *sigma_high);

/* ---- the formats and steps either side of the stack (F3, F4) ----
 * internal/fits/read.go:172-445   FITS payload -> fp32: big-endian BITPIX 8/16/32/64/-32/-64,
 *                                 v = float32(val)*BSCALE + BZERO; min, max, mean (fp64 sum, in order)
 * internal/fits/write.go:182-200  fp32 -> big-endian bytes, NaN -> 0
 * internal/fits/pixelops.go:601-605  MatchHistogram: x*multiplier + offset, in place
 * internal/star/coord.go:141-145, :159-199  Transform2D.Apply / Invert
 * internal/fits/project.go:26-76  bilinear resampling, out of bounds -> given value (NaN) */
int  nlo_fits_decode(const unsigned char *raw, int bitpix, int64_t n, float bscale, float bzero,
                     float *out, float *min, float *max, float *mean);
void nlo_fits_encode(const float *data, int64_t n, int replace_nans, unsigned char *raw);
void nlo_affine(float *data, int64_t n, float multiplier, float offset);
int  nlo_transform_invert(const float t[6], float inv[6]);
int  nlo_project_bilinear(const float *src, int32_t src_w, int32_t src_h,
                          float *dst, int32_t dst_w, int32_t dst_h,
                          const float trans[6], float out_of_bounds);

#ifdef __cplusplus
}
#endif
#endif
