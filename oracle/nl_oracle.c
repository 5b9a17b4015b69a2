/*
 * nl_oracle.c -- CPU oracle: plain-C restatement of Nightlight's per-pixel
 * stacking path.  TEST INFRASTRUCTURE ONLY (see nl_oracle.h).
 *
 * Arithmetic contract: IEEE fp32, round to nearest,
 * no FMA contraction (build with -ffp-contract=off, no -ffast-math), every
 * expression evaluated left to right as the Go source writes it, arrays
 * modified in place so later steps see the permuted order.
 * Go's float32(math.Sqrt(float64(x))) == (float)sqrt((double)x).
 */
#include "nl_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* internal/qsort/qsort.go                                             */
/* ------------------------------------------------------------------ */

/* qsort.go:38-56  Hoare partition around the middle element */
int nlo_qpartition_f32(float *a, int n)
{
    int left = 0, right = n - 1;
    int mid = (left + right) >> 1;
    float pivot = a[mid];
    int l = left - 1, r = right + 1;
    for (;;) {
        do { l++; } while (!(a[l] >= pivot));
        do { r--; } while (!(a[r] <= pivot));
        if (l >= r) return r;
        float t = a[l]; a[l] = a[r]; a[r] = t;
    }
}

/* qsort.go:26-32 */
void nlo_qsort_f32(float *a, int n)
{
    if (n > 1) {
        int index = nlo_qpartition_f32(a, n);
        nlo_qsort_f32(a, index + 1);
        nlo_qsort_f32(a + index + 1, n - (index + 1));
    }
}

/* qsort.go:94-126  k is 1-based */
float nlo_qselect_f32(float *a, int n, int k)
{
    int left = 0, right = n - 1;
    while (left < right) {
        int mid = (left + right) >> 1;
        float pivot = a[mid];
        int l = left - 1, r = right + 1;
        for (;;) {
            do { l++; } while (!(a[l] >= pivot));
            do { r--; } while (!(a[r] <= pivot));
            if (l >= r) break;
            float t = a[l]; a[l] = a[r]; a[r] = t;
        }
        int index = r;
        int offset = index - left + 1;
        if (k <= offset) {
            right = index;
        } else {
            left = index + 1;
            k = k - offset;
        }
    }
    return a[left];
}

/* qsort.go:61-63 */
float nlo_qselect_first_quartile_f32(float *a, int n)
{
    return nlo_qselect_f32(a, n, (n >> 2) + 1);
}

/* qsort.go:68-82 */
float nlo_qselect_median_f32(float *a, int n)
{
    int k = (n >> 1) + 1;
    float upper = nlo_qselect_f32(a, n, k);
    if ((n & 1) != 0) return upper;
    float lower = a[0];
    for (int i = 1; i < k - 1; i++) {
        if (a[i] > lower) lower = a[i];
    }
    return 0.5f * (lower + upper);
}

/* ------------------------------------------------------------------ */
/* internal/stats/stats.go                                             */
/* ------------------------------------------------------------------ */

/* stats.go:246-261 */
void nlo_mean_stddev(const float *xs, int n, float *mean, float *stddev)
{
    float xmean = 0.0f;
    for (int i = 0; i < n; i++) xmean += xs[i];
    xmean /= (float)n;
    float xvar = 0.0f;
    for (int i = 0; i < n; i++) {
        float diff = xs[i] - xmean;
        xvar += diff * diff;
    }
    xvar /= (float)n;
    *mean = xmean;
    *stddev = (float)sqrt((double)xvar);
}

/* stats.go:569-586 (note the n+1 divisor) */
void nlo_linear_regression(const float *xs, const float *ys, int n,
                           float *slope, float *intercept, float *xmean,
                           float *xstddev, float *ymean, float *ystddev)
{
    float xm, xs_, ym, ys_;
    nlo_mean_stddev(xs, n, &xm, &xs_);
    nlo_mean_stddev(ys, n, &ym, &ys_);
    float corr = 0.0f;
    for (int i = 0; i < n; i++) {
        float diff = (xs[i] - xm) * (ys[i] - ym);
        corr += diff;
    }
    float denom = xs_ * ys_;
    denom = denom * ((float)n + 1.0f);
    corr /= denom;
    float sl = corr * ys_;
    sl = sl / xs_;
    float prod = sl * xm;
    *slope = sl;
    *intercept = ym - prod;
    *xmean = xm; *xstddev = xs_; *ymean = ym; *ystddev = ys_;
}

/* stats.go:264-277 */
void nlo_min_mean_max(const float *data, int64_t n, float *mn, float *mean, float *mx)
{
    float mmin = data[0], mmax = data[0];
    double mmean = 0.0;
    for (int64_t i = 0; i < n; i++) {
        float mv = data[i];
        if (mv < mmin) mmin = mv;
        if (mv > mmax) mmax = mv;
        mmean += (double)mv;
    }
    *mn = mmin;
    *mean = (float)(mmean / (double)n);
    *mx = mmax;
}

/* stats.go:280-287 */
double nlo_variance(const float *data, int64_t n, float mean)
{
    double variance = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double diff = (double)(data[i] - mean);
        variance += diff * diff;
    }
    return variance / (double)n;
}

/* stats_amd64.s:28-92: four fp32 min/max lanes, four fp64 sum lanes over
 * elements i%4; horizontal (l0+l1)+(l2+l3); min/max via MINPS/MAXPS
 * (second operand returned when either is NaN -- inputs here are NaN-free) */
void nlo_min_mean_max_lanes4(const float *data, int64_t n, float *mn, float *mean, float *mx)
{
    float lmin[4], lmax[4];
    double lsum[4] = {0, 0, 0, 0};
    for (int j = 0; j < 4; j++) { lmin[j] = data[j]; lmax[j] = data[j]; }
    for (int64_t i = 0; i < n; i += 4) {
        for (int j = 0; j < 4; j++) {
            float v = data[i + j];
            if (v < lmin[j]) lmin[j] = v;
            if (v > lmax[j]) lmax[j] = v;
            lsum[j] += (double)v;
        }
    }
    float m = lmin[0], M = lmax[0];
    for (int j = 1; j < 4; j++) { if (lmin[j] < m) m = lmin[j]; if (lmax[j] > M) M = lmax[j]; }
    double s = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
    *mn = m; *mx = M;
    *mean = (float)(s / (double)n);
}

/* stats_amd64.s:102-143 */
double nlo_variance_lanes4(const float *data, int64_t n, float mean)
{
    double lsum[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; i += 4) {
        for (int j = 0; j < 4; j++) {
            double d = (double)(data[i + j] - mean);
            lsum[j] += d * d;
        }
    }
    double s = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
    return s / (double)n;
}

/* ------------------------------------------------------------------ */
/* internal/stats/noise.go:24-55  (pure-Go path)                       */
/* ------------------------------------------------------------------ */
float nlo_estimate_noise(const float *data, int64_t n, int32_t width)
{
    static const float w[9] = { 1, -2, 1, -2, 4, -2, 1, -2, 1 };
    const int32_t off[9] = { -width - 1, -width, -width + 1, -1, 0, 1,
                             width - 1, width, width + 1 };
    int32_t height = (int32_t)(n / width);
    float sum = 0.0f;
    for (int32_t y = 1; y < height - 1; y++) {
        float row_sum = 0.0f;
        for (int32_t x = 1; x < width - 1; x++) {
            int64_t i = (int64_t)y * width + x;
            float conv = 0.0f;
            for (int j = 0; j < 9; j++) {
                float p = data[i + off[j]] * w[j];
                conv += p;
            }
            row_sum += fabsf(conv);
        }
        sum += row_sum;
    }
    float c = (float)sqrt(0.5 * M_PI);
    float d = 6.0f * (float)(width - 2);
    d = d * (float)(height - 2);
    float factor = c / d;
    return sum * factor;
}

/* ------------------------------------------------------------------ */
/* internal/median/median3x3.go                                        */
/* ------------------------------------------------------------------ */
#define NLO_SWAP(i, j) do { if (a[i] > a[j]) { float t_ = a[i]; a[i] = a[j]; a[j] = t_; } } while (0)
#define NLO_MAXTO(i, j) do { if (a[i] > a[j]) { a[j] = a[i]; } } while (0) /* a[j]=max */
#define NLO_MINTO(i, j) do { if (a[i] > a[j]) { a[i] = a[j]; } } while (0) /* a[i]=min */

/* median3x3.go:85-110 : 19-step median-of-9 network */
float nlo_median9(float *a)
{
    NLO_SWAP(0, 1); NLO_SWAP(3, 4); NLO_SWAP(6, 7);
    NLO_SWAP(1, 2); NLO_SWAP(4, 5); NLO_SWAP(7, 8);
    NLO_SWAP(0, 1); NLO_SWAP(3, 4); NLO_SWAP(6, 7);
    NLO_MAXTO(0, 3);
    NLO_MAXTO(3, 6);
    NLO_SWAP(1, 4);
    NLO_MINTO(4, 7);
    NLO_MAXTO(1, 4);
    NLO_MINTO(5, 8);
    NLO_MINTO(2, 5);
    NLO_SWAP(2, 4);
    NLO_MINTO(4, 6);
    NLO_MAXTO(2, 4);
    return a[4];
}

/* median3x3.go:115-119 */
float nlo_median_f32(float *a, int n)
{
    if (n == 0) return NAN;
    if (n == 9) return nlo_median9(a);
    return nlo_qselect_median_f32(a, n);
}

/* gather.go:26-38.  NOTE the reference takes the median of the WHOLE buffer (len(mask) values),
 * not of the `num` gathered ones: where part of the neighbourhood falls outside the data the tail
 * of the buffer still holds what earlier calls left there (permuted by their quickselects). */
float nlo_gather_and_median(const float *data, int64_t n, int32_t index, const int32_t *mask, int mask_len,
                            float *buffer)
{
    int num = 0;
    for (int j = 0; j < mask_len; j++) {
        int32_t io = index + mask[j];
        if (io >= 0 && (int64_t)io < n) buffer[num++] = data[io];
    }
    return nlo_median_f32(buffer, mask_len);
}

/* ops/pre/badpixels.go:54-77 MedianFilter, as ONE goroutine walking 0..n-1 with one buffer
 * (the reference splits the range over NumCPU goroutines, each with a fresh zeroed buffer).
 * full[i] = 1 where the whole neighbourhood lies inside the data, i.e. where the result does
 * not depend on the buffer's history. */
void nlo_median_filter_mask(float *out, const float *data, int64_t n, const int32_t *mask, int mask_len,
                            unsigned char *full)
{
    float *buffer = (float *)calloc((size_t)(mask_len > 0 ? mask_len : 1), sizeof(float));
    for (int64_t i = 0; i < n; i++) {
        int ok = 1;
        for (int j = 0; j < mask_len; j++) {
            int64_t io = i + mask[j];
            if (io < 0 || io >= n) ok = 0;
        }
        if (full) full[i] = (unsigned char)ok;
        out[i] = nlo_gather_and_median(data, n, (int32_t)i, mask, mask_len, buffer);
    }
    free(buffer);
}

/* star/findstars.go:187-200 CreateMask: offsets of a disc of the given radius */
int nlo_create_mask(int32_t width, float radius, int32_t *mask, int cap)
{
    int cnt = 0;
    int32_t rad = (int32_t)radius;
    for (int32_t y = -rad; y <= rad; y++)
        for (int32_t x = -rad; x <= rad; x++) {
            float dist = (float)sqrt((double)(y * y + x * x));
            if (dist <= radius + 1e-8f) {
                if (cnt < cap) mask[cnt] = y * width + x;
                cnt++;
            }
        }
    return cnt;
}

/* median3x3.go:26-77 : interior = median of 3x3, border rows/cols copied */
void nlo_median_filter_3x3(float *out, const float *data, int64_t n, int32_t width)
{
    int64_t height = n / width;
    memcpy(out, data, (size_t)width * sizeof(float));
    for (int64_t line = 0; line < height - 2; line++) {
        const float *r0 = data + line * width;
        const float *r1 = r0 + width;
        const float *r2 = r1 + width;
        float *o = out + (line + 1) * width;
        o[0] = r1[0];
        for (int32_t x = 1; x < width - 1; x++) {
            float g[9] = { r0[x - 1], r0[x], r0[x + 1],
                           r1[x - 1], r1[x], r1[x + 1],
                           r2[x - 1], r2[x], r2[x + 1] };
            o[x] = nlo_median9(g);
        }
        o[width - 1] = r1[width - 1];
    }
    memcpy(out + (height - 1) * width, data + (height - 1) * width, (size_t)width * sizeof(float));
}

/* ------------------------------------------------------------------ */
/* internal/ops/stack/stack.go : per-pixel stackers                    */
/* ------------------------------------------------------------------ */

/* gather step shared by all modes (stack.go:280-287 and twins):
 * frame order, keep iff not NaN (math.IsNaN only; +-Inf are data) */
static inline int gather(const float *const *lights, int n_frames, int64_t i, float *dst)
{
    int n = 0;
    for (int li = 0; li < n_frames; li++) {
        float v = lights[li][i];
        if (!isnan(v)) dst[n++] = v;
    }
    return n;
}

static inline int gather_w(const float *const *lights, const float *weights, int n_frames,
                           int64_t i, float *dst, float *wdst)
{
    int n = 0;
    for (int li = 0; li < n_frames; li++) {
        float v = lights[li][i];
        if (!isnan(v)) { dst[n] = v; wdst[n] = weights[li]; n++; }
    }
    return n;
}

/* stack.go:274-303 */
void nlo_stack_median(const float *const *lights, int n_frames, int64_t npix, float ref_loc, float *res)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)n_frames);
    for (int64_t i = 0; i < npix; i++) {
        int n = gather(lights, n_frames, i, g);
        if (n == 0) { res[i] = ref_loc; continue; }
        res[i] = nlo_qselect_median_f32(g, n);
    }
    free(g);
}

/* stack.go:307-333 */
void nlo_stack_mean(const float *const *lights, int n_frames, int64_t npix, float ref_loc, float *res)
{
    for (int64_t i = 0; i < npix; i++) {
        int n = 0;
        float sum = 0.0f;
        for (int li = 0; li < n_frames; li++) {
            float v = lights[li][i];
            if (!isnan(v)) { sum += v; n++; }
        }
        if (n == 0) { res[i] = ref_loc; continue; }
        res[i] = sum / (float)n;
    }
}

/* stack.go:337-366 */
void nlo_stack_mean_weighted(const float *const *lights, const float *weights, int n_frames,
                             int64_t npix, float ref_loc, float *res)
{
    for (int64_t i = 0; i < npix; i++) {
        int n = 0;
        float sum = 0.0f, wsum = 0.0f;
        for (int li = 0; li < n_frames; li++) {
            float v = lights[li][i];
            if (!isnan(v)) {
                float w = weights[li];
                float p = v * w;
                sum += p;
                wsum += w;
                n++;
            }
        }
        if (n == 0) { res[i] = ref_loc; continue; }
        res[i] = sum / wsum;
    }
}

/* the clip pass of the sigma family (stack.go:411-424): swap-with-last,
 * re-test the same index; optional mirrored weight array */
static inline int clip_pass(float *a, float *w, int n, float lo, float hi,
                            int64_t *c_lo, int64_t *c_hi)
{
    int j = 0;
    while (j < n) {
        float g = a[j];
        if (g < lo) {
            a[j] = a[n - 1];
            if (w) w[j] = w[n - 1];
            n--;
            (*c_lo)++;
        } else if (g > hi) {
            a[j] = a[n - 1];
            if (w) w[j] = w[n - 1];
            n--;
            (*c_hi)++;
        } else {
            j++;
        }
    }
    return n;
}

/* stack.go:372-436 */
void nlo_stack_sigma(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                     float sigma_low, float sigma_high, float *res,
                     int64_t *clip_low, int64_t *clip_high)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)n_frames);
    int64_t c_lo = 0, c_hi = 0;
    for (int64_t i = 0; i < npix; i++) {
        int n = gather(lights, n_frames, i, g);
        if (n == 0) { res[i] = ref_loc; continue; }
        for (;;) {
            float median = nlo_qselect_median_f32(g, n);
            float mean, std;
            nlo_mean_stddev(g, n, &mean, &std);
            float t_lo = sigma_low * std, t_hi = sigma_high * std;
            float lo = median - t_lo, hi = median + t_hi;
            int64_t prev = c_lo + c_hi;
            n = clip_pass(g, NULL, n, lo, hi, &c_lo, &c_hi);
            if ((c_lo + c_hi) == prev || n <= 1) { res[i] = mean; break; }
        }
    }
    free(g);
    *clip_low = c_lo; *clip_high = c_hi;
}

/* weighted mean over the survivors (stack.go:514-522) */
static inline float weighted_mean(const float *a, const float *w, int n)
{
    float ws = 0.0f, s = 0.0f;
    for (int i = 0; i < n; i++) {
        float p = a[i] * w[i];
        s += p;
        ws += w[i];
    }
    return s / ws;
}

/* stack.go:442-531 */
void nlo_stack_sigma_weighted(const float *const *lights, const float *weights, int n_frames,
                              int64_t npix, float ref_loc, float sigma_low, float sigma_high,
                              float *res, int64_t *clip_low, int64_t *clip_high)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)n_frames);
    float *w = (float *)malloc(sizeof(float) * (size_t)n_frames);
    int64_t c_lo = 0, c_hi = 0;
    for (int64_t i = 0; i < npix; i++) {
        int n = gather_w(lights, weights, n_frames, i, g, w);
        if (n == 0) { res[i] = ref_loc; continue; }
        for (;;) {
            /* QSelect permutes g only: the weights are NOT permuted with it
             * (stack.go:487), they follow only the clip swaps */
            float median = nlo_qselect_median_f32(g, n);
            float mean, std;
            nlo_mean_stddev(g, n, &mean, &std);
            (void)mean;
            float t_lo = sigma_low * std, t_hi = sigma_high * std;
            float lo = median - t_lo, hi = median + t_hi;
            int64_t prev = c_lo + c_hi;
            n = clip_pass(g, w, n, lo, hi, &c_lo, &c_hi);
            if ((c_lo + c_hi) == prev || n <= 1) { res[i] = weighted_mean(g, w, n); break; }
        }
    }
    free(g); free(w);
    *clip_low = c_lo; *clip_high = c_hi;
}

/* stack.go:536-605 */
void nlo_stack_mad_sigma(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                         float sigma_low, float sigma_high, float *res,
                         int64_t *clip_low, int64_t *clip_high)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)n_frames);
    float *ad = (float *)malloc(sizeof(float) * (size_t)n_frames);
    int64_t c_lo = 0, c_hi = 0;
    for (int64_t i = 0; i < npix; i++) {
        int n = gather(lights, n_frames, i, g);
        if (n == 0) { res[i] = ref_loc; continue; }
        float median = nlo_qselect_median_f32(g, n);
        for (int j = 0; j < n; j++) {
            float d = g[j] - median;
            if (d < 0) d = -d;
            ad[j] = d;
        }
        float mad = nlo_qselect_median_f32(ad, n);
        float std = mad * 1.4826f;
        float t_lo = sigma_low * std, t_hi = sigma_high * std;
        float lo = median - t_lo, hi = median + t_hi;
        n = clip_pass(g, NULL, n, lo, hi, &c_lo, &c_hi);
        float mean = 0.0f;
        for (int j = 0; j < n; j++) mean += g[j];
        mean /= (float)n;
        res[i] = mean;
    }
    free(g); free(ad);
    *clip_low = c_lo; *clip_high = c_hi;
}

/* winsorized standard deviation (stack.go:646-672): returns the new stdDev */
static inline float winsorized_stddev(const float *a, float *wz, int n, float median, float std)
{
    memcpy(wz, a, sizeof(float) * (size_t)n);
    for (;;) {
        float t = 1.5f * std;
        float lo = median - t, hi = median + t;
        int changed = 0;
        for (int i = 0; i < n; i++) {
            float v = wz[i];
            if (v < lo) { wz[i] = lo; changed++; }
            else if (v > hi) { wz[i] = hi; changed++; }
        }
        float old = std, m_;
        nlo_mean_stddev(wz, n, &m_, &std);
        std = 1.134f * std;
        float diff = std - old;
        float factor = (float)fabs((double)diff) / old;
        if (changed == 0 || factor <= 0.0005f) break;
    }
    return std;
}

/* stack.go:611-705 */
void nlo_stack_winsor_sigma(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                            float sigma_low, float sigma_high, float *res,
                            int64_t *clip_low, int64_t *clip_high)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)n_frames);
    float *wz = (float *)malloc(sizeof(float) * (size_t)n_frames);
    int64_t c_lo = 0, c_hi = 0;
    for (int64_t i = 0; i < npix; i++) {
        int n = gather(lights, n_frames, i, g);
        if (n == 0) { res[i] = ref_loc; continue; }
        for (;;) {
            float median = nlo_qselect_median_f32(g, n);
            float mean, std;
            nlo_mean_stddev(g, n, &mean, &std);
            std = winsorized_stddev(g, wz, n, median, std);
            float t_lo = sigma_low * std, t_hi = sigma_high * std;
            float lo = median - t_lo, hi = median + t_hi;
            int64_t prev = c_lo + c_hi;
            n = clip_pass(g, NULL, n, lo, hi, &c_lo, &c_hi);
            if ((c_lo + c_hi) == prev || n <= 1) { res[i] = mean; break; }
        }
    }
    free(g); free(wz);
    *clip_low = c_lo; *clip_high = c_hi;
}

/* stack.go:710-829 */
void nlo_stack_winsor_sigma_weighted(const float *const *lights, const float *weights,
                                     int n_frames, int64_t npix, float ref_loc,
                                     float sigma_low, float sigma_high, float *res,
                                     int64_t *clip_low, int64_t *clip_high)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)n_frames);
    float *w = (float *)malloc(sizeof(float) * (size_t)n_frames);
    float *wz = (float *)malloc(sizeof(float) * (size_t)n_frames);
    int64_t c_lo = 0, c_hi = 0;
    for (int64_t i = 0; i < npix; i++) {
        int n = gather_w(lights, weights, n_frames, i, g, w);
        if (n == 0) { res[i] = ref_loc; continue; }
        for (;;) {
            float median = nlo_qselect_median_f32(g, n);
            float mean, std;
            nlo_mean_stddev(g, n, &mean, &std);
            (void)mean;
            std = winsorized_stddev(g, wz, n, median, std);
            float t_lo = sigma_low * std, t_hi = sigma_high * std;
            float lo = median - t_lo, hi = median + t_hi;
            int64_t prev = c_lo + c_hi;
            n = clip_pass(g, w, n, lo, hi, &c_lo, &c_hi);
            if ((c_lo + c_hi) == prev || n <= 1) { res[i] = weighted_mean(g, w, n); break; }
        }
    }
    free(g); free(w); free(wz);
    *clip_low = c_lo; *clip_high = c_hi;
}

/* stack.go:834-918 */
void nlo_stack_linear_fit(const float *const *lights, int n_frames, int64_t npix, float ref_loc,
                          float sigma_low, float sigma_high, float *res,
                          int64_t *clip_low, int64_t *clip_high)
{
    float *gfull = (float *)malloc(sizeof(float) * (size_t)n_frames);
    float *xs = (float *)malloc(sizeof(float) * (size_t)n_frames);
    for (int i = 0; i < n_frames; i++) xs[i] = (float)i;
    int64_t c_lo = 0, c_hi = 0;
    for (int64_t i = 0; i < npix; i++) {
        int n = gather(lights, n_frames, i, gfull);
        if (n == 0) { res[i] = ref_loc; continue; }
        float *g = gfull;
        float mean = 0.0f;
        for (;;) {
            nlo_qsort_f32(g, n);
            float slope, intercept, xm, xsd, ysd;
            nlo_linear_regression(xs, g, n, &slope, &intercept, &xm, &xsd, &mean, &ysd);
            float sigma = 0.0f;
            for (int j = 0; j < n; j++) {
                float lin = (float)j * slope;
                lin = lin + intercept;
                float diff = g[j] - lin;
                sigma += (float)fabs((double)diff);
            }
            sigma /= (float)n;
            int left = 0;
            float lb = sigma_low * sigma, hb = sigma_high * sigma;
            for (int j = 0; j < n; j++) {
                float v = g[j];
                float lin = (float)j * slope;
                lin = lin + intercept;
                if (lin - v > lb) { g[j] = g[left]; left++; c_lo++; }
                else if (v - lin > hb) { g[j] = g[left]; left++; c_hi++; }
            }
            if (left == 0 || n < 3) break;
            g += left; n -= left;
        }
        res[i] = mean;
    }
    free(gfull); free(xs);
    *clip_low = c_lo; *clip_high = c_hi;
}

/* ------------------------------------------------------------------ */
/* stack.go:45-55, 231-270, 115-227, 924-944                           */
/* ------------------------------------------------------------------ */

int nlo_auto_select_mode(int l)
{
    if (l >= 25) return NLO_ST_LINEAR_FIT;
    if (l >= 15) return NLO_ST_WINSOR_SIGMA;
    if (l >= 6) return NLO_ST_SIGMA;
    return NLO_ST_MEAN;
}

/* stack.go:231-270 */
int nlo_get_weights(int weighting, const float *per_frame, int n_frames,
                    float *weights_out, int *has_weights, int *bad_index)
{
    *has_weights = 0;
    if (bad_index) *bad_index = -1;
    if (weighting == NLO_WEIGHT_NONE) return NLO_OK;
    if (weighting == NLO_WEIGHT_EXPOSURE) {
        for (int i = 0; i < n_frames; i++) {
            if (per_frame[i] == 0) { if (bad_index) *bad_index = i; return NLO_ERR_MISSING_EXPOSURE; }
            weights_out[i] = per_frame[i];
        }
        *has_weights = 1;
        return NLO_OK;
    }
    if (weighting == NLO_WEIGHT_INVERSE_NOISE || weighting == NLO_WEIGHT_INVERSE_HFR) {
        float mn = 3.40282346638528859811704183484516925440e+38f, mx = -mn;
        for (int i = 0; i < n_frames; i++) {
            float v = per_frame[i];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        for (int i = 0; i < n_frames; i++) {
            float num = per_frame[i] - mn;
            num = 4.0f * num;
            float q = num / (mx - mn);
            weights_out[i] = 1.0f / (1.0f + q);
        }
        *has_weights = 1;
        return NLO_OK;
    }
    return NLO_ERR_INVALID_WEIGHTING;
}

typedef struct {
    int mode;
    const float *const *lights;
    const float *weights;
    int n_frames;
    int64_t npix;
    float ref_loc, sigma_low, sigma_high;
    float *res;
    int64_t batch_size;
    int64_t next_lower;      /* guarded by lock */
    int64_t clip_low, clip_high;
    pthread_mutex_t lock;
} apply_job;

static void run_batch(apply_job *job, int64_t lower, int64_t upper,
                      const float **ld, int64_t *c_lo, int64_t *c_hi)
{
    int64_t np = upper - lower;
    for (int i = 0; i < job->n_frames; i++) ld[i] = job->lights[i] + lower;
    float *out = job->res + lower;
    *c_lo = 0; *c_hi = 0;
    switch (job->mode) {
    case NLO_ST_MEDIAN:
        nlo_stack_median(ld, job->n_frames, np, job->ref_loc, out);
        break;
    case NLO_ST_MEAN:
        if (!job->weights) nlo_stack_mean(ld, job->n_frames, np, job->ref_loc, out);
        else nlo_stack_mean_weighted(ld, job->weights, job->n_frames, np, job->ref_loc, out);
        break;
    case NLO_ST_SIGMA:
        if (!job->weights)
            nlo_stack_sigma(ld, job->n_frames, np, job->ref_loc, job->sigma_low, job->sigma_high, out, c_lo, c_hi);
        else
            nlo_stack_sigma_weighted(ld, job->weights, job->n_frames, np, job->ref_loc, job->sigma_low, job->sigma_high, out, c_lo, c_hi);
        break;
    case NLO_ST_WINSOR_SIGMA:
        if (!job->weights)
            nlo_stack_winsor_sigma(ld, job->n_frames, np, job->ref_loc, job->sigma_low, job->sigma_high, out, c_lo, c_hi);
        else
            nlo_stack_winsor_sigma_weighted(ld, job->weights, job->n_frames, np, job->ref_loc, job->sigma_low, job->sigma_high, out, c_lo, c_hi);
        break;
    case NLO_ST_MAD_SIGMA:
        nlo_stack_mad_sigma(ld, job->n_frames, np, job->ref_loc, job->sigma_low, job->sigma_high, out, c_lo, c_hi);
        break;
    case NLO_ST_LINEAR_FIT: /* weights ignored: stack.go:188-189 (quirk Q1) */
        nlo_stack_linear_fit(ld, job->n_frames, np, job->ref_loc, job->sigma_low, job->sigma_high, out, c_lo, c_hi);
        break;
    }
}

/* Timing aid (bench.py cpu_baseline, tools/cpu_probe.py): worker t of the pool is pinned to the t-th CPU of the
 * process's affinity mask, so that a measurement does not depend on how fast the scheduler spreads freshly created
 * threads (Go's runtime keeps GOMAXPROCS long-lived OS threads that are spread already).  Results never depend on it. */
static int g_pin_workers = 0;
void nlo_set_pin_workers(int on) { g_pin_workers = on; }

static void pin_worker(int t)
{
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    int n = CPU_COUNT(&allowed);
    if (n <= 0) return;
    int want = t % n, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &allowed)) continue;
        if (seen++ == want) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(c, &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
            return;
        }
    }
}

typedef struct { void *job; int index; } worker_arg;

static void *apply_worker(void *arg);
static void *apply_worker_pinned(void *arg)
{
    worker_arg *wa = (worker_arg *)arg;
    pin_worker(wa->index);
    return apply_worker(wa->job);
}

static void *apply_worker(void *arg)
{
    apply_job *job = (apply_job *)arg;
    const float **ld = (const float **)malloc(sizeof(float *) * (size_t)job->n_frames);
    for (;;) {
        pthread_mutex_lock(&job->lock);
        int64_t lower = job->next_lower;
        job->next_lower += job->batch_size;
        pthread_mutex_unlock(&job->lock);
        if (lower >= job->npix) break;
        int64_t upper = lower + job->batch_size;
        if (upper > job->npix) upper = job->npix;
        int64_t c_lo, c_hi;
        run_batch(job, lower, upper, ld, &c_lo, &c_hi);
        if (c_lo > 0 || c_hi > 0) {
            pthread_mutex_lock(&job->lock);
            job->clip_low += c_lo;
            job->clip_high += c_hi;
            pthread_mutex_unlock(&job->lock);
        }
    }
    free(ld);
    return NULL;
}

/* stack.go:115-227 (the numeric part: mode resolution, batching rule,
 * worker pool, counter totals).  Results do not depend on the split. */
int nlo_stack_apply(int mode, const float *const *lights, const float *weights,
                    int n_frames, int64_t npix, float ref_loc,
                    float sigma_low, float sigma_high, int num_cpu,
                    float *res, int64_t *clip_low, int64_t *clip_high, int *mode_used)
{
    if (n_frames <= 0) return NLO_ERR_NO_INPUTS;
    if (mode < NLO_ST_MEDIAN || mode > NLO_ST_AUTO) return NLO_ERR_INVALID_MODE;
    if (mode == NLO_ST_AUTO) mode = nlo_auto_select_mode(n_frames);
    if (mode_used) *mode_used = mode;
    if (mode == NLO_ST_MAD_SIGMA && weights) return NLO_ERR_WEIGHTED_MAD;
    if (num_cpu < 1) num_cpu = 1;

    int64_t num_batches = 4 * (int64_t)n_frames * npix / (8192 * 1024);
    if (num_batches < 8 * (int64_t)num_cpu) num_batches = 8 * (int64_t)num_cpu;
    int64_t batch_size = (npix + num_batches - 1) / num_batches;
    if (batch_size < 1) batch_size = 1;

    apply_job job;
    job.mode = mode; job.lights = lights; job.weights = weights;
    job.n_frames = n_frames; job.npix = npix;
    job.ref_loc = ref_loc; job.sigma_low = sigma_low; job.sigma_high = sigma_high;
    job.res = res; job.batch_size = batch_size; job.next_lower = 0;
    job.clip_low = 0; job.clip_high = 0;
    pthread_mutex_init(&job.lock, NULL);

    if (num_cpu == 1) {
        apply_worker(&job);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)num_cpu);
        worker_arg *wa = (worker_arg *)malloc(sizeof(worker_arg) * (size_t)num_cpu);
        for (int t = 0; t < num_cpu; t++) {
            wa[t].job = &job; wa[t].index = t;
            if (g_pin_workers) pthread_create(&th[t], NULL, apply_worker_pinned, &wa[t]);
            else pthread_create(&th[t], NULL, apply_worker, &job);
        }
        for (int t = 0; t < num_cpu; t++) pthread_join(th[t], NULL);
        free(wa);
        free(th);
    }
    pthread_mutex_destroy(&job.lock);
    if (clip_low) *clip_low = job.clip_low;
    if (clip_high) *clip_high = job.clip_high;
    return NLO_OK;
}

/* stack.go:924-937 */
void nlo_stack_incremental(float *stack, const float *light, int64_t npix, float weight, int first)
{
    if (first) {
        for (int64_t i = 0; i < npix; i++) stack[i] = light[i] * weight;
    } else {
        for (int64_t i = 0; i < npix; i++) {
            float p = light[i] * weight;
            stack[i] += p;
        }
    }
}

/* stack.go:940-943 (factor is a float32: 1.0/weightSum) */
void nlo_stack_incremental_finalize(float *stack, int64_t npix, float weight_sum)
{
    float factor = 1.0f / weight_sum;
    for (int64_t i = 0; i < npix; i++) stack[i] = stack[i] * factor;
}

/* ------------------------------------------------------------------ */
/* goal-seek, stackfindsigma.go:48-98 (commented-out spec)              */
/* ------------------------------------------------------------------ */
int nlo_find_sigmas_bisect(int mode, const float *const *lights, const float *weights,
                           int n_frames, int64_t npix, float ref_loc,
                           float clip_perc_low, float clip_perc_high, int num_cpu,
                           float *res, int64_t *clip_low, int64_t *clip_high,
                           float *sigma_low, float *sigma_high)
{
    float low_left = 1.0f, low_right = 11.0f;
    float low_mid = 0.5f * (low_left + low_right);
    float high_left = 1.0f, high_right = 11.0f;
    float high_mid = 0.5f * (high_left + high_right);
    int passes = 0;
    for (int i = 0;; i++) {
        int64_t c_lo = 0, c_hi = 0;
        nlo_stack_apply(mode, lights, weights, n_frames, npix, ref_loc, low_mid, high_mid,
                        num_cpu, res, &c_lo, &c_hi, NULL);
        passes++;
        float total = (float)(npix * (int64_t)n_frames);
        float perc_l = (float)c_lo * 100.0f / total;
        float perc_h = (float)c_hi * 100.0f / total;
        int delta_l = (int)(100 * perc_l + 0.5f) - (int)(100 * clip_perc_low);
        int delta_h = (int)(100 * perc_h + 0.5f) - (int)(100 * clip_perc_high);
        if ((delta_l == 0 && delta_h == 0) || i >= 20) {
            *clip_low = c_lo; *clip_high = c_hi;
            *sigma_low = low_mid; *sigma_high = high_mid;
            return passes;
        }
        if (delta_l > 0) { low_left = low_mid; low_mid = 0.5f * (low_left + low_right); }
        else if (delta_l < 0) { low_right = low_mid; low_mid = 0.5f * (low_left + low_right); }
        if (delta_h > 0) { high_left = high_mid; high_mid = 0.5f * (high_left + high_right); }
        else if (delta_h < 0) { high_right = high_mid; high_mid = 0.5f * (high_left + high_right); }
    }
}

/* stackfindsigma.go:101-170 (commented-out spec): Newton's method on both sigmas for the
 * linear fit.  Restated WITH the reference's quirks: deltaH and deltaH3 subtract the LOW
 * target (:114, :155), and the loop counter advances by three per iteration (the body's two
 * i++ plus the for statement's), so the "i>=20" test fires at the 8th base evaluation.
 * Returns the number of stack passes made; res / counters are those of the last BASE pass. */
int nlo_find_sigmas_newton(int mode, const float *const *lights, const float *weights,
                           int n_frames, int64_t npix, float ref_loc,
                           float clip_perc_low, float clip_perc_high, int num_cpu,
                           float *res, int64_t *clip_low, int64_t *clip_high,
                           float *sigma_low, float *sigma_high)
{
    (void)clip_perc_high;                                /* never read by the reference either */
    float sig_low = 6.0f, sig_high = 6.0f;
    const float epsilon = 0.005f;
    const float total = (float)(npix * (int64_t)n_frames);
    float *scratch = (float *)malloc(sizeof(float) * (size_t)npix);
    int passes = 0;
    for (int i = 0;; i++) {
        int64_t c_lo = 0, c_hi = 0;
        nlo_stack_apply(mode, lights, weights, n_frames, npix, ref_loc, sig_low, sig_high, num_cpu, res, &c_lo, &c_hi, NULL);
        passes++;
        float perc_l = (float)c_lo * 100.0f / total;
        float perc_h = (float)c_hi * 100.0f / total;
        float delta_l = perc_l - clip_perc_low;
        float delta_h = perc_h - clip_perc_low;          /* sic, :114 */
        int delta_li = (int)(100 * delta_l + 0.5f);
        int delta_hi = (int)(100 * delta_h + 0.5f);
        *clip_low = c_lo; *clip_high = c_hi;
        *sigma_low = sig_low; *sigma_high = sig_high;
        if ((delta_li == 0 && delta_hi == 0) || i >= 20) break;

        i++;
        int64_t c_lo2 = 0, c_hi2 = 0;
        nlo_stack_apply(mode, lights, weights, n_frames, npix, ref_loc, sig_low + epsilon, sig_high, num_cpu, scratch, &c_lo2, &c_hi2, NULL);
        passes++;
        float perc_l2 = (float)c_lo2 * 100.0f / total;
        float delta_l2 = perc_l2 - clip_perc_low;
        float delta_l_diff = (delta_l2 - delta_l) / epsilon;
        if (delta_l_diff == 0) break;
        float new_low = sig_low - delta_l / delta_l_diff;
        if (new_low < 0.1f) new_low = 0.1f;
        if (new_low > 20) new_low = 20;

        i++;
        int64_t c_lo3 = 0, c_hi3 = 0;
        nlo_stack_apply(mode, lights, weights, n_frames, npix, ref_loc, sig_low, sig_high + epsilon, num_cpu, scratch, &c_lo3, &c_hi3, NULL);
        passes++;
        float perc_h3 = (float)c_hi3 * 100.0f / total;
        float delta_h3 = perc_h3 - clip_perc_low;        /* sic, :155 */
        float delta_h_diff = (delta_h3 - delta_h) / epsilon;
        if (delta_h_diff == 0) break;
        float new_high = sig_high - delta_h / delta_h_diff;
        if (new_high < 0.1f) new_high = 0.1f;
        if (new_high > 20) new_high = 20;
        sig_low = new_low; sig_high = new_high;
    }
    free(scratch);
    return passes;
}


/* ======================================================================== *
 *  Formats and steps either side of the stack
 * ======================================================================== */

/* internal/fits/read.go:172-445 (readUint8Data ... readFloat64Data): every variant is
 * "assemble the big-endian value, v = float32(val)*Bscale + Bzero, track min / max,
 * sum in float64 in file order"; mean = float32(sum / n) (:210, :255, ...). */
int nlo_fits_decode(const unsigned char *raw, int bitpix, int64_t n, float bscale, float bzero,
                    float *out, float *min, float *max, float *mean)
{
    float mn = FLT_MAX, mx = -FLT_MAX;
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        float val;
        switch (bitpix) {
        case 8:
            val = (float)raw[i];                                                   /* :192-193 */
            break;
        case 16: {
            const unsigned char *b = raw + 2 * i;
            int16_t x = (int16_t)(uint16_t)(((uint16_t)b[0] << 8) | (uint16_t)b[1]); /* :234 */
            val = (float)x;
            break;
        }
        case 32: {
            const unsigned char *b = raw + 4 * i;
            uint32_t u = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
            val = (float)(int32_t)u;
            break;
        }
        case 64: {
            const unsigned char *b = raw + 8 * i;
            uint64_t u = 0;
            for (int j = 0; j < 8; j++) u = (u << 8) | (uint64_t)b[j];            /* :325-326 */
            val = (float)(int64_t)u;
            break;
        }
        case -32: {
            const unsigned char *b = raw + 4 * i;
            uint32_t u = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
            memcpy(&val, &u, 4);                                                   /* :372-373 */
            break;
        }
        case -64: {
            const unsigned char *b = raw + 8 * i;
            uint64_t u = 0;
            double d;
            for (int j = 0; j < 8; j++) u = (u << 8) | (uint64_t)b[j];
            memcpy(&d, &u, 8);
            val = (float)d;                                                        /* :421-423 */
            break;
        }
        default:
            return -1;                                                             /* :169 */
        }
        float t = val * bscale;
        float v = t + bzero;
        if (v < mn) mn = v;
        if (v > mx) mx = v;
        sum += (double)v;
        out[i] = v;
    }
    if (min) *min = mn;
    if (max) *max = mx;
    if (mean) *mean = (float)(sum / (double)n);
    return 0;
}

/* internal/fits/write.go:182-200 */
void nlo_fits_encode(const float *data, int64_t n, int replace_nans, unsigned char *raw)
{
    for (int64_t i = 0; i < n; i++) {
        float d = data[i];
        if (replace_nans && d != d) d = 0.0f;
        uint32_t u;
        memcpy(&u, &d, 4);
        raw[4 * i + 0] = (unsigned char)(u >> 24);
        raw[4 * i + 1] = (unsigned char)(u >> 16);
        raw[4 * i + 2] = (unsigned char)(u >> 8);
        raw[4 * i + 3] = (unsigned char)u;
    }
}

/* internal/fits/pixelops.go:601-605 */
void nlo_affine(float *data, int64_t n, float multiplier, float offset)
{
    for (int64_t i = 0; i < n; i++) {
        float t = data[i] * multiplier;
        data[i] = t + offset;
    }
}

/* internal/star/coord.go:159-199 */
int nlo_transform_invert(const float t[6], float inv[6])
{
    const float A = t[0], B = t[1], C = t[2], D = t[3], E = t[4], F = t[5];
    float bd = B * D, ae = A * E;
    float eps = bd - ae;
    if (eps < 1e-8f && -eps < 1e-8f) return -1;
    float den1 = bd - ae;                 /* b*d - a*e */
    float den2 = ae - bd;                 /* a*e - b*d */
    float ce = C * E, bf = B * F, cd = C * D, af = A * F;
    inv[0] = -E / den1;
    inv[1] = B / den1;
    inv[2] = (ce - bf) / den1;
    inv[3] = -D / den2;
    inv[4] = A / den2;
    inv[5] = (cd - af) / den2;
    return 0;
}

/* internal/fits/project.go:26-76 with Transform2D.Apply (coord.go:141-145).
 * int32(math.Floor(x)) of a value outside the int32 range (or NaN) is 0x80000000 on
 * amd64, i.e. negative, i.e. out of bounds: the range test below says the same. */
int nlo_project_bilinear(const float *src, int32_t src_w, int32_t src_h,
                         float *dst, int32_t dst_w, int32_t dst_h,
                         const float trans[6], float out_of_bounds)
{
    float inv[6];
    if (nlo_transform_invert(trans, inv) != 0) return -1;
    for (int32_t row = 0; row < dst_h; row++) {
        for (int32_t col = 0; col < dst_w; col++) {
            float px = (float)col, py = (float)row;
            float ax = inv[0] * px, bx = inv[1] * py;
            float sx = ax + bx;
            float X = sx + inv[2];
            float ay = inv[3] * px, by = inv[4] * py;
            float sy = ay + by;
            float Y = sy + inv[5];
            double fx = floor((double)X), fy = floor((double)Y);
            int ok = fx >= 0.0 && fy >= 0.0 && fx < 2147483647.0 && fy < 2147483647.0;
            int32_t xl = 0, yl = 0;
            if (ok) {
                xl = (int32_t)fx;
                yl = (int32_t)fy;
                if (xl + 1 >= src_w || yl + 1 >= src_h) ok = 0;
            }
            if (!ok) {
                dst[(int64_t)col + (int64_t)row * dst_w] = out_of_bounds;
                continue;
            }
            float xr = X - (float)xl, yr = Y - (float)yl;
            int64_t xlyl = (int64_t)xl + (int64_t)yl * src_w;
            float omx = 1.0f - xr, omy = 1.0f - yr;
            float p0 = src[xlyl] * omx, p1 = src[xlyl + 1] * xr;
            float vyl = p0 + p1;
            float p2 = src[xlyl + src_w] * omx, p3 = src[xlyl + src_w + 1] * xr;
            float vyh = p2 + p3;
            float q0 = vyl * omy, q1 = vyh * yr;
            dst[(int64_t)col + (int64_t)row * dst_w] = q0 + q1;
        }
    }
    return 0;
}
