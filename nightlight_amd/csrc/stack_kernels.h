// stack_kernels.h -- shared declarations of libnlstack's HIP kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/nlstack.h"

// Result stores of the streaming kernels are NONTEMPORAL (round 6): the one store per pixel a stack pass makes is
// what the slow allocations of section 11.9 are slow beside -- reads + plain stores 6.12 against 6.77 TB/s, reads + nontemporal
// stores 6.52 against 6.93 (tools/ubench/alloc_probe_modes.hip): +2.4 % on every frame buffer, +6.5 % on the unlucky ones.
// NL_PLAIN_STORES (build switch) restores plain stores for A/B libraries.
#if defined(__HIPCC__)
#ifdef NL_PLAIN_STORES
#define NL_STORE_RESULT(ptr, val) (*(ptr) = (val))
#else
#define NL_STORE_RESULT(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif
#endif

namespace nl {

// 160 KiB LDS per CU (MI355X); a single workgroup may use all of it.
constexpr size_t kLdsBudgetBytes = 160 * 1024;
// clip counters: sharded accumulators [kClipSlots][2], zeroed before every pass
constexpr int kClipSlots = 256;

struct StackArgs {
    const float *frames;          // planar [n_frames][stride] fp32
    int64_t stride;               // floats between consecutive frames (>= tile pixels: the owned buffer is padded, nlstack_api.hip padded_frame_stride)
    int64_t npix;                 // pixels in the tile
    int64_t tiles;                // work items (wave tiles) in the launch
    int n_frames;
    int n_pad;                    // next power of two >= n_frames (sorting modes)
    const float *weights;         // device, n_frames floats, or nullptr
    const float *xstat;           // device, [n_frames+1][2]: MeanStdDev of 0..n-1 (linear fit)
    float sig_lo, sig_hi, ref_loc;
    float *out;                   // [npix]
    unsigned long long *partial;  // [kClipSlots][2] sharded clip counters (atomic adds)
    const unsigned *list;         // optional: pixel indices to process instead of 0..npix-1
    const unsigned *list_count;   // device-side length of `list`
    unsigned list_capacity;
    // wave-per-pixel replay only: a cell (zero before the pass) shared by the two replays of a pass.  The
    // first workgroup of either to look stores 1 + the list's length at that moment; part 0 replays the
    // items before that snapshot (what the dominant kernel handed over), part 1 the items from it on
    // (what the generic pass added) -- no host-enqueued snapshot copy between the kernels
    unsigned *list_snap;
    int list_part;
    // Fused pass protocol of the sigma / winsorized fast path (nlstack_api.hip): no memset before and no
    // reduction kernel after a pass.  The dominant kernel's first workgroup zeroes `final` and the scratch set
    // of the NEXT pass (`zero_next`, kScratchWords words: the sets alternate); the dominant kernel adds its
    // clip counts to this pass's sharded `partial` slots; the first workgroup of the generic pass sums them
    // into `final`; every kernel after the dominant one (generic pass, replays) adds to `final` directly.
    // The replays add to one address, which is only cheap while their lists are short (thousands of workgroups
    // adding to one word take longer than a reduction kernel): nlstack_api.hip runs a pass fused only if the last
    // finished pass reported a short exact list.  All nullptr: the plain protocol (memset, sharded slots,
    // reduce_counters_kernel).
    unsigned long long *final;    // [2] clip totals of the pass
    unsigned long long *zero_next;
    // Weighted stacks, decision pass + permutation replay (stack_fast_decide.hip): thresholds that reproduce the
    // reference's clip decisions, [round][pixel] (round stride = npix), and the number of rounds per pixel (0 = not
    // decided: the replay computes its own bounds, as without these).  nullptr: off.
    float2 *bounds;
    unsigned char *nrounds;
};

// words (64 bit) of one per-pass scratch set: clip accumulators + {exact-list length, generic-list length,
// list snapshot, spare} (32 bit each)
constexpr int kScratchWords = 2 * kClipSlots + 2;

// fallback list written by the fast kernels, consumed by the exact kernel
struct FastArgs {
    unsigned *fb_list;            // [fb_capacity] pixels for the exact kernel
    unsigned *fb_count;           // device counter, zeroed before every pass
    unsigned fb_capacity;
    unsigned *fb_snap;            // optional, see StackArgs::list_snap: a generic pass stores 1 + *fb_count here before it appends
    unsigned *gen_list;           // [gen_capacity] pixels a zonal wave hands to the generic pass
    unsigned *gen_count;          // device counter, zeroed before every pass
    unsigned gen_capacity;
    unsigned gen_hint;            // 1 + the generic list's length in the last finished pass (0 = unknown): sizes the generic grid
    const unsigned *in_list;      // generic pass: list to process (nullptr = the whole tile)
    const unsigned *in_count;
    unsigned in_capacity;
    // Winsorization cascade of the one-lane winsorized kernels (stack_fast_sigma_impl.hpp).  Host-side plan, filled in by
    // nlstack_api.hip: two continuation lists (pixel, clip counts so far) with their per-workgroup lengths and the wave
    // budgets of the first two stages (winsorization rounds; the third stage runs to the end).  cas_list[0] == nullptr:
    // no cascade.  NO global atomics: a workgroup of a stage compacts its unfinished pixels into a region of its own
    // (through an LDS counter) and stores the region's length; a workgroup of the next stage takes cas_group consecutive
    // regions.  (One device counter for the 262 144 waves of a 4096^2 tile costs 3 ms -- 11 ns per atomic, all in one L2
    // channel -- whether it is one word or 256 neighbouring ones: measured, the dominant kernel got slower with a budget.)
    // Stage k (0 = the dominant kernel) appends to list k % 2 and, from k = 1 on, reads list (k - 1) % 2; it may run
    // cas_pass[k] clipping passes per wave with at most cas_cap[k] winsorization rounds each (0 = no limit: the last stage).
    unsigned *cas_list[2];
    unsigned *cas_state[2];
    unsigned *cas_count[2];       // lengths of the regions of the two lists (one region per workgroup of the stage that filled it)
    int cas_stages;               // stages in all, 2 ... kCascadeStages
    int cas_pass[6], cas_cap[6];
    int cas_group[6];             // [k]: regions of its input list per workgroup of stage k (k >= 1), at most 16
    // ... and what one launch of the cascade works with (set by the launcher from the plan above)
    unsigned *cont_list;          // where this stage appends its unfinished pixels: region blockIdx.x, cont_region entries long
    unsigned *cont_state;
    unsigned *cont_count;         // [workgroups of this launch]
    unsigned cont_region;
    const unsigned *in_state;     // CONT kernels: the states that belong to in_list; in_count = the regions' lengths
    unsigned in_region, in_regions, in_group;      // entries per input region, number of regions, regions per workgroup
    int pass_budget, round_cap;   // clipping passes a wave may run in this stage / winsorization rounds per pass (0 = no limit)
    int record_only = 0;          // LDS-column kernels as the DECISION pass of a weighted stack (129 ... 512 frames): no outputs, no
                                  // lists, no counters -- only StackArgs::bounds / nrounds (0 rounds for a pixel they would hand over)
    // The split pass of the LDS-column kernel whose columns are selected (497 ... 512 frames, plain sigma; stack_fast_mlz_impl.hpp):
    // the sorting kernel leaves mlz_split_rows() rows x 64 pixels per workgroup here (blocks of 64 pixels, contiguous; cols_stride =
    // pixels the buffer holds, a multiple of 64) and a second kernel, one wave per block, runs the clipping rounds over them.
    // nullptr: one kernel does both.
    float *cols = nullptr;
    long long cols_stride = 0;
    // The same class as PERSISTENT workgroups (three per CU, each looping over blocks of 64 pixels; the rounds of a block run in one
    // of its waves while the others are sorting the next block -- stack_fast_mlz_impl.hpp, PHASE 3).  0: one workgroup per block.
    int cert_first = 0, cert_every = 1;      // invariant-interval certificate of the winsorization loops (stack_fast_sigma_impl.hpp): first trial after this many rounds of a loop (0: off), then every so many
    int gen_round_cap = 0;        // generic pass of the one-lane winsorized kernels: winsorization rounds per clipping pass before a pixel
                                  // is handed to the exact replay instead (0: 100, the limit of every kernel)
    int persistent = 0;
    unsigned *ticket = nullptr;   // persistent workgroups: the next block to hand out (zero at the start of the pass)
};

// sets what nl_last_error() returns on this thread (nlstack_api.hip)
void set_last_error(const char *msg);

// Bisection of the goal-seek (spec: internal/ops/stack/stackfindsigma.go:48-98): sigma_low and
// sigma_high in [1,11], percentages in the reference's fp32 arithmetic, 21 passes at most.
// Host arithmetic shared by nl_stack_find_sigmas and nl_group_find_sigmas.
struct SigmaBisection {
    float low_left = 1.0f, low_right = 11.0f, low_mid;
    float high_left = 1.0f, high_right = 11.0f, high_mid;
    float perc_low, perc_high, total;
    int i = 0;
    SigmaBisection(float clip_perc_low, float clip_perc_high, int64_t total_samples)
        : perc_low(clip_perc_low), perc_high(clip_perc_high), total((float)total_samples)
    {
        low_mid = 0.5f * (low_left + low_right);
        high_mid = 0.5f * (high_left + high_right);
    }
    // counters of the pass run with (low_mid, high_mid); true = done, else the mids have moved
    bool step(int64_t clip_low, int64_t clip_high)
    {
        const float pl = (float)clip_low * 100.0f / total;
        const float ph = (float)clip_high * 100.0f / total;
        const int delta_l = (int)(100 * pl + 0.5f) - (int)(100 * perc_low);
        const int delta_h = (int)(100 * ph + 0.5f) - (int)(100 * perc_high);
        if ((delta_l == 0 && delta_h == 0) || i >= 20) return true;
        i++;
        if (delta_l > 0) { low_left = low_mid; low_mid = 0.5f * (low_left + low_right); }
        else if (delta_l < 0) { low_right = low_mid; low_mid = 0.5f * (low_left + low_right); }
        if (delta_h > 0) { high_left = high_mid; high_mid = 0.5f * (high_left + high_right); }
        else if (delta_h < 0) { high_right = high_mid; high_mid = 0.5f * (high_left + high_right); }
        return false;
    }
};

// Newton's method of the goal-seek for the linear fit (spec: stackfindsigma.go:101-170), as a state
// machine fed with the counters of one pass at a time.  The reference's quirks are kept: both high
// deltas subtract the LOW target (:114, :155) and the loop counter advances by three per iteration.
struct SigmaNewton {
    float sig_low = 6.0f, sig_high = 6.0f;
    const float epsilon = 0.005f;
    float perc_low, total;
    int i = 0, phase = 0;                 // phase 0: base pass, 1: sig_low + eps, 2: sig_high + eps
    float delta_l = 0, delta_h = 0, new_low = 0;
    int64_t base_lo = 0, base_hi = 0;     // counters of the last base pass (what the reference returns)
    SigmaNewton(float clip_perc_low, int64_t total_samples) : perc_low(clip_perc_low), total((float)total_samples) {}
    // sigmas of the pass to run next
    float next_low() const { return phase == 1 ? sig_low + epsilon : sig_low; }
    float next_high() const { return phase == 2 ? sig_high + epsilon : sig_high; }
    // counters of that pass.  0: go on; 1: done, the pass just run was the base pass; 2: done, but
    // the result image has to be re-made with (sig_low, sig_high) -- a probe pass overwrote it
    int step(int64_t clip_low, int64_t clip_high)
    {
        if (phase == 0) {
            base_lo = clip_low; base_hi = clip_high;
            const float pl = (float)clip_low * 100.0f / total, ph = (float)clip_high * 100.0f / total;
            delta_l = pl - perc_low;
            delta_h = ph - perc_low;                             // sic
            const int li = (int)(100 * delta_l + 0.5f), hi = (int)(100 * delta_h + 0.5f);
            if ((li == 0 && hi == 0) || i >= 20) return 1;
            i++; phase = 1;
            return 0;
        }
        if (phase == 1) {
            const float d2 = (float)clip_low * 100.0f / total - perc_low;
            const float diff = (d2 - delta_l) / epsilon;
            if (diff == 0) return 2;
            new_low = sig_low - delta_l / diff;
            if (new_low < 0.1f) new_low = 0.1f;
            if (new_low > 20) new_low = 20;
            i++; phase = 2;
            return 0;
        }
        const float d3 = (float)clip_high * 100.0f / total - perc_low;      // sic
        const float diff = (d3 - delta_h) / epsilon;
        if (diff == 0) return 2;
        float new_high = sig_high - delta_h / diff;
        if (new_high < 0.1f) new_high = 0.1f;
        if (new_high > 20) new_high = 20;
        sig_low = new_low; sig_high = new_high;
        i++; phase = 0;
        return 0;
    }
};

// ---- stack_exact.hip ----
// Picks lanes-per-wave and LDS bytes for the exact kernel; -1 if it cannot fit.
int exact_plan(int mode, bool weighted, int n_frames, int n_pad, int max_lanes, int *lanes,
               size_t *lds_bytes);
hipError_t launch_stack_exact(int mode, bool weighted, StackArgs &args, int lanes, int grid,
                              size_t lds_bytes, hipStream_t stream, const char **name);
// list_counts (optional): {exact-list length, generic-list length} of the pass, left in counters[2] (low | high << 32);
// a chunked pass has n_lists such pairs, list_stride words apart, and leaves their sums
// zero_after: the kernel leaves the scratch set (kScratchWords words at `partial`) zeroed for the next pass
hipError_t launch_reduce_counters(unsigned long long *partial, int n_blocks,
                                  unsigned long long *counters, hipStream_t stream,
                                  const unsigned *list_counts = nullptr, int n_lists = 1, int list_stride = 0,
                                  bool zero_after = false);

// ---- stack_fast.hip ----
// one-lane register kernels address a group of 4 frames through one buffer descriptor with
// 32-bit offsets (3 frames + the pixel): tiles of 2^27 pixels or more take the int64-indexed kernels
constexpr int64_t kFastMaxPixels = (int64_t)1 << 27;
int fast_supported(int mode, bool weighted, int n_frames, int64_t npix);
hipError_t launch_stack_median_fast(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                                    const char **name, hipEvent_t dominant_done);
// dominant_done (optional) is recorded right after the first, dominant kernel
int mad_fast_supported(int mode, bool weighted, int n_frames, int64_t npix);
hipError_t launch_stack_mad_fast(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name);
// after_dominant(user) is called between the launch of the dominant (zonal) kernel and
// the generic pass, so the caller can start work that only depends on the former
typedef void (*AfterDominant)(void *user);
// 249..256 / 505..512 frames: zonal sigma / winsorized sigma pass with the clipping rounds on LDS
// columns (stack_fast_mlz.hip); hand-over lists as the other fast kernels
// generic pass of the multi-lane sigma / winsor kernels over fargs.in_list, whole columns in LDS (stack_fast_mlg.hip)
hipError_t launch_stack_sigma_mlg(const StackArgs &args, const FastArgs &fargs, unsigned grid, hipStream_t stream,
                                  bool winsor);
int fast_mlz_supported(int mode, bool weighted, int n_frames);
// rows per pixel the split pass of this mode / frame count keeps in FastArgs::cols (0: the pass is not split)
int mlz_split_rows(int mode, int n_frames);
hipError_t launch_stack_sigma_mlz(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name,
                                  bool winsor);
// tail (optional): the stream the generic pass is launched on instead of `stream` -- chunked passes
// (nlstack_api.hip), whose after_dominant callback orders it behind the dominant kernel
// fused_replay (optional; tail_fused_supported): the generic pass and the replay of the exact list as the dominant kernel left
// it run as ONE launch (stack_tail_fused.hip) -- *fused_replay are the replay's arguments (list part 0), in fused_replay_blocks
// workgroups; after_dominant is then not needed for the replay
hipError_t launch_stack_sigma_fast(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                                   const char **name, hipEvent_t dominant_done,
                                   bool winsor, AfterDominant after_dominant, void *user, hipStream_t tail = nullptr,
                                   const StackArgs *fused_replay = nullptr, unsigned fused_replay_blocks = 0);
// ---- stack_tail_fused.hip: generic pass (one lane per pixel, LDS columns) + first replay in one grid; plain sigma, 65 ... 128 frames
int tail_fused_supported(int mode, bool weighted, int n_frames);
hipError_t launch_stack_sigma_tail(const StackArgs &generic, const FastArgs &fargs, unsigned gen_blocks,
                                   const StackArgs &replay, unsigned replay_blocks, hipStream_t stream);

constexpr int kCascadeStages = 6;   // most stages of a winsorization cascade, the dominant kernel included
// rounds of clip bounds a decision pass records per pixel (pixels that need more are replayed in full)
constexpr int kBoundRounds = 8;
// ---- stack_fast_decide.hip: the register-resident kernels as the DECISION pass of weighted sigma / winsorized
// stacks (StackArgs::bounds / nrounds); 33 ... 128 frames (decide_supported); 129 ... 512 frames: the LDS-column kernel
// of the frame-count class, record-only (decide_ml_supported) ----
int decide_supported(int mode, int n_frames, int64_t npix);
int decide_ml_supported(int mode, int n_frames, int64_t npix);      // 129 ... 512 frames: the LDS-column kernel of the class, FastArgs::record_only
hipError_t launch_stack_sigma_decide(const StackArgs &args, hipStream_t stream, bool winsor, const char **name);

// ---- stack_fast_ml.hip (129..512 frames, 2 or 4 lanes per pixel) ----
int fast_ml_supported(int mode, bool weighted, int n_frames, int64_t npix);
hipError_t launch_stack_median_ml(const StackArgs &args, hipStream_t stream, const char **name);
hipError_t launch_stack_mad_ml(const StackArgs &args, const FastArgs &fargs, hipStream_t stream, const char **name);
hipError_t launch_stack_sigma_ml(const StackArgs &args, const FastArgs &fargs, hipStream_t stream,
                                 const char **name, hipEvent_t dominant_done, bool winsor,
                                 AfterDominant after_dominant, void *user, hipStream_t tail = nullptr);

// ---- stack_exact_coop.hip (bit-exact sigma replay, one wave per pixel) ----
int coop_supported(int mode, bool weighted, int n_frames);
hipError_t launch_stack_median_coop(const StackArgs &args, int grid, hipStream_t stream, const char **name);
hipError_t launch_stack_sigma_coop(int mode, const StackArgs &args, int grid, hipStream_t stream, const char **name);
int coop_group(const StackArgs &args);      // pixels per work item of that launch (4: whole-tile replay with aligned 16-byte loads)

// ---- stack_exact_coop4.hip (the same replay, four pixels per wave on 16-lane rows: the sequential sums cost a
// third of the instructions per pixel) ----
int coop4_supported(int mode, bool weighted, int n_frames);
hipError_t launch_stack_sigma_coop4(int mode, const StackArgs &args, int grid, hipStream_t stream, const char **name);

// ---- stack_exact_tile.hip (bit-exact sigma / winsorized clipping over whole tiles: one wave = 64
// consecutive pixels, columns in LDS, one pixel per lane; the weighted modes' default path) ----
// frame counts up to which it beats the other replays over a whole tile (tools/replay_probe2.py, weighted stacks,
// ms per 512 x 4096 pixels, tile / four pixels per wave / one pixel per wave -- sigma: 40 frames 1.54 / 2.16 / 2.79,
// 56 frames 2.82 / 2.84 / 3.17, 64 frames 3.5 / 2.9 / 3.2; winsorized: 40 frames 4.37 / 5.22 / 8.11, 48 frames
// 5.34 / 5.28 / 8.25, 56 frames 6.70 / 6.27 / 8.54)
constexpr int kTileMaxFramesSigma = 40, kTileMaxFramesWinsor = 32;
// (those numbers are from before the decision pass.  With it -- 33 ... 128 frames: the register-resident kernel decides
// every round, the replay only permutes -- the one-pixel-per-wave replay needs no sequential sums any more and, with
// its partition passes in registers, wins wherever the decision pass exists, whole 4096^2 images in ms: sigma 36
// frames 9.3 (tile) vs 11.5, 44: 13.7 vs 13.7, 56: 21.7 vs 16.9, 64: 21.2 (four pixels per wave) vs 18.9, 96: 30.6
// vs 26.0; winsorized 36 frames 27.2 (tile) vs 16.6, 44: 36.2 vs 17.4, 96: 35.5 (four) vs 28.9, 128: 52.6 vs 38.3;
// with four pixels per work item (stack_exact_coop.hip, GROUP): sigma 34 frames 8.5 (tile) vs 11.1, 40: 11.4 vs 11.7,
// 44: 13.7 vs 12.0.)  Four pixels per wave stays for winsorized stacks WITHOUT a decision pass (since the LDS-column
// kernels decide 129 ... 512 frames: only with developer switch 4 or without memory for the bounds), for
// kCoop4MinFrames ... kCoop4MaxFrames frames (ms per 2048 x 4096 pixels, four pixels per wave vs one: 136 frames 65 vs 89,
// 160 frames 79 vs 93, 192 frames 115 vs 98).
constexpr int kCoop4MinFrames = 129, kCoop4MaxFrames = 176;
int tile_supported(int mode, bool weighted, int n_frames);
hipError_t launch_stack_sigma_tile(int mode, const StackArgs &args, int grid, hipStream_t stream, const char **name);

// ---- stack_linfit.hip (register-resident linear fit, bit-exact) ----
// Cascade: a stage runs at most max_iters fit iterations; pixels that are not done are
// appended to out_list together with their liveness mask (4 words) and continued by the
// next stage in freshly packed waves (lists and states: npix entries each).
struct LinfitCascade {
    unsigned *list[2];            // ping-pong pixel lists
    uint4 *state[2];              // liveness masks of the listed pixels
    unsigned *count;              // device: [kLinfitStages] list lengths, zeroed per pass
    unsigned capacity;
};
constexpr int kLinfitStages = 4;
constexpr int kLinfitCounters = 8;  // device list lengths of one pass (the bit-exact cascade uses the first kLinfitStages)
int linfit_fast_supported(int mode, int n_frames, int64_t npix);
int linfit_ml_supported(int mode, int n_frames, int64_t npix);

// ---- stack_linfit_guard.hip (guarded stages in front of the bit-exact cascade, 17 ... 128 frames) ----
// Three pixel lists with liveness masks (npix entries each) and kLinfitCounters list lengths, zeroed per pass:
// two continuation lists ping-pong between the guarded stages, the third collects the pixels a guarded stage cannot
// decide; the bit-exact stages over it reuse the first two.
struct LinfitGuardBufs {
    unsigned *list[3];
    uint4 *state[3];
    unsigned *count;              // device: [kLinfitCounters]
    unsigned capacity;
};
struct LinfitGuardLists {          // where a guarded stage hands its undecidable pixels over
    unsigned *x_list; unsigned *x_count; uint4 *x_state; unsigned x_capacity;
};
struct LinfitStage;
int linfit_guard_supported(int mode, int n_frames, int64_t npix);
hipError_t launch_stack_linfit_guarded(const StackArgs &args, const FastArgs &fargs, const LinfitGuardBufs &bufs,
                                       hipStream_t stream, const char **name, hipEvent_t dominant_done);
// one continuation stage of the bit-exact one-lane kernel over stage.in_list (stack_linfit.hip)
void launch_linfit_exact_stage(const StackArgs &args, const FastArgs &fargs, const LinfitStage &stage, unsigned blocks,
                               hipStream_t stream);

hipError_t launch_stack_linfit_fast(const StackArgs &args, const FastArgs &fargs, const LinfitCascade *cascade,
                                    hipStream_t stream, const char **name, hipEvent_t dominant_done);
hipError_t launch_stack_linfit_ml(const StackArgs &args, const FastArgs &fargs, const LinfitCascade *cascade,
                                  hipStream_t stream, const char **name, hipEvent_t dominant_done);

// ---- stack_mean.hip ----
hipError_t launch_stack_mean(bool weighted, const StackArgs &args, hipStream_t stream,
                             const char **name);
hipError_t launch_axpy(float *acc, const float *x, float weight, int first, int64_t n,
                       hipStream_t stream);
hipError_t launch_scale(float *acc, float factor, int64_t n, hipStream_t stream);

// ---- ingest.hip (FITS payload decode / encode, MatchHistogram, Project) ----
int fits_bytes_per_value(int bitpix);
hipError_t launch_fits_decode(const void *raw, int bitpix, int64_t n, float bscale, float bzero, bool affine,
                              float mult, float off, float *out, double *partial /*[blocks][3]*/, int blocks,
                              hipStream_t stream);
hipError_t launch_fits_encode(const float *data, int64_t n, int replace_nans, void *raw, hipStream_t stream);
hipError_t launch_affine(float *data, int64_t n, float mult, float off, hipStream_t stream);
hipError_t launch_project(const float *src, int src_w, int src_h, float *dst, int dst_w, int row0, int rows,
                          const float inv[6], float oob, bool affine, float mult, float off,
                          hipStream_t stream);

// ---- synth.hip ----
hipError_t launch_fill_synthetic(float *frames, int64_t stride, int n_frames, int width,
                                 int height, int row0, int rows, uint64_t seed,
                                 hipStream_t stream);

// ---- frame_stats.hip ----
hipError_t launch_min_sum_max(const float *data, int64_t n, double *partial /*[blocks][3]*/,
                              int blocks, hipStream_t stream);
hipError_t launch_variance(const float *data, int64_t n, float mean, double *partial /*[blocks]*/,
                           int blocks, hipStream_t stream);
hipError_t launch_noise(const float *data, int width, int height, double *partial /*[blocks]*/,
                        int blocks, hipStream_t stream);
hipError_t launch_median3x3(const float *in, float *out, int width, int height,
                            hipStream_t stream);
constexpr int kMedianMaskMax = 32;
hipError_t launch_median_mask(const float *in, float *out, int64_t n, const int *mask, int len, hipStream_t stream);

}  // namespace nl
