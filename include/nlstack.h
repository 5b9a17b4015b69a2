/*
 * nlstack.h -- C ABI of libnlstack.so: MI355X (gfx950) implementation of
 * Nightlight's per-pixel stacking hot path.
 *
 * This is the drop-in boundary.  A thin cgo shim (go/ and INTEGRATION.md)
 * binds exactly these entry points from a replacement of the reference's
 * `internal/ops/stack` package, keeping the ops.Operator surface
 * (internal/ops/operator.go:135-138) and OpStack's JSON fields
 * (internal/ops/stack/stack.go:66-73) unchanged.
 *
 * Conventions
 *   - plain C types only; every pointer is caller-owned and is NOT retained
 *     after the call returns (cgo pointer rules), except device pointers the
 *     caller explicitly lends with nl_stack_attach_device_frames();
 *   - functions returning int return NL_OK (0) or a negative NL_ERR_* code and
 *     never abort; nl_last_error() gives the message of the calling thread's
 *     last failure (the Go shim turns it into an `error`);
 *   - every entry point selects its handle's device itself (goroutines migrate
 *     between OS threads; SURVEY.md section 8b "Threading");
 *   - stack modes are numbered exactly as StackMode, stack.go:33-42.
 *
 * Data layout in HBM: frames are planar [n_frames][rows*width] fp32 for the
 * row tile [row0, row0+rows) of a width x height image (whole image: row0=0,
 * rows=height).  NaN = "no data" (alignment out-of-bounds), as the reference.
 */
#ifndef NLSTACK_H
#define NLSTACK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* StackMode, internal/ops/stack/stack.go:33-42 */
#define NL_ST_MEDIAN        0
#define NL_ST_MEAN          1
#define NL_ST_SIGMA         2
#define NL_ST_WINSOR_SIGMA  3
#define NL_ST_MAD_SIGMA     4
#define NL_ST_LINEAR_FIT    5
#define NL_ST_AUTO          6

/* StackWeighting, internal/ops/stack/stack.go:57-63 */
#define NL_WEIGHT_NONE           0
#define NL_WEIGHT_EXPOSURE       1
#define NL_WEIGHT_INVERSE_NOISE  2
#define NL_WEIGHT_INVERSE_HFR    3

#define NL_OK                      0
#define NL_ERR_INVALID_MODE       -1  /* "invalid stacking mode", stack.go:119 */
#define NL_ERR_MISSING_EXPOSURE   -2  /* "%d: Missing exposure information ...", stack.go:238 */
#define NL_ERR_INVALID_WEIGHTING  -3  /* "Invalid weighting mode %d", stack.go:267 */
#define NL_ERR_WEIGHTED_MAD       -4  /* the reference panics (stack.go:185); we return an error */
#define NL_ERR_NO_INPUTS          -5  /* "stack operator needs inputs", stack.go:103 */
#define NL_ERR_INVALID_ARG        -6
#define NL_ERR_HIP                -7  /* HIP runtime failure; message has the hipError string */
#define NL_ERR_TOO_MANY_FRAMES    -8  /* per-pixel column does not fit the 160 KiB LDS */
#define NL_ERR_NO_DEVICE          -9

typedef struct nl_stack nl_stack_t;

/* message of the calling thread's last error ("" if none) */
const char *nl_last_error(void);
/* number of visible HIP devices, or a negative error */
int nl_device_count(void);
/* library version string.  0.2.0 (round 5 / 6): the OWNED frame buffer is padded -- frame k starts at
 * nl_stack_frames_device_ptr(h) + k * nl_stack_frame_stride(h) floats, NOT at k * rows * width; a producer written
 * against 0.1.0's dense layout must read the stride (or lend its own dense buffer with nl_stack_attach_device_frames). */
const char *nl_version(void);
/* nl_stack_destroy parks the large device buffers of a handle (frames, result, hand-over lists, the scratch of the
 * winsorized / linear-fit cascades and of weighted passes; per device at most 64 blocks and NL_MEM_CACHE_MB MiB -- default:
 * a sixteenth of the device's memory, 0 turns the cache off) for the next nl_stack_create / nl_group_create of the same
 * geometry on the same device: a drop-in that creates one handle per OpStack.Apply (stack.go:131-138 allocates per
 * call, too) otherwise pays more for hipMalloc + hipFree than for the stack pass.  This returns the parked buffers to
 * HIP.  The library does so itself whenever ANY of its own device allocations fails (every one of them goes through
 * one helper that releases the cache and retries); allocations of OTHER code in the process (torch, RCCL) do not see
 * the parked blocks as free memory -- call this, or set NL_MEM_CACHE_MB=0, when the process shares the device.
 * The streams of destroyed handles are parked the same way (main + side stream as the pair they were created as, copy
 * streams in a list of their own, up to 16 of each per device: destroying a handle's two or three streams was 0.5 ms of
 * its 0.55 ms, creating them 0.2 of 0.24; NL_STREAM_POOL=0 turns it off), and so are the pinned staging buffers of the
 * asynchronous uploads (at most 16 blocks / 2 GiB per process: hipHostMalloc + hipHostFree of the ring were 20 ms of an
 * Apply from host memory); all are released here as well.  The limits are per device (a sixteenth of THAT device's
 * memory); a buffer above the limit -- the 32 GiB frame buffer of a 512 x 4096 x 4096 stack -- is never parked.
 * No counterpart in the reference. */
void nl_release_cached_memory(void);

/* ---- handle: replaces the per-call state of OpStack.Apply (stack.go:115-227) ---- */

/* Allocates the planar [n_frames][rows*width] device buffer, the result tile
 * and the counter scratch on `device`.  Returns NULL on failure. */
nl_stack_t *nl_stack_create(int n_frames, int width, int height, int row0, int rows, int device);
void nl_stack_destroy(nl_stack_t *h);

/* Copies the handle's row tile out of one full host frame (width*height
 * floats = fits.Image.Data, internal/fits/fits.go:42) into slot `idx`.
 * Synchronous; the pointer is not retained.  One call per Go slice. */
int nl_stack_upload_frame(nl_stack_t *h, int idx, const float *host_frame);
/* Same, but the host buffer holds only the tile (rows*width floats). */
int nl_stack_upload_tile(nl_stack_t *h, int idx, const float *host_tile);
/* Overlapped upload (caller side of the path, OpStackBatches' frame loop,
 * internal/ops/stack/stackbatches.go:68-96): the frame's tile is copied into a
 * pinned staging buffer and the call returns -- host_frame is not retained --
 * while the DMA proceeds on a copy stream.  The next nl_stack_run* waits for
 * the uploads on the device; nl_stack_upload_wait waits on the host. */
int nl_stack_upload_frame_async(nl_stack_t *h, int idx, const float *host_frame);
int nl_stack_upload_wait(nl_stack_t *h);
/* Device address of the planar frame buffer (for in-place producers that
 * already live on the GPU); valid until destroy/attach.  Frame k starts at
 * nl_stack_frame_stride(h) * k floats, see below. */
void *nl_stack_frames_device_ptr(nl_stack_t *h);
/* ---- FITS framing (host): where the payload sits in a file image, and the frame around a result ------------------
 * internal/fits/read.go:445-469 reads the header in 2880-byte units of 80-byte cards up to END; :97-147 take SIMPLE,
 * BITPIX, NAXIS, NAXISn (mandatory) and BZERO (default 0), BSCALE (default 1), EXPOSURE or else EXPTIME (default 0) from
 * it.  internal/fits/write.go:54-89 writes SIMPLE, BITPIX -32, NAXIS, NAXISn, BZERO, BSCALE, EXPOSURE (if non-zero),
 * PROGRAM, END, pads header and payload to 2880 bytes with spaces.  The payload itself is decoded / encoded on the device
 * (nl_stack_upload_frame_fits, nl_stack_download_result_fits). */
#define NL_FITS_MAX_AXES 8
typedef struct nl_fits_header {
    int32_t bitpix, naxis, naxisn[NL_FITS_MAX_AXES];
    float bzero, bscale, exposure;
    int64_t pixels;                  /* product of the axes */
    int64_t header_bytes;            /* = offset of the payload in the file, a multiple of 2880 */
    int64_t payload_bytes;           /* pixels * |bitpix| / 8 */
    int64_t padded_payload_bytes;    /* the same, rounded up to 2880 */
} nl_fits_header_t;
/* parses the header at the start of a file image; `id` is the frame number the reference puts in front of its messages */
int nl_fits_parse_header(const void *file_bytes, int64_t n_bytes, int id, nl_fits_header_t *out);
/* the header Image.Write emits for a BITPIX -32 image; returns its length (a multiple of 2880; with dst == NULL only
 * that), -1 on error */
int64_t nl_fits_write_header(void *dst, int64_t capacity, int naxis, const int32_t *naxisn, float bzero, float bscale,
                             float exposure);
int64_t nl_fits_padded_bytes(int64_t payload_bytes);

/* Device memory (bytes) the handle holds right now: the buffers of nl_stack_create plus what passes and upload paths
 * have allocated since and keep until nl_stack_destroy (e.g. 65 bytes per pixel of the tile for the thresholds of the
 * weighted clip modes' decision pass).  The reference sizes its batches to host memory (stackbatches.go:121-187); a
 * caller doing the same for the device reads this.  No counterpart in the reference. */
int64_t nl_stack_device_bytes(nl_stack_t *h);
/* Lends an existing DENSE device buffer -- frame k at device_frames + k * rows * width floats -- instead of the
 * owned one (NULL restores the owned buffer).  The caller keeps it alive. */
int nl_stack_attach_device_frames(nl_stack_t *h, void *device_frames);
/* Frame layout.  The owned buffer is planar with nl_stack_frame_stride(h) floats between consecutive frames:
 * rows*width rounded up, plus a fixed padding, so that the same pixel of consecutive frames does not fall into the
 * same HBM channel and bank (a power-of-two frame size such as 4096 x 4096 x 4 bytes otherwise costs the 512-frame
 * pass 12 %).  Producers that write through nl_stack_frames_device_ptr, and handles that
 * borrow another handle's frames, use this stride: frame k starts at ptr + k * stride floats.  The _strided attach
 * lends a buffer of any stride >= rows*width (a multiple of 4 floats when rows*width is one); the plain attach above
 * is the stride rows*width.  All upload / download / ingest / statistics entry points follow the handle's current
 * stride.  No counterpart in the reference (its frames are separate Go slices). */
int64_t nl_stack_frame_stride(nl_stack_t *h);
int nl_stack_attach_device_frames_strided(nl_stack_t *h, void *device_frames, int64_t frame_stride);
/* Fills all frames on the device with the deterministic synthetic stack of
 * SURVEY.md section 8d (sky gradient + per-frame gain/offset/noise, 0.4 % hot
 * and 0.1 % cold outliers, NaN borders, one all-NaN 8x8 patch).  Pixel
 * coordinates are those of the full image, so tiles agree with the whole. */
int nl_stack_fill_synthetic(nl_stack_t *h, uint64_t seed);
/* Downloads frame `idx`'s tile (rows*width floats). */
int nl_stack_download_tile(nl_stack_t *h, int idx, float *host_tile);
/* Downloads n_rows rows starting at tile-relative row first_row (n_rows*width
 * floats) of frame `idx`, or of the result tile of the last finished pass when
 * idx == -1.  Rows are contiguous in the planar layout, so this is one DMA;
 * it lets a caller inspect parts of stacks far larger than host memory. */
int nl_stack_download_rows(nl_stack_t *h, int idx, int first_row, int n_rows, float *host_rows);

/* Frames in use by the next uploads / passes: slots [0, n), 1 <= n <= the n_frames given
 * to nl_stack_create.  Lets a batch loop (OpStackBatches, stackbatches.go:69-111) keep one
 * handle and its device accumulator across batches of different sizes.  Clears the weights
 * when n changes. */
int nl_stack_set_active_frames(nl_stack_t *h, int n);
/* getWeights (stack.go:231-270).  weights: n_frames floats or NULL = none. */
int nl_stack_set_weights(nl_stack_t *h, const float *weights);
/* Computes the weights from per-frame scalars exactly as getWeights does:
 * NL_WEIGHT_EXPOSURE: w=exposure (error if 0, *bad_index = frame);
 * NL_WEIGHT_INVERSE_NOISE / _HFR: w = 1/(1+4*(v-min)/(max-min)).
 * Pure host arithmetic, no device work. */
int nl_weights_from_scalars(int weighting, const float *per_frame, int n_frames,
                            float *weights_out, int *bad_index);

/* One stack pass = the numeric core of OpStack.Apply (stack.go:142-218):
 * runs Stack{Median,Mean,MeanWeighted,Sigma,SigmaWeighted,MADSigma,
 * WinsorSigma,WinsorSigmaWeighted,LinearFit} (stack.go:274-918) over the tile.
 * mode NL_ST_AUTO is resolved from n_frames as stack.go:45-55.
 * out_host: full-image buffer (width*height floats); the tile's rows are
 * written at offset row0*width.  NULL leaves the result on the device.
 * clip_low/high: totals for THIS tile (sum them across tiles/ranks). */
int nl_stack_run(nl_stack_t *h, int mode, float sigma_low, float sigma_high, float ref_loc,
                 float *out_host, int64_t *clip_low, int64_t *clip_high);
/* Asynchronous split of the same: enqueue on the handle's stream ... */
int nl_stack_run_async(nl_stack_t *h, int mode, float sigma_low, float sigma_high, float ref_loc);
/* ... then wait and fetch.  Any of the three outputs may be NULL. */
int nl_stack_finish(nl_stack_t *h, float *out_host, int64_t *clip_low, int64_t *clip_high);
/* Device address of the result tile (rows*width floats). */
void *nl_stack_result_device_ptr(nl_stack_t *h);
/* Mode actually run by the last pass (after NL_ST_AUTO resolution). */
int nl_stack_last_mode(nl_stack_t *h);
/* GPU time of the last pass's kernels in ms, from HIP events recorded on the
 * handle's stream around the launches (valid after finish/run). */
float nl_stack_last_kernel_ms(nl_stack_t *h);
/* Same, for the dominant kernel of the pass alone (the one named by
 * nl_stack_last_kernel_name); the difference is the hand-over passes. */
float nl_stack_last_dominant_kernel_ms(nl_stack_t *h);
/* GPU times of the pass enqueued `back` passes ago (0 = the last one): the handle keeps the
 * HIP events of its last 64 passes, so a caller may queue passes back to back without a host
 * sync and read every pass's kernel time afterwards.  Either output may be NULL. */
int nl_stack_pass_times(nl_stack_t *h, int back, float *pass_ms, float *dominant_ms);
/* The handle's hipStream_t (passes are enqueued on it) and the device address of the
 * {clip_low, clip_high} totals of the last pass (2 x uint64, valid once the pass has run on that
 * stream): a multi-process caller reduces them across ranks ON THE DEVICE -- e.g. RCCL
 * ncclAllReduce on this stream -- instead of a host round trip per pass (stack.go:193-198). */
void *nl_stack_stream(nl_stack_t *h);
void *nl_stack_counters_device_ptr(nl_stack_t *h);
/* Enqueues, behind the last pass on the handle's stream, a copy of those 16 bytes into a
 * caller-owned device buffer (zeros for modes without counters). */
int nl_stack_copy_counters_async(nl_stack_t *h, void *device_dst);
/* The passes enqueued from now on leave their {clip_low, clip_high} (two int64; then two more 64-bit words of
 * bookkeeping) in `device_buf` -- 32 bytes of device memory of the caller, on the handle's device -- instead of
 * the handle's own buffer; NULL goes back to the own buffer.  For a per-process launcher that reduces the counters
 * of every pass over the ranks on the device (RCCL, stack.go:193-198 is a mutex-protected sum over goroutines): with a
 * small ring of buffers -- pass i into buffer i mod 3, the all-reduce of pass i in place on the first 16 bytes while
 * pass i + 1 runs -- no copy kernel sits between a pass and its collective (bench.py; 11 us of a 0.27 ms pass on one of
 * eight row tiles of the headline stack).  nl_stack_finish reads the buffer of the LAST pass, i.e. whatever the
 * caller's collective has made of it by then.  A buffer must not be handed to a new pass before the collective of
 * the pass that wrote it last has finished. */
int nl_stack_set_counters_buffer(nl_stack_t *h, void *device_buf);
/* Makes `hip_stream` (a hipStream_t of the handle's device) wait for everything enqueued on the handle so far -- the passes
 * and what they leave in the counters buffer -- without the handle's stream waiting for anything: one event record on it, with
 * no system-scope fence (device work ordered against device work).  For the launcher above: the stream a collective is
 * issued from waits for pass i this way, and pass i + 1 follows pass i on the handle's stream undisturbed (bench.py; measured on
 * a 512-row share of the headline stack with a world-size-1 RCCL group: 0.2634 -> 0.2605 ms per step -- most of the 20 us a
 * step takes beyond its pass is the collective's own work on the device, not its bookkeeping).  No counterpart in the
 * reference. */
int nl_stack_order_stream_after(nl_stack_t *h, void *hip_stream);
/* on != 0: run every mode with the bit-exact kernels only (per-pixel replay
 * of the reference's permutation; slow, used for verification; 1 = one pixel
 * per lane with the column in LDS, 2 = one wavefront per pixel, 3 = one
 * wavefront per 64 consecutive pixels with their columns in LDS, 4 = four
 * pixels per wavefront on 16-lane rows -- 2, 3 and 4 exist for sigma and
 * winsorized clipping, weighted or not; by default the weighted clip modes run 3
 * for shallow stacks, a decision pass + 2 for 33 ... 512 frames, 2 above).  Default 0:
 * sigma clipping uses the register-resident kernel, which keeps the clip
 * counters identical to the reference's and the output within summation-order
 * rounding, and hands undecidable pixels to the exact kernel. */
int nl_stack_set_exact(nl_stack_t *h, int on);
/* Developer switches of the sigma / winsorized fast path, for A/B timing inside one process (results are the
 * same either way): bit 0 = plain pass protocol (memset before, reduction kernel after every pass) instead of
 * the fused one, bit 1 = the exact replay of the dominant kernel's hand-overs runs in front of the generic pass
 * on the same stream instead of beside it (kernel traces then show each kernel's own duration), bit 2 = weighted
 * stacks replay every clipping round in full (no decision pass), bit 3 (8) = weighted stacks skip the
 * four-pixels-per-wave replay, bit 4 (16) = they skip the 64-pixels-per-wave tile replay (the next engine of the
 * table in DESIGN.md section 3 runs), bit 5 (32) = a pass records none of its three timing events (start, dominant
 * kernel start / end; nl_stack_pass_times then fails with NL_ERR_INVALID_ARG for that pass.  The event at the END of a
 * pass stays: asynchronous uploads order themselves behind it.  tools/wall_probe.py measures what the events cost),
 * bit 6 (64) = no chunked pass even where the environment variable NL_CHUNKS asks for one,
 * bit 7 (128) = winsorized passes of 16 ... 128 frames without the winsorization cascade,
 * bit 9 (512) = the first pass on a handle takes no list-length hints from earlier handles of the same geometry.
 * bit 10 (1024) = sigma clipping of 497 ... 512 frames as TWO kernels (sorting kernel, then a rounds kernel over columns kept in
 * device memory, 352 bytes per pixel; also NL_MLZ_SPLIT=1) -- measured slower than the one-kernel pass, DESIGN.md section 5n.
 * bit 11 (2048) = the same class with persistent workgroups (three per CU looping over blocks of 64 pixels, no barrier: a block's
 * rounds run in one wave while the others sort the next block; also NL_MLZ_PERSIST=1) -- slower as well, same section.
 * bit 13 (8192) = generic pass and first replay of a short-listed sigma pass on two streams (the protocol of rounds 2 - 4)
 * instead of one launch (stack_tail_fused.hip; also NL_TAIL_FUSED=0), for A/B runs.
 * Default 0.  Bits 10 and 11 (and nl_stack_set_exact(h, 4), and the environment switches NL_CHUNKS, NL_MLZ_SPLIT, NL_MLZ_PERSIST,
 * NL_COOP4, NL_LFG) select code of the EXPERIMENTS build (make EXPERIMENTS=1 -> libnlstack_exp.so): the default library
 * rejects the two bits and the flavour with NL_ERR_INVALID_ARG instead of running its one kernel under another name.
 * No counterpart in the reference. */
int nl_stack_set_dev_flags(nl_stack_t *h, unsigned flags);
/* Pixels of the last pass that were re-done by the exact kernel. */
int64_t nl_stack_last_fallback_pixels(nl_stack_t *h);
/* Pixels of the last pass that the dominant kernel handed to the generic pass (all positions
 * masked by rank: pixels that miss many samples or clip more than the clip zones hold). */
int64_t nl_stack_last_generic_pixels(nl_stack_t *h);
/* How the last pass was enqueued (diagnostics; the results do not depend on it): bit 0 = fused protocol (no memset in front, no
 * reduction kernel behind: sigma / winsorized passes once a handle knows its list lengths), bit 1 = generic pass and first
 * replay as one launch (plain sigma, 65 ... 128 frames, short exact lists; stack_tail_fused.hip), bit 2 = chunked (experiments
 * build).  No counterpart in the reference. */
int nl_stack_last_pass_protocol(nl_stack_t *h);
/* Linear-fit cascade of the last pass (stack_linfit.hip; StackLinearFit stack.go:834-918 has no
 * counterpart, diagnostics only): counts[s] = pixels stage s handed to stage s+1 (4 stages; entries 4 ... 7 belong to the
 * guarded stages of the experiments build and are zero in the default library).  Writes min(n, 8) values -- pass a buffer
 * of 8 -- and returns how many, 0 when the last pass ran no cascade. */
int nl_stack_linfit_stage_counts(nl_stack_t *h, unsigned *counts, int n);
/* Name of the dominant kernel launched by the last pass (for profiles). */
const char *nl_stack_last_kernel_name(nl_stack_t *h);

/* ---- goal-seek (spec: internal/ops/stack/stackfindsigma.go:27-170, FindSigmasAndStack) ----
 * Sigma / winsorized sigma (:48-98): bisection on sigma_low / sigma_high in [1,11] until
 * the clipped percentages match the targets to 0.01 % or 21 passes were made.
 * Linear fit (:101-170): Newton's method from (6, 6) with probe passes at +0.005, the
 * reference's quirks included (both high deltas are taken against the LOW target; the
 * step counter advances by three per iteration).  Other modes "do not support sigmas":
 * one pass with 0, 0; the returned sigmas are 0 (:42-46).  After each
 * pass the tile's {clip_low, clip_high} are handed to `reduce` (may be NULL
 * for a single tile) which must replace them with the totals over all tiles
 * -- e.g. an RCCL all-reduce -- so every rank takes the same branch.
 * The percentages are taken of width*height*n_frames (the WHOLE image) when a
 * reducer is given, of this handle's tile (rows*width*n_frames) when it is NULL. */
typedef int (*nl_reduce_fn)(int64_t *counters2, void *user);
int nl_stack_find_sigmas(nl_stack_t *h, int mode, float ref_loc,
                         float clip_perc_low, float clip_perc_high,
                         nl_reduce_fn reduce, void *user,
                         float *out_host, int64_t *clip_low, int64_t *clip_high,
                         float *sigma_low, float *sigma_high, int *passes);

/* ---- one stack over several GPUs from ONE process (stack.go:142-152, 193-198) ----
 * The reference's Apply splits the pixel range over goroutines and sums the two clip
 * counters over them; nl_group_* is that split over the GPUs of the node for a
 * single-process host (the Go CLI behind cgo, the C++ operator mirror): tile t owns the
 * rows nl_group_tile_rows(height, n_tiles, t) of all frames on devices[t] (devices ==
 * NULL: device t modulo the device count; n_tiles <= 0: one tile per device).  All
 * tiles' passes are enqueued before any is awaited; result tiles land in disjoint rows
 * of out_host; the counters are summed on the host.  Same arguments, error codes and
 * messages as the nl_stack_* calls they fan out to. */
typedef struct nl_group nl_group_t;
void nl_group_tile_rows(int height, int n_tiles, int t, int *row0, int *rows);
nl_group_t *nl_group_create(int n_frames, int width, int height, int n_tiles, const int *devices);
void nl_group_destroy(nl_group_t *g);
int nl_group_size(nl_group_t *g);
nl_stack_t *nl_group_tile(nl_group_t *g, int t);            /* borrowed, owned by the group */
int nl_group_upload_frame(nl_group_t *g, int idx, const float *host_frame);   /* overlapped, pointer not retained */
/* The ingest fast paths on the group (rows F3 / F4: internal/fits/read.go:351-395, project.go:26-76), overlapped
 * like nl_group_upload_frame -- one host thread per tile stages its share, nothing is awaited on the devices.
 * raw_host: the big-endian payload of the WHOLE frame (every tile takes the byte range of its rows);
 * src_host: the whole unaligned source frame (every tile projects its own rows). */
int nl_group_upload_frame_fits(nl_group_t *g, int idx, const void *raw_host, int bitpix, float bscale, float bzero,
                               float multiplier, float offset);
int nl_group_upload_frame_projected(nl_group_t *g, int idx, const float *src_host, int src_w, int src_h,
                                    const float trans[6], float out_of_bounds, float multiplier, float offset);
int nl_group_fill_synthetic(nl_group_t *g, uint64_t seed);
int nl_group_set_active_frames(nl_group_t *g, int n);
int nl_group_set_weights(nl_group_t *g, const float *weights);
int nl_group_set_exact(nl_group_t *g, int on);
int nl_group_run(nl_group_t *g, int mode, float sigma_low, float sigma_high, float ref_loc,
                 float *out_host, int64_t *clip_low, int64_t *clip_high);
int nl_group_last_mode(nl_group_t *g);
int nl_group_find_sigmas(nl_group_t *g, int mode, float ref_loc, float clip_perc_low, float clip_perc_high,
                         float *out_host, int64_t *clip_low, int64_t *clip_high,
                         float *sigma_low, float *sigma_high, int *passes);
int nl_group_accumulate(nl_group_t *g, float weight, int first);
int nl_group_accumulate_finalize(nl_group_t *g, float weight_sum, float *out_host);

/* ---- stack of stacks (StackIncremental / Finalize, stack.go:924-944) ----
 * acc += result_of_last_pass * weight (first != 0: acc = result*weight),
 * on the device; finalize multiplies by 1/weight_sum and downloads. */
int nl_stack_accumulate(nl_stack_t *h, float weight, int first);
int nl_stack_accumulate_finalize(nl_stack_t *h, float weight_sum, float *out_host);

/* ---- per-frame statistics on resident frames (internal/stats) ----
 * calcMinMeanMax + calcVariance (stats_amd64.s:28-143, stats.go:264-287):
 * min/max fp32, mean with fp64 accumulation, variance = sum((x-mean)^2)/n in
 * fp64.  NaN-free input is assumed, as in the reference. */
int nl_stack_frame_stats(nl_stack_t *h, int idx, float *mn, float *mean, float *mx,
                         double *variance);
/* EstimateNoise (stats/noise.go:32-55, noise_amd64.s:78-195).  Needs a
 * whole-image handle (row0=0, rows=height). */
int nl_stack_frame_noise(nl_stack_t *h, int idx, float *noise);
/* Noise of every frame -> inverse-noise weights -> nl_stack_set_weights
 * (stack.go:241-253).  noise_out: n_frames floats or NULL. */
int nl_stack_weights_from_noise(nl_stack_t *h, float *noise_out);

/* ---- formats and steps either side of the stack (F3, F4) ----
 * A frame goes from its on-disk bytes to its slot of the stack buffer without
 * a CPU pass.  All of these are bit-exact restatements (elementwise fp32, the
 * reference's operation order).
 *
 * nl_stack_upload_frame_fits: internal/fits/read.go:172-445 (readUint8Data ..
 * readFloat64Data).  raw_host = the big-endian FITS payload bytes of exactly the
 * handle's tile (rows*width values of BITPIX 8/16/32/64/-32/-64; row-major, so
 * a row tile is a contiguous byte range of the file).  v = float32(val)*bscale
 * + bzero.  stats_out (3 floats or NULL) = min, max, mean of the decoded tile
 * (mean through an fp64 sum, read.go:210).  multiplier/offset: MatchHistogram
 * (internal/fits/pixelops.go:601-605) fused behind the decode; pass 1, 0 for
 * none (then nothing is applied). */
int nl_stack_upload_frame_fits(nl_stack_t *h, int idx, const void *raw_host, int bitpix,
                               float bscale, float bzero, float multiplier, float offset,
                               float *stats_out);
/* nl_stack_upload_frame_projected: Image.Project (internal/fits/project.go:26-76)
 * straight into the frame slot.  src_host = the WHOLE unaligned frame
 * (src_w x src_h fp32); trans = the forward Transform2D {A,B,C,D,E,F}
 * (internal/star/coord.go:52-59), inverted as coord.go:159-199 (singular ->
 * NL_ERR_INVALID_ARG, the reference returns an error too); bilinear taps with
 * the reference's fp32 expressions; destination pixels whose taps leave the
 * source get out_of_bounds (NaN in the pipeline = "no data" for the stack).
 * Only the handle's rows are produced.  multiplier/offset as above. */
/* Overlapped forms of the two calls above (no statistics): the bytes go through the pinned staging ring of
 * nl_stack_upload_frame_async, the DMA, the decode / projection kernel run on the copy stream, the call returns
 * without waiting for the device; the next nl_stack_run* waits for them on the device. */
int nl_stack_upload_frame_fits_async(nl_stack_t *h, int idx, const void *raw_host, int bitpix,
                                     float bscale, float bzero, float multiplier, float offset);
int nl_stack_upload_frame_projected_async(nl_stack_t *h, int idx, const float *src_host, int src_w, int src_h,
                                          const float trans[6], float out_of_bounds, float multiplier,
                                          float offset);
int nl_stack_upload_frame_projected(nl_stack_t *h, int idx, const float *src_host, int src_w,
                                    int src_h, const float trans[6], float out_of_bounds,
                                    float multiplier, float offset);
/* MatchHistogram on a resident frame: x = x*multiplier + offset (pixelops.go:601-605). */
int nl_stack_frame_affine(nl_stack_t *h, int idx, float multiplier, float offset);
/* The result tile of the last pass as FITS payload bytes: big-endian fp32, NaN
 * replaced by 0 (internal/fits/write.go:88, 182-200).  raw_host: rows*width*4 bytes. */
int nl_stack_download_result_fits(nl_stack_t *h, void *raw_host);
/* Stand-alone forms (host in, host out) of the decode and the projection. */
int nl_fits_decode(const void *raw_host, int bitpix, int64_t n, float bscale, float bzero,
                   float *out_host, float *stats_out, int device);
int nl_project_bilinear(const float *src_host, int src_w, int src_h, float *dst_host, int dst_w,
                        int dst_h, const float trans[6], float out_of_bounds, int device);

/* ---- 3x3 spatial median filter (internal/median/median3x3.go:26-110) ----
 * host in/out, width*height floats each; border rows/columns copied. */
int nl_median_filter_3x3(const float *in_host, float *out_host, int width, int height, int device);
/* ---- MedianFilter = GatherAndMedian over every pixel (internal/median/gather.go:26-38,
 * internal/ops/pre/badpixels.go:54-77, MedianFloat32 median3x3.go:115-119) ----
 * out[i] = median of in[i + mask[j]] over the offsets that fall inside [0, n); mask as
 * star.CreateMask builds it (findstars.go:187-200), at most 32 offsets.  Even counts
 * average the two middle values (qsort.go:68-82).  Where the whole neighbourhood exists
 * this equals the reference; at the data's edges the reference's value depends on the
 * leftovers of earlier calls in its scratch buffer, here it is the median of what exists. */
int nl_median_filter_mask(const float *in_host, float *out_host, int64_t n, const int32_t *mask,
                          int mask_len, int device);

/* ---- host-side operator mirror (nightlight_amd/host/, C++) ----
 * The reference's stack operator decoded from its JSON form and run through
 * MakePromises/Apply exactly as OpSequence would drive it
 * (internal/ops/operator.go:484-513, internal/ops/stack/stack.go:92-227), on
 * host frames (frames[i] == NULL is a frame skipped upstream, "(nil, nil)").
 * Returns 0 on success; on failure err_buf holds the reference's message.
 * log_buf receives what the operator wrote to Context.Log. */
int nl_host_op_stack_apply_json(const char *json, int n_frames, int width, int height,
                                const float *const *frames, const float *exposure,
                                const float *hfr, int device, int max_threads,
                                float *out, float *exposure_out,
                                char *log_buf, int log_cap, char *err_buf, int err_cap);
/* Devices every host-side operator of this process stacks on from now on: one row tile of
 * each stack per entry (nl_group_*); a device may repeat.  n <= 0: back to the `device`
 * argument of the calls below. */
int nl_host_set_devices(const int *devices, int n);
/* Unmarshal with defaults (stack.go:92-99), marshal back. */
const char *nl_host_op_stack_roundtrip_json(const char *json);
/* OpStackBatches (internal/ops/stack/stackbatches.go:46-217): partition the inputs
 * into batches that fit stack_memory_mb (Context.StackMemoryMB, operator.go:41),
 * stack every batch with the per-batch "stack" operator given as JSON, combine
 * the batch results with StackIncremental / StackIncrementalFinalize weighted by
 * the batch frame counts (stack.go:924-944) ON THE DEVICES (nl_group_accumulate).  Same log
 * lines and error strings; the permutation comes from a fixed-seed generator instead of Go's
 * math/rand and is sorted inside every batch as stackbatches.go:199-209 does.  perm_out
 * (n_frames ints or NULL): input index of every position, batches = consecutive runs. */
int nl_host_op_stack_batches_apply_json(const char *per_batch_json, int n_frames, int width,
                                        int height, const float *const *frames,
                                        const float *exposure, int device, int max_threads,
                                        int memory_mb, int stack_memory_mb, float *out,
                                        float *exposure_out, int *perm_out, char *log_buf, int log_cap,
                                        char *err_buf, int err_cap);

#ifdef __cplusplus
}
#endif
#endif
