// nlstack_api.hip -- the C ABI of libnlstack.so (include/nlstack.h): handle
// management, uploads, one-pass stacking, goal-seek, per-frame statistics.
// Host-side restatement of the bookkeeping in OpStack.Apply
// (internal/ops/stack/stack.go:115-227); all pixel arithmetic runs in the HIP
// kernels of this directory.  There is no CPU fallback: without a HIP device
// every compute entry point fails with NL_ERR_NO_DEVICE / NL_ERR_HIP.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "stack_kernels.h"

// Kernels and pass protocols that were built, tested and measured slower than what is dispatched live in the
// experiments build (make EXPERIMENTS=1, -DNL_EXPERIMENTS): the four-pixels-per-wave replay (stack_exact_coop4.hip),
// chunked passes, the split / persistent LDS-column pass, the round-1 multi-lane kernel, the guarded linear fit.
#ifdef NL_EXPERIMENTS
#define NL_COOP4_SUPPORTED(mode, weighted, n) (nl::coop4_supported(mode, weighted, n) != 0)
#define NL_LAUNCH_COOP4(...) nl::launch_stack_sigma_coop4(__VA_ARGS__)
#else
#define NL_COOP4_SUPPORTED(mode, weighted, n) false
#define NL_LAUNCH_COOP4(...) hipErrorNotSupported
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

// the thread's error message, for the other translation units of the library (nlstack_group.hip)
namespace nl { void set_last_error(const char *msg) { g_err = msg ? msg : ""; } }

namespace {

#define NL_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(NL_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

constexpr int kTimingRing = 64;     // passes whose HIP-event times can be read back after the fact
constexpr int kStageSlots = 4;      // pinned staging buffers of the asynchronous upload
constexpr int kStageThreads = 4;    // host threads filling one staging buffer
constexpr int kStatBlocks = 2048;
constexpr int kListGrid = 2048;     // workgroups of the exact kernel in fallback-list mode
constexpr int kCoopGrid = 16384;    // workgroups (one wave each) of the wave-per-pixel exact replay
constexpr int kListLanes = 4;       // pixels per wave there: few pixels, keep divergence low
constexpr int kOrderRing = 8;              // events nl_stack_order_stream_after cycles through
constexpr unsigned kFusedMaxList = 512;    // exact-list length up to which a pass runs the fused protocol
constexpr unsigned kTailFusedMaxList = 512;    // ... up to which generic pass and first replay share one launch (stack_tail_fused.hip)
constexpr int kMaxChunks = 16;             // pixel ranges of a chunked pass
// winsorization cascade, "clipping passes : winsorization rounds per pass : regions of the previous stage's list per
// workgroup" for every stage (the last one runs to the end): measured on 4096^2 -- up to 40 frames
// 16 / 24 frames 3.68 / 3.94 -> 2.97 / 3.07 ms, 41 ... 96 frames (64: 5.22 -> 4.59 ms); beyond that a continuing stage
// re-reads every cache line of the stack for an eighth of its pixels and the cascade loses (128 frames: 5.43 -> 5.83 ms)
constexpr const char *kWinsorPlanShallow = "1:6,1:12:4,0:0:4";   // (round 5, with the certificate: first stage 8 -> 6 rounds, three stages instead of four: 16 / 24 / 32 frames 2.34 / 2.78 / 2.97 -> 2.17 / 2.71 / 2.80 ms)
constexpr const char *kWinsorPlanDeep = "2:12,2:16:8,3:24:4,0:0:4";
constexpr int kWinsorCascadeMaxFrames = 96;
// per-pass device scratch, zeroed by one memset (or, in the fused protocol of the sigma / winsorized fast path, by
// the previous pass's dominant kernel -- two sets alternate): clip accumulators + list lengths + snapshot
constexpr size_t kScratchBytes = sizeof(unsigned long long) * nl::kScratchWords;

// ---- device-memory cache ---------------------------------------------------------------------------------------------
// The cgo drop-in creates a handle per OpStack.Apply (stack.go:131-138 allocates per call as well) and destroys it
// afterwards: hipMalloc + hipFree of the frame buffer alone cost more than the headline pass (measured, bench.py
// "fresh_handle": create 1.0 - 1.6 ms, destroy 1.3 - 1.8 ms, pass 1.7 ms).  The large buffers of a destroyed handle are
// therefore parked -- per device at most kCacheBlocks of them and NL_MEM_CACHE_MB MiB (default: a sixteenth of the device's memory; 0 = off) -- and the
// next handle with the same sizes on the same device takes them over.  nl_release_cached_memory() returns them to HIP.
constexpr int kCacheBlocks = 64;
constexpr size_t kCacheMinBytes = (size_t)1 << 20;
struct CachedBlock { int device; size_t bytes; void *ptr; };
std::mutex g_cache_mu;
std::vector<CachedBlock> g_cache;
size_t g_cache_bytes = 0;

// Limit of the parked bytes: NL_MEM_CACHE_MB if set (0 = off), else a sixteenth of the device's memory (18 GB of an
// MI355X's 288: the frame buffer of the headline stack is 8 GiB) -- blocks parked by this library are invisible to the
// other allocators of the process (torch, RCCL) until an allocation of OURS fails, so the default stays small.
size_t cache_limit(int device = -1)
{
    static const long long env_mb = [] { const char *e = getenv("NL_MEM_CACHE_MB"); return e ? atoll(e) : -1ll; }();
    if (env_mb >= 0) return env_mb > 0 ? (size_t)env_mb << 20 : (size_t)0;
    // a sixteenth of THAT device's memory (a table per device: the limit used to be whatever device was current at the
    // first free, applied to all)
    static std::mutex mu;
    static std::vector<size_t> per_device;
    std::lock_guard<std::mutex> lk(mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (device < 0) device = cur;
    if ((size_t)device >= per_device.size()) per_device.resize((size_t)device + 1, 0);
    if (per_device[(size_t)device] == 0) {
        size_t free_b = 0, total_b = 0;
        if (device != cur) (void)hipSetDevice(device);
        const bool ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        if (device != cur) (void)hipSetDevice(cur);
        per_device[(size_t)device] = ok ? total_b / 16 : (size_t)4096 << 20;
    }
    return per_device[(size_t)device];
}

void cache_release_all()
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const CachedBlock &b : g_cache) {
        (void)hipSetDevice(b.device);
        (void)hipFree(b.ptr);
    }
    g_cache.clear();
    g_cache_bytes = 0;
    (void)hipSetDevice(cur);
}

// EVERY device allocation of the library goes through here: when HIP is out of memory while blocks are parked, they are
// handed back and the allocation is tried again (the caller has selected the device).
hipError_t dev_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    bool parked;
    { std::lock_guard<std::mutex> lk(g_cache_mu); parked = !g_cache.empty(); }
    if (!parked) return e;
    (void)hipGetLastError();
    cache_release_all();
    return hipMalloc(p, bytes);
}
template <class T>
hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc(reinterpret_cast<void **>(p), bytes); }

// (the caller has selected `device`)
hipError_t cached_malloc(void **p, size_t bytes, int device)
{
    if (bytes >= kCacheMinBytes) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_cache.size(); i++)
            if (g_cache[i].device == device && g_cache[i].bytes == bytes) {
                *p = g_cache[i].ptr;
                g_cache_bytes -= bytes;
                g_cache.erase(g_cache.begin() + (long)i);
                return hipSuccess;
            }
    }
    return dev_malloc(p, bytes);
}

void cached_free(void *p, size_t bytes, int device)
{
    if (!p) return;
    if (bytes >= kCacheMinBytes) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        // (both limits per DEVICE: a group of one tile per GPU parks the buffers of all its tiles, nlstack_group.hip)
        int blocks = 0;
        size_t parked = 0;
        for (const CachedBlock &b : g_cache)
            if (b.device == device) { blocks++; parked += b.bytes; }
        static const int max_blocks = [] { const char *e = getenv("NL_CACHE_BLOCKS"); return e ? atoi(e) : kCacheBlocks; }();
        if (blocks < max_blocks && parked + bytes <= cache_limit(device)) {
            g_cache.push_back({device, bytes, p});
            g_cache_bytes += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

// Pinned staging buffers are parked as well (round 6): a handle per Apply allocated its ring of kStageSlots pinned buffers
// inside its first uploads and freed it in destroy -- hipHostMalloc + hipHostFree of 4 x 64 MiB are 10 + 10 ms of the 180 ms an
// Apply of 128 frames from host memory takes (bench.py apply_from_host).  At most kPinnedBlocks blocks / kPinnedLimit bytes
// stay (process-wide: pinned memory belongs to no device); a request takes the smallest parked block that is large enough.
constexpr size_t kPinnedBlocks = 16;
constexpr size_t kPinnedLimit = (size_t)2 << 30;
std::mutex g_pinned_mu;
std::vector<std::pair<size_t, void *>> g_pinned;
size_t g_pinned_bytes = 0;

hipError_t pinned_malloc(void **p, size_t bytes, size_t *cap)
{
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        size_t best = g_pinned.size();
        for (size_t i = 0; i < g_pinned.size(); i++)
            if (g_pinned[i].first >= bytes && (best == g_pinned.size() || g_pinned[i].first < g_pinned[best].first)) best = i;
        if (best < g_pinned.size() && g_pinned[best].first <= 2 * bytes + ((size_t)1 << 20)) {
            *p = g_pinned[best].second;
            *cap = g_pinned[best].first;
            g_pinned_bytes -= g_pinned[best].first;
            g_pinned.erase(g_pinned.begin() + (long)best);
            return hipSuccess;
        }
    }
    *cap = bytes;
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

void pinned_free(void *p, size_t cap)
{
    if (!p) return;
    if (cache_limit() > 0) {                               // (NL_MEM_CACHE_MB=0 turns every cache of the library off)
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        if (g_pinned.size() < kPinnedBlocks && g_pinned_bytes + cap <= kPinnedLimit) {
            g_pinned.emplace_back(cap, p);
            g_pinned_bytes += cap;
            return;
        }
    }
    (void)hipHostFree(p);
}

void pinned_release_all()
{
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    for (const auto &e : g_pinned) (void)hipHostFree(e.second);
    g_pinned.clear();
    g_pinned_bytes = 0;
}

// Streams are parked like the buffers: destroying the two or three streams of a handle is most of what nl_stack_destroy
// costs once the buffers stay (tools/group_create_probe.py), and the drop-in makes a handle per Apply.  A parked stream is idle
// (synchronised before it is parked); nl_release_cached_memory destroys them.
// A stream keeps its ROLE (round 6): HIP binds a stream to one of its few hardware queues when it is created, and the main and
// side stream of a handle -- created back to back -- sit on different ones, which is what lets the replay of a pass run beside its
// generic pass.  Round 5 parked all streams in one list: the next handle's main stream could be a former copy stream (created
// lazily, any queue) next to a former side stream on the SAME queue, and the tail of every pass with a long replay serialised
// (C3 tile 4.10 -> 4.58 ms, winsor 24 2.65 -> 2.88, tools/bisect_tail.sh, profiles/r06_bisect_tail.txt).  Main + side are
// therefore parked and handed out as the PAIR they were created as; copy streams have a list of their own.
// NL_STREAM_POOL=0 turns the pool off.
constexpr size_t kStreamPool = 32;
struct StreamPair { int device; hipStream_t main, side; };
std::mutex g_stream_mu;
std::vector<StreamPair> g_stream_pairs;
std::vector<std::pair<int, hipStream_t>> g_copy_streams;           // (device, non-blocking stream at default priority)

bool stream_pool_on()
{
    static const bool on = [] { const char *e = getenv("NL_STREAM_POOL"); return !e || atoi(e) != 0; }();
    return on;
}

hipError_t pooled_stream_pair(hipStream_t *main, hipStream_t *side, int device)
{
    if (stream_pool_on()) {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        for (size_t i = g_stream_pairs.size(); i-- > 0;)
            if (g_stream_pairs[i].device == device) {
                *main = g_stream_pairs[i].main;
                *side = g_stream_pairs[i].side;
                g_stream_pairs.erase(g_stream_pairs.begin() + (long)i);
                return hipSuccess;
            }
    }
    hipError_t e = hipStreamCreateWithFlags(main, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    e = hipStreamCreateWithFlags(side, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipStreamDestroy(*main); *main = nullptr; }
    return e;
}

hipError_t pooled_copy_stream(hipStream_t *s, int device)
{
    if (stream_pool_on()) {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        for (size_t i = g_copy_streams.size(); i-- > 0;)
            if (g_copy_streams[i].first == device) {
                *s = g_copy_streams[i].second;
                g_copy_streams.erase(g_copy_streams.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

static bool stream_idle(hipStream_t s)
{
    if (hipStreamSynchronize(s) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

void park_stream_pair(hipStream_t main, hipStream_t side, int device)
{
    if (main && side && stream_pool_on() && stream_idle(side) && stream_idle(main)) {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        size_t n = 0;
        for (const auto &e : g_stream_pairs) n += e.device == device;
        if (n < kStreamPool / 2) { g_stream_pairs.push_back({device, main, side}); return; }
    }
    if (side) (void)hipStreamDestroy(side);
    if (main) (void)hipStreamDestroy(main);
}

void park_copy_stream(hipStream_t s, int device)
{
    if (!s) return;
    if (stream_pool_on() && stream_idle(s)) {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        size_t n = 0;
        for (const auto &e : g_copy_streams) n += e.first == device;
        if (n < kStreamPool / 2) { g_copy_streams.emplace_back(device, s); return; }
    }
    (void)hipStreamDestroy(s);
}

void stream_pool_release_all()
{
    std::lock_guard<std::mutex> lk(g_stream_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const auto &e : g_copy_streams) { (void)hipSetDevice(e.first); (void)hipStreamDestroy(e.second); }
    g_copy_streams.clear();
    for (const auto &e : g_stream_pairs) { (void)hipSetDevice(e.device); (void)hipStreamDestroy(e.side); (void)hipStreamDestroy(e.main); }
    g_stream_pairs.clear();
    (void)hipSetDevice(cur);
}

// ---- list-length hints across handles ---------------------------------------------------------------------------------
// A pass sizes its replay grids and picks its protocol from the list lengths the last FINISHED pass on the handle
// reported.  A handle that lives for one Apply never has one: its pass ran with 16 384-workgroup replay grids and the
// plain protocol (headline stack: 2.04 instead of 1.73 ms).  The lengths are therefore also remembered per geometry --
// frames, tile pixels, mode, weighted -- in a small process-wide table: the next handle of that geometry starts from
// what the last one saw (stacks of one session resemble each other; a wrong hint costs time, never correctness).
struct HintKey { int frames; int64_t npix; int mode; bool weighted; };
struct HintEntry { HintKey key; unsigned fb, gen; };
std::mutex g_hint_mu;
std::vector<HintEntry> g_hints;
constexpr size_t kHintEntries = 32;

void hints_store(const HintKey &k, unsigned fb, unsigned gen)
{
    std::lock_guard<std::mutex> lk(g_hint_mu);
    for (HintEntry &e : g_hints)
        if (e.key.frames == k.frames && e.key.npix == k.npix && e.key.mode == k.mode && e.key.weighted == k.weighted) {
            e.fb = fb; e.gen = gen;
            return;
        }
    if (g_hints.size() >= kHintEntries) g_hints.erase(g_hints.begin());
    g_hints.push_back({k, fb, gen});
}

bool hints_load(const HintKey &k, unsigned *fb, unsigned *gen)
{
    std::lock_guard<std::mutex> lk(g_hint_mu);
    for (const HintEntry &e : g_hints)
        if (e.key.frames == k.frames && e.key.npix == k.npix && e.key.mode == k.mode && e.key.weighted == k.weighted) {
            *fb = e.fb; *gen = e.gen;
            return true;
        }
    return false;
}

int next_pow2(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace

struct nl_stack {
    int device = 0;
    int n_frames = 0, width = 0, height = 0, row0 = 0, rows = 0;
    int n_capacity = 0;               // frame slots allocated; n_frames <= n_capacity are in use (nl_stack_set_active_frames)
    int64_t npix = 0;                 // rows*width
    int64_t fstride = 0;              // floats between consecutive frames of the buffer d_frames points at
    int64_t fstride_owned = 0;        // ... of the owned buffer (padded_frame_stride); a lent buffer brings its own
    hipStream_t stream = nullptr;
    // HIP events of the last kTimingRing passes (whole pass; dominant kernel only), so a caller can
    // queue many passes without a host sync and read every pass's GPU time afterwards
    hipEvent_t ring_start[kTimingRing] = {}, ring_stop[kTimingRing] = {};
    hipEvent_t ring_dom0[kTimingRing] = {}, ring_dom1[kTimingRing] = {};
    bool ring_dom0_is_start[kTimingRing] = {};            // the pass recorded one event for both (nothing ran in between)
    bool ring_timed[kTimingRing] = {};                    // the pass in this slot recorded its timing events (not with developer switch 32)
    int64_t pass_seq = 0;                                  // passes enqueued so far
    int64_t copy_waits_pass = 0;                           // pass the copy stream has been ordered behind
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;      // = the ring slot of the current / last pass
    hipEvent_t ev_dom0 = nullptr, ev_dom1 = nullptr;
    hipStream_t side_stream = nullptr;                     // replay of the dominant kernel's hand-overs,
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;        // concurrent with the generic pass
    hipEvent_t ev_order[kOrderRing] = {};                   // nl_stack_order_stream_after
    int order_seq = 0;
    unsigned ev_rel = 0;                                   // creation flag of the pass's events (hipEventDisableSystemFence or 0)
    float *d_frames_owned = nullptr;  // [n_capacity][fstride_owned], the first npix floats of a slot in use
    float *d_frames = nullptr;        // owned or lent
    float *d_out = nullptr;           // [npix]
    float *d_acc = nullptr;           // stack-of-stacks accumulator, lazily allocated
    float *d_weights = nullptr;       // [n_frames]
    bool has_weights = false;
    float *d_xstat = nullptr;         // [(n_frames+1)*2]
    unsigned long long *d_sets = nullptr;      // two scratch sets of kScratchWords; d_partial = the current one
    int cur_set = 0;
    bool sets_clean = false;                   // both sets as a fused pass leaves them: the current one used, the other zeroed
    bool partial_clean = false;                // the current set is all zeros: the last pass's reduction kernel left it so (plain protocol of the sigma fast path)
    unsigned long long *d_partial = nullptr;   // [kClipSlots][2] clip accumulators + 2 words of list lengths
    float2 *d_bounds = nullptr;                // decision pass of weighted stacks: [kBoundRounds][npix] thresholds, lazily allocated
    unsigned char *d_nrounds = nullptr;        // [npix]
    bool bounds_tried = false;
    unsigned fb_hint = 0;                      // exact-list length of the last finished fast pass + 1 (0 = unknown)
    unsigned gen_hint = 0;                     // same for the generic list
    bool last_weighted = false;                // the last pass ran with weights (key of the hints it leaves)
    bool last_fused = false;
    bool last_tail_fused = false;              // generic pass + first replay ran as one launch (stack_tail_fused.hip)
    bool last_lists = false;                   // the last pass left its list lengths behind the totals (d_counters[2])
    unsigned dev_flags = 0;                    // nl_stack_set_dev_flags (A/B measurements)
    unsigned *d_fb_list = nullptr;             // [npix] pixels the fast kernel handed to the exact kernel
    unsigned *d_fb_count = nullptr;            // [2]: exact-list length, generic-list length (inside d_partial)
    unsigned *d_gen_list = nullptr;            // [npix] pixels zonal waves handed to the generic pass
    float *d_cols = nullptr;                   // split LDS-column pass (FastArgs::cols): rows of cols_stride floats, allocated on first use
    size_t cols_bytes = 0;
    int64_t cols_stride = 0;
    bool cols_tried = false;
    bool force_exact = false;
    int exact_flavour = 0;            // nl_stack_set_exact argument: 1 = LDS column kernel, 2 = wave-per-pixel replay
    bool last_used_fast = false;
    unsigned long long *d_counters = nullptr;  // [4]: where a pass leaves {clip_low, clip_high, list lengths, -}: the handle's own buffer or the caller's (nl_stack_set_counters_buffer)
    unsigned long long *d_counters_own = nullptr;
    double *d_stat_partial = nullptr;          // [kStatBlocks*3]
    // linear-fit cascade (stack_linfit.hip): ping-pong pixel lists + liveness masks, lazily allocated
    // (a third list + masks for the guarded stages' hand-overs, stack_linfit_guard.hip: up to 128 frames)
    unsigned *d_lf_list[3] = {nullptr, nullptr, nullptr};
    uint4 *d_lf_state[3] = {nullptr, nullptr, nullptr};
    unsigned *d_lf_count = nullptr;
    int lf_lanes = 0;                          // liveness masks per listed pixel the state arrays were sized for
    int lf_lists = 0;                          // lists allocated (2, or 3 with the guarded stages)
    bool lf_tried = false, lf_no_third = false;
    void *d_ingest = nullptr;                  // raw FITS bytes / unaligned source frame, grown on demand
    size_t ingest_bytes = 0;
    // asynchronous uploads: pinned staging ring + copy stream (nl_stack_upload_frame_async)
    hipStream_t copy_stream = nullptr;
    size_t stage_cap[kStageSlots] = {0, 0, 0, 0};
    void *d_ingest_async = nullptr;            // raw bytes / source frame of the overlapped ingest (copy stream)
    size_t ingest_async_bytes = 0;
    double *d_stat_partial_async = nullptr;
    void *h_stage[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t stage_done[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_used[kStageSlots] = {false, false, false, false};
    int stage_next = 0;
    bool uploads_pending = false;
    // chunked passes (sigma / winsorized fast path, see chunk_plan): the dominant kernel runs as a few launches over
    // consecutive pixel ranges, each range with hand-over lists of its own, and the tail of a range (generic pass,
    // exact replays) runs on two more streams while the next range's dominant kernel has the device
    hipStream_t chunk_stream[2] = {nullptr, nullptr};      // [0]: generic pass + replay of its additions, [1]: replay of the dominant kernel's list
    hipEvent_t ev_chunk[kMaxChunks] = {};                  // behind the dominant kernel of a chunk
    hipEvent_t ev_chunk_join[2] = {nullptr, nullptr};
    unsigned *d_chunk_counts = nullptr;                    // [kMaxChunks][4]: exact-list length, generic-list length, snapshot, spare
    int last_chunks = 0;                                   // chunks of the last pass (0: not chunked)
    int max_grid = 0;
    int last_mode = -1;
    bool last_has_counters = false;
    bool pending = false;
    const char *last_kernel = "";
};

// stats.MeanStdDev over xs = 0..n-1 (stats.go:246-261, called from :570) depends on n only: tabulated once per frame count
// of the process, in the same fp32 operation order ({mean, stddev} for n = 1 .. n_frames at [2n], [2n + 1]).  (The table is
// O(n^2) scalar operations: 0.45 ms of a 512-frame handle's create.)  Returned by value: 8 bytes per frame.
static std::vector<float> xstat_table(int n_frames)
{
    static std::mutex mu;
    static std::vector<std::pair<int, std::vector<float>>> tables;        // (a handful of frame counts per process)
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &t : tables)
        if (t.first == n_frames) return t.second;
    std::vector<float> xstat(2 * (size_t)(n_frames + 1), 0.0f);
    for (int n = 1; n <= n_frames; n++) {
        volatile float s = 0.0f;
        for (int i = 0; i < n; i++) s = s + (float)i;
        const float mean = s / (float)n;
        volatile float v = 0.0f;
        for (int i = 0; i < n; i++) {
            volatile float d = (float)i - mean;
            volatile float dd = d * d;
            v = v + dd;
        }
        const float var = v / (float)n;
        xstat[2 * (size_t)n] = mean;
        xstat[2 * (size_t)n + 1] = (float)sqrt((double)var);
    }
    if (tables.size() >= 64) tables.erase(tables.begin());
    tables.emplace_back(n_frames, xstat);
    return xstat;
}

// Floats between consecutive frames of the owned planar buffer.  A stride that is a multiple of a large power of two
// -- 4096 x 4096 floats = 2^26 bytes, or a 512-row tile's 2^23 -- puts the same pixel of every frame into the same
// HBM channel / bank: the 4 frames one load of the LDS-column kernels reads, and the 128-512 loads a wave has in
// flight, then queue on a few banks while the rest idle.  Measured on sigma 512 x 4096^2 (profiles/r05_stride_pad.txt):
// stride + 0: 10.9 ms, + 16 KiB: 10.6, + 32 KiB: 10.2, + 64 KiB: 9.75, + 64 KiB + 256 B: 9.60, more: no further gain;
// 32 frames are indifferent, 128 frames gain 1-3 %.  So the frame is rounded up to 128 KiB and 64 KiB + 256 B are added:
// the residue of the stride modulo 128 KiB is the measured optimum for every tile size, at a cost of at most 192 KiB
// per frame.  Small tiles (< 1 MiB per frame) are left dense.  NL_STRIDE_PAD=<floats> (developer) sets the padding
// added to the tile's pixel count instead (0 = the dense layout of rounds 1-4).
static bool frame_stride_addressable(int64_t npix, int64_t stride)
{
    // pixel offset + 3 frames in one signed 32-bit byte offset (multi-lane gathers), stride a multiple of 16 bytes
    // where the dense layout is one (the float4 loads of the mean and the replay kernels)
    return (npix + 3 * stride) * 4 < ((int64_t)1 << 31) && (npix % 4 != 0 || stride % 4 == 0);
}

static int64_t padded_frame_stride(int64_t npix)
{
    const char *e = getenv("NL_STRIDE_PAD");                  // read per handle: the tests switch it
    const long env_pad = e ? atol(e) : -1L;
    int64_t stride = npix;
    if (env_pad >= 0) stride = npix + env_pad;
    else if (npix >= (1 << 18)) stride = ((npix + 32767) & ~(int64_t)32767) + 16384 + 64;
    if (stride != npix && npix < nl::kFastMaxPixels && !frame_stride_addressable(npix, stride)) stride = npix;
    return stride;
}


// Grid of a dense replay whose workgroups stride through the pixels (item = workgroup + i * grid): with a grid that
// is a multiple of the image width a workgroup would visit ONE image column throughout, and the few workgroups of the
// alignment borders -- NaN columns, every pixel a full replay -- would run three times as long as the rest with the
// device draining around them (measured: 5 400 of 8 192 waves in flight on average).  A multiple of 8 (the
// XCD-contiguous mapping wants whole sweeps) that shares no large factor with the width walks through the columns
// (at least 64 of them per workgroup).
static int dense_grid(int64_t items, int64_t max_grid, int width, int pixels_per_item)
{
    int64_t g = items < max_grid ? items : max_grid;
    if (g <= 8 || items <= g) return (int)g;                 // no second sweep: nothing to align with
    g &= ~(int64_t)7;
    auto gcd = [](int64_t x, int64_t y) { while (y) { const int64_t t = x % y; x = y; y = t; } return x; };
    const int64_t most = (int64_t)width / 64 > 8 * pixels_per_item ? (int64_t)width / 64 : 8 * pixels_per_item;       // >= 64 columns per workgroup
    for (int tries = 0; tries < 64 && g > 8 && gcd(g * pixels_per_item, (int64_t)width) > most; tries++) g -= 8;
    return (int)g;
}

extern "C" {

const char *nl_last_error(void) { return g_err.c_str(); }

const char *nl_version(void)
{
#ifdef NL_EXPERIMENTS
    return "nlstack 0.2.0 (gfx950) +experiments";
#else
    return "nlstack 0.2.0 (gfx950)";
#endif
}

void nl_release_cached_memory(void) { cache_release_all(); stream_pool_release_all(); pinned_release_all(); }

int nl_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(NL_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

static int destroy_impl(nl_stack_t *h)
{
    if (!h) return NL_OK;
    if (!h->stream) {             // create failed before anything existed on a device (e.g. a bad ordinal):
        delete h;                 // no HIP call, so no stale error is left behind for hipGetLastError
        return NL_OK;
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    for (int i = 0; i < 2; i++)
        if (h->chunk_stream[i]) (void)hipStreamSynchronize(h->chunk_stream[i]);
    // (the large create-time buffers are parked for the next handle of the same geometry, see cached_free)
    cached_free(h->d_frames_owned, (size_t)h->fstride_owned * sizeof(float) * (size_t)h->n_capacity, h->device);
    cached_free(h->d_out, (size_t)h->npix * sizeof(float), h->device);
    if (h->d_acc) (void)hipFree(h->d_acc);
    if (h->d_weights) (void)hipFree(h->d_weights);
    if (h->d_xstat) (void)hipFree(h->d_xstat);
    if (h->d_sets) (void)hipFree(h->d_sets);
    cached_free(h->d_bounds, (size_t)nl::kBoundRounds * (size_t)h->npix * sizeof(float2), h->device);
    cached_free(h->d_nrounds, (size_t)h->npix, h->device);
    cached_free(h->d_fb_list, sizeof(unsigned) * (size_t)h->npix, h->device);
    cached_free(h->d_gen_list, sizeof(unsigned) * (size_t)h->npix, h->device);
    if (h->d_cols) cached_free(h->d_cols, h->cols_bytes, h->device);
    if (h->d_counters_own) (void)hipFree(h->d_counters_own);
    if (h->d_stat_partial) (void)hipFree(h->d_stat_partial);
    if (h->d_ingest) (void)hipFree(h->d_ingest);
    if (h->d_ingest_async) (void)hipFree(h->d_ingest_async);
    if (h->d_stat_partial_async) (void)hipFree(h->d_stat_partial_async);
    for (int i = 0; i < 3; i++) {          // (parked like the create-time buffers: a handle per Apply pays no hipMalloc for them)
        cached_free(h->d_lf_list[i], sizeof(unsigned) * (size_t)h->npix, h->device);
        cached_free(h->d_lf_state[i], sizeof(uint4) * (size_t)h->npix * (size_t)h->lf_lanes, h->device);
    }
    if (h->d_lf_count) (void)hipFree(h->d_lf_count);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    for (int i = 0; i < kStageSlots; i++) {
        pinned_free(h->h_stage[i], h->stage_cap[i]);
        if (h->stage_done[i]) (void)hipEventDestroy(h->stage_done[i]);
    }
    park_copy_stream(h->copy_stream, h->device);
    for (int i = 0; i < kTimingRing; i++) {
        if (h->ring_start[i]) (void)hipEventDestroy(h->ring_start[i]);
        if (h->ring_stop[i]) (void)hipEventDestroy(h->ring_stop[i]);
        if (h->ring_dom0[i]) (void)hipEventDestroy(h->ring_dom0[i]);
        if (h->ring_dom1[i]) (void)hipEventDestroy(h->ring_dom1[i]);
    }
    for (int i = 0; i < 2; i++) {
        if (h->chunk_stream[i]) { (void)hipStreamSynchronize(h->chunk_stream[i]); (void)hipStreamDestroy(h->chunk_stream[i]); }
        if (h->ev_chunk_join[i]) (void)hipEventDestroy(h->ev_chunk_join[i]);
    }
    for (int i = 0; i < kMaxChunks; i++)
        if (h->ev_chunk[i]) (void)hipEventDestroy(h->ev_chunk[i]);
    if (h->d_chunk_counts) (void)hipFree(h->d_chunk_counts);
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (hipEvent_t &ev : h->ev_order) if (ev) (void)hipEventDestroy(ev);
    park_stream_pair(h->stream, h->side_stream, h->device);
    delete h;
    return NL_OK;
}

void nl_stack_destroy(nl_stack_t *h) { destroy_impl(h); }

static int create_impl(nl_stack_t *h)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(NL_ERR_NO_DEVICE, "no HIP device available (%s); libnlstack has no CPU path",
                    hipGetErrorString(e));
    if (h->device < 0 || h->device >= ndev)
        return fail(NL_ERR_INVALID_ARG, "device %d out of range (have %d)", h->device, ndev);
    NL_HIP(hipSetDevice(h->device));
    // (main and side stream as the pair they were created as: different hardware queues, see the stream pool)
    NL_HIP(pooled_stream_pair(&h->stream, &h->side_stream, h->device));
    // (the timing events of a ring slot are created by the first pass that uses it: a handle that lives for ONE
    // Apply -- the cgo drop-in -- creates 4 events instead of 256)
    {
        // Events that only order device work against device work (fork / join of the side stream) or only take times: no
        // system-scope fence when they complete (hipEventDisableSystemFence) -- the writeback / invalidate it stands for costs
        // the next kernel 4 - 9 us per pass (512-row tile: 0.268 -> 0.259 ms, 32 frames 0.106 -> 0.099).  What the HOST reads
        // (counters, results) is ordered by the stream synchronisation of nl_stack_finish, not by these events.
        // NL_EV_FENCE=1 (developer switch): default events, for A/B runs.
        static const unsigned nofence = [] { const char *e = getenv("NL_EV_FENCE"); return e && e[0] == '1' ? 0u : (unsigned)hipEventDisableSystemFence; }();
        NL_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming | nofence));
        NL_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming | nofence));
        h->ev_rel = nofence;
    }
    const size_t frame_bytes = (size_t)h->npix * sizeof(float);
    h->fstride = h->fstride_owned = padded_frame_stride(h->npix);
    NL_HIP(cached_malloc((void **)&h->d_frames_owned, (size_t)h->fstride * sizeof(float) * (size_t)h->n_frames, h->device));
    h->d_frames = h->d_frames_owned;
    NL_HIP(cached_malloc((void **)&h->d_out, frame_bytes, h->device));
    NL_HIP(dev_malloc(&h->d_weights, sizeof(float) * (size_t)h->n_frames));
    h->max_grid = 256 * 64;
    NL_HIP(dev_malloc(&h->d_sets, 2 * kScratchBytes));
    NL_HIP(hipMemsetAsync(h->d_sets, 0, 2 * kScratchBytes, h->stream));
    h->cur_set = 0;
    h->d_partial = h->d_sets;
    h->d_fb_count = reinterpret_cast<unsigned *>(h->d_partial + 2 * nl::kClipSlots);
    if (h->npix < (int64_t)0xFFFFFFFFll) {
        NL_HIP(cached_malloc((void **)&h->d_fb_list, sizeof(unsigned) * (size_t)h->npix, h->device));
        NL_HIP(cached_malloc((void **)&h->d_gen_list, sizeof(unsigned) * (size_t)h->npix, h->device));
    }
    NL_HIP(dev_malloc(&h->d_counters_own, sizeof(unsigned long long) * 4));      // {clip_low, clip_high, list lengths (fused passes), -}
    h->d_counters = h->d_counters_own;
    NL_HIP(hipMemsetAsync(h->d_counters, 0, sizeof(unsigned long long) * 4, h->stream));
    NL_HIP(dev_malloc(&h->d_stat_partial, sizeof(double) * 3 * kStatBlocks));

    const std::vector<float> xstat = xstat_table(h->n_frames);
    NL_HIP(dev_malloc(&h->d_xstat, xstat.size() * sizeof(float)));
    NL_HIP(hipMemcpy(h->d_xstat, xstat.data(), xstat.size() * sizeof(float), hipMemcpyHostToDevice));
    return NL_OK;
}

nl_stack_t *nl_stack_create(int n_frames, int width, int height, int row0, int rows, int device)
{
    if (n_frames <= 0) { fail(NL_ERR_NO_INPUTS, "stack operator needs inputs"); return nullptr; }
    if (width <= 0 || height <= 0 || row0 < 0 || rows <= 0 || row0 + rows > height) {
        fail(NL_ERR_INVALID_ARG, "bad geometry %dx%d rows [%d,%d)", width, height, row0, row0 + rows);
        return nullptr;
    }
    nl_stack_t *h = new nl_stack();
    h->device = device; h->n_frames = h->n_capacity = n_frames; h->width = width; h->height = height;
    h->row0 = row0; h->rows = rows; h->npix = (int64_t)rows * width;
    if (create_impl(h) != NL_OK) {
        std::string keep = g_err;
        destroy_impl(h);
        g_err = keep;
        return nullptr;
    }
    return h;
}

#define NL_CHECK_HANDLE(h)                                              \
    do {                                                                \
        if (!(h)) return fail(NL_ERR_INVALID_ARG, "null handle");       \
        NL_HIP(hipSetDevice((h)->device));                              \
    } while (0)

// entry points that read frames on h->stream outside a stack pass first let pending
// asynchronous uploads land
#define NL_SETTLE_UPLOADS(h)                                            \
    do {                                                                \
        if ((h)->uploads_pending) {                                     \
            NL_HIP(hipStreamSynchronize((h)->copy_stream));             \
            (h)->uploads_pending = false;                               \
        }                                                               \
    } while (0)

int nl_stack_upload_frame(nl_stack_t *h, int idx, const float *host_frame)
{
    NL_CHECK_HANDLE(h);
    if (idx < 0 || idx >= h->n_frames || !host_frame)
        return fail(NL_ERR_INVALID_ARG, "upload_frame: bad index %d or null frame", idx);
    return nl_stack_upload_tile(h, idx, host_frame + (int64_t)h->row0 * h->width);
}

int nl_stack_upload_tile(nl_stack_t *h, int idx, const float *host_tile)
{
    NL_CHECK_HANDLE(h);
    if (idx < 0 || idx >= h->n_frames || !host_tile)
        return fail(NL_ERR_INVALID_ARG, "upload_tile: bad index %d or null tile", idx);
    NL_HIP(hipMemcpyAsync(h->d_frames + (int64_t)idx * h->fstride, host_tile,
                          (size_t)h->npix * sizeof(float), hipMemcpyHostToDevice, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));   // pointer must not be retained (cgo rules)
    return NL_OK;
}

// Overlapped uploads (the caller side of the path).  The
// caller's frame is copied into a pinned staging buffer by a few host threads
// and this call returns (the caller's pointer is not retained, cgo rules); the
// DMA runs on its own stream while the caller prepares the next frame, and the
// next stack pass waits for it on the device, not on the host.
// Overlapped uploads: `bytes` of host memory go into a pinned staging slot (ring of kStageSlots; a slot is
// re-used once its last DMA has left it) and the call returns; the copy stream is ordered behind the last
// pass that may still read the frames.  *staged = the pinned copy, *slot_out = its slot (record stage_done on
// the copy stream after the last operation that reads it).
static int stage_host_bytes(nl_stack_t *h, const void *src_v, size_t bytes, char **staged, int *slot_out)
{
    if (!h->copy_stream) NL_HIP(pooled_copy_stream(&h->copy_stream, h->device));
    if (h->pass_seq > 0 && h->copy_waits_pass != h->pass_seq) {
        // a pass enqueued earlier may still be reading the frames: the copy stream waits for its
        // end on the device (staging batch b+1 while batch b is stacked must not overwrite b)
        NL_HIP(hipStreamWaitEvent(h->copy_stream, h->ev_stop, 0));
        h->copy_waits_pass = h->pass_seq;
    }
    const int slot = h->stage_next;
    h->stage_next = (slot + 1) % kStageSlots;
    if (h->stage_used[slot]) NL_HIP(hipEventSynchronize(h->stage_done[slot]));     // its last DMA has left the buffer
    if (!h->stage_done[slot]) NL_HIP(hipEventCreateWithFlags(&h->stage_done[slot], hipEventDisableTiming));
    if (h->stage_cap[slot] < bytes) {
        if (h->h_stage[slot]) { pinned_free(h->h_stage[slot], h->stage_cap[slot]); h->h_stage[slot] = nullptr; h->stage_cap[slot] = 0; }
        NL_HIP(pinned_malloc(&h->h_stage[slot], bytes, &h->stage_cap[slot]));
    }
    const char *src = static_cast<const char *>(src_v);
    char *dst = static_cast<char *>(h->h_stage[slot]);
    if (bytes < ((size_t)4 << 20)) {
        memcpy(dst, src, bytes);
    } else {
        std::thread workers[kStageThreads - 1];
        const size_t part = (bytes / kStageThreads + 63) & ~(size_t)63;
        for (int t = 1; t < kStageThreads; t++) {
            const size_t b = (size_t)t * part, e = (b + part < bytes) ? b + part : bytes;
            workers[t - 1] = std::thread([=] { if (b < e) memcpy(dst + b, src + b, e - b); });
        }
        memcpy(dst, src, part < bytes ? part : bytes);
        for (auto &w : workers) w.join();
    }
    *staged = dst;
    *slot_out = slot;
    return NL_OK;
}

static int stage_done(nl_stack_t *h, int slot)
{
    NL_HIP(hipEventRecord(h->stage_done[slot], h->copy_stream));
    h->stage_used[slot] = true;
    h->uploads_pending = true;
    return NL_OK;
}

int nl_stack_upload_frame_async(nl_stack_t *h, int idx, const float *host_frame)
{
    NL_CHECK_HANDLE(h);
    if (idx < 0 || idx >= h->n_frames || !host_frame)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_async: bad index %d or null frame", idx);
    if (h->d_frames != h->d_frames_owned)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_async: frames are attached, not owned");
    const size_t bytes = (size_t)h->npix * sizeof(float);
    char *dst = nullptr;
    int slot = 0;
    int rc = stage_host_bytes(h, host_frame + (int64_t)h->row0 * h->width, bytes, &dst, &slot);
    if (rc != NL_OK) return rc;
    // (fp32 frames keep the copy engine: a kernel that pulls the frame out of the pinned buffer itself, as the FITS decode
    // below does, measured 41.6 against 43.3 GiB/s on the same box -- profiles/r06_apply_from_host.txt)
    NL_HIP(hipMemcpyAsync(h->d_frames + (int64_t)idx * h->fstride, dst, bytes, hipMemcpyHostToDevice, h->copy_stream));
    return stage_done(h, slot);
}

// Host-side wait for every asynchronous upload issued so far.
int nl_stack_upload_wait(nl_stack_t *h)
{
    NL_CHECK_HANDLE(h);
    if (h->copy_stream) NL_HIP(hipStreamSynchronize(h->copy_stream));
    h->uploads_pending = false;
    return NL_OK;
}

int nl_stack_download_tile(nl_stack_t *h, int idx, float *host_tile)
{
    NL_CHECK_HANDLE(h);
    NL_SETTLE_UPLOADS(h);
    if (idx < 0 || idx >= h->n_frames || !host_tile)
        return fail(NL_ERR_INVALID_ARG, "download_tile: bad index %d or null tile", idx);
    NL_HIP(hipMemcpyAsync(host_tile, h->d_frames + (int64_t)idx * h->fstride,
                          (size_t)h->npix * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

int nl_stack_download_rows(nl_stack_t *h, int idx, int first_row, int n_rows, float *host_rows)
{
    NL_CHECK_HANDLE(h);
    NL_SETTLE_UPLOADS(h);
    if (idx < -1 || idx >= h->n_frames || !host_rows || first_row < 0 || n_rows <= 0 ||
        (int64_t)first_row + n_rows > h->rows)
        return fail(NL_ERR_INVALID_ARG, "download_rows: bad index %d, rows [%d,%d) of %d, or null buffer",
                    idx, first_row, first_row + n_rows, h->rows);
    const float *src = (idx < 0 ? h->d_out : h->d_frames + (int64_t)idx * h->fstride) + (int64_t)first_row * h->width;
    NL_HIP(hipMemcpyAsync(host_rows, src, (size_t)n_rows * h->width * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

// Device memory the handle holds right now: the buffers of nl_stack_create plus everything a pass, an upload path or
// the stack of stacks has allocated since (decision-pass thresholds, linear-fit cascade lists, accumulator, ingest
// staging) -- those stay until nl_stack_destroy, so a caller that sizes batches to the device (OpStackBatches,
// stackbatches.go:121-187 does it for host memory) can see what is left.
int64_t nl_stack_device_bytes(nl_stack_t *h)
{
    if (!h) return 0;
    const int64_t np = h->npix;
    int64_t b = 0;
    if (h->d_frames_owned) b += h->fstride_owned * 4 * h->n_capacity;
    if (h->d_out) b += np * 4;
    if (h->d_acc) b += np * 4;
    if (h->d_weights) b += 4 * (int64_t)h->n_capacity;
    if (h->d_xstat) b += 8 * (int64_t)(h->n_capacity + 1);
    if (h->d_sets) b += 2 * (int64_t)kScratchBytes;
    if (h->d_bounds) b += (int64_t)nl::kBoundRounds * np * 8;
    if (h->d_nrounds) b += np;
    if (h->d_fb_list) b += np * 4;
    if (h->d_gen_list) b += np * 4;
    if (h->d_cols) b += (int64_t)h->cols_bytes;
    if (h->d_counters) b += 32;
    if (h->d_stat_partial) b += 8 * 3 * kStatBlocks;
    if (h->d_stat_partial_async) b += 8 * 3 * kStatBlocks;
    const int lanes = h->n_capacity <= 128 ? 1 : h->n_capacity <= 256 ? 2 : 4;
    for (int i = 0; i < 3; i++) {
        if (h->d_lf_list[i]) b += np * 4;
        if (h->d_lf_state[i]) b += np * 16 * lanes;
    }
    if (h->d_lf_count) b += 4 * nl::kLinfitCounters;
    if (h->d_chunk_counts) b += 16 * kMaxChunks;
    b += (int64_t)h->ingest_bytes + (int64_t)h->ingest_async_bytes;
    return b;
}

void *nl_stack_frames_device_ptr(nl_stack_t *h) { return h ? h->d_frames : nullptr; }
void *nl_stack_result_device_ptr(nl_stack_t *h) { return h ? h->d_out : nullptr; }
int nl_stack_last_mode(nl_stack_t *h) { return h ? h->last_mode : -1; }
const char *nl_stack_last_kernel_name(nl_stack_t *h) { return h ? h->last_kernel : ""; }

int nl_stack_attach_device_frames_strided(nl_stack_t *h, void *device_frames, int64_t frame_stride)
{
    NL_CHECK_HANDLE(h);
    if (!device_frames) {
        h->d_frames = h->d_frames_owned;
        h->fstride = h->fstride_owned;
        return NL_OK;
    }
    // the kernels address 4 frames + a pixel with one signed 32-bit byte offset (fast_common.hpp gather_sorted,
    // fast_ml_common.hpp ml_gather_raw) and load 16 bytes per lane where the stride allows; the owned buffer's stride
    // is chosen inside these limits, a lent one is checked
    if (frame_stride < h->npix)
        return fail(NL_ERR_INVALID_ARG, "attach_device_frames: stride %lld < %lld pixels of the tile",
                    (long long)frame_stride, (long long)h->npix);
    if (h->npix < nl::kFastMaxPixels && !frame_stride_addressable(h->npix, frame_stride))
        return fail(NL_ERR_INVALID_ARG, "attach_device_frames: stride %lld too large for the tile's 32-bit frame offsets",
                    (long long)frame_stride);
    h->d_frames = static_cast<float *>(device_frames);
    h->fstride = frame_stride;
    return NL_OK;
}

int nl_stack_attach_device_frames(nl_stack_t *h, void *device_frames)
{
    NL_CHECK_HANDLE(h);
    return nl_stack_attach_device_frames_strided(h, device_frames, h->npix);
}

int64_t nl_stack_frame_stride(nl_stack_t *h)
{
    return h ? h->fstride : 0;
}

int nl_stack_fill_synthetic(nl_stack_t *h, uint64_t seed)
{
    NL_CHECK_HANDLE(h);
    NL_HIP(nl::launch_fill_synthetic(h->d_frames, h->fstride, h->n_frames, h->width, h->height,
                                     h->row0, h->rows, seed, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

// Frames in use for the next passes: slots [0, n) of the n_capacity allocated at create.  The
// caller of a batch loop (OpStackBatches, stackbatches.go:69-111) keeps ONE handle -- buffers,
// accumulator -- across batches whose last one is smaller.
int nl_stack_set_active_frames(nl_stack_t *h, int n)
{
    NL_CHECK_HANDLE(h);
    if (n < 1 || n > h->n_capacity)
        return fail(NL_ERR_INVALID_ARG, "set_active_frames: %d not in [1, %d]", n, h->n_capacity);
    if (n != h->n_frames) h->has_weights = false;      // weights are per frame of a given batch
    h->n_frames = n;
    return NL_OK;
}

int nl_stack_set_weights(nl_stack_t *h, const float *weights)
{
    NL_CHECK_HANDLE(h);
    if (!weights) { h->has_weights = false; return NL_OK; }
    NL_HIP(hipMemcpyAsync(h->d_weights, weights, sizeof(float) * (size_t)h->n_frames,
                          hipMemcpyHostToDevice, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    h->has_weights = true;
    return NL_OK;
}

// getWeights, stack.go:231-270 (scalar host arithmetic, fp32, same order)
int nl_weights_from_scalars(int weighting, const float *per_frame, int n_frames,
                            float *weights_out, int *bad_index)
{
    if (bad_index) *bad_index = -1;
    if (weighting == NL_WEIGHT_NONE) return NL_OK;
    if (!per_frame || !weights_out || n_frames <= 0)
        return fail(NL_ERR_INVALID_ARG, "weights_from_scalars: null argument");
    if (weighting == NL_WEIGHT_EXPOSURE) {
        for (int i = 0; i < n_frames; i++) {
            if (per_frame[i] == 0) {
                if (bad_index) *bad_index = i;
                return fail(NL_ERR_MISSING_EXPOSURE,
                            "%d: Missing exposure information for exposure-weighted stacking", i);
            }
            weights_out[i] = per_frame[i];
        }
        return NL_OK;
    }
    if (weighting == NL_WEIGHT_INVERSE_NOISE || weighting == NL_WEIGHT_INVERSE_HFR) {
        float mn = 3.40282346638528859811704183484516925440e+38f, mx = -mn;
        for (int i = 0; i < n_frames; i++) {
            const float v = per_frame[i];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        const float range = mx - mn;
        for (int i = 0; i < n_frames; i++) {
            volatile float num = per_frame[i] - mn;
            num = 4.0f * num;
            volatile float q = num / range;
            volatile float den = 1.0f + q;
            weights_out[i] = 1.0f / den;
        }
        return NL_OK;
    }
    return fail(NL_ERR_INVALID_WEIGHTING, "Invalid weighting mode %d\n", weighting);
}

// Linear-fit cascade buffers (stack_linfit.hip, stack_linfit_guard.hip): pixel lists and state arrays with
// lanes_per_pixel liveness masks (16 B) per pixel, allocated on first use -- two for the bit-exact cascade, a third
// for the guarded stages' hand-overs (up to 128 frames).  Without them (allocation failure) the kernels run as a
// single bit-exact stage.
static bool linfit_buffers(nl_stack_t *h, int lists = 2)
{
    if (!h->lf_tried) {
        // sized for the most lanes per pixel any active frame count of this handle can need
        h->lf_lanes = h->n_capacity <= 128 ? 1 : h->n_capacity <= 256 ? 2 : 4;
        h->lf_tried = true;
        if (dev_malloc(&h->d_lf_count, sizeof(unsigned) * nl::kLinfitCounters) != hipSuccess) {
            (void)hipGetLastError();
            h->d_lf_count = nullptr;
        }
    }
    if (!h->d_lf_count) return false;
    const size_t np = (size_t)h->npix;
    if (lists > 2 && h->lf_no_third) return false;
    while (h->lf_lists < lists) {                     // (the third list only when a guarded linear fit asks for it)
        const int i = h->lf_lists;
        if (cached_malloc((void **)&h->d_lf_list[i], sizeof(unsigned) * np, h->device) != hipSuccess ||
            cached_malloc((void **)&h->d_lf_state[i], sizeof(uint4) * np * (size_t)h->lf_lanes, h->device) != hipSuccess) {
            (void)hipGetLastError();
            if (h->d_lf_list[i]) { (void)hipFree(h->d_lf_list[i]); h->d_lf_list[i] = nullptr; }
            h->d_lf_state[i] = nullptr;
            if (i < 2) {                               // no cascade at all on this handle
                (void)hipFree(h->d_lf_count);
                h->d_lf_count = nullptr;
            } else {
                h->lf_no_third = true;                 // the bit-exact cascade keeps its two lists
            }
            return false;
        }
        h->lf_lists++;
    }
    return true;
}

static const nl::LinfitCascade *linfit_cascade(nl_stack_t *h, int /*lanes_per_pixel*/, nl::LinfitCascade *out)
{
    if (!linfit_buffers(h, 2)) return nullptr;
    out->list[0] = h->d_lf_list[0]; out->list[1] = h->d_lf_list[1];
    out->state[0] = h->d_lf_state[0]; out->state[1] = h->d_lf_state[1];
    out->count = h->d_lf_count;
    out->capacity = (unsigned)h->npix;
    return out;
}

#ifdef NL_EXPERIMENTS
// the guarded stages' buffers (three lists); false: run the bit-exact cascade alone
static bool linfit_guard_bufs(nl_stack_t *h, nl::LinfitGuardBufs *out)
{
    static const bool on = [] { const char *e = getenv("NL_LFG"); return e && e[0] == '1'; }();
    if (!on || (h->dev_flags & 4096u)) return false;                  // developer switch 4096: bit-exact cascade only (A/B)
    if (h->n_capacity > 128 || !linfit_buffers(h, 3)) return false;
    for (int i = 0; i < 3; i++) { out->list[i] = h->d_lf_list[i]; out->state[i] = h->d_lf_state[i]; }
    out->count = h->d_lf_count;
    out->capacity = (unsigned)h->npix;
    return true;
}
#endif

// Weighted sigma / winsorized stacks of 33 ... 512 frames run a decision pass in front of the bit-exact replay
// (33 ... 128 frames: stack_fast_decide.hip, 129 ... 512: the LDS-column kernel of the class, record-only), and
// unweighted winsorized passes above 128 frames put their decided rounds on record for the list replay: scratch for the
// thresholds, kBoundRounds * 8 + 1 bytes per pixel of the tile (1.1 GB for 4096^2), allocated by the first pass that
// wants it and held until the handle is destroyed; nl_stack_device_bytes() reports what a handle holds at any time.  false: off (NL_WDECIDE=0, developer switch 4, allocation failed: those passes then run without it).
// The split pass of the selected LDS-column kernel (stack_fast_mlz_impl.hpp, FastArgs::cols): 352 bytes per pixel for the
// columns between the sorting kernel and the rounds kernel.  OFF by default -- measured slower than the one-kernel pass
// (sigma 512 x 4096^2 10.72 against 10.20 ms); NL_MLZ_SPLIT=1 or developer switch 1024 turn it on.
static void set_split_cols(nl_stack *h, int mode, int n_frames, nl::FastArgs &f)
{
#ifndef NL_EXPERIMENTS
    (void)h; (void)mode; (void)n_frames; (void)f;          // (the default library carries the one-kernel pass only)
#else
    const int rows = nl::mlz_split_rows(mode, n_frames);
    static const bool on = [] { const char *e = getenv("NL_MLZ_SPLIT"); return e && e[0] == '1'; }();
    // (the same class as persistent workgroups -- three per CU, no barrier, the rounds of a block behind the sorting of the
    // next: also built, also slower; NL_MLZ_PERSIST=1 / developer switch 2048)
    static const bool persist = [] { const char *e = getenv("NL_MLZ_PERSIST"); return e && e[0] == '1'; }();
    if (rows != 0 && (persist || (h->dev_flags & 2048u)) && !(h->dev_flags & 1024u) && f.fb_count) {
        f.persistent = 1;
        f.ticket = f.fb_count + 3;                     // (fourth word of the pass's list counters: zeroed with them)
    }
    if (rows == 0 || !(on || (h->dev_flags & 1024u))) return;
    if (!h->d_cols && !h->cols_tried) {
        h->cols_tried = true;
        h->cols_stride = (h->npix + 63) / 64 * 64;
        h->cols_bytes = (size_t)rows * (size_t)h->cols_stride * sizeof(float);
        if (cached_malloc((void **)&h->d_cols, h->cols_bytes, h->device) != hipSuccess) {
            (void)hipGetLastError();
            h->d_cols = nullptr;
        }
    }
    if (!h->d_cols) return;
    f.cols = h->d_cols;
    f.cols_stride = h->cols_stride;
#endif
}

static bool ensure_bounds(nl_stack *h)
{
    static const bool on = [] { const char *e = getenv("NL_WDECIDE"); return !(e && e[0] == '0'); }();
    if (!on || (h->dev_flags & 4u)) return false;
    if (h->d_bounds) return true;
    if (h->bounds_tried) return false;
    h->bounds_tried = true;
    // (through the cache: a handle per Apply of a weighted stack pays no hipMalloc / hipFree of 1.1 GB at 4096^2)
    if (cached_malloc((void **)&h->d_bounds, (size_t)nl::kBoundRounds * (size_t)h->npix * sizeof(float2), h->device) != hipSuccess ||
        cached_malloc((void **)&h->d_nrounds, (size_t)h->npix, h->device) != hipSuccess) {
        (void)hipGetLastError();
        if (h->d_bounds) { (void)hipFree(h->d_bounds); h->d_bounds = nullptr; }
        h->d_nrounds = nullptr;
        return false;
    }
    return true;
}

// Chunked passes (developer switch NL_CHUNKS; off by default, see chunk_plan).  The tail of a sigma / winsorized fast pass -- generic pass, bit-exact replay of the undecidable
// pixels -- is latency- and gather-bound and depends on the dominant kernel, which is bound by VALU issue: run one after
// the other they leave each other's resource idle (sigma 512 x 4096^2: 10.6 ms + 0.76 ms).  A chunked pass launches the
// dominant kernel over a few consecutive pixel ranges; every range has hand-over lists of its own, and its tail runs
// on two more streams while the next range's dominant kernel has the device.  Only the tail of the LAST range is
// exposed, so the ranges shrink towards the end.  The plan: per cent of the tile per range (the last takes the rest).
struct ChunkPlan {
    int n = 0;
    int64_t off[kMaxChunks], len[kMaxChunks];
};

static void chunk_plan(const nl_stack *h, int mode, bool weighted, int n_frames, ChunkPlan *plan)
{
    plan->n = 0;
#ifdef NL_EXPERIMENTS
    // NL_CHUNKS (developer switch): "0" = never, "p1,p2,..." = these ranges whenever the fast path runs
    static const std::vector<double> env_plan = [] {
        std::vector<double> v;
        const char *e = getenv("NL_CHUNKS");
        if (!e) return v;
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const double x = strtod(p, &end);
            if (end == p) break;
            v.push_back(x);
            p = (*end == ',') ? end + 1 : end;
            if (*end != ',') break;
        }
        if (v.empty()) v.push_back(0.0);
        return v;
    }();
    if (h->dev_flags & 64u) return;                    // developer switch 64: no chunks (A/B inside one process)
    if (!(mode == NL_ST_SIGMA || mode == NL_ST_WINSOR_SIGMA) || weighted) return;
    // OFF unless NL_CHUNKS asks for it: measured (round 4), the overlap LOSES -- a replay wave
    // (64 registers, latency-bound, resident for ~100 us) takes the slot of a dominant-kernel wave (168 registers,
    // three per SIMD), and the VALU-bound kernel slows down by more than the tail it hides: sigma 512 x 4096^2
    // 11.47 ms unchunked, 11.97 ms chunked at default stream priority, 13.6 - 14.3 ms with high-priority tails.
    (void)n_frames;
    if (env_plan.empty() || (env_plan.size() == 1 && env_plan[0] <= 0.0)) return;
    const double *pc = env_plan.data();
    int n = (int)env_plan.size();
    if (n < 2) return;
    if (n > kMaxChunks) n = kMaxChunks;
    int64_t at = 0;
    for (int k = 0; k < n && at < h->npix; k++) {
        int64_t len = (int64_t)((double)h->npix * pc[k] / 100.0);
        len = (len + 1023) & ~(int64_t)1023;           // whole workgroups of every dominant kernel, aligned loads
        if (len <= 0) continue;
        if (k == n - 1 || at + len > h->npix) len = h->npix - at;
        plan->off[plan->n] = at;
        plan->len[plan->n] = len;
        plan->n++;
        at += len;
    }
    if (plan->n > 0 && at < h->npix) plan->len[plan->n - 1] += h->npix - at;
    if (plan->n < 2) plan->n = 0;
#else
    (void)h; (void)mode; (void)weighted; (void)n_frames;      // (the default library has no chunked passes: measured slower)
#endif
}

#ifdef NL_EXPERIMENTS
static int ensure_chunk_resources(nl_stack *h)
{
    if (h->d_chunk_counts) return NL_OK;
    // the tails get the wave slots the dominant kernel's retiring workgroups free BEFORE its own next workgroups do
    // (NL_CHUNK_PRIO=0: default priority, for A/B runs)
    static const bool prio = [] { const char *e = getenv("NL_CHUNK_PRIO"); return !(e && e[0] == '0'); }();
    int least = 0, greatest = 0;
    NL_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int i = 0; i < 2; i++) {
        NL_HIP(hipStreamCreateWithPriority(&h->chunk_stream[i], hipStreamNonBlocking, prio ? greatest : 0));
        NL_HIP(hipEventCreateWithFlags(&h->ev_chunk_join[i], hipEventDisableTiming));
    }
    for (int i = 0; i < kMaxChunks; i++) NL_HIP(hipEventCreateWithFlags(&h->ev_chunk[i], hipEventDisableTiming));
    NL_HIP(dev_malloc(&h->d_chunk_counts, sizeof(unsigned) * 4 * kMaxChunks));
    return NL_OK;
}
#endif  // NL_EXPERIMENTS

static int auto_select_mode(int l)   // stack.go:45-55
{
    if (l >= 25) return NL_ST_LINEAR_FIT;
    if (l >= 15) return NL_ST_WINSOR_SIGMA;
    if (l >= 6) return NL_ST_SIGMA;
    return NL_ST_MEAN;
}

static int run_async_impl(nl_stack_t *h, int mode, float sigma_low, float sigma_high, float ref_loc);

// A pass that fails half-way (a launch or an event call after the first kernel) must not hand control back with work in
// flight on the handle's streams and its bookkeeping half-updated: whatever was enqueued is waited for, the scratch
// sets count as dirty, no list lengths or hints are taken from the broken pass.  The error of the failing call is kept.
int nl_stack_run_async(nl_stack_t *h, int mode, float sigma_low, float sigma_high, float ref_loc)
{
    const int rc = run_async_impl(h, mode, sigma_low, sigma_high, ref_loc);
    if (rc != NL_OK && h && h->stream) {
        const std::string keep = g_err;
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
        for (int i = 0; i < 2; i++)
            if (h->chunk_stream[i]) (void)hipStreamSynchronize(h->chunk_stream[i]);
        (void)hipGetLastError();
        h->sets_clean = false;
        h->partial_clean = false;
        h->last_lists = false;
        h->last_fused = false;
        h->last_has_counters = false;
        h->last_used_fast = false;
        h->last_chunks = 0;
        h->pending = false;
        g_err = keep;
    }
    return rc;
}

static int run_async_impl(nl_stack_t *h, int mode, float sigma_low, float sigma_high, float ref_loc)
{
    NL_CHECK_HANDLE(h);
    if (mode < NL_ST_MEDIAN || mode > NL_ST_AUTO) return fail(NL_ERR_INVALID_MODE, "invalid stacking mode");
    if (mode == NL_ST_AUTO) mode = auto_select_mode(h->n_frames);
    bool weighted = h->has_weights;
    if (mode == NL_ST_MAD_SIGMA && weighted)
        return fail(NL_ERR_WEIGHTED_MAD, "MADSigma stacking with weights is still unimplemented");
    if (mode == NL_ST_LINEAR_FIT || mode == NL_ST_MEDIAN) weighted = false;  // stack.go:158,188-189

    if (h->uploads_pending) {
        // asynchronous uploads: the pass waits for the last DMA on the device
        const int last = (h->stage_next + kStageSlots - 1) % kStageSlots;
        NL_HIP(hipStreamWaitEvent(h->stream, h->stage_done[last], 0));
        h->uploads_pending = false;
    }

    nl::StackArgs a;
    a.frames = h->d_frames;
    a.stride = h->fstride;
    a.npix = h->npix;
    a.n_frames = h->n_frames;
    a.n_pad = next_pow2(h->n_frames);
    a.weights = weighted ? h->d_weights : nullptr;
    a.xstat = h->d_xstat;
    a.sig_lo = sigma_low; a.sig_hi = sigma_high; a.ref_loc = ref_loc;
    a.out = h->d_out;
    a.partial = h->d_partial;
    a.tiles = 0;
    a.list = nullptr;
    a.list_count = nullptr;
    a.list_capacity = 0;
    a.list_snap = nullptr;
    a.list_part = 0;
    a.final = nullptr;
    a.zero_next = nullptr;
    a.bounds = nullptr;
    a.nrounds = nullptr;

    {
        const int slot = (int)(h->pass_seq % kTimingRing);
        if (!h->ring_start[slot]) {
            NL_HIP(hipEventCreateWithFlags(&h->ring_start[slot], hipEventDefault | h->ev_rel));
            NL_HIP(hipEventCreateWithFlags(&h->ring_stop[slot], hipEventDefault | h->ev_rel));
            NL_HIP(hipEventCreateWithFlags(&h->ring_dom0[slot], hipEventDefault | h->ev_rel));
            NL_HIP(hipEventCreateWithFlags(&h->ring_dom1[slot], hipEventDefault | h->ev_rel));
        }
        h->ev_start = h->ring_start[slot]; h->ev_stop = h->ring_stop[slot];
        h->ev_dom0 = h->ring_dom0[slot]; h->ev_dom1 = h->ring_dom1[slot];
    }
    const bool timed = !(h->dev_flags & 32u);         // developer switch 32: a pass without its timing events
    h->ring_timed[h->pass_seq % kTimingRing] = timed;
    if (timed) NL_HIP(hipEventRecord(h->ev_start, h->stream));
    // The sigma / winsorized fast path from 17 frames on (a zonal kernel followed by a generic pass) runs the
    // FUSED protocol (StackArgs::final): no memset in front of the pass -- the previous fused pass's dominant
    // kernel zeroed this pass's scratch set, the two sets alternate -- and no reduction kernel behind it.
    // NL_FUSED=0 (developer switch) keeps memset + reduce_counters_kernel for A/B runs.
    static const bool fused_on = [] {
        for (const char *name : {"NL_FUSED", "NL_MLG", "NL_MLZ"}) { const char *e = getenv(name); if (e && e[0] == '0') return false; }
        return true;
    }();
    const bool sigma_fast = !h->force_exact && h->d_fb_list && (mode == NL_ST_SIGMA || mode == NL_ST_WINSOR_SIGMA) &&
                            (nl::fast_supported(mode, weighted, a.n_frames, a.npix) ||
                             nl::fast_ml_supported(mode, weighted, a.n_frames, a.npix));
    // (only while the exact list is short -- the length the last finished pass reported: its replays add their
    // counts to ONE word, and thousands of workgroups doing that take longer than a reduction kernel)
    if (sigma_fast && h->fb_hint == 0 && !(h->dev_flags & 512u)) {          // (developer switch 512: no hints from other handles)
        unsigned fb = 0, gen = 0;
        if (hints_load({a.n_frames, a.npix, mode, weighted}, &fb, &gen)) { h->fb_hint = fb; h->gen_hint = gen; }
    }
    h->last_weighted = weighted;
    ChunkPlan plan;
    if (sigma_fast && a.n_frames > 16 && nl::coop_supported(mode, weighted, a.n_frames) != 0) chunk_plan(h, mode, weighted, a.n_frames, &plan);
    const bool chunked = plan.n > 1;
    const bool fused = fused_on && !chunked && !(h->dev_flags & 1u) && sigma_fast && a.n_frames > 8 && h->fb_hint != 0 &&
                       h->fb_hint - 1u < kFusedMaxList && nl::coop_supported(mode, weighted, a.n_frames) != 0;
    // Every event recorded on the pass's stream costs a few microseconds of it (three of them: 17 us of a 277 us pass on
    // a 512-row tile, tools/wall_probe.py): a fused pass that finds its scratch set clean has nothing between "start" and
    // "dominant kernel starts", and the event behind the dominant kernel is also the fork of the side stream.
    const bool one_start = timed && fused && h->sets_clean;
    h->ring_dom0_is_start[h->pass_seq % kTimingRing] = one_start;
    if (one_start) h->ev_dom0 = h->ev_start;
    if (fused) {
        if (h->sets_clean) h->cur_set ^= 1;
        else NL_HIP(hipMemsetAsync(h->d_sets, 0, 2 * kScratchBytes, h->stream));
        h->d_partial = h->d_sets + (size_t)h->cur_set * nl::kScratchWords;
        h->d_fb_count = reinterpret_cast<unsigned *>(h->d_partial + 2 * nl::kClipSlots);
        a.partial = h->d_partial;
        a.final = h->d_counters;
        a.zero_next = h->d_sets + (size_t)(h->cur_set ^ 1) * nl::kScratchWords;
    } else if (!h->partial_clean) {
        NL_HIP(hipMemsetAsync(h->d_partial, 0, kScratchBytes, h->stream));
    }
    h->sets_clean = false;                       // until this pass is enqueued completely
    bool zeroed_behind = false;
    const bool keep_clean = h->partial_clean && mode == NL_ST_MEAN;     // (a mean pass does not touch the scratch set)
    h->partial_clean = false;
    if (timed && !one_start) NL_HIP(hipEventRecord(h->ev_dom0, h->stream));
    if (mode == NL_ST_MEAN) {
        NL_HIP(nl::launch_stack_mean(weighted, a, h->stream, &h->last_kernel));
        NL_HIP(hipEventRecord(h->ev_dom1, h->stream));
        h->last_has_counters = false;
    } else if (!h->force_exact && mode == NL_ST_MEDIAN && nl::fast_supported(mode, weighted, a.n_frames, a.npix)) {
        // register-resident sorting network, bit-exact; pixels with many missing samples are
        // handed from the pruned-network kernel to the full-sort one
        nl::FastArgs f;
        memset(&f, 0, sizeof f);
        f.gen_list = h->d_gen_list;                 // nullptr for huge tiles: full sort everywhere
        f.gen_count = h->d_fb_count + 1;
        f.gen_capacity = (unsigned)h->npix;
        NL_HIP(nl::launch_stack_median_fast(a, f, h->stream, &h->last_kernel, h->ev_dom1));
        h->last_has_counters = false;
        h->last_used_fast = false;
    } else if (!h->force_exact && mode == NL_ST_MEDIAN && nl::fast_ml_supported(mode, weighted, a.n_frames, a.npix)) {
        // 129..512 frames: 2 or 4 lanes per pixel, bit-exact
        NL_HIP(nl::launch_stack_median_ml(a, h->stream, &h->last_kernel));
        NL_HIP(hipEventRecord(h->ev_dom1, h->stream));
        h->last_has_counters = false;
        h->last_used_fast = false;
    } else if (!h->force_exact && h->d_fb_list &&
               (nl::mad_fast_supported(mode, weighted, a.n_frames, a.npix) ||
                (mode == NL_ST_MAD_SIGMA && nl::fast_ml_supported(mode, weighted, a.n_frames, a.npix)))) {
        // register-resident MAD clipping: counters exact (the bounds come from two medians);
        // pixels with a non-finite median are replayed by the LDS kernel
        nl::FastArgs f;
        memset(&f, 0, sizeof f);
        f.fb_list = h->d_fb_list;
        f.fb_count = h->d_fb_count;
        f.fb_capacity = (unsigned)h->npix;
        f.gen_list = h->d_gen_list;                 // 128 frames: pixels with too few samples for the selection kernel
        f.gen_count = h->d_fb_count + 1;
        f.gen_capacity = (unsigned)h->npix;
        if (a.n_frames <= 128) NL_HIP(nl::launch_stack_mad_fast(a, f, h->stream, &h->last_kernel));
        else                   NL_HIP(nl::launch_stack_mad_ml(a, f, h->stream, &h->last_kernel));
        NL_HIP(hipEventRecord(h->ev_dom1, h->stream));
        {
            int lanes = 0;
            size_t lds = 0;
            if (nl::exact_plan(mode, weighted, a.n_frames, a.n_pad, kListLanes, &lanes, &lds) != 0)
                return fail(NL_ERR_TOO_MANY_FRAMES,
                            "%d frames do not fit the per-pixel LDS column (mode %d)", a.n_frames, mode);
            nl::StackArgs e = a;
            e.list = h->d_fb_list;
            e.list_count = h->d_fb_count;
            e.list_capacity = (unsigned)h->npix;
            const char *exact_name = "";
            NL_HIP(nl::launch_stack_exact(mode, weighted, e, lanes, kListGrid, lds, h->stream, &exact_name));
        }
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream));
        h->last_has_counters = true;
        h->last_used_fast = true;
    } else if (!h->force_exact && h->d_fb_list && nl::linfit_ml_supported(mode, a.n_frames, a.npix)) {
        // 129..512 frames: 2 or 4 lanes per pixel, the sums chained through the lanes; bit-exact
        nl::FastArgs f;
        memset(&f, 0, sizeof f);
        f.fb_list = h->d_fb_list;
        f.fb_count = h->d_fb_count;
        f.fb_capacity = (unsigned)h->npix;
        nl::LinfitCascade cascade;
        const nl::LinfitCascade *cas = linfit_cascade(h, a.n_frames <= 256 ? 2 : 4, &cascade);
        if (cas) NL_HIP(hipMemsetAsync(h->d_lf_count, 0, sizeof(unsigned) * nl::kLinfitCounters, h->stream));
        NL_HIP(nl::launch_stack_linfit_ml(a, f, cas, h->stream, &h->last_kernel, h->ev_dom1));
        int lanes = 0;
        size_t lds = 0;
        if (nl::exact_plan(mode, weighted, a.n_frames, a.n_pad, kListLanes, &lanes, &lds) != 0)
            return fail(NL_ERR_TOO_MANY_FRAMES,
                        "%d frames do not fit the per-pixel LDS column (mode %d)", a.n_frames, mode);
        nl::StackArgs e = a;
        e.list = h->d_fb_list;
        e.list_count = h->d_fb_count;
        e.list_capacity = (unsigned)h->npix;
        const char *exact_name = "";
        NL_HIP(nl::launch_stack_exact(mode, weighted, e, lanes, kListGrid, lds, h->stream, &exact_name));
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream));
        h->last_has_counters = true;
        h->last_used_fast = true;
    } else if (!h->force_exact && h->d_fb_list && nl::linfit_fast_supported(mode, a.n_frames, a.npix)) {
        // register-resident linear fit: bit-exact (sums run in sorted order);
        // only pixels with an infinite sample are replayed by the LDS kernel
        nl::FastArgs f;
        memset(&f, 0, sizeof f);
        f.fb_list = h->d_fb_list;
        f.fb_count = h->d_fb_count;
        f.fb_capacity = (unsigned)h->npix;
#ifdef NL_EXPERIMENTS
        // Guarded stages in front of the bit-exact cascade (stack_linfit_guard.hip; round 5): exact ymean, enclosed slope /
        // sigma, undecidable pixels continue bit-exactly from their state.  Parity-green, decides 94 % of the pixels --
        // and issues as many vector instructions as the cascade it replaces (8.32 against 8.37 * 10^9
        // for the first stage, pass 18.9 against 17.0 ms): experiments build only, NL_LFG=1.
        nl::LinfitGuardBufs gb;
        if (nl::linfit_guard_supported(mode, a.n_frames, a.npix) && linfit_guard_bufs(h, &gb)) {
            NL_HIP(hipMemsetAsync(h->d_lf_count, 0, sizeof(unsigned) * nl::kLinfitCounters, h->stream));
            NL_HIP(nl::launch_stack_linfit_guarded(a, f, gb, h->stream, &h->last_kernel, h->ev_dom1));
        } else
#endif
        {
            nl::LinfitCascade cascade;
            const nl::LinfitCascade *cas = linfit_cascade(h, 1, &cascade);
            if (cas) NL_HIP(hipMemsetAsync(h->d_lf_count, 0, sizeof(unsigned) * nl::kLinfitCounters, h->stream));
            NL_HIP(nl::launch_stack_linfit_fast(a, f, cas, h->stream, &h->last_kernel, h->ev_dom1));
        }
        int lanes = 0;
        size_t lds = 0;
        if (nl::exact_plan(mode, weighted, a.n_frames, a.n_pad, kListLanes, &lanes, &lds) != 0)
            return fail(NL_ERR_TOO_MANY_FRAMES,
                        "%d frames do not fit the per-pixel LDS column (mode %d)", a.n_frames, mode);
        nl::StackArgs e = a;
        e.list = h->d_fb_list;
        e.list_count = h->d_fb_count;
        e.list_capacity = (unsigned)h->npix;
        const char *exact_name = "";
        NL_HIP(nl::launch_stack_exact(mode, weighted, e, lanes, kListGrid, lds, h->stream, &exact_name));
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream));
        h->last_has_counters = true;
        h->last_used_fast = true;
#ifdef NL_EXPERIMENTS
    } else if (chunked) {
        // the sigma / winsorized fast path as a chunked pass (see chunk_plan): plain protocol (the memset above,
        // sharded clip counters, a reduction kernel at the end), one set of list lengths per chunk
        int rc = ensure_chunk_resources(h);
        if (rc != NL_OK) return rc;
        NL_HIP(hipMemsetAsync(h->d_chunk_counts, 0, sizeof(unsigned) * 4 * kMaxChunks, h->stream));
        const bool record = mode == NL_ST_WINSOR_SIGMA && a.n_frames > 128 && fused_on && ensure_bounds(h);     // (as the unchunked pass below)
        static const bool coop4_env = [] { const char *e = getenv("NL_COOP4"); return e && e[0] == '1'; }();
        const bool coop4 = coop4_env && NL_COOP4_SUPPORTED(mode, weighted, a.n_frames);
        struct Fork { nl_stack *h; nl::StackArgs e; int mode; int grid0; bool coop4; hipEvent_t done; hipError_t err; };
        for (int k = 0; k < plan.n; k++) {
            const int64_t off = plan.off[k], len = plan.len[k];
            const double share = (double)len / (double)h->npix;
            nl::StackArgs ak = a;
            ak.frames = a.frames + off;
            ak.out = a.out + off;
            ak.npix = len;
            if (record) {
                ak.bounds = h->d_bounds + (size_t)nl::kBoundRounds * (size_t)off;      // [round][pixel of the chunk]
                ak.nrounds = h->d_nrounds + off;
            }
            unsigned *cc = h->d_chunk_counts + 4 * k;
            nl::FastArgs f;
            memset(&f, 0, sizeof f);
            f.fb_list = h->d_fb_list + off;
            f.fb_count = cc;
            f.fb_capacity = (unsigned)len;
            f.fb_snap = cc + 2;
            f.gen_list = h->d_gen_list + off;
            f.gen_count = cc + 1;
            f.gen_capacity = (unsigned)len;
            f.gen_hint = h->gen_hint ? (unsigned)((double)(h->gen_hint - 1u) * share * 1.25) + 1u : 0u;
            if (!weighted) set_split_cols(h, mode, ak.n_frames, f);     // (chunks run one after the other on the pass's stream: one buffer)
            if (mode == NL_ST_WINSOR_SIGMA) f.gen_round_cap = ak.n_frames >= 48 ? 24 : (ak.n_frames > 20 ? 32 : 40);
            nl::StackArgs e = ak;
            e.list = f.fb_list;
            e.list_count = cc;
            e.list_capacity = (unsigned)len;
            int grid0 = kCoopGrid, grid1 = kCoopGrid / 4;
            if (h->fb_hint) {
                const int want = next_pow2((int)(2.0 * share * (double)(h->fb_hint - 1u)) + 64);
                grid0 = want < 1024 ? 1024 : (want > kCoopGrid ? kCoopGrid : want);
                grid1 = grid0 / 4 < 512 ? 512 : grid0 / 4;
            }
            if (coop4) { grid0 = (grid0 + 3) / 4; grid1 = (grid1 + 3) / 4; }
            const bool last = k == plan.n - 1;
            hipEvent_t done = (last && timed) ? h->ev_dom1 : h->ev_chunk[k];
            Fork fork{h, e, mode, grid0, coop4, done, hipSuccess};
            nl::AfterDominant after = [](void *u) {
                // behind the chunk's dominant kernel: its hand-overs are replayed on one stream, the generic pass
                // (launched by the caller of this callback) and the replay of what it adds run on the other
                Fork *fk = static_cast<Fork *>(u);
                nl_stack *hh = fk->h;
                const char *ignored = "";
                hipError_t err = hipStreamWaitEvent(hh->chunk_stream[1], fk->done, 0);
                nl::StackArgs first = fk->e;
                first.list_snap = const_cast<unsigned *>(fk->e.list_count) + 2;
                first.list_part = 0;
                if (err == hipSuccess) err = fk->coop4 ? NL_LAUNCH_COOP4(fk->mode, first, fk->grid0, hh->chunk_stream[1], &ignored)
                                                       : nl::launch_stack_sigma_coop(fk->mode, first, fk->grid0, hh->chunk_stream[1], &ignored);
                if (err == hipSuccess) err = hipStreamWaitEvent(hh->chunk_stream[0], fk->done, 0);
                fk->err = err;
            };
            const char *name_k = "";
            if (a.n_frames <= 128)
                NL_HIP(nl::launch_stack_sigma_fast(ak, f, h->stream, &name_k, done, mode == NL_ST_WINSOR_SIGMA, after, &fork, h->chunk_stream[0]));
            else
                NL_HIP(nl::launch_stack_sigma_ml(ak, f, h->stream, &name_k, done, mode == NL_ST_WINSOR_SIGMA, after, &fork, h->chunk_stream[0]));
            NL_HIP(fork.err);
            if (k == 0) h->last_kernel = name_k;
            const char *exact_name = "";
            e.list_snap = cc + 2;                         // the generic pass's additions
            e.list_part = 1;
            if (coop4) NL_HIP(NL_LAUNCH_COOP4(mode, e, grid1, h->chunk_stream[0], &exact_name));
            else       NL_HIP(nl::launch_stack_sigma_coop(mode, e, grid1, h->chunk_stream[0], &exact_name));
        }
        for (int i = 0; i < 2; i++) {
            NL_HIP(hipEventRecord(h->ev_chunk_join[i], h->chunk_stream[i]));
            NL_HIP(hipStreamWaitEvent(h->stream, h->ev_chunk_join[i], 0));
        }
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream, h->d_chunk_counts, plan.n, 4));
        h->last_lists = true;
        h->last_fused = false;
        h->last_has_counters = true;
        h->last_used_fast = true;
#endif  // NL_EXPERIMENTS
    } else if (!h->force_exact && h->d_fb_list &&
               (nl::fast_supported(mode, weighted, a.n_frames, a.npix) || nl::fast_ml_supported(mode, weighted, a.n_frames, a.npix))) {
        // register-resident fast kernel; pixels it cannot decide go to the exact kernel
        nl::FastArgs f;
        memset(&f, 0, sizeof f);
        // winsorized passes: the fast kernels put the thresholds of every round they decide on record, so that the
        // replay of a pixel that turns undecidable later skips the winsorization loops of the decided rounds
        // (not with the developer switches that bring back round-1 kernels, which write no round counts)
        // (from 129 frames on: C3 tile 5.28 -> 5.14 ms; at 128 frames most undecidable pixels are undecidable in
        // their first round and the stores cost the dominant kernel 0.6 %)
        if (mode == NL_ST_WINSOR_SIGMA && a.n_frames > 128 && fused_on && ensure_bounds(h)) {
            a.bounds = h->d_bounds;
            a.nrounds = h->d_nrounds;
        }
        f.fb_list = h->d_fb_list;
        f.fb_count = h->d_fb_count;
        f.fb_capacity = (unsigned)h->npix;
        f.fb_snap = h->d_fb_count + 2;               // see the replay below
        f.gen_list = h->d_gen_list;
        f.gen_count = h->d_fb_count + 1;
        f.gen_capacity = (unsigned)h->npix;
        f.gen_hint = h->gen_hint;
        f.in_list = nullptr;
        f.in_count = nullptr;
        f.in_capacity = 0;
        if (!weighted) set_split_cols(h, mode, a.n_frames, f);
        // winsorized generic passes and the stages of the cascade behind the dominant kernel: a wave runs for its slowest pixel,
        // and the few pixels whose winsorization loops take dozens of rounds are cheaper in the replay (NL_GEN_ROUND_CAP: rounds per clipping pass; 100 = the limit of every kernel)
        if (mode == NL_ST_WINSOR_SIGMA) {
            static const int cap_env = [] { const char *e = getenv("NL_GEN_ROUND_CAP"); return e ? atoi(e) : 0; }();
            // (measured per frame count on the bench stack; 12 / 13 frames -- the smallest stacks with a zonal kernel -- lose with 40)
            f.gen_round_cap = cap_env > 0 ? cap_env : (a.n_frames >= 48 ? 24 : (a.n_frames > 20 ? 32 : ((a.n_frames == 12 || a.n_frames == 13) ? 60 : 40)));
        }
        // the invariant-interval certificate of the winsorization loops (stack_fast_sigma_impl.hpp): first trial after
        // cert_first rounds of a loop, then every cert_every; NL_WCERT="first,every" ("0" = off), developer switch 16384: off
        // (3, 3: measured best at 16 frames and within 2 % of the best at 24, profiles/r05_winsor_cert.txt)
        if (mode == NL_ST_WINSOR_SIGMA) {
            static const int cert_env[2] = {[] { const char *e = getenv("NL_WCERT"); return e ? atoi(e) : 3; }(),
                                            [] { const char *e = getenv("NL_WCERT"); const char *c = e ? strchr(e, ',') : nullptr; const int v = c ? atoi(c + 1) : 3; return v > 0 ? v : 1; }()};
            f.cert_first = (h->dev_flags & 16384u) ? 0 : cert_env[0];
            f.cert_every = cert_env[1];
        }
        // winsorized clipping of 16 ... 128 frames: the winsorization cascade (stack_fast_sigma_impl.hpp) -- the dominant
        // kernel and a second stage stop at a budget of rounds per wave and hand their unfinished pixels on, a third
        // stage finishes them.  Lists and states live in the buffers of the linear-fit cascade (same sizes, never in
        // use at the same time); their lengths in the scratch set.  NL_WCAS="b1,b2" sets the budgets, "0" turns it off;
        // developer switch 128: off (A/B inside one process)
        bool cascade = false;
        if (mode == NL_ST_WINSOR_SIGMA && a.n_frames <= 128 && a.n_frames >= 12 && !(h->dev_flags & 128u)) {
            // plan: "passes:cap[:group]" per stage, comma-separated, the dominant kernel first; the last stage runs to the end
            struct Plan { int stages; int pass[nl::kCascadeStages], cap[nl::kCascadeStages], group[nl::kCascadeStages]; };
            auto parse = [](const char *e, Plan *pl) {
                pl->stages = 0;
                const char *p = e;
                while (*p && pl->stages < nl::kCascadeStages) {
                    char *end = nullptr;
                    const long a1 = strtol(p, &end, 10);
                    if (end == p || *end != ':') break;
                    p = end + 1;
                    const long a2 = strtol(p, &end, 10);
                    if (end == p) break;
                    long a3 = 4;
                    if (*end == ':') { p = end + 1; a3 = strtol(p, &end, 10); if (end == p) break; }
                    pl->pass[pl->stages] = (int)a1;
                    pl->cap[pl->stages] = (int)a2;
                    pl->group[pl->stages] = a3 < 1 ? 1 : (a3 > 16 ? 16 : (int)a3);
                    pl->stages++;
                    if (*end != ',') break;
                    p = end + 1;
                }
            };
            static const Plan env_plan = [&] { Plan p0{}; const char *e = getenv("NL_WCAS"); if (e) parse(e, &p0); return p0; }();
            static const bool env_off = [] { const char *e = getenv("NL_WCAS"); return e && e[0] == '0' && e[1] == 0; }();
            Plan pl{};
            if (env_plan.stages >= 2) pl = env_plan;
            else if (a.n_frames <= kWinsorCascadeMaxFrames) parse(a.n_frames <= 40 ? kWinsorPlanShallow : kWinsorPlanDeep, &pl);
            nl::LinfitCascade cb;
            // (a list holds at most one entry per pixel of the tile, rounded up to whole workgroups: list and states of a
            // stage share one of the cascade's state arrays, 4 words per pixel; the region lengths take its pixel lists)
            if (!env_off && pl.stages >= 2 && h->npix >= 65536 && linfit_cascade(h, 1, &cb)) {
                cascade = true;
                for (int i = 0; i < 2; i++) {
                    unsigned *base = reinterpret_cast<unsigned *>(cb.state[i]);
                    f.cas_list[i] = base;
                    f.cas_state[i] = base + 2 * (size_t)h->npix;
                    f.cas_count[i] = cb.list[i];
                }
                f.cas_stages = pl.stages;
                for (int k = 0; k < pl.stages; k++) { f.cas_pass[k] = pl.pass[k]; f.cas_cap[k] = pl.cap[k]; f.cas_group[k] = pl.group[k]; }
            }
        }
        // exact replay of the undecidable pixels: one wave per pixel where available.  The
        // hand-overs of the dominant kernel are replayed on a side stream WHILE the generic
        // pass runs (both only depend on the dominant kernel); what the generic pass adds
        // to the list is replayed after it.
        nl::StackArgs e = a;
        e.list = h->d_fb_list;
        e.list_count = h->d_fb_count;
        e.list_capacity = (unsigned)h->npix;
        const bool coop = nl::coop_supported(mode, weighted, a.n_frames) != 0;
        unsigned *snap = h->d_fb_count + 2;               // 1 + list length when the first replay started (set on the device)
        // replay grids: one wave per workgroup, grid-stride over a list whose length is only known on the device;
        // launching 16 k workgroups for a few hundred pixels costs more than replaying them, so the length the
        // last finished pass reported (nl_stack_finish) sizes the grid
        int grid0 = kCoopGrid, grid1 = kCoopGrid / 4;
        if (h->fb_hint) {
            const int want = next_pow2((int)(2u * (h->fb_hint - 1u) + 64u));
            grid0 = want < 1024 ? 1024 : (want > kCoopGrid ? kCoopGrid : want);      // (a 256-workgroup floor measured the same)
            grid1 = grid0 / 4 < 512 ? 512 : grid0 / 4;
        }
        // NL_COOP4=1 (developer switch): replay the lists four pixels per wave (stack_exact_coop4.hip).  Its
        // sequential sums cost a third of the instructions, but with a quarter of the waves in flight the replay
        // turns latency-bound: C3 tile 5.32 -> 6.74 ms, sigma 512 tail 0.82 -> 1.45 ms -- measured, off by default
        static const bool coop4_env = [] { const char *e = getenv("NL_COOP4"); return e && e[0] == '1'; }();
        const bool coop4 = coop && coop4_env && NL_COOP4_SUPPORTED(mode, weighted, a.n_frames);
        if (coop4) { grid0 = (grid0 + 3) / 4; grid1 = (grid1 + 3) / 4; }
        struct Fork { nl_stack *h; nl::StackArgs e; int mode; unsigned *snap; int grid0; bool coop4; bool cascade; hipError_t err; } fork{h, e, mode, snap, grid0, coop4, cascade, hipSuccess};
        nl::AfterDominant after = nullptr;
        if (coop) after = [](void *u) {
            Fork *k = static_cast<Fork *>(u);
            nl_stack *hh = k->h;
            const char *ignored = "";
            // (ev_dom1: recorded behind the dominant kernel.  With a winsorization cascade two more kernels have filled the
            // lists since: an event of its own)
            const bool own = (hh->dev_flags & 32u) || k->cascade;
            hipEvent_t fork_ev = own ? hh->ev_fork : hh->ev_dom1;
            hipError_t err = own ? hipEventRecord(hh->ev_fork, hh->stream) : hipSuccess;
            if (hh->dev_flags & 2u) {            // developer switch: the first replay in front of the generic pass, same stream
                nl::StackArgs first = k->e;
                first.list_snap = k->snap;
                first.list_part = 0;
                if (err == hipSuccess) err = k->coop4 ? NL_LAUNCH_COOP4(k->mode, first, k->grid0, hh->stream, &ignored)
                                                      : nl::launch_stack_sigma_coop(k->mode, first, k->grid0, hh->stream, &ignored);
                if (err == hipSuccess) err = hipEventRecord(hh->ev_join, hh->stream);
                k->err = err;
                return;
            }
            if (err == hipSuccess) err = hipStreamWaitEvent(hh->side_stream, fork_ev, 0);
            nl::StackArgs first = k->e;
            first.list_snap = k->snap;                    // the list as the dominant kernel left it (snapshot on the device)
            first.list_part = 0;
            if (err == hipSuccess) err = k->coop4 ? NL_LAUNCH_COOP4(k->mode, first, k->grid0, hh->side_stream, &ignored)
                                                  : nl::launch_stack_sigma_coop(k->mode, first, k->grid0, hh->side_stream, &ignored);
            if (err == hipSuccess) err = hipEventRecord(hh->ev_join, hh->side_stream);
            k->err = err;
        };
        // Short exact lists (plain sigma, 65 ... 128 frames, fused protocol): generic pass and first replay as the lower and the
        // upper workgroups of ONE launch (stack_tail_fused.hip) instead of two streams -- no fork, no join: the join alone costs
        // a 512-row tile 14 us of its 257.  Every workgroup of that launch claims the generic pass's 48 KiB of LDS (three per
        // CU), hence only while one wave per listed pixel fits the device at that rate.
        // NL_TAIL_FUSED=0 / developer switch 8192: the two-stream protocol (A/B).
        static const bool tail_fused_on = [] { const char *e = getenv("NL_TAIL_FUSED"); return !(e && e[0] == '0'); }();
        const bool tail_fused = tail_fused_on && !(h->dev_flags & (8192u | 2u)) && fused && coop && !coop4 && !cascade &&
                                nl::tail_fused_supported(mode, weighted, a.n_frames) != 0 && h->fb_hint != 0 &&
                                h->fb_hint - 1u <= kTailFusedMaxList;
        nl::StackArgs first_replay = e;
        first_replay.list_snap = snap;                    // the list as the dominant kernel left it (snapshot on the device)
        first_replay.list_part = 0;
        unsigned replay_blocks = h->fb_hint + 31u;        // one wave per listed pixel and some: the list's length is last pass's
        replay_blocks = replay_blocks < 64u ? 64u : replay_blocks > 768u ? 768u : replay_blocks;
        if (tail_fused) after = nullptr;
        if (a.n_frames <= 128)
            NL_HIP(nl::launch_stack_sigma_fast(a, f, h->stream, &h->last_kernel, timed ? h->ev_dom1 : nullptr,
                                               mode == NL_ST_WINSOR_SIGMA, after, &fork, nullptr,
                                               tail_fused ? &first_replay : nullptr, replay_blocks));
        else   // 129..512 frames: 2 or 4 lanes per pixel
            NL_HIP(nl::launch_stack_sigma_ml(a, f, h->stream, &h->last_kernel, timed ? h->ev_dom1 : nullptr,
                                             mode == NL_ST_WINSOR_SIGMA, after, &fork));
        NL_HIP(fork.err);
        const char *exact_name = "";
        if (coop) {
            e.list_snap = snap;                           // the generic pass's additions
            e.list_part = 1;
            if (coop4) NL_HIP(NL_LAUNCH_COOP4(mode, e, grid1, h->stream, &exact_name));
            else       NL_HIP(nl::launch_stack_sigma_coop(mode, e, grid1, h->stream, &exact_name));
            if (!tail_fused) NL_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        } else {
            int lanes = 0;
            size_t lds = 0;
            if (nl::exact_plan(mode, weighted, a.n_frames, a.n_pad, kListLanes, &lanes, &lds) != 0)
                return fail(NL_ERR_TOO_MANY_FRAMES,
                            "%d frames do not fit the per-pixel LDS column (mode %d)", a.n_frames, mode);
            NL_HIP(nl::launch_stack_exact(mode, weighted, e, lanes, kListGrid, lds, h->stream, &exact_name));
        }
        if (fused) h->sets_clean = true;             // (fused implies coop: every kernel of the pass is enqueued)
        else {
            // (the reduction zeroes the scratch set behind itself: no memset in front of the next pass)
            NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream, h->d_fb_count, 1, 0, true));
            zeroed_behind = true;
        }
        h->last_lists = true;
        h->last_fused = fused;
        h->last_tail_fused = tail_fused;
        h->last_has_counters = true;
        h->last_used_fast = true;
    } else if ((h->exact_flavour == 3 ||
                (!h->force_exact && weighted && !(h->dev_flags & 16u) &&
                 a.n_frames <= (mode == NL_ST_WINSOR_SIGMA ? nl::kTileMaxFramesWinsor : nl::kTileMaxFramesSigma))) &&
               nl::tile_supported(mode, weighted, a.n_frames)) {
        // bit-exact replay over the whole tile, 64 consecutive pixels per wave with their columns in
        // LDS, one pixel per lane: the default for weighted sigma / winsorized clipping (their result
        // depends on the reference's permutation, so there is no register-resident shortcut) up to
        // 64 frames -- the LDS column limits it to one wave per SIMD at 128 frames, where the
        // wave-per-pixel replay below is faster (tools/replay_probe.py: 0.5 vs 1.5 ms per Mpixel at
        // 32 frames, 7.7 vs 3.8 at 128); nl_stack_set_exact(h, 3) forces it (verification)
        h->last_used_fast = false;
        const int64_t tiles = (a.npix + 63) / 64;
        const int64_t g = tiles < (1 << 20) ? tiles : (1 << 20);
        NL_HIP(nl::launch_stack_sigma_tile(mode, a, (int)g, h->stream, &h->last_kernel));
        NL_HIP(hipEventRecord(h->ev_dom1, h->stream));
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream));
        h->last_has_counters = true;
    } else if (NL_COOP4_SUPPORTED(mode, weighted, a.n_frames) &&           // (experiments build only)
               (h->exact_flavour == 4 ||
                (!h->force_exact && weighted && !(h->dev_flags & 8u) && mode == NL_ST_WINSOR_SIGMA &&
                 a.n_frames >= nl::kCoop4MinFrames && a.n_frames <= nl::kCoop4MaxFrames &&
                 !(nl::decide_ml_supported(mode, a.n_frames, a.npix) && ensure_bounds(h))))) {      // (only without a decision pass)
        // the four-pixels-per-wave replay over the whole tile: weighted stacks of medium depth (the sequential sums
        // are a large share of a dense replay, and a row of 16 lanes wastes fewer of them on short ranges);
        // nl_stack_set_exact(h, 4) forces it (verification)
        h->last_used_fast = false;
        if (weighted && h->exact_flavour == 0 && nl::decide_supported(mode, a.n_frames, a.npix) && ensure_bounds(h)) {
            // decision pass: the register-resident kernel leaves the clip bounds of every round it can decide
            a.bounds = h->d_bounds;
            a.nrounds = h->d_nrounds;
            const char *ignored = "";
            NL_HIP(nl::launch_stack_sigma_decide(a, h->stream, mode == NL_ST_WINSOR_SIGMA, &ignored));
        }
        const int g = dense_grid((a.npix + 3) / 4, 65536, h->width, 4);
        (void)g;
        NL_HIP(NL_LAUNCH_COOP4(mode, a, (int)g, h->stream, &h->last_kernel));
        NL_HIP(hipEventRecord(h->ev_dom1, h->stream));
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream));
        h->last_has_counters = true;
    } else if ((h->exact_flavour == 2 || (!h->force_exact && (weighted || a.n_frames > 512))) &&
               nl::coop_supported(mode, weighted, a.n_frames)) {
        // the wave-per-pixel exact replay over the whole tile: the default for weighted sigma /
        // winsorized clipping (their result depends on the reference's permutation, so there is
        // no register-resident shortcut) and beyond 512 frames; with nl_stack_set_exact(h, 2) a
        // verification path
        h->last_used_fast = false;
        if (weighted && h->exact_flavour == 0 && mode != NL_ST_MEDIAN && nl::decide_supported(mode, a.n_frames, a.npix) &&
            ensure_bounds(h)) {
            a.bounds = h->d_bounds;                    // decision pass, see the four-pixels-per-wave branch above
            a.nrounds = h->d_nrounds;
            const char *ignored = "";
            NL_HIP(nl::launch_stack_sigma_decide(a, h->stream, mode == NL_ST_WINSOR_SIGMA, &ignored));
        } else if (weighted && h->exact_flavour == 0 && mode != NL_ST_MEDIAN && nl::decide_ml_supported(mode, a.n_frames, a.npix) &&
                   ensure_bounds(h)) {
            // 129 ... 512 frames: the LDS-column kernel of the frame-count class decides (FastArgs::record_only: no
            // outputs, lists or counters; a pixel it would hand to the generic pass has no round on record)
            a.bounds = h->d_bounds;
            a.nrounds = h->d_nrounds;
            nl::FastArgs f;
            memset(&f, 0, sizeof f);
            f.record_only = 1;
            set_split_cols(h, mode, a.n_frames, f);
            const char *ignored = "";
            NL_HIP(nl::launch_stack_sigma_mlz(a, f, h->stream, &ignored, mode == NL_ST_WINSOR_SIGMA));
        }
        const int per_item = mode == NL_ST_MEDIAN ? 1 : nl::coop_group(a);
        // many short workgroups: neighbours that start together share the sectors they fetch, long-lived workgroups
        // drift apart (128 frames x 4096^2, weighted sigma: 34.7 ms with 8 192 workgroups, 31.6 with 16 384, 28.0 with
        // 65 536, 27.2 with 262 144; a 512-row tile of 64 frames: 2.50 / 2.26 / 2.07 / 2.08 ms)
        const int64_t items = a.npix / per_item;
        const int64_t most = 262144;
        const int g = dense_grid(items, most, h->width, per_item);
        if (mode == NL_ST_MEDIAN) NL_HIP(nl::launch_stack_median_coop(a, (int)g, h->stream, &h->last_kernel));
        else                      NL_HIP(nl::launch_stack_sigma_coop(mode, a, (int)g, h->stream, &h->last_kernel));
        NL_HIP(hipEventRecord(h->ev_dom1, h->stream));
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream));
        h->last_has_counters = (mode != NL_ST_MEDIAN);
    } else {
        h->last_used_fast = false;
        int lanes = 0;
        size_t lds = 0;
        if (nl::exact_plan(mode, weighted, a.n_frames, a.n_pad, 64, &lanes, &lds) != 0)
            return fail(NL_ERR_TOO_MANY_FRAMES,
                        "%d frames do not fit the per-pixel LDS column (mode %d)", a.n_frames, mode);
        a.tiles = (a.npix + lanes - 1) / lanes;
        int grid = (int)(a.tiles < (int64_t)h->max_grid ? a.tiles : (int64_t)h->max_grid);
        NL_HIP(nl::launch_stack_exact(mode, weighted, a, lanes, grid, lds, h->stream, &h->last_kernel));
        NL_HIP(hipEventRecord(h->ev_dom1, h->stream));
        NL_HIP(nl::launch_reduce_counters(h->d_partial, nl::kClipSlots, h->d_counters, h->stream));
        h->last_has_counters = (mode != NL_ST_MEDIAN);
    }
    NL_HIP(hipEventRecord(h->ev_stop, h->stream));
    h->partial_clean = zeroed_behind || keep_clean;
    if (!fused) h->last_fused = false;
    if (!fused || !sigma_fast) h->last_tail_fused = false;
    if (!sigma_fast) h->last_lists = false;
    h->last_chunks = chunked ? plan.n : 0;
    h->pass_seq++;
    h->last_mode = mode;
    h->pending = true;
    return NL_OK;
}

int nl_stack_finish(nl_stack_t *h, float *out_host, int64_t *clip_low, int64_t *clip_high)
{
    NL_CHECK_HANDLE(h);
    unsigned long long c[4] = {0, 0, 0, 0};
    // (a fast sigma / winsorized pass leaves its list lengths behind the totals: c[2] = exact list | generic list << 32)
    if (h->last_has_counters && (clip_low || clip_high || h->last_lists))
        NL_HIP(hipMemcpyAsync(c, h->d_counters, h->last_lists ? 3 * sizeof c[0] : 2 * sizeof c[0], hipMemcpyDeviceToHost, h->stream));
    if (out_host)
        NL_HIP(hipMemcpyAsync(out_host + (int64_t)h->row0 * h->width, h->d_out,
                              (size_t)h->npix * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    h->pending = false;
    if (h->last_has_counters && h->last_lists) {
        h->fb_hint = (unsigned)(c[2] & 0xffffffffull) + 1u;
        h->gen_hint = (unsigned)(c[2] >> 32) + 1u;
        hints_store({h->n_frames, h->npix, h->last_mode, h->last_weighted}, h->fb_hint, h->gen_hint);
    }
    if (clip_low) *clip_low = (int64_t)c[0];
    if (clip_high) *clip_high = (int64_t)c[1];
    return NL_OK;
}

int nl_stack_run(nl_stack_t *h, int mode, float sigma_low, float sigma_high, float ref_loc,
                 float *out_host, int64_t *clip_low, int64_t *clip_high)
{
    int rc = nl_stack_run_async(h, mode, sigma_low, sigma_high, ref_loc);
    if (rc != NL_OK) return rc;
    return nl_stack_finish(h, out_host, clip_low, clip_high);
}

float nl_stack_last_dominant_kernel_ms(nl_stack_t *h)
{
    if (!h || !h->ev_dom0) return -1.0f;
    if (hipSetDevice(h->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(h->ev_dom1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, h->ev_dom0, h->ev_dom1) != hipSuccess) return -1.0f;
    return ms;
}

int nl_stack_set_exact(nl_stack_t *h, int on)
{
    NL_CHECK_HANDLE(h);
    if (on < 0 || on > 4) return fail(NL_ERR_INVALID_ARG, "set_exact: unknown flavour %d (0 ... 4)", on);
#ifndef NL_EXPERIMENTS
    // (a switch this build does not carry must not fall through to another kernel silently: an A/B run would time the same
    // kernel twice)
    if (on == 4)
        return fail(NL_ERR_INVALID_ARG, "set_exact: flavour 4 (four pixels per wave) is in the experiments build only (make EXPERIMENTS=1)");
#endif
    h->force_exact = on != 0;
    h->exact_flavour = on;
    return NL_OK;
}

int nl_stack_set_dev_flags(nl_stack_t *h, unsigned flags)
{
    NL_CHECK_HANDLE(h);
#ifndef NL_EXPERIMENTS
    if (flags & (1024u | 2048u))
        return fail(NL_ERR_INVALID_ARG, "set_dev_flags: switches 1024 / 2048 (split / persistent LDS-column pass) are in the experiments "
                                        "build only (make EXPERIMENTS=1)");
#endif
    h->dev_flags = flags;
    return NL_OK;
}

// list lengths of the last fast pass: a sigma / winsorized pass leaves them behind its totals (d_counters[2] = exact list |
// generic list << 32 -- its own counters may be zeroed again by then), the other fast passes keep them in the scratch set
static int64_t last_list_length(nl_stack_t *h, int which)
{
    if (hipSetDevice(h->device) != hipSuccess) return -1;
    if (h->last_lists) {
        unsigned long long c = 0;
        if (hipMemcpyAsync(&c, h->d_counters + 2, sizeof c, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
        return which == 0 ? (int64_t)(c & 0xffffffffull) : (int64_t)(c >> 32);
    }
    unsigned c = 0;
    if (hipMemcpyAsync(&c, h->d_fb_count + which, sizeof c, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    return (int64_t)c;
}

int64_t nl_stack_last_fallback_pixels(nl_stack_t *h)
{
    if (!h || !h->last_used_fast || !h->d_fb_count) return 0;
    return last_list_length(h, 0);
}

int nl_stack_last_pass_protocol(nl_stack_t *h)
{
    if (!h) return 0;
    return (h->last_fused ? 1 : 0) | (h->last_tail_fused ? 2 : 0) | (h->last_chunks > 0 ? 4 : 0);
}

int64_t nl_stack_last_generic_pixels(nl_stack_t *h)
{
    if (!h || !h->last_used_fast || !h->d_fb_count || !h->d_gen_list) return 0;
    return last_list_length(h, 1);
}

int nl_stack_linfit_stage_counts(nl_stack_t *h, unsigned *counts, int n)
{
    if (!h || !counts || n <= 0 || !h->d_lf_count || h->last_mode != NL_ST_LINEAR_FIT || !h->last_used_fast) return 0;
    if (hipSetDevice(h->device) != hipSuccess) return -1;
    unsigned c[nl::kLinfitCounters] = {};
    if (hipMemcpyAsync(c, h->d_lf_count, sizeof c, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    const int m = n < nl::kLinfitCounters ? n : nl::kLinfitCounters;
    for (int i = 0; i < m; i++) counts[i] = c[i];
    return m;
}

// GPU times of a pass that is `back` passes old (0 = the last one enqueued); -1 where unavailable
int nl_stack_pass_times(nl_stack_t *h, int back, float *pass_ms, float *dominant_ms)
{
    NL_CHECK_HANDLE(h);
    if (back < 0 || back >= kTimingRing || (int64_t)back >= h->pass_seq)
        return fail(NL_ERR_INVALID_ARG, "pass_times: pass %d back is not in the ring of %d", back, kTimingRing);
    const int slot = (int)((h->pass_seq - 1 - back) % kTimingRing);
    if (!h->ring_timed[slot])
        return fail(NL_ERR_INVALID_ARG, "pass_times: pass %d back ran without timing events (developer switch 32)", back);
    NL_HIP(hipEventSynchronize(h->ring_stop[slot]));
    float ms = -1.0f;
    if (pass_ms) {
        NL_HIP(hipEventElapsedTime(&ms, h->ring_start[slot], h->ring_stop[slot]));
        *pass_ms = ms;
    }
    if (dominant_ms) {
        NL_HIP(hipEventElapsedTime(&ms, h->ring_dom0_is_start[slot] ? h->ring_start[slot] : h->ring_dom0[slot], h->ring_dom1[slot]));
        *dominant_ms = ms;
    }
    return NL_OK;
}

// enqueues, behind the last pass on the handle's stream, a 16-byte device-to-device copy of its
// {clip_low, clip_high} totals into a caller-owned device buffer (e.g. the tensor an RCCL
// all-reduce runs on): no host round trip between the pass and the reduction
int nl_stack_copy_counters_async(nl_stack_t *h, void *device_dst)
{
    NL_CHECK_HANDLE(h);
    if (!device_dst) return fail(NL_ERR_INVALID_ARG, "copy_counters_async: null destination");
    if (h->last_has_counters)
        NL_HIP(hipMemcpyAsync(device_dst, h->d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    else
        NL_HIP(hipMemsetAsync(device_dst, 0, 2 * sizeof(unsigned long long), h->stream));
    return NL_OK;
}

void *nl_stack_stream(nl_stack_t *h) { return h ? (void *)h->stream : nullptr; }
void *nl_stack_counters_device_ptr(nl_stack_t *h) { return h ? (void *)h->d_counters : nullptr; }

int nl_stack_set_counters_buffer(nl_stack_t *h, void *device_buf)
{
    NL_CHECK_HANDLE(h);
    h->d_counters = device_buf ? static_cast<unsigned long long *>(device_buf) : h->d_counters_own;
    return NL_OK;
}

int nl_stack_order_stream_after(nl_stack_t *h, void *hip_stream)
{
    NL_CHECK_HANDLE(h);
    if (!hip_stream) return fail(NL_ERR_INVALID_ARG, "order_stream_after: null stream");
    // a ring of events: the waiting stream may still be working off an older one when the next pass is enqueued
    const int slot = h->order_seq++ % kOrderRing;
    if (!h->ev_order[slot]) NL_HIP(hipEventCreateWithFlags(&h->ev_order[slot], hipEventDisableTiming | h->ev_rel));
    NL_HIP(hipEventRecord(h->ev_order[slot], h->stream));
    NL_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), h->ev_order[slot], 0));
    return NL_OK;
}

float nl_stack_last_kernel_ms(nl_stack_t *h)
{
    if (!h || !h->ev_start) return -1.0f;
    if (hipSetDevice(h->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(h->ev_stop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, h->ev_start, h->ev_stop) != hipSuccess) return -1.0f;
    return ms;
}

// stackfindsigma.go:48-98 (commented-out reference code = the spec)
int nl_stack_find_sigmas(nl_stack_t *h, int mode, float ref_loc,
                         float clip_perc_low, float clip_perc_high,
                         nl_reduce_fn reduce, void *user,
                         float *out_host, int64_t *clip_low, int64_t *clip_high,
                         float *sigma_low, float *sigma_high, int *passes)
{
    NL_CHECK_HANDLE(h);
    if (mode == NL_ST_AUTO) mode = auto_select_mode(h->n_frames);
    if (mode < NL_ST_MEDIAN || mode > NL_ST_LINEAR_FIT) return fail(NL_ERR_INVALID_MODE, "invalid stacking mode");
    // the counters cover the samples the percentages are taken of: with a reducer the whole
    // image (every tile contributes), without one only this handle's tile
    const int64_t total = reduce ? (int64_t)h->width * h->height * (int64_t)h->n_frames
                                 : h->npix * (int64_t)h->n_frames;
    if (mode != NL_ST_SIGMA && mode != NL_ST_WINSOR_SIGMA) {
        // stackfindsigma.go:40-46: Newton's method for the linear fit; the other modes "do not support
        // sigmas" and are stacked once with 0, 0
        nl::SigmaNewton nw(clip_perc_low, total);
        int n_pass = 0;
        for (;;) {
            const bool newton = mode == NL_ST_LINEAR_FIT;
            int64_t c[2] = {0, 0};
            int rc = nl_stack_run(h, mode, newton ? nw.next_low() : 0.0f, newton ? nw.next_high() : 0.0f, ref_loc,
                                  nullptr, &c[0], &c[1]);
            if (rc != NL_OK) return rc;
            n_pass++;
            if (reduce) {
                rc = reduce(c, user);
                if (rc != 0) return fail(NL_ERR_INVALID_ARG, "counter reduction callback failed (%d)", rc);
            }
            const int st = newton ? nw.step(c[0], c[1]) : 1;
            if (st == 0) continue;
            if (st == 2) {                       // a probe pass overwrote the result: re-make the base pass
                rc = nl_stack_run(h, mode, nw.sig_low, nw.sig_high, ref_loc, nullptr, nullptr, nullptr);
                if (rc != NL_OK) return rc;
            }
            if (clip_low) *clip_low = newton ? nw.base_lo : c[0];
            if (clip_high) *clip_high = newton ? nw.base_hi : c[1];
            if (sigma_low) *sigma_low = newton ? nw.sig_low : 0.0f;
            if (sigma_high) *sigma_high = newton ? nw.sig_high : 0.0f;
            if (passes) *passes = n_pass;
            if (out_host) return nl_stack_finish(h, out_host, nullptr, nullptr);
            return NL_OK;
        }
    }
    nl::SigmaBisection bis(clip_perc_low, clip_perc_high, total);
    int n_pass = 0;
    for (;;) {
        int64_t c[2] = {0, 0};
        int rc = nl_stack_run(h, mode, bis.low_mid, bis.high_mid, ref_loc, nullptr, &c[0], &c[1]);
        if (rc != NL_OK) return rc;
        n_pass++;
        if (reduce) {
            rc = reduce(c, user);
            if (rc != 0) return fail(NL_ERR_INVALID_ARG, "counter reduction callback failed (%d)", rc);
        }
        if (bis.step(c[0], c[1])) {
            if (clip_low) *clip_low = c[0];
            if (clip_high) *clip_high = c[1];
            if (sigma_low) *sigma_low = bis.low_mid;
            if (sigma_high) *sigma_high = bis.high_mid;
            if (passes) *passes = n_pass;
            if (out_host) return nl_stack_finish(h, out_host, nullptr, nullptr);
            return NL_OK;
        }
    }
}

// StackIncremental / StackIncrementalFinalize, stack.go:924-944
int nl_stack_accumulate(nl_stack_t *h, float weight, int first)
{
    NL_CHECK_HANDLE(h);
    if (!h->d_acc) NL_HIP(dev_malloc(&h->d_acc, (size_t)h->npix * sizeof(float)));
    NL_HIP(nl::launch_axpy(h->d_acc, h->d_out, weight, first, h->npix, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

int nl_stack_accumulate_finalize(nl_stack_t *h, float weight_sum, float *out_host)
{
    NL_CHECK_HANDLE(h);
    if (!h->d_acc) return fail(NL_ERR_INVALID_ARG, "accumulate_finalize before accumulate");
    volatile float factor = 1.0f / weight_sum;
    NL_HIP(nl::launch_scale(h->d_acc, factor, h->npix, h->stream));
    if (out_host)
        NL_HIP(hipMemcpyAsync(out_host + (int64_t)h->row0 * h->width, h->d_acc,
                              (size_t)h->npix * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

// ---- per-frame statistics ---------------------------------------------------

static int frame_stats_impl(nl_stack_t *h, const float *d, int64_t n, float *mn, float *mean,
                            float *mx, double *variance)
{
    std::vector<double> part(3 * kStatBlocks);
    NL_HIP(nl::launch_min_sum_max(d, n, h->d_stat_partial, kStatBlocks, h->stream));
    NL_HIP(hipMemcpyAsync(part.data(), h->d_stat_partial, sizeof(double) * 3 * kStatBlocks,
                          hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    float lo = (float)part[0], hi = (float)part[2];
    double sum = 0.0;
    for (int b = 0; b < kStatBlocks; b++) {
        const float bl = (float)part[3 * b], bh = (float)part[3 * b + 2];
        if (bl < lo) lo = bl;
        if (bh > hi) hi = bh;
        sum += part[3 * b + 1];
    }
    const float m = (float)(sum / (double)n);
    if (mn) *mn = lo;
    if (mx) *mx = hi;
    if (mean) *mean = m;
    if (variance) {
        NL_HIP(nl::launch_variance(d, n, m, h->d_stat_partial, kStatBlocks, h->stream));
        NL_HIP(hipMemcpyAsync(part.data(), h->d_stat_partial, sizeof(double) * kStatBlocks,
                              hipMemcpyDeviceToHost, h->stream));
        NL_HIP(hipStreamSynchronize(h->stream));
        double s = 0.0;
        for (int b = 0; b < kStatBlocks; b++) s += part[b];
        *variance = s / (double)n;
    }
    return NL_OK;
}

int nl_stack_frame_stats(nl_stack_t *h, int idx, float *mn, float *mean, float *mx,
                         double *variance)
{
    NL_CHECK_HANDLE(h);
    NL_SETTLE_UPLOADS(h);
    if (idx < 0 || idx >= h->n_frames) return fail(NL_ERR_INVALID_ARG, "frame_stats: bad index %d", idx);
    return frame_stats_impl(h, h->d_frames + (int64_t)idx * h->fstride, h->npix, mn, mean, mx, variance);
}

static int frame_noise_impl(nl_stack_t *h, const float *d, float *noise)
{
    std::vector<double> part(kStatBlocks);
    NL_HIP(nl::launch_noise(d, h->width, h->height, h->d_stat_partial, kStatBlocks, h->stream));
    NL_HIP(hipMemcpyAsync(part.data(), h->d_stat_partial, sizeof(double) * kStatBlocks,
                          hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    double s = 0.0;
    for (int b = 0; b < kStatBlocks; b++) s += part[b];
    // noise.go:53: factor = float32(sqrt(pi/2)) / (6*float32(w-2)*float32(h-2)), fp32
    const float c = (float)sqrt(0.5 * M_PI);
    volatile float den = 6.0f * (float)(h->width - 2);
    den = den * (float)(h->height - 2);
    const float factor = c / den;
    *noise = (float)s * factor;
    return NL_OK;
}

int nl_stack_frame_noise(nl_stack_t *h, int idx, float *noise)
{
    NL_CHECK_HANDLE(h);
    NL_SETTLE_UPLOADS(h);
    if (idx < 0 || idx >= h->n_frames || !noise)
        return fail(NL_ERR_INVALID_ARG, "frame_noise: bad index %d or null output", idx);
    if (h->row0 != 0 || h->rows != h->height)
        return fail(NL_ERR_INVALID_ARG, "frame_noise needs a whole-image handle (3x3 stencil)");
    if (h->width < 3 || h->height < 3) return fail(NL_ERR_INVALID_ARG, "frame_noise: image too small");
    return frame_noise_impl(h, h->d_frames + (int64_t)idx * h->fstride, noise);
}

int nl_stack_weights_from_noise(nl_stack_t *h, float *noise_out)
{
    NL_CHECK_HANDLE(h);
    std::vector<float> noise((size_t)h->n_frames), w((size_t)h->n_frames);
    for (int i = 0; i < h->n_frames; i++) {
        int rc = nl_stack_frame_noise(h, i, &noise[(size_t)i]);
        if (rc != NL_OK) return rc;
    }
    if (noise_out) memcpy(noise_out, noise.data(), sizeof(float) * noise.size());
    int rc = nl_weights_from_scalars(NL_WEIGHT_INVERSE_NOISE, noise.data(), h->n_frames, w.data(), nullptr);
    if (rc != NL_OK) return rc;
    return nl_stack_set_weights(h, w.data());
}

// ---- formats and steps either side of the stack (ingest.hip) ----------------------
static int ingest_reserve(nl_stack_t *h, size_t bytes)
{
    if (bytes <= h->ingest_bytes) return NL_OK;
    if (h->d_ingest) { (void)hipFree(h->d_ingest); h->d_ingest = nullptr; h->ingest_bytes = 0; }
    NL_HIP(dev_malloc(&h->d_ingest, bytes));
    h->ingest_bytes = bytes;
    return NL_OK;
}

// min / max / mean from the decode kernel's per-block partials (read.go:210: mean = float32(sum/len))
static int decode_stats(nl_stack_t *h, int64_t n, float *stats_out)
{
    std::vector<double> part(3 * kStatBlocks);
    NL_HIP(hipMemcpyAsync(part.data(), h->d_stat_partial, sizeof(double) * 3 * kStatBlocks,
                          hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    float lo = (float)part[0], hi = (float)part[1];
    double sum = 0.0;
    for (int b = 0; b < kStatBlocks; b++) {
        const float bl = (float)part[3 * b], bh = (float)part[3 * b + 1];
        if (bl < lo) lo = bl;
        if (bh > hi) hi = bh;
        sum += part[3 * b + 2];
    }
    stats_out[0] = lo;
    stats_out[1] = hi;
    stats_out[2] = (float)(sum / (double)n);
    return NL_OK;
}

// internal/star/coord.go:159-199, fp32 as written there
static int invert_transform(const float t[6], float inv[6])
{
    const volatile float bd = t[1] * t[3], ae = t[0] * t[4];
    const float eps = bd - ae;
    if (eps < 1e-8f && -eps < 1e-8f) return fail(NL_ERR_INVALID_ARG, "Matrix has no inverse, epsilon=%g", eps);
    const volatile float den1 = bd - ae, den2 = ae - bd;
    const volatile float ce = t[2] * t[4], bf = t[1] * t[5], cd = t[2] * t[3], af = t[0] * t[5];
    const volatile float n1 = ce - bf, n2 = cd - af;
    inv[0] = -t[4] / den1;
    inv[1] = t[1] / den1;
    inv[2] = n1 / den1;
    inv[3] = -t[3] / den2;
    inv[4] = t[0] / den2;
    inv[5] = n2 / den2;
    return NL_OK;
}

int nl_stack_upload_frame_fits(nl_stack_t *h, int idx, const void *raw_host, int bitpix, float bscale,
                               float bzero, float multiplier, float offset, float *stats_out)
{
    NL_CHECK_HANDLE(h);
    if (idx < 0 || idx >= h->n_frames || !raw_host)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_fits: bad index %d or null payload", idx);
    const int bpv = nl::fits_bytes_per_value(bitpix);
    if (bpv == 0) return fail(NL_ERR_INVALID_ARG, "Unknown BITPIX value %d", bitpix);      // read.go:169
    if (h->d_frames != h->d_frames_owned)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_fits: frames are attached, not owned");
    const size_t bytes = (size_t)h->npix * (size_t)bpv;
    int rc = ingest_reserve(h, bytes);
    if (rc != NL_OK) return rc;
    NL_HIP(hipMemcpyAsync(h->d_ingest, raw_host, bytes, hipMemcpyHostToDevice, h->stream));
    const bool affine = !(multiplier == 1.0f && offset == 0.0f);
    NL_HIP(nl::launch_fits_decode(h->d_ingest, bitpix, h->npix, bscale, bzero, affine, multiplier, offset,
                                  h->d_frames + (int64_t)idx * h->fstride, h->d_stat_partial, kStatBlocks,
                                  h->stream));
    if (stats_out) return decode_stats(h, h->npix, stats_out);
    NL_HIP(hipStreamSynchronize(h->stream));          // the caller's buffer must not be read after return
    return NL_OK;
}

int nl_stack_upload_frame_projected(nl_stack_t *h, int idx, const float *src_host, int src_w, int src_h,
                                    const float trans[6], float out_of_bounds, float multiplier, float offset)
{
    NL_CHECK_HANDLE(h);
    if (idx < 0 || idx >= h->n_frames || !src_host || !trans || src_w < 1 || src_h < 1)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_projected: bad argument (frame %d)", idx);
    if (h->d_frames != h->d_frames_owned)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_projected: frames are attached, not owned");
    float inv[6];
    int rc = invert_transform(trans, inv);
    if (rc != NL_OK) return rc;
    const size_t bytes = (size_t)src_w * (size_t)src_h * sizeof(float);
    rc = ingest_reserve(h, bytes);
    if (rc != NL_OK) return rc;
    NL_HIP(hipMemcpyAsync(h->d_ingest, src_host, bytes, hipMemcpyHostToDevice, h->stream));
    const bool affine = !(multiplier == 1.0f && offset == 0.0f);
    NL_HIP(nl::launch_project(static_cast<const float *>(h->d_ingest), src_w, src_h,
                              h->d_frames + (int64_t)idx * h->fstride, h->width, h->row0, h->rows, inv,
                              out_of_bounds, affine, multiplier, offset, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

// device scratch of the overlapped ingest (copy stream: operations on it are stream-ordered, one buffer serves
// every frame in flight)
static int ingest_async_reserve(nl_stack_t *h, size_t bytes)
{
    if (!h->d_stat_partial_async) NL_HIP(dev_malloc(&h->d_stat_partial_async, sizeof(double) * 3 * kStatBlocks));
    if (h->ingest_async_bytes >= bytes) return NL_OK;
    if (h->copy_stream) NL_HIP(hipStreamSynchronize(h->copy_stream));
    if (h->d_ingest_async) { NL_HIP(hipFree(h->d_ingest_async)); h->d_ingest_async = nullptr; h->ingest_async_bytes = 0; }
    NL_HIP(dev_malloc(&h->d_ingest_async, bytes));
    h->ingest_async_bytes = bytes;
    return NL_OK;
}

int nl_stack_upload_frame_fits_async(nl_stack_t *h, int idx, const void *raw_host, int bitpix, float bscale,
                                     float bzero, float multiplier, float offset)
{
    NL_CHECK_HANDLE(h);
    if (idx < 0 || idx >= h->n_frames || !raw_host)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_fits: bad index %d or null payload", idx);
    const int bpv = nl::fits_bytes_per_value(bitpix);
    if (bpv == 0) return fail(NL_ERR_INVALID_ARG, "Unknown BITPIX value %d", bitpix);      // read.go:169
    if (h->d_frames != h->d_frames_owned)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_fits: frames are attached, not owned");
    const size_t bytes = (size_t)h->npix * (size_t)bpv;
    char *staged = nullptr;
    int slot = 0;
    int rc = stage_host_bytes(h, raw_host, bytes, &staged, &slot);
    if (rc != NL_OK) return rc;
    // The decode kernel reads the payload straight out of the pinned staging buffer (round 6): DMA into a device scratch
    // and a kernel behind it on one stream took turns -- copy engine, compute queue, copy engine ... with a dependency
    // hand-over each way -- and ran at 0.9 ms per 32 MiB int16 frame where the link needs 0.6 (bench.py apply_from_host:
    // 34.7 GiB/s).  One kernel per frame that pulls its bytes over the link itself (8- or 16-byte loads per lane) has no hand-over at all.
    // NL_FITS_ZEROCOPY=0: the DMA + kernel pair, for A/B runs.
    static const bool zero_copy = [] { const char *e = getenv("NL_FITS_ZEROCOPY"); return !e || atoi(e) != 0; }();
    const void *raw_dev = staged;
    if (!zero_copy) {
        rc = ingest_async_reserve(h, bytes);
        if (rc != NL_OK) return rc;
        NL_HIP(hipMemcpyAsync(h->d_ingest_async, staged, bytes, hipMemcpyHostToDevice, h->copy_stream));
        raw_dev = h->d_ingest_async;
    } else if (!h->d_stat_partial_async) {
        NL_HIP(dev_malloc(&h->d_stat_partial_async, sizeof(double) * 3 * kStatBlocks));
    }
    const bool affine = !(multiplier == 1.0f && offset == 0.0f);
    NL_HIP(nl::launch_fits_decode(raw_dev, bitpix, h->npix, bscale, bzero, affine, multiplier, offset,
                                  h->d_frames + (int64_t)idx * h->fstride, h->d_stat_partial_async, kStatBlocks,
                                  h->copy_stream));
    return stage_done(h, slot);
}

int nl_stack_upload_frame_projected_async(nl_stack_t *h, int idx, const float *src_host, int src_w, int src_h,
                                          const float trans[6], float out_of_bounds, float multiplier, float offset)
{
    NL_CHECK_HANDLE(h);
    if (idx < 0 || idx >= h->n_frames || !src_host || !trans || src_w < 1 || src_h < 1)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_projected: bad argument (frame %d)", idx);
    if (h->d_frames != h->d_frames_owned)
        return fail(NL_ERR_INVALID_ARG, "upload_frame_projected: frames are attached, not owned");
    float inv[6];
    int rc = invert_transform(trans, inv);
    if (rc != NL_OK) return rc;
    const size_t bytes = (size_t)src_w * (size_t)src_h * sizeof(float);
    char *staged = nullptr;
    int slot = 0;
    rc = stage_host_bytes(h, src_host, bytes, &staged, &slot);
    if (rc != NL_OK) return rc;
    rc = ingest_async_reserve(h, bytes);
    if (rc != NL_OK) return rc;
    NL_HIP(hipMemcpyAsync(h->d_ingest_async, staged, bytes, hipMemcpyHostToDevice, h->copy_stream));
    const bool affine = !(multiplier == 1.0f && offset == 0.0f);
    NL_HIP(nl::launch_project(static_cast<const float *>(h->d_ingest_async), src_w, src_h,
                              h->d_frames + (int64_t)idx * h->fstride, h->width, h->row0, h->rows, inv,
                              out_of_bounds, affine, multiplier, offset, h->copy_stream));
    return stage_done(h, slot);
}

int nl_stack_frame_affine(nl_stack_t *h, int idx, float multiplier, float offset)
{
    NL_CHECK_HANDLE(h);
    NL_SETTLE_UPLOADS(h);
    if (idx < 0 || idx >= h->n_frames) return fail(NL_ERR_INVALID_ARG, "frame_affine: bad index %d", idx);
    NL_HIP(nl::launch_affine(h->d_frames + (int64_t)idx * h->fstride, h->npix, multiplier, offset, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

int nl_stack_download_result_fits(nl_stack_t *h, void *raw_host)
{
    NL_CHECK_HANDLE(h);
    if (!raw_host) return fail(NL_ERR_INVALID_ARG, "download_result_fits: null buffer");
    if (h->pending) return fail(NL_ERR_INVALID_ARG, "download_result_fits: a pass is still pending (call nl_stack_finish)");
    const size_t bytes = (size_t)h->npix * sizeof(float);
    int rc = ingest_reserve(h, bytes);
    if (rc != NL_OK) return rc;
    NL_HIP(nl::launch_fits_encode(h->d_out, h->npix, 1, h->d_ingest, h->stream));
    NL_HIP(hipMemcpyAsync(raw_host, h->d_ingest, bytes, hipMemcpyDeviceToHost, h->stream));
    NL_HIP(hipStreamSynchronize(h->stream));
    return NL_OK;
}

int nl_fits_decode(const void *raw_host, int bitpix, int64_t n, float bscale, float bzero, float *out_host,
                   float *stats_out, int device)
{
    if (!raw_host || !out_host || n < 1 || n > 0x7fffffff)
        return fail(NL_ERR_INVALID_ARG, "fits_decode: bad argument");
    if (nl::fits_bytes_per_value(bitpix) == 0) return fail(NL_ERR_INVALID_ARG, "Unknown BITPIX value %d", bitpix);
    // a one-frame handle of n x 1 pixels carries the stream and the scratch buffers
    nl_stack_t *h = nl_stack_create(1, (int)n, 1, 0, 1, device);
    if (!h) return NL_ERR_HIP;
    int rc = nl_stack_upload_frame_fits(h, 0, raw_host, bitpix, bscale, bzero, 1.0f, 0.0f, stats_out);
    if (rc == NL_OK) rc = nl_stack_download_tile(h, 0, out_host);
    nl_stack_destroy(h);
    return rc;
}

int nl_project_bilinear(const float *src_host, int src_w, int src_h, float *dst_host, int dst_w, int dst_h,
                        const float trans[6], float out_of_bounds, int device)
{
    if (!src_host || !dst_host || dst_w < 1 || dst_h < 1)
        return fail(NL_ERR_INVALID_ARG, "project_bilinear: bad argument");
    nl_stack_t *h = nl_stack_create(1, dst_w, dst_h, 0, dst_h, device);
    if (!h) return NL_ERR_HIP;
    int rc = nl_stack_upload_frame_projected(h, 0, src_host, src_w, src_h, trans, out_of_bounds, 1.0f, 0.0f);
    if (rc == NL_OK) rc = nl_stack_download_tile(h, 0, dst_host);
    nl_stack_destroy(h);
    return rc;
}

// MedianFilter / GatherAndMedian, ops/pre/badpixels.go:54-77 and internal/median/gather.go:26-38
int nl_median_filter_mask(const float *in_host, float *out_host, int64_t n, const int32_t *mask, int mask_len, int device)
{
    if (!in_host || !out_host || n < 1 || !mask || mask_len < 1 || mask_len > nl::kMedianMaskMax)
        return fail(NL_ERR_INVALID_ARG, "median_filter_mask: bad argument (mask of 1..%d offsets)", nl::kMedianMaskMax);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NL_ERR_NO_DEVICE, "no HIP device available; libnlstack has no CPU path");
    NL_HIP(hipSetDevice(device));
    const size_t bytes = (size_t)n * sizeof(float);
    float *d_in = nullptr, *d_out = nullptr;
    NL_HIP(dev_malloc(&d_in, bytes));
    hipError_t e = dev_malloc(&d_out, bytes);
    if (e != hipSuccess) { (void)hipFree(d_in); return fail(NL_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    do {
        if ((e = hipMemcpy(d_in, in_host, bytes, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = nl::launch_median_mask(d_in, d_out, n, mask, mask_len, nullptr)) != hipSuccess) break;
        if ((e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(NL_ERR_HIP, "median_filter_mask: %s", hipGetErrorString(e));
    return NL_OK;
}

int nl_median_filter_3x3(const float *in_host, float *out_host, int width, int height, int device)
{
    if (!in_host || !out_host || width < 1 || height < 1)
        return fail(NL_ERR_INVALID_ARG, "median_filter_3x3: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NL_ERR_NO_DEVICE, "no HIP device available; libnlstack has no CPU path");
    NL_HIP(hipSetDevice(device));
    const size_t bytes = (size_t)width * height * sizeof(float);
    float *d_in = nullptr, *d_out = nullptr;
    NL_HIP(dev_malloc(&d_in, bytes));
    hipError_t e = dev_malloc(&d_out, bytes);
    if (e != hipSuccess) { (void)hipFree(d_in); return fail(NL_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    int rc = NL_OK;
    do {
        if ((e = hipMemcpy(d_in, in_host, bytes, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = nl::launch_median3x3(d_in, d_out, width, height, nullptr)) != hipSuccess) break;
        if ((e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    if (e != hipSuccess) rc = fail(NL_ERR_HIP, "median_filter_3x3: %s", hipGetErrorString(e));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
