// nlstack_group.hip -- nl_group_*: one stack fanned out over several GPUs from ONE process.
//
// The reference's Apply splits the pixel range over goroutines
// (internal/ops/stack/stack.go:142-152) and sums the two clip counters over them
// (:193-198).  A single-process host -- the Go CLI behind the cgo shim, the C++
// operator mirror -- gets the same split over the GPUs of the node here: tile t
// owns the rows tile_rows(height, n_tiles, t) of ALL frames on its own device, the
// passes of all tiles are enqueued before any is awaited, the result tiles land in
// disjoint row ranges of the caller's buffer, and the counters are summed on the
// host (two int64 per tile).  No pixel crosses devices, so there is no collective;
// the multi-process form of the same split (torch.distributed / RCCL) is
// nightlight_amd/dist.py.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <thread>
#include <vector>

#include "stack_kernels.h"

struct nl_group {
    int n_frames = 0, width = 0, height = 0;
    std::vector<nl_stack_t *> tiles;
    std::vector<int> row0, rows;
    int distinct_devices = 1;
};

extern "C" {

void nl_group_tile_rows(int height, int n_tiles, int t, int *row0, int *rows)
{
    const int base = height / n_tiles, extra = height % n_tiles;
    if (rows) *rows = base + (t < extra ? 1 : 0);
    if (row0) *row0 = t * base + (t < extra ? t : extra);
}

void nl_group_destroy(nl_group_t *g)
{
    if (!g) return;
    for (nl_stack_t *h : g->tiles) nl_stack_destroy(h);
    delete g;
}

nl_group_t *nl_group_create(int n_frames, int width, int height, int n_tiles, const int *devices)
{
    int ndev = nl_device_count();
    if (ndev <= 0) return nullptr;                       // nl_last_error: no HIP device, no CPU path
    if (n_tiles <= 0) n_tiles = ndev;
    if (n_tiles > height) n_tiles = height > 0 ? height : 1;
    nl_group_t *g = new nl_group();
    g->n_frames = n_frames; g->width = width; g->height = height;
    for (int t = 0; t < n_tiles; t++) {
        int r0 = 0, nr = 0;
        nl_group_tile_rows(height, n_tiles, t, &r0, &nr);
        nl_stack_t *h = nl_stack_create(n_frames, width, height, r0, nr, devices ? devices[t] : t % ndev);
        if (!h) {
            nl_group_destroy(g);                          // destroy never touches the thread's error message
            return nullptr;
        }
        g->tiles.push_back(h);
        g->row0.push_back(r0);
        g->rows.push_back(nr);
    }
    {
        std::vector<int> seen;
        for (int t = 0; t < n_tiles; t++) {
            const int d = devices ? devices[t] : t % ndev;
            bool have = false;
            for (int s : seen) have = have || s == d;
            if (!have) seen.push_back(d);
        }
        g->distinct_devices = (int)seen.size();
    }
    return g;
}

int nl_group_size(nl_group_t *g) { return g ? (int)g->tiles.size() : 0; }

nl_stack_t *nl_group_tile(nl_group_t *g, int t)
{
    return (g && t >= 0 && t < (int)g->tiles.size()) ? g->tiles[(size_t)t] : nullptr;
}

}  // extern "C"

// Runs f(tile) for every tile, one host thread per tile beyond the first: staging a frame is a host memcpy
// into pinned memory per tile, and with one tile per device a serial loop would feed one DMA engine at a time.
// The first error (lowest tile) wins; its message becomes the calling thread's nl_last_error().
template <class F>
static int for_each_tile(nl_group_t *g, F &&f)
{
    const size_t n = g->tiles.size();
    std::vector<int> rc(n, NL_OK);
    std::vector<std::string> msg(n);
    auto one = [&](size_t t) {
        rc[t] = f(t);
        if (rc[t] != NL_OK) msg[t] = nl_last_error();        // (thread-local: fetch it on the worker)
    };
    std::vector<std::thread> workers;
    for (size_t t = 1; t < n; t++) workers.emplace_back(one, t);
    if (n > 0) one(0);
    for (auto &w : workers) w.join();
    for (size_t t = 0; t < n; t++)
        if (rc[t] != NL_OK) {
            nl::set_last_error(msg[t].c_str());
            return rc[t];
        }
    return NL_OK;
}

// The finish of a group pass on worker threads: when the tiles sit on more than one device (the copies of the result rows then
// run over separate links); NL_GROUP_PARALLEL_FINISH=1 / 0 forces it on / off (the tests force it on one device).
static bool finish_in_parallel(const nl_group_t *g)
{
    if (const char *e = getenv("NL_GROUP_PARALLEL_FINISH")) return e[0] == '1';
    return g->distinct_devices > 1;
}

extern "C" {

// every tile copies its rows out of the caller's frame into its own pinned staging buffer
// (pointer not retained, cgo rules) and starts its DMA; nothing is awaited on the devices
int nl_group_upload_frame(nl_group_t *g, int idx, const float *host_frame)
{
    if (!g) return NL_ERR_INVALID_ARG;
    return for_each_tile(g, [&](size_t t) { return nl_stack_upload_frame_async(g->tiles[t], idx, host_frame); });
}

// F3 on the group: raw_host = the big-endian FITS payload of the WHOLE frame (width*height values); every tile
// takes the byte range of its rows (row-major: contiguous) through its pinned ring, decodes on its device
int nl_group_upload_frame_fits(nl_group_t *g, int idx, const void *raw_host, int bitpix, float bscale, float bzero,
                               float multiplier, float offset)
{
    if (!g || !raw_host) return NL_ERR_INVALID_ARG;
    const int bpv = bitpix < 0 ? -bitpix / 8 : bitpix / 8;
    return for_each_tile(g, [&](size_t t) {
        const char *part = static_cast<const char *>(raw_host) + (size_t)g->row0[t] * (size_t)g->width * (size_t)bpv;
        return nl_stack_upload_frame_fits_async(g->tiles[t], idx, part, bitpix, bscale, bzero, multiplier, offset);
    });
}

// F4 on the group: every tile receives the whole unaligned source frame and projects its own rows
int nl_group_upload_frame_projected(nl_group_t *g, int idx, const float *src_host, int src_w, int src_h,
                                    const float trans[6], float out_of_bounds, float multiplier, float offset)
{
    if (!g) return NL_ERR_INVALID_ARG;
    return for_each_tile(g, [&](size_t t) {
        return nl_stack_upload_frame_projected_async(g->tiles[t], idx, src_host, src_w, src_h, trans, out_of_bounds,
                                                     multiplier, offset);
    });
}

int nl_group_fill_synthetic(nl_group_t *g, uint64_t seed)
{
    if (!g) return NL_ERR_INVALID_ARG;
    for (size_t t = 0; t < g->tiles.size(); t++) {
        int rc = nl_stack_fill_synthetic(g->tiles[t], seed);
        if (rc != NL_OK) return rc;
    }
    return NL_OK;
}

int nl_group_set_active_frames(nl_group_t *g, int n)
{
    if (!g) return NL_ERR_INVALID_ARG;
    for (size_t t = 0; t < g->tiles.size(); t++) {
        int rc = nl_stack_set_active_frames(g->tiles[t], n);
        if (rc != NL_OK) return rc;
    }
    g->n_frames = n;
    return NL_OK;
}

int nl_group_set_weights(nl_group_t *g, const float *weights)
{
    if (!g) return NL_ERR_INVALID_ARG;
    for (size_t t = 0; t < g->tiles.size(); t++) {
        int rc = nl_stack_set_weights(g->tiles[t], weights);
        if (rc != NL_OK) return rc;
    }
    return NL_OK;
}

int nl_group_set_exact(nl_group_t *g, int on)
{
    if (!g) return NL_ERR_INVALID_ARG;
    for (size_t t = 0; t < g->tiles.size(); t++) {
        int rc = nl_stack_set_exact(g->tiles[t], on);
        if (rc != NL_OK) return rc;
    }
    return NL_OK;
}

// stack.go:142-210 over the devices: enqueue every tile's pass, then collect
int nl_group_run(nl_group_t *g, int mode, float sigma_low, float sigma_high, float ref_loc,
                 float *out_host, int64_t *clip_low, int64_t *clip_high)
{
    if (!g) return NL_ERR_INVALID_ARG;
    // A failing tile must not leave the other tiles' passes enqueued and pending: every pass that was
    // started is also finished, and the FIRST error (with its message) is what the caller gets.
    int first_rc = NL_OK;
    std::string first_msg;
    auto note = [&](int rc) {
        if (rc != NL_OK && first_rc == NL_OK) { first_rc = rc; first_msg = nl_last_error(); }
    };
    size_t started = 0;
    for (; started < g->tiles.size() && first_rc == NL_OK; started++)
        note(nl_stack_run_async(g->tiles[started], mode, sigma_low, sigma_high, ref_loc));
    if (first_rc != NL_OK) started--;                     // (nl_stack_run_async settles a handle whose pass failed: nothing of it is in flight)
    int64_t lo = 0, hi = 0;
    if (first_rc == NL_OK && finish_in_parallel(g)) {
        // one host thread per tile: a tile's finish is a wait and the copy of its rows into the caller's (pageable) buffer --
        // 8 MiB of a 4096 x 4096 result per tile of eight, each over its own device's link
        std::vector<int64_t> l(g->tiles.size(), 0), h(g->tiles.size(), 0);
        const int rc = for_each_tile(g, [&](size_t t) { return nl_stack_finish(g->tiles[t], out_host, &l[t], &h[t]); });
        if (rc != NL_OK) return rc;                       // (for_each_tile finished every tile and kept the first message)
        for (size_t t = 0; t < g->tiles.size(); t++) { lo += l[t]; hi += h[t]; }
        if (clip_low) *clip_low = lo;
        if (clip_high) *clip_high = hi;
        return NL_OK;
    }
    for (size_t t = 0; t < started; t++) {
        int64_t l = 0, h = 0;
        note(nl_stack_finish(g->tiles[t], first_rc == NL_OK ? out_host : nullptr, &l, &h));
        lo += l;                                          // stack.go:193-198
        hi += h;
    }
    if (first_rc != NL_OK) {
        nl::set_last_error(first_msg.c_str());
        return first_rc;
    }
    if (clip_low) *clip_low = lo;
    if (clip_high) *clip_high = hi;
    return NL_OK;
}

int nl_group_last_mode(nl_group_t *g) { return (g && !g->tiles.empty()) ? nl_stack_last_mode(g->tiles[0]) : -1; }

// stackfindsigma.go:48-98 with the counters summed over the tiles after every pass
int nl_group_find_sigmas(nl_group_t *g, int mode, float ref_loc, float clip_perc_low, float clip_perc_high,
                         float *out_host, int64_t *clip_low, int64_t *clip_high,
                         float *sigma_low, float *sigma_high, int *passes)
{
    if (!g || g->tiles.empty()) return NL_ERR_INVALID_ARG;
    const int64_t total = (int64_t)g->width * g->height * (int64_t)g->n_frames;
    auto finish_all = [&](float *out) -> int {
        if (out)
            for (size_t t = 0; t < g->tiles.size(); t++) {
                int rc = nl_stack_finish(g->tiles[t], out, nullptr, nullptr);
                if (rc != NL_OK) return rc;
            }
        return NL_OK;
    };
    int m = mode;
    if (m == NL_ST_AUTO) {                                        // stack.go:45-55
        const int l = g->n_frames;
        m = l >= 25 ? NL_ST_LINEAR_FIT : (l >= 15 ? NL_ST_WINSOR_SIGMA : (l >= 6 ? NL_ST_SIGMA : NL_ST_MEAN));
    }
    int64_t lo = 0, hi = 0;
    int rc = NL_OK, n_pass = 0;
    if (m == NL_ST_SIGMA || m == NL_ST_WINSOR_SIGMA) {
        nl::SigmaBisection bis(clip_perc_low, clip_perc_high, total);
        for (;;) {                                            // stackfindsigma.go:48-98
            rc = nl_group_run(g, m, bis.low_mid, bis.high_mid, ref_loc, nullptr, &lo, &hi);
            if (rc != NL_OK) return rc;
            n_pass++;
            if (bis.step(lo, hi)) {
                if (clip_low) *clip_low = lo;
                if (clip_high) *clip_high = hi;
                if (sigma_low) *sigma_low = bis.low_mid;
                if (sigma_high) *sigma_high = bis.high_mid;
                if (passes) *passes = n_pass;
                return finish_all(out_host);
            }
        }
    }
    // stackfindsigma.go:40-46, 101-170: Newton's method for the linear fit; the other modes "do not
    // support sigmas" and are stacked once with 0, 0
    const bool newton = m == NL_ST_LINEAR_FIT;
    nl::SigmaNewton nw(clip_perc_low, total);
    for (;;) {
        rc = nl_group_run(g, m, newton ? nw.next_low() : 0.0f, newton ? nw.next_high() : 0.0f, ref_loc, nullptr, &lo, &hi);
        if (rc != NL_OK) return rc;
        n_pass++;
        const int st = newton ? nw.step(lo, hi) : 1;
        if (st == 0) continue;
        if (st == 2) {
            rc = nl_group_run(g, m, nw.sig_low, nw.sig_high, ref_loc, nullptr, nullptr, nullptr);
            if (rc != NL_OK) return rc;
        }
        if (clip_low) *clip_low = newton ? nw.base_lo : lo;
        if (clip_high) *clip_high = newton ? nw.base_hi : hi;
        if (sigma_low) *sigma_low = newton ? nw.sig_low : 0.0f;
        if (sigma_high) *sigma_high = newton ? nw.sig_high : 0.0f;
        if (passes) *passes = n_pass;
        return finish_all(out_host);
    }
}

// StackIncremental / StackIncrementalFinalize (stack.go:924-944) on every tile's device
int nl_group_accumulate(nl_group_t *g, float weight, int first)
{
    if (!g) return NL_ERR_INVALID_ARG;
    for (size_t t = 0; t < g->tiles.size(); t++) {
        int rc = nl_stack_accumulate(g->tiles[t], weight, first);
        if (rc != NL_OK) return rc;
    }
    return NL_OK;
}

int nl_group_accumulate_finalize(nl_group_t *g, float weight_sum, float *out_host)
{
    if (!g) return NL_ERR_INVALID_ARG;
    for (size_t t = 0; t < g->tiles.size(); t++) {
        int rc = nl_stack_accumulate_finalize(g->tiles[t], weight_sum, out_host);
        if (rc != NL_OK) return rc;
    }
    return NL_OK;
}

}  // extern "C"
