// rcf_api.cpp -- the C ABI of librcf.so (include/rcf.h): front-end state, stream bookkeeping and
// kernel scheduling.  One rcf_t == one SDR source of /root/reference/rc_frontend/receiver.py; every
// push/commit runs all open channels, the filterbank, the discriminators and an armed scan over the
// new block on the handle's HIP stream.
//
// Memory plan (sized for 288 GB of HBM3E): two wideband buffers [hist | block] ping-pong so the
// producer can fill the next block while kernels chew on the current one; the tail of every block is
// copied behind the other buffer's block as its history, which lets every kernel address the stream
// linearly (no wrap handling on the hot loads).  Narrowband outputs live in per-channel power-of-two
// rings addressed by the absolute output index.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <unordered_map>
#include <memory>
#include <mutex>
#include <vector>

#include "rcf_internal.h"

namespace rcfx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return false;
}

static const double kTwoPi = 6.283185307179586476925286766559;

static inline int64_t ceil_div(int64_t a, int64_t b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }
static inline int64_t floor_div(int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

static size_t pow2_at_least(size_t v)
{
    size_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct Chan {
    int id = -1;
    uint64_t many_stamp = 0;      // the rcf_chan_read_many call that last listed this channel
    int src = -1;                 // -1 wideband; RCF_SRC_PFB_BIN0 + bin; else source channel id
    int D = 0, T = 0;
    double src_rate = 0, offset_hz = 0;
    bool is_tap = false;          // a bin of a frame-major filterbank open as a channel: the bank's kernel copies it
                                  // into the launch's tap matrix, tap_finalize_kernel fills the rings
    std::vector<float> proto;     // prototype taps (host)
    float2 *d_ctaps = nullptr;
    uint64_t taps_version = 0;    // bumped whenever d_ctaps changes (bank-matrix cache key)
    float2 *d_iq = nullptr;
    float *d_fm = nullptr;
    int64_t start_sample = 0;     // in source index space
    int64_t k_abs0 = 0;
    int64_t produced = 0;         // relative output count
    int64_t rd_iq = 0, rd_fm = 0;
    // optional real FIR over gain * fm (P25 symbol filter)
    float *d_sym = nullptr, *d_symtaps = nullptr;
    int sym_ntaps = 0;
    float sym_gain = 1.f;
    int64_t sym_from = 0;         // first relative output index the filter is defined for
    int64_t rd_sym = 0;
    // analog voice chain (rcf_chan_audio_open)
    struct Audio {
        AudioState *d_state = nullptr;
        float *d_rings = nullptr;       // a | l | h | o | c (cf32), out_cap samples each
        float *d_taps = nullptr;        // lpf | hpf | rs (padded)
        int n_lpf = 0, n_hpf = 0, nt_rs = 0, interp = 1, decim = 1;
        float gain = 1.f;
        double thr = 0, alpha = 0, b0 = 1, b1 = 0, fb1 = 0;
        int64_t from = 0;               // first relative channel output the chain consumes
        int64_t rd = 0;                 // audio samples handed to the reader
    };
    std::unique_ptr<Audio> audio;
    // exact rotator (rcf_set_rotator): phase ring + {phase, counter} state, one pool slice; incr = what GNU Radio iterates
    float2 *d_rot = nullptr;
    float incr[2] = {1.f, 0.f};
    // rotator model
    double extra_dangle = 0, extra_dlogmag = 0;   // added to the increment's own angle / log magnitude (filterbank taps)
    double dangle = 0, dlogmag = 0;
    long double angle0 = 0;
    double logmag0 = 0;
    int64_t n_seg0 = 0;
    int depth = 0;
    // output range [blk_before, blk_after) the block with serial blk_serial gave this channel (its derived channels'
    // input range; process_block)
    uint64_t blk_serial = 0;
    int64_t blk_before = 0, blk_after = 0;
};

struct Pfb {
    bool open = false;
    bool frame_major = false;      // output ring layout (PfbLaunch.frame_major)
    int NB = 0, D = 0, T = 0, P = 0, Ppad = 0;
    std::vector<float> proto;      // prototype taps (host): rcf_pfb_tap_open's GNU-Radio phase model needs them
    float *d_ptaps = nullptr;
    float2 *d_tw = nullptr;
    float2 *d_bins = nullptr;
    float2 *d_stage = nullptr;     // frame-major banks: contiguous staging for rcf_pfb_read_bin
    std::vector<int64_t> rd;       // per-bin read cursors
    int64_t start_sample = 0, n_abs0 = 0, produced = 0;
    int64_t produced_before = 0;   // value of `produced` before the current commit (for derived channels)
};

struct Scan {
    bool armed = false, done = false;
    int N = 0, n_frames = 0, L = 0, R = 0, chunk = 0;
    int frames_done = 0;
    int64_t start_sample = 0;
    float *d_window = nullptr, *d_vring = nullptr, *d_sum = nullptr, *d_out = nullptr;
    float2 *d_tw = nullptr, *d_scratch = nullptr;
    int64_t *d_peaks = nullptr;
    void *d_peak_ws = nullptr;
};

}  // namespace rcfx

using namespace rcfx;

struct rcf {
    int device = 0;
    double fs = 0, fc = 0;
    size_t block_cap = 0, hist_cap = 0, out_cap = 0;
    uint64_t blk_serial = 0;       // process_block count (Chan::blk_serial)
    uint64_t ring_mask = 0;
    hipStream_t stream = nullptr;
    float2 *d_buf[2] = {nullptr, nullptr};
    int cur = 0;
    int64_t total_in = 0;
    double shift_hz = 0;          // accumulated rcf_source_shift
    float *d_atan = nullptr;
    float *d_level = nullptr;     // rcf_chan_fm_level result
    void *d_raw = nullptr;        // wire-format staging (rcf_push_raw), block_cap * 4 bytes, lazily allocated
    // launch-parameter arenas (pinned host + device), double buffered
    size_t arena_cap = 8u << 20;
    unsigned char *h_arena[2] = {nullptr, nullptr};
    unsigned char *h_arena_dev[2] = {nullptr, nullptr};   // the same pinned memory as the device sees it
    bool copy_kernels = true;     // RCF_COPY_KERNELS=0: hipMemcpyAsync for the launch records and the history (A/B)
    unsigned char *d_arena[2] = {nullptr, nullptr};
    hipEvent_t arena_ev[2] = {nullptr, nullptr};
    bool arena_used[2] = {false, false};
    int arena_cur = 0;
    size_t arena_fill = 0;        // bytes of the current arena taken by earlier commits (records are appended: the event
                                  // that guards an arena's reuse is recorded when it is LEFT, not once per commit --
                                  // every hipEventRecord costs ~6 us of queue gap, rocprof trace of the timed configuration)
    std::map<int, std::unique_ptr<Chan>> chans;
    int next_id = 1;
    Pfb pfb;
    Scan scan;
    // bank matrices of the matrix-core FIR path, one per (D, T) class, rebuilt when membership or taps change
    struct BankCache { std::vector<std::pair<int, uint64_t>> key; float *d = nullptr; size_t cap = 0; };
    std::map<std::pair<int, int>, BankCache> banks;
    uint64_t taps_clock = 0;
    // device buffers to release once the stream is idle: (pointer, pool slice bytes; 0 = plain hipFree)
    std::vector<std::pair<void *, size_t>> graveyard;
    // Channel buffers (rings, composite taps) come from slabs cut into equal slices, one pool per slice size:
    // opening a channel is a free-list pop instead of three hipMalloc + two memsets, closing one returns the
    // slices once the stream has passed them (create / release is what the reference's own self-test times,
    // frontend_connector.py:242-251)
    struct SlicePool { std::vector<void *> slabs, free_; };
    std::map<size_t, SlicePool> pools;
    std::map<int, std::vector<float>> proto_cache;   // channel_rate -> low_pass_2 prototype (rcf_chan_open)
    // H2D of block n+1 runs on its own stream while block n's kernels run (push_iq / push_raw)
    hipStream_t copy_stream = nullptr;
    hipEvent_t buf_done[2] = {nullptr, nullptr};   // the kernels that read d_buf[i] have finished
    hipEvent_t copy_ev = nullptr, raw_done = nullptr;
    bool buf_done_set[2] = {false, false};
    bool buf_dirty[2] = {false, false};   // kernels that read d_buf[i] were queued after buf_done[i] was last recorded
    bool eager_buf_done = false;          // a handle that is fed by rcf_push_iq records buf_done after every block (the
                                          // next block's copy overlaps this block's kernels); one fed in place
                                          // (rcf_ingest_ptr / rcf_commit) has no copy to order and records nothing
    bool raw_done_set = false;
    // RCCL communicator for the peak-list all-gather (rcf_comm_init); librccl is dlopen'ed on first use
    void *comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    int64_t *d_gather = nullptr;
    size_t gather_cap = 0;
    // rcf_chan_read_many: pinned staging the gather kernel writes (and reads its records from) across PCIe
    unsigned char *h_many = nullptr, *h_many_dev = nullptr;
    size_t many_cap = 0;
    uint64_t many_stamp = 0;
    // optional per-kernel-class HIP-event timing (rcf_timing_*)
    bool timing = false;
    unsigned timing_mask = ~0u;
    int mfma_min = 8;             // fewest channels of a class worth a matrix-core launch (RCF_FIR_MFMA_MIN)
    int mfma_nt = 0, mfma_parts = 0;   // RCF_FIR_MFMA_NT / RCF_FIR_MFMA_PARTS: override the launch plan (measurements)
    bool exact_rot = false;       // rcf_set_rotator / RCF_ROTATOR=exact: channels iterate GNU Radio's float32 rotator
    int decim_rule = RCF_DECIM_EXACT;   // rcf_set_decim_rule / RCF_DECIM_FLOOR=1
    uint64_t plan_calls = 0;            // blocks planned so far (RCF_FAIL_PLAN_AT)
    float2 *d_tapmat = nullptr;   // filterbank taps: the current launch's compact tap matrix (PfbLaunch::tap_mat)
    size_t tapmat_cap = 0;        // in float2
    float2 *d_partial = nullptr;  // split-K slabs of the matrix-core bank
    size_t partial_cap = 0;       // in float2
    bool no_mfma = false;         // RCF_FIR_NOMFMA=1: keep the vector-FMA bank kernel (A/B measurements)
    struct TimeRec { int what; hipEvent_t a, b; };
    std::vector<TimeRec> time_pending;
    std::vector<hipEvent_t> time_pool;
    unsigned timing_stride = 1;   // rcf_timing_stride: events around every n-th launch of a class only
    unsigned time_seen[RCF_T_COUNT] = {0};
    double time_ms[RCF_T_COUNT] = {0};
    int64_t time_n[RCF_T_COUNT] = {0};
    std::mutex mu;
};

namespace {

int set_dev(rcf_t *h)
{
    RCF_HIP(hipSetDevice(h->device));
    return RCF_OK;
}

void bury(rcf_t *h, void *p, size_t slice = 0)
{
    if (p) h->graveyard.push_back({p, slice});
}

// the stream is known to be idle (the caller just synchronised it): buried buffers can go
void free_graveyard_idle(rcf_t *h)
{
    for (auto &e : h->graveyard) {
        if (e.second) h->pools[e.second].free_.push_back(e.first);
        else (void)hipFree(e.first);
    }
    h->graveyard.clear();
}

void drain_graveyard(rcf_t *h)
{
    if (h->graveyard.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    free_graveyard_idle(h);
}

size_t slice_round(size_t bytes) { return (bytes + 255) & ~size_t(255); }

// one slice of `bytes` (a multiple of 256) from the handle's pools; nullptr + error set on failure
void *pool_get(rcf_t *h, size_t bytes)
{
    rcf::SlicePool &p = h->pools[bytes];
    if (p.free_.empty()) {
        size_t n = (size_t(64) << 20) / bytes;               // ~64 MiB slabs
        n = std::max<size_t>(1, std::min<size_t>(n, 512));
        void *slab = nullptr;
        if (!hip_ok(hipMalloc(&slab, n * bytes), "hipMalloc(channel slab)")) return nullptr;
        p.slabs.push_back(slab);
        for (size_t i = n; i-- > 0;) p.free_.push_back(static_cast<unsigned char *>(slab) + i * bytes);
    }
    void *r = p.free_.back();
    p.free_.pop_back();
    return r;
}

hipEvent_t time_event(rcf_t *h)
{
    hipEvent_t e = nullptr;
    if (!h->time_pool.empty()) { e = h->time_pool.back(); h->time_pool.pop_back(); return e; }
    (void)hipEventCreate(&e);
    return e;
}

struct Timed {   // RAII: brackets the launches issued in its scope with two events on the stream
    rcf_t *h; int what; hipEvent_t a = nullptr;
    Timed(rcf_t *h_, int what_) : h(h_), what(what_)
    {
        if (h->timing && (h->timing_mask >> what & 1u) && (h->time_seen[what]++ % h->timing_stride) == h->timing_stride - 1) {   // the LAST of each group: never the first launch after a sync
            a = time_event(h);
            (void)hipEventRecord(a, h->stream);
        }
    }
    ~Timed()
    {
        if (!a) return;
        hipEvent_t b = time_event(h);
        (void)hipEventRecord(b, h->stream);
        h->time_pending.push_back({what, a, b});
    }
};

// The filterbank's launch is timed with the events ATTACHED to its dispatch (PfbLaunch::ev_start / ev_stop) instead of a
// bracket of two event records: one barrier packet less inside the measured interval (bracket 102.9 us, attached
// 101.1-102.2 on one box; rocprofv3's kernel trace reads another 2.5-5 us less).  RCF_TIMING_BRACKET=1 keeps the bracket.
struct TimedAttached {
    rcf_t *h; int what; PfbLaunch &pl; hipEvent_t a = nullptr, b = nullptr; bool bracket = false;
    TimedAttached(rcf_t *h_, int what_, PfbLaunch &pl_) : h(h_), what(what_), pl(pl_)
    {
        static const bool use_bracket = [] { const char *e = getenv("RCF_TIMING_BRACKET"); return e && atoi(e) != 0; }();
        pl.ev_start = pl.ev_stop = nullptr;
        if (h->timing && (h->timing_mask >> what & 1u) && (h->time_seen[what]++ % h->timing_stride) == h->timing_stride - 1) {
            a = time_event(h);
            bracket = use_bracket;
            if (bracket) { (void)hipEventRecord(a, h->stream); }
            else { b = time_event(h); pl.ev_start = a; pl.ev_stop = b; }
        }
    }
    ~TimedAttached()
    {
        pl.ev_start = pl.ev_stop = nullptr;
        if (!a) return;
        if (bracket) { b = time_event(h); (void)hipEventRecord(b, h->stream); }
        h->time_pending.push_back({what, a, b});
    }
};

void time_collect(rcf_t *h)
{
    for (auto &r : h->time_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { h->time_ms[r.what] += ms; h->time_n[r.what] += 1; }
        h->time_pool.push_back(r.a);
        h->time_pool.push_back(r.b);
    }
    h->time_pending.clear();
}

// source description for one commit
struct SrcRange {
    StreamView view;
    int64_t p0, p1;      // new samples [p0, p1) in the source's index space
};

bool source_range(rcf_t *h, int src, int64_t S0, int64_t S1, SrcRange *out)
{
    if (src < 0) {
        out->view.base = h->d_buf[h->cur];
        out->view.mask = ~0ull;
        out->view.origin = S0 - (int64_t)h->hist_cap;
        out->view.stride = 1;
        out->p0 = S0;
        out->p1 = S1;
        return true;
    }
    if (src >= RCF_SRC_PFB_BIN0) {
        if (!h->pfb.open) return false;
        const int bin = src - RCF_SRC_PFB_BIN0;
        if (h->pfb.frame_major) {                           // bins_ring[i NB + bin]
            out->view.base = h->pfb.d_bins + bin;
            out->view.stride = h->pfb.NB;
            out->view.tshift = 0;
        } else {                                            // bins_ring[(i >> 4) tile_pitch + 16 bin + (i & 15)]
            out->view.base = h->pfb.d_bins + ((size_t)bin << kPfbTileLog2);
            out->view.stride = pfb_tile_pitch(h->pfb.NB);
            out->view.tshift = kPfbTileLog2;
        }
        out->view.mask = h->ring_mask;
        out->view.origin = 0;
        out->p0 = h->pfb.produced_before;
        out->p1 = h->pfb.produced;
        return true;
    }
    return false;   // channel-sourced: resolved by the caller (needs per-commit bookkeeping)
}

int upload_composite(rcf_t *h, Chan *c)
{
    std::vector<float> ct;
    float incr[2];
    // rcf_source_shift moves every signal of the source by -shift at baseband: the wideband channels' NCOs follow,
    // and so do the channels fed by filterbank bins (same Hz, at the bin rate)
    const bool shifted = c->src < 0 || c->src >= RCF_SRC_PFB_BIN0;
    design_composite(c->proto.data(), c->T, c->D, c->offset_hz + (shifted ? h->shift_hz : 0.0), c->src_rate,
                     ct, incr);
    const size_t slice = slice_round(sizeof(float2) * (size_t)c->T);
    float2 *fresh = static_cast<float2 *>(pool_get(h, slice));
    if (!fresh) return RCF_ENOMEM;
    if (!hip_ok(hipMemcpy(fresh, ct.data(), sizeof(float2) * (size_t)c->T, hipMemcpyHostToDevice), "hipMemcpy(taps)")) {
        h->pools[slice].free_.push_back(fresh);
        return RCF_EHIP;
    }
    bury(h, c->d_ctaps, slice);
    c->d_ctaps = fresh;
    c->taps_version = ++h->taps_clock;
    // GR iterates phase *= incr in float32; model it by the increment's actual angle and magnitude
    c->incr[0] = incr[0];
    c->incr[1] = incr[1];
    c->dangle = std::atan2((double)incr[1], (double)incr[0]) + c->extra_dangle;
    c->dlogmag = std::log(std::hypot((double)incr[0], (double)incr[1])) + c->extra_dlogmag;
    return RCF_OK;
}

int new_channel(rcf_t *h, int src, int D, const float *taps, int T, double offset_hz, int *chan_id)
{
    if (D < 1 || T < 1 || !taps || !chan_id) { set_error("bad channel arguments"); return RCF_EINVAL; }
    std::unique_ptr<Chan> c(new Chan);
    c->src = src;
    c->D = D;
    c->T = T;
    c->offset_hz = offset_hz;
    c->proto.assign(taps, taps + T);
    if (src < 0) {
        c->src_rate = h->fs;
        c->start_sample = h->total_in;
        c->depth = 0;
        if ((size_t)(T - 1 + D) > h->hist_cap) { set_error("history capacity %zu < T-1+D", h->hist_cap); return RCF_ECAP; }
    } else if (src >= RCF_SRC_PFB_BIN0) {
        if (!h->pfb.open || src - RCF_SRC_PFB_BIN0 >= h->pfb.NB) { set_error("no such PFB bin"); return RCF_EINVAL; }
        c->src_rate = h->fs / h->pfb.D;
        c->start_sample = h->pfb.produced;
        c->depth = 1;
    } else {
        auto it = h->chans.find(src);
        if (it == h->chans.end()) { set_error("no such source channel %d", src); return RCF_ENOCHAN; }
        c->src_rate = it->second->src_rate / it->second->D;
        c->start_sample = it->second->produced;
        c->depth = it->second->depth + 1;
    }
    if (src >= 0 && (size_t)(T + D) * 2 > h->out_cap) { set_error("source ring too small for T=%d", T); return RCF_ECAP; }
    c->k_abs0 = ceil_div(c->start_sample, D);
    // iq ring + discriminator ring in one slice.  Not cleared: readers never go past `produced`, and every
    // kernel masks what lies before a channel's first output (GR zero history)
    const size_t ring_slice = slice_round(12 * h->out_cap);
    c->d_iq = static_cast<float2 *>(pool_get(h, ring_slice));
    if (!c->d_iq) return RCF_ENOMEM;
    c->d_fm = reinterpret_cast<float *>(c->d_iq + h->out_cap);
    if (h->exact_rot) {
        const size_t rot_slice = slice_round(sizeof(float2) * h->out_cap + 256);
        c->d_rot = static_cast<float2 *>(pool_get(h, rot_slice));
        if (!c->d_rot) { h->pools[ring_slice].free_.push_back(c->d_iq); return RCF_ENOMEM; }
        const float st0[4] = {1.0f, 0.0f, 0.0f, 0.0f};       // phase 1 + 0j, counter 0 (bit pattern of 0.0f)
        if (!hip_ok(hipMemcpy(c->d_rot + h->out_cap, st0, sizeof(st0), hipMemcpyHostToDevice), "hipMemcpy(rotator state)")) {
            h->pools[rot_slice].free_.push_back(c->d_rot);
            h->pools[ring_slice].free_.push_back(c->d_iq);
            return RCF_EHIP;
        }
    }
    int rc = upload_composite(h, c.get());
    if (rc != RCF_OK) {
        h->pools[ring_slice].free_.push_back(c->d_iq);      // never seen by the stream: straight back
        if (c->d_rot) h->pools[slice_round(sizeof(float2) * h->out_cap + 256)].free_.push_back(c->d_rot);
        return rc;
    }
    c->id = h->next_id++;
    *chan_id = c->id;
    h->chans[c->id] = std::move(c);
    return RCF_OK;
}

void free_channel(rcf_t *h, Chan *c)
{
    bury(h, c->d_ctaps, slice_round(sizeof(float2) * (size_t)c->T));
    bury(h, c->d_iq, slice_round(12 * h->out_cap));       // d_fm lives in the same slice
    bury(h, c->d_rot, slice_round(sizeof(float2) * h->out_cap + 256));
    c->d_rot = nullptr;
    bury(h, c->d_sym);
    bury(h, c->d_symtaps);
    if (c->audio) { bury(h, c->audio->d_state); bury(h, c->audio->d_rings); bury(h, c->audio->d_taps); c->audio.reset(); }
    c->d_sym = nullptr;
    c->d_symtaps = nullptr;
    c->d_ctaps = nullptr;
    c->d_iq = nullptr;
    c->d_fm = nullptr;
    (void)h;
}

struct Arena {
    unsigned char *h, *d;
    size_t used = 0, cap;
    template <class T>
    bool put(const std::vector<T> &v, const T **dev)
    {
        const size_t bytes = sizeof(T) * v.size();
        const size_t at = (used + 63) & ~size_t(63);
        if (at + bytes > cap) return false;
        std::memcpy(h + at, v.data(), bytes);
        *dev = reinterpret_cast<const T *>(d + at);
        used = at + bytes;
        return true;
    }
};

int choose_kt(int D, int T)
{
    int kt = (8192 - T) / D + 1;
    if (kt < 1) kt = 1;
    if (kt > 256) kt = 256;
    if (kt >= 8) kt &= ~7;
    return kt;
}

// ------------------------------------------------------------------ the per-block schedule
// Everything one commit schedules is built on the host first (a BlockPlan: launch records in the pinned arena, jobs per
// dependency depth), then uploaded with one copy and launched in dependency order.  process_block() is the sequence;
// the plan_*() functions below each build one part of it.
struct FirJob {
    FirLaunchDims dims; const ChanLaunch *dev; bool repack; const unsigned char *dirty;
    // bank-matrix cache entry to mark current once the pack launch has been queued (not before: an error
    // return in between must not leave a key that claims a matrix nobody built)
    rcf::BankCache *bc; std::vector<std::pair<int, uint64_t>> key;
};
struct DiscJob { const DiscLaunch *dev; int n; int max_n; };

struct BlockPlan {
    int64_t S0 = 0, S1 = 0;            // the block's samples [S0, S1)
    size_t n = 0;

    // how far back every consumer of a ring reaches beyond the block's new samples (a block's writes must not
    // overwrite what the same block's readers still need): derived channels T - 1 + D of source output, the
    // discriminator one sample, the symbol filter its taps.  Every ring reaches back 1 (the discriminator); only
    // sources of other channels and channels with a symbol filter reach further -- the map holds just those (with
    // 131072 plain wideband channels it stays empty: a std::map entry per channel and block was a tenth of the
    // host's schedule time).
    std::unordered_map<int, size_t> reach_x;                // source id (channel id / RCF_SRC_PFB_BIN0) -> samples
    int max_depth = 0;
    int min_d0 = 0;                    // smallest decimation among the channels on the wideband stream (0: none)
    size_t max_reach = 1;              // largest consumer reach of any ring (see reach_x) / voice-chain filter
    size_t arena_need = 0;
    int a = 0;                         // arena in use, and where this commit's records start in it
    size_t arena_base = 0;
    Arena ar{nullptr, nullptr, 0, 0};
    uint64_t serial = 0;               // Chan::blk_before / blk_after of this block carry it
    std::vector<std::vector<FirJob>> fir_by_depth;
    std::vector<DiscJob> disc_jobs;
    std::vector<FmFirLaunch> symf;     // symbol filters, all channels in one launch
    int symf_max_n = 0;
    std::vector<RotFill> rot_fills;    // exact rotator: one record per launched channel, one launch before the FIRs
    std::vector<TapLaunch> tap_list;   // filterbank taps: copied out by the bank's kernel, finished by tap_finalize
    std::vector<int32_t> tap_bins;
    std::vector<AudioLaunch> audf;     // analog voice chains, all channels in one set of launches
    int audf_max_n = 0;
    double audf_ratio = 0;
    int audf_num = 1, audf_den = 1;
    PfbLaunch pl{};
    bool run_pfb = false;
    const TapLaunch *d_tap_list = nullptr;
    const int32_t *d_group_bin0 = nullptr;   // per group of 16 tap slots: first bin of a run read straight from the ring, or -1
    std::vector<int32_t> tap_first_of_bin;
    std::vector<TapLaunch> tap_ordered;
    const RotFill *d_rot_fills = nullptr;
    const FmFirLaunch *d_symf = nullptr;
    const AudioLaunch *d_audf = nullptr;
    bool history_done = false;         // launch_plan copied the history tail together with the launch records

    size_t reach(int id) const
    {
        if (reach_x.empty()) return 1;
        auto it = reach_x.find(id);
        return it == reach_x.end() ? (size_t)1 : std::max<size_t>(1, it->second);
    }
};

// channels of one (depth, D, T) class collected for launching
struct ClassPlan {
    std::vector<ChanLaunch> launches;
    std::vector<Chan *> launched;
    std::vector<DiscLaunch> discs;
    int max_n = 0;
    bool shared_src = true;
};

// arena for this commit: sized for every channel's launch records before anything is scheduled, so the schedule
// cannot run out half way (it mutates channel state as it goes); also the consumers' reach and the deepest chain
int plan_arena(rcf_t *h, BlockPlan &bp)
{
    auto &reach_x = bp.reach_x;
    int &max_depth = bp.max_depth;
    size_t &arena_need = bp.arena_need;

    {
        // (one pass over the channel map for everything that needs one: at 196608 channels each pass is ~8 ms of
        // pointer chasing)
        size_t need = 4096;
        for (auto &kv : h->chans) {
            const Chan &c = *kv.second;
            need += 2 * sizeof(ChanLaunch) + sizeof(TapLaunch) + sizeof(DiscLaunch) + sizeof(RotFill) + 12 + 128;
            if (c.d_sym) need += sizeof(FmFirLaunch);
            if (c.audio) need += sizeof(AudioLaunch);
            max_depth = std::max(max_depth, c.depth);
            if (c.src < 0 && (bp.min_d0 == 0 || c.D < bp.min_d0)) bp.min_d0 = c.D;
            if (c.audio) bp.max_reach = std::max<size_t>(bp.max_reach, (size_t)std::max(std::max(c.audio->n_lpf, c.audio->n_hpf), c.audio->nt_rs));
            if (c.d_sym) { size_t &own = reach_x[c.id]; own = std::max<size_t>(own, std::max<size_t>(1, (size_t)c.sym_ntaps)); }
            if (c.src >= 0) {
                size_t &r = reach_x[c.src >= RCF_SRC_PFB_BIN0 ? RCF_SRC_PFB_BIN0 : c.src];
                r = std::max<size_t>(r, (size_t)(c.T - 1 + c.D));
                bp.max_reach = std::max(bp.max_reach, r);
            }
            if (c.d_sym) bp.max_reach = std::max<size_t>(bp.max_reach, (size_t)c.sym_ntaps);
        }
        need += 64 * (h->chans.size() / 4 + 64);              // per-class alignment slack
        arena_need = need;
        if (need > h->arena_cap) {
            RCF_HIP(hipStreamSynchronize(h->stream));
            size_t cap = h->arena_cap;
            while (cap < need) cap *= 2;
            for (int i = 0; i < 2; ++i) {
                unsigned char *nh = nullptr, *nd = nullptr;
                RCF_HIP(hipHostMalloc(&nh, cap, hipHostMallocDefault));
                RCF_HIP(hipMalloc(&nd, cap));
                (void)hipHostFree(h->h_arena[i]);
                (void)hipFree(h->d_arena[i]);
                h->h_arena[i] = nh;
                h->d_arena[i] = nd;
                h->arena_used[i] = false;
                void *dv = nullptr;
                h->h_arena_dev[i] = hipHostGetDevicePointer(&dv, nh, 0) == hipSuccess ? static_cast<unsigned char *>(dv) : nullptr;
                if (!h->h_arena_dev[i]) h->copy_kernels = false;
            }
            h->arena_cap = cap;
            h->arena_fill = 0;
        }
    }
    if (h->arena_fill + arena_need > h->arena_cap) {
        // this arena is full: everything queued so far may still read it -- one event now guards its reuse -- and the
        // other one must have been drained
        RCF_HIP(hipEventRecord(h->arena_ev[h->arena_cur], h->stream));
        h->arena_used[h->arena_cur] = true;
        h->arena_cur ^= 1;
        h->arena_fill = 0;
        if (h->arena_used[h->arena_cur]) RCF_HIP(hipEventSynchronize(h->arena_ev[h->arena_cur]));
    }
    bp.a = h->arena_cur;
    bp.arena_base = h->arena_fill;
    bp.ar = Arena{h->h_arena[bp.a], h->d_arena[bp.a], bp.arena_base, h->arena_cap};
    return RCF_OK;
}

// the filterbank's share of the block (derived channels need its new range)
int plan_pfb(rcf_t *h, BlockPlan &bp)
{
    const int64_t S0 = bp.S0, S1 = bp.S1;
    const size_t n = bp.n;
    auto &reach_x = bp.reach_x;
    PfbLaunch &pl = bp.pl;
    bool &run_pfb = bp.run_pfb;

    if (h->pfb.open) {
        Pfb &p = h->pfb;
        const int64_t n_lo = std::max(ceil_div(S0, p.D), p.n_abs0);
        const int64_t n_hi = floor_div(S1 - 1, p.D);
        p.produced_before = p.produced;
        if (n_hi >= n_lo) {
            const int64_t cnt = n_hi - n_lo + 1;
            if ((size_t)cnt + (reach_x.count(RCF_SRC_PFB_BIN0) ? reach_x[RCF_SRC_PFB_BIN0] : 0) > h->out_cap) {
                set_error("block yields %lld PFB frames (+%zu of history its stage-2 channels need) > ring capacity %zu",
                          (long long)cnt, reach_x.count(RCF_SRC_PFB_BIN0) ? reach_x[RCF_SRC_PFB_BIN0] : (size_t)0, h->out_cap);
                return RCF_ECAP;
            }
            pl.src.base = h->d_buf[h->cur];
            pl.src.mask = ~0ull;
            pl.src.origin = S0 - (int64_t)h->hist_cap;
            pl.src.stride = 1;
            pl.frame_major = p.frame_major ? 1 : 0;
            pl.ptaps = p.d_ptaps;
            pl.tw = p.d_tw;
            pl.bins_ring = p.d_bins;
            pl.ring_mask = h->ring_mask;
            pl.tile_pitch = pfb_tile_pitch(p.NB);
            pl.n_lo = n_lo;
            pl.n_abs0 = p.n_abs0;
            pl.start_sample = p.start_sample;
            pl.src_len = (int64_t)(h->hist_cap + n);
            pl.n_frames = (int32_t)cnt;
            pl.NB = p.NB; pl.D = p.D; pl.P = p.P;
            run_pfb = true;
            p.produced = n_hi - p.n_abs0 + 1;
        }
    }
    return RCF_OK;
}

// one channel's launch records (FIR / tap, discriminator, symbol filter, voice chain, exact rotator) and the advance of
// its state.  Returns RCF_OK also when the channel has nothing to do in this block.
int plan_channel(rcf_t *h, BlockPlan &bp, ClassPlan &cp, Chan *c, int D)
{
    const int64_t S0 = bp.S0, S1 = bp.S1;
    const uint64_t serial = bp.serial;
    auto &launches = cp.launches;
    auto &launched = cp.launched;
    auto &discs = cp.discs;
    int &max_n = cp.max_n;
    bool &shared_src = cp.shared_src;
    auto &rot_fills = bp.rot_fills;
    auto &tap_list = bp.tap_list;
    auto &tap_bins = bp.tap_bins;
    auto &symf = bp.symf;
    int &symf_max_n = bp.symf_max_n;
    auto &audf = bp.audf;
    int &audf_max_n = bp.audf_max_n;
    double &audf_ratio = bp.audf_ratio;
    int &audf_num = bp.audf_num, &audf_den = bp.audf_den;
    auto reach = [&](int id) { return bp.reach(id); };

    SrcRange sr{};
    if (c->src >= 0 && c->src < RCF_SRC_PFB_BIN0) {
        auto it = h->chans.find(c->src);
        if (it == h->chans.end()) return RCF_OK;            // source closed: channel starves
        const Chan &sc_ = *it->second;
        const bool fresh = sc_.blk_serial == serial;
        sr.view.base = sc_.d_iq;
        sr.view.mask = h->ring_mask;
        sr.view.origin = 0;
        sr.view.stride = 1;
        sr.p0 = fresh ? sc_.blk_before : sc_.produced;
        sr.p1 = fresh ? sc_.blk_after : sc_.produced;
    } else if (!source_range(h, c->src, S0, S1, &sr)) {
        return RCF_OK;
    }
    if (c->src >= 0) shared_src = false;
    const int64_t k_lo = std::max(ceil_div(sr.p0, D), c->k_abs0);
    const int64_t k_hi = floor_div(sr.p1 - 1, D);
    const int64_t before = c->produced;
    if (sr.p1 <= sr.p0 || k_hi < k_lo) { c->blk_serial = serial; c->blk_before = c->blk_after = before; return RCF_OK; }
    const int64_t cnt = k_hi - k_lo + 1;
    if ((size_t)cnt + reach(c->id) > h->out_cap) {
        set_error("block yields %lld outputs (+%zu of history its consumers need) > ring capacity %zu",
                  (long long)cnt, reach(c->id), h->out_cap);
        return RCF_ECAP;
    }
    ChanLaunch L{};
    L.ctaps = c->d_ctaps;
    L.fm_ring = c->d_fm;
    L.iq_ring = c->d_iq;
    L.src = sr.view;
    L.k_lo = k_lo;
    L.k_abs0 = c->k_abs0;
    L.start_sample = c->start_sample;
    L.n_seg0 = c->n_seg0;
    L.angle0 = (double)c->angle0;
    L.dangle = c->dangle;
    L.logmag0 = c->logmag0;
    L.dlogmag = c->dlogmag;
    L.n_k = (int32_t)cnt;
    // exact rotator: plain channels only (a filterbank tap's rotator carries the bank's own phases too)
    if (c->d_rot && !c->is_tap && c->extra_dangle == 0.0 && c->extra_dlogmag == 0.0) {
        L.rot_ring = c->d_rot;
        L.rot_mask = h->ring_mask;
        RotFill rf{};
        rf.ring = c->d_rot;
        rf.state = reinterpret_cast<float *>(c->d_rot + h->out_cap);
        rf.n_from = k_lo - c->k_abs0;
        rf.n_k = (int32_t)cnt;
        rf.incr_re = c->incr[0];
        rf.incr_im = c->incr[1];
        rot_fills.push_back(rf);
    }
    DiscLaunch dl{};
    dl.iq_ring = c->d_iq;
    dl.fm_ring = c->d_fm;
    dl.n_lo = k_lo - c->k_abs0;
    dl.n_k = (int32_t)cnt;
    if (c->is_tap) {                            // served through the tap matrix, not by a FIR launch
        TapLaunch tl{};
        tl.iq_ring = c->d_iq;
        tl.fm_ring = c->d_fm;
        tl.k_lo = L.k_lo; tl.k_abs0 = L.k_abs0; tl.n_seg0 = L.n_seg0;
        tl.angle0 = L.angle0; tl.dangle = L.dangle; tl.logmag0 = L.logmag0; tl.dlogmag = L.dlogmag;
        tl.n_k = L.n_k;
        tl.bin = c->src - RCF_SRC_PFB_BIN0;
        tap_list.push_back(tl);
        tap_bins.push_back(tl.bin);
    } else {
        launches.push_back(L);
        launched.push_back(c);
        discs.push_back(dl);
        max_n = std::max(max_n, (int)cnt);
    }
    if (c->d_sym) {
        FmFirLaunch fl{};
        fl.fm_ring = c->d_fm;
        fl.sym_ring = c->d_sym;
        fl.taps = c->d_symtaps;
        fl.gain = c->sym_gain;
        fl.ntaps = c->sym_ntaps;
        fl.n_lo = std::max(dl.n_lo, c->sym_from);
        fl.n_first = c->sym_from;
        fl.n_k = (int32_t)(dl.n_lo + dl.n_k - fl.n_lo);
        if (fl.n_k > 0) symf.push_back(fl);
        symf_max_n = std::max(symf_max_n, (int)cnt);
    }
    if (c->audio) {
        Chan::Audio &au = *c->audio;
        AudioLaunch al{};
        al.iq_ring = c->d_iq;
        al.st = au.d_state;
        al.a_ring = au.d_rings;
        al.l_ring = au.d_rings + h->out_cap;
        al.h_ring = au.d_rings + 2 * h->out_cap;
        al.o_ring = au.d_rings + 3 * h->out_cap;
        al.c_ring = reinterpret_cast<float2 *>(au.d_rings + 4 * h->out_cap);
        al.lpf = au.d_taps;
        al.hpf = au.d_taps + au.n_lpf;
        al.rs = au.d_taps + au.n_lpf + au.n_hpf;
        al.n_lo = std::max(dl.n_lo, au.from);
        al.n_k = (int32_t)(dl.n_lo + dl.n_k - al.n_lo);
        al.n_lpf = au.n_lpf; al.n_hpf = au.n_hpf; al.nt_rs = au.nt_rs;
        al.interp = au.interp; al.decim = au.decim;
        al.gain = au.gain;
        al.thr = au.thr; al.alpha = au.alpha; al.b0 = au.b0; al.b1 = au.b1; al.fb1 = au.fb1;
        if (al.n_k > 0) {
            const size_t reach = (size_t)std::max(std::max(au.n_lpf, au.n_hpf), au.nt_rs);
            if ((size_t)al.n_k + reach > h->out_cap) {
                set_error("block yields %d channel samples: audio rings of %zu too small", al.n_k, h->out_cap);
                return RCF_ECAP;
            }
            audf.push_back(al);
            audf_max_n = std::max(audf_max_n, (int)al.n_k);
            if ((double)au.interp / au.decim > audf_ratio) {
                audf_ratio = (double)au.interp / au.decim; audf_num = au.interp; audf_den = au.decim;
            }
        }
    }
    // advance channel state: rebase the rotator model at the next output index
    const int64_t n_next = k_hi - c->k_abs0 + 1;
    const int64_t r512 = n_next & ~(int64_t)511;
    const long double adv = (long double)(n_next - c->n_seg0) * (long double)c->dangle;
    c->logmag0 = (r512 > c->n_seg0) ? (double)(n_next - r512) * c->dlogmag
                                    : c->logmag0 + (double)(n_next - c->n_seg0) * c->dlogmag;
    c->angle0 = fmodl(c->angle0 + adv, (long double)kTwoPi);
    c->n_seg0 = n_next;
    c->produced = n_next;
    c->blk_serial = serial; c->blk_before = before; c->blk_after = n_next;
    return RCF_OK;
}

// the launches of one (depth, D, T) class: matrix-core job (+ zero-history fix-ups), vector job, discriminator job
int plan_class_jobs(rcf_t *h, BlockPlan &bp, ClassPlan &cp, int depth, std::pair<int, int> cls_key)
{
    const int D = cls_key.first, T = cls_key.second;
    const size_t n = bp.n;
    Arena &ar = bp.ar;
    auto &fir_by_depth = bp.fir_by_depth;
    auto &disc_jobs = bp.disc_jobs;
    auto &launches = cp.launches;
    auto &launched = cp.launched;
    auto &discs = cp.discs;
    const int max_n = cp.max_n;
    const bool shared_src = cp.shared_src;

    if (launches.empty()) return RCF_OK;
    FirJob job{};
    job.dims.D = D; job.dims.T = T; job.dims.KT = choose_kt(D, T);
    job.dims.chans_per_wg = shared_src ? 16 : 1;
    job.dims.max_n_k = max_n;
    job.dims.ring_mask = h->ring_mask;
    job.dims.atan_tab = h->d_atan;
    // Matrix-core path: channels on one shared source with one common output range.  A channel that was just
    // opened still has outputs whose taps reach before its start (GR zero history) -- at most ceil((T-1)/D)
    // of them, four for the reference's shapes.  It joins the matrix-core launch anyway (which computes those
    // few outputs from real history, i.e. wrongly) and a vector-kernel launch AFTER it on the same stream
    // rewrites just those outputs with the per-tap mask: opening 16384 channels at once used to put one
    // whole block (70 ms) on the vector kernel.  Channels that start later inside the block keep the vector
    // kernel for that block.
    std::vector<ChanLaunch> clean, rest, fixups;
    std::vector<Chan *> clean_ch;
    int n_common_of_clean = max_n;
    if (shared_src && depth == 0 && mfma2_applicable(D, T, h->hist_cap, h->hist_cap + h->block_cap) && !h->no_mfma) {
        int64_t k_common = -1;
        int32_t n_common = 0;
        for (auto &L : launches)                                   // the range most channels share: the earliest
            if (k_common < 0 || L.k_lo < k_common) { k_common = L.k_lo; n_common = L.n_k; }
        n_common_of_clean = n_common;
        size_t n_ok = 0;
        for (const ChanLaunch &L : launches) n_ok += (L.k_lo == k_common && L.n_k == n_common) ? 1 : 0;
        if (n_ok == launches.size()) {              // the steady state: the whole class, no record copied
            clean.swap(launches);
            clean_ch.swap(launched);
        } else {
            clean.reserve(n_ok);
            clean_ch.reserve(n_ok);
            for (size_t i = 0; i < launches.size(); ++i) {
                const ChanLaunch &L = launches[i];
                const bool ok = L.k_lo == k_common && L.n_k == n_common;
                if (!ok) { rest.push_back(L); continue; }
                clean.push_back(L);
                clean_ch.push_back(launched[i]);
            }
        }
        for (const ChanLaunch &L : clean)
            if (L.k_lo * D - L.start_sample < (int64_t)(T - 1)) {
                // outputs k with k D - (T-1) < start: k < ceil((start + T - 1) / D)
                const int64_t k_end = ceil_div(L.start_sample + (int64_t)(T - 1), D);
                ChanLaunch F = L;
                F.n_k = (int32_t)std::min<int64_t>(L.n_k, std::max<int64_t>(0, k_end - L.k_lo));
                if (F.n_k > 0) fixups.push_back(F);
            }
        // (no size limit on a class: every group of 32 channels has its own tap slab)
        if ((int)clean.size() < h->mfma_min) {
            rest.insert(rest.end(), clean.begin(), clean.end());   // (order within a vector launch is free)
            clean.clear();
            clean_ch.clear();
            fixups.clear();
        }
    } else {
        rest.swap(launches);
    }
    if (!clean.empty()) {
        FirJob mj = job;
        mj.bc = nullptr;
        rcf::BankCache &bc = h->banks[cls_key];
        std::vector<std::pair<int, uint64_t>> key;
        key.reserve(clean_ch.size());
        for (Chan *c : clean_ch) key.push_back({c->id, c->taps_version});
        mj.repack = key != bc.key;
        mj.dirty = nullptr;
        if (mj.repack) {
            // + one chunk of slack: the kernel prefetches one chunk past a group's last
            const size_t need = (size_t)((clean.size() + kM2Group - 1) / kM2Group) * bank2_group_floats(T) +
                                (size_t)kM2ChunkSteps * 1024;
            bool fresh = false;
            if (need > bc.cap) {
                // grow with headroom: a class that gains channels one by one must not reallocate each time
                const size_t want = std::max(need, bc.cap + bc.cap / 2);
                float *nd = nullptr;
                RCF_HIP(hipMalloc(&nd, sizeof(float) * want));
                bury(h, bc.d);
                bc.d = nd;
                bc.cap = want;
                fresh = true;
            }
            if (!fresh) {
                // rebuild only the groups of 32 whose membership or taps changed
                const size_t ng = (clean.size() + kM2Group - 1) / kM2Group;
                std::vector<unsigned char> dirty(ng, 0);
                for (size_t i = 0; i < key.size(); ++i)
                    if (i >= bc.key.size() || bc.key[i] != key[i]) dirty[i / kM2Group] = 1;
                if (bc.key.size() > key.size())                      // the class shrank: its last group lost rows
                    dirty[ng - 1] = 1;
                if (!ar.put(dirty, &mj.dirty)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
            }
            bc.key.clear();                           // stale until the pack launch below is queued
            mj.bc = &bc;
            mj.key = key;
        }
        mj.dims.n_chans = (int)clean.size();
        mj.dims.mfma = 1;
        mj.dims.chans_per_wg = 128;
        mj.dims.bank = bc.d;
        mj.dims.max_n_k = n_common_of_clean;
        mj.dims.src_len = (int64_t)(h->hist_cap + n);
        {
            const MfmaPlan plan = mfma_plan((int)clean.size(), n_common_of_clean, T, h->mfma_nt, h->mfma_parts);
            mj.dims.mfma_nt = plan.nt;
            mj.dims.mfma_parts = plan.parts;
            mj.dims.partial = nullptr;
            if (plan.parts > 1) {
                const size_t need = (size_t)plan.parts * clean.size() * (size_t)n_common_of_clean;
                if (need > h->partial_cap) {
                    float2 *np_ = nullptr;
                    RCF_HIP(hipMalloc(&np_, sizeof(float2) * need));
                    bury(h, h->d_partial);
                    h->d_partial = np_;
                    h->partial_cap = need;
                }
                mj.dims.partial = h->d_partial;
            }
        }
        if (!ar.put(clean, &mj.dev)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
        fir_by_depth[depth].push_back(mj);
    }
    if (!fixups.empty()) {                          // queued behind the matrix-core launch: see above
        FirJob fj = job;
        fj.bc = nullptr;
        fj.repack = false;
        fj.dirty = nullptr;
        fj.dims.n_chans = (int)fixups.size();
        fj.dims.max_n_k = 0;
        for (auto &F : fixups) fj.dims.max_n_k = std::max(fj.dims.max_n_k, (int)F.n_k);
        fj.dims.small = 0;
        fj.dims.mfma = 0;
        fj.dims.chans_per_wg = 1;                   // per-channel n_k differ: one channel per workgroup
        if (!ar.put(fixups, &fj.dev)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
        fir_by_depth[depth].push_back(fj);
    }
    if (!rest.empty()) {
        job.dims.n_chans = (int)rest.size();
        job.dims.small = (!shared_src && fir_small_outputs(D, T) > 0) ? 1 : 0;
        if (!ar.put(rest, &job.dev)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
        fir_by_depth[depth].push_back(job);
    }
    if (!(job.dims.small && clean.empty())) {   // the small-T kernel writes the discriminator ring itself
        DiscJob dj{};
        dj.n = (int)discs.size(); dj.max_n = max_n;
        if (!ar.put(discs, &dj.dev)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
        disc_jobs.push_back(dj);
    }
    return RCF_OK;
}

// filterbank taps (matrix + records) and the records that go out as one launch each
int plan_tail(rcf_t *h, BlockPlan &bp)
{
    Arena &ar = bp.ar;
    auto &tap_list = bp.tap_list;
    auto &tap_bins = bp.tap_bins;
    auto &rot_fills = bp.rot_fills;
    auto &symf = bp.symf;
    auto &audf = bp.audf;
    PfbLaunch &pl = bp.pl;
    const bool run_pfb = bp.run_pfb;
    const TapLaunch *&d_tap_list = bp.d_tap_list;
    const RotFill *&d_rot_fills = bp.d_rot_fills;
    const FmFirLaunch *&d_symf = bp.d_symf;
    const AudioLaunch *&d_audf = bp.d_audf;

    if (!tap_list.empty() && run_pfb) {
        const size_t pitch = (tap_list.size() + 15) & ~size_t(15);      // slots, whole groups of 16
        // Slot order: first every aligned run of 16 bins that is tapped completely (tap_finalize reads those from the
        // bank's ring: PfbLaunch::tap_first), then the remaining taps, which go through the matrix.
        const int NB = pl.NB;
        auto &first = bp.tap_first_of_bin;
        first.assign((size_t)NB, -1);
        for (size_t i = 0; i < tap_list.size(); ++i)
            if (first[tap_list[i].bin] < 0) first[tap_list[i].bin] = (int32_t)i;
        auto &ordered = bp.tap_ordered;
        ordered.clear();
        ordered.reserve(tap_list.size());
        std::vector<int32_t> group_bin0;
        group_bin0.reserve(pitch / 16);
        for (int b0 = 0; b0 + 16 <= NB; b0 += 16) {
            bool full = true;
            for (int j = 0; j < 16 && full; ++j) full = first[b0 + j] >= 0;
            if (!full) continue;
            for (int j = 0; j < 16; ++j) {
                ordered.push_back(tap_list[first[b0 + j]]);
                tap_list[first[b0 + j]].bin = -1;               // taken
            }
            group_bin0.push_back(b0);
        }
        pl.tap_first = (int32_t)ordered.size();
        for (const TapLaunch &t : tap_list)
            if (t.bin >= 0) ordered.push_back(t);
        tap_list.swap(ordered);
        group_bin0.resize(pitch / 16, -1);
        for (size_t i = 0; i < tap_list.size(); ++i) tap_bins[i] = tap_list[i].bin;
        if (!ar.put(tap_bins, &pl.tap_bins) || !ar.put(tap_list, &d_tap_list) || !ar.put(group_bin0, &bp.d_group_bin0)) {
            set_error("launch arena exhausted");
            return RCF_ENOMEM;
        }
        // the matrix holds the slots from tap_first on and nothing else (every bin of a 1600-bin bank tapped, 2^25-sample
        // blocks: no matrix at all instead of 537 MB of it)
        const size_t mat_pitch = pitch - (size_t)pl.tap_first;
        const size_t need = mat_pitch * (size_t)pl.n_frames;
        if (need > h->tapmat_cap) {
            float2 *nm = nullptr;
            RCF_HIP(hipMalloc(&nm, sizeof(float2) * need));
            bury(h, h->d_tapmat);
            h->d_tapmat = nm;
            h->tapmat_cap = need;
        }
        pl.tap_mat = h->d_tapmat;
        pl.tap_pitch = (int32_t)mat_pitch;
        pl.n_taps = (int32_t)tap_list.size();
    }
    if (!rot_fills.empty() && !ar.put(rot_fills, &d_rot_fills)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
    if (!symf.empty() && !ar.put(symf, &d_symf)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
    if (!audf.empty() && !ar.put(audf, &d_audf)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
    return RCF_OK;
}

// upload all launch parameters in one copy, then launch in dependency order
int launch_plan(rcf_t *h, BlockPlan &bp)
{
    hipStream_t st = h->stream;
    Arena &ar = bp.ar;
    const int a = bp.a;
    const size_t arena_base = bp.arena_base;
    auto &fir_by_depth = bp.fir_by_depth;
    auto &disc_jobs = bp.disc_jobs;
    auto &symf = bp.symf;
    auto &audf = bp.audf;
    auto &rot_fills = bp.rot_fills;
    PfbLaunch &pl = bp.pl;
    const bool run_pfb = bp.run_pfb;
    const TapLaunch *d_tap_list = bp.d_tap_list;
    const RotFill *d_rot_fills = bp.d_rot_fills;
    const FmFirLaunch *d_symf = bp.d_symf;
    const AudioLaunch *d_audf = bp.d_audf;
    const int symf_max_n = bp.symf_max_n, audf_max_n = bp.audf_max_n, audf_num = bp.audf_num, audf_den = bp.audf_den;

    {
        // the block's launch records host -> device, and -- in the same launch -- its history tail behind the OTHER input
        // buffer's block (nothing in this block reads that place, and the kernels that did read it are earlier in
        // the stream): one small launch per block instead of two
        const size_t from = arena_base & ~size_t(63);
        const size_t bytes = ar.used > arena_base ? ((ar.used + 63) & ~size_t(63)) - from : 0;
        static const bool merge = [] { const char *e = getenv("RCF_COPY_MERGE"); return !e || atoi(e) != 0; }();   // A/B
        // ... or none at all: when the filterbank's launch is the first of the block that needs neither (no direct
        // channels, no exact-rotator fill before it, no tap matrix whose slot list the bank itself reads from the
        // arena), its first workgroups do both copies on the way in (PfbLaunch::rider_*)
        static const bool ride_env = [] { const char *e = getenv("RCF_COPY_RIDE"); return !e || atoi(e) != 0; }();    // A/B
        const bool ride = ride_env && merge && h->copy_kernels && run_pfb && !d_rot_fills &&
                          (fir_by_depth.empty() || fir_by_depth[0].empty()) && pl.n_taps == pl.tap_first &&
                          bytes / 8 < (1u << 31) && h->hist_cap < (1u << 28) && pfb_takes_rider(pl);
        if (ride) {
            pl.rider_dst[0] = reinterpret_cast<unsigned long long *>(ar.d + from);
            pl.rider_src[0] = reinterpret_cast<const unsigned long long *>(h->h_arena_dev[a] + from);
            pl.rider_n8[0] = (uint32_t)((bytes + 7) / 8);
            pl.rider_dst[1] = reinterpret_cast<unsigned long long *>(h->d_buf[h->cur ^ 1]);
            pl.rider_src[1] = reinterpret_cast<const unsigned long long *>(h->d_buf[h->cur] + bp.n);
            pl.rider_n8[1] = (uint32_t)(sizeof(float2) * h->hist_cap / 8);
            bp.history_done = true;
        } else if (h->copy_kernels && !merge) {
            if (bytes) launch_copy8(ar.d + from, h->h_arena_dev[a] + from, bytes, st);
        } else if (h->copy_kernels) {
            Timed t(h, RCF_T_HISTORY);
            launch_copy8x2(ar.d + from, h->h_arena_dev[a] + from, bytes, h->d_buf[h->cur ^ 1], h->d_buf[h->cur] + bp.n,
                           sizeof(float2) * h->hist_cap, st);
            bp.history_done = true;
        } else if (bytes) {
            RCF_HIP(hipMemcpyAsync(ar.d + from, ar.h + from, bytes, hipMemcpyHostToDevice, st));
        }
        if (bytes) h->arena_fill = (ar.used + 63) & ~size_t(63);
    }
    if (d_rot_fills) launch_rot_fill(d_rot_fills, (int)rot_fills.size(), h->ring_mask, st);
    if (!fir_by_depth.empty())
        for (auto &j : fir_by_depth[0]) {
            if (j.repack) {
                launch_fir_pack(j.dev, j.dims.n_chans, j.dims.T, const_cast<float *>(j.dims.bank), j.dirty, st);
                if (j.bc) j.bc->key = std::move(j.key);
            }
            Timed t(h, j.dims.mfma ? RCF_T_FIR_MFMA : RCF_T_FIR);
            launch_fir_bank(j.dev, j.dims, st);
        }
    if (run_pfb) { TimedAttached t(h, RCF_T_PFB, pl); launch_pfb(pl, st); }
    if (run_pfb && pl.n_taps > 0) {
        Timed t(h, RCF_T_TAPS);
        launch_tap_finalize(d_tap_list, pl.n_taps, pl.tap_mat, pl.tap_pitch, pl.n_frames, pl.n_lo - pl.n_abs0,
                            h->ring_mask, h->d_atan, bp.d_group_bin0, pl.tap_first, pl.bins_ring, pl.NB, st);
    }
    for (size_t d = 1; d < fir_by_depth.size(); ++d)
        for (auto &j : fir_by_depth[d]) { Timed t(h, RCF_T_FIR_DERIVED); launch_fir_bank(j.dev, j.dims, st); }
    for (auto &dj : disc_jobs) {
        Timed t(h, RCF_T_DISC);
        launch_discriminator(dj.dev, dj.n, dj.max_n, h->ring_mask, h->d_atan, st);
    }
    if (d_symf) {
        Timed t(h, RCF_T_DISC);
        launch_fm_fir(d_symf, (int)symf.size(), symf_max_n, h->ring_mask, st);
    }
    if (d_audf) {
        Timed t(h, RCF_T_AUDIO);
        launch_audio(d_audf, (int)audf.size(), audf_max_n, audf_num, audf_den, h->ring_mask, h->d_atan, st);
    }
    return RCF_OK;
}

// the scan's share of the block: every frame that is complete now
int run_scan(rcf_t *h, const BlockPlan &bp)
{
    hipStream_t st = h->stream;
    const int64_t S0 = bp.S0, S1 = bp.S1;

    Scan &sc = h->scan;
    if (sc.armed && !sc.done) {
        int64_t avail = (S1 - sc.start_sample) / sc.N;
        if (avail > sc.n_frames) avail = sc.n_frames;
        while (sc.frames_done < avail) {
            const int cnt = (int)std::min<int64_t>(sc.chunk, avail - sc.frames_done);
            ScanLaunch sl{};
            sl.src.base = h->d_buf[h->cur];
            sl.src.mask = ~0ull;
            sl.src.origin = S0 - (int64_t)h->hist_cap;
            sl.src.stride = 1;
            sl.s0 = sc.start_sample + (int64_t)sc.frames_done * sc.N;
            sl.window = sc.d_window;
            sl.tw = sc.d_tw;
            sl.vring = sc.d_vring;
            sl.N = sc.N; sl.R = sc.R;
            sl.f0 = sc.frames_done; sl.n_frames = cnt;
            sl.scratch = sc.d_scratch;
            { Timed t(h, RCF_T_SCAN_FFT); launch_scan_fft(sl, st); }
            {
                Timed t(h, RCF_T_SCAN_MOVSUM);
                launch_scan_movsum(sc.d_vring, sc.N, sc.R, sc.L, sc.frames_done, cnt, sc.n_frames - 1, sc.d_sum,
                                   sc.d_out, st);
            }
            sc.frames_done += cnt;
        }
        if (sc.frames_done >= sc.n_frames) sc.done = true;
    }
    return RCF_OK;
}

// history for the next block, flip buffers
int finish_block(rcf_t *h, const BlockPlan &bp)
{
    hipStream_t st = h->stream;
    const size_t n = bp.n;
    const int64_t S1 = bp.S1;

    const int other = h->cur ^ 1;
    if (!bp.history_done) {
        Timed t(h, RCF_T_HISTORY);
        if (h->copy_kernels) launch_copy8(h->d_buf[other], h->d_buf[h->cur] + n, sizeof(float2) * h->hist_cap, st);
        else RCF_HIP(hipMemcpyAsync(h->d_buf[other], h->d_buf[h->cur] + n, sizeof(float2) * h->hist_cap,
                                    hipMemcpyDeviceToDevice, st));
    }
    if (h->eager_buf_done) {
        RCF_HIP(hipEventRecord(h->buf_done[h->cur], st));    // everything that reads this buffer is queued
        h->buf_done_set[h->cur] = true;
        h->buf_dirty[h->cur] = false;
    } else {
        h->buf_dirty[h->cur] = true;
    }
    h->cur = other;
    h->total_in = S1;
    RCF_HIP(hipGetLastError());
    return RCF_OK;
}

// plan_channel() advances a channel's state as it goes, so a block must not be refused half way through the schedule
// (the channels planned before the refusal would have counted outputs nobody computed).  The one refusal that depends on
// the block is a ring too small for what the block yields: this pass walks the channels in dependency order WITHOUT
// touching them and reports it first.  It only runs when the cheap bound in process_block() says a ring could overflow.
int check_block_capacity(rcf_t *h, const BlockPlan &bp)
{
    std::unordered_map<int, std::pair<int64_t, int64_t>> dry;      // channel id -> produced (before, after) this block
    for (int depth = 0; depth <= bp.max_depth; ++depth)
        for (auto &kv : h->chans) {
            const Chan &c = *kv.second;
            if (c.depth != depth) continue;
            int64_t p0, p1;
            if (c.src < 0) { p0 = bp.S0; p1 = bp.S1; }
            else if (c.src >= RCF_SRC_PFB_BIN0) {
                if (!h->pfb.open) continue;
                p0 = h->pfb.produced_before; p1 = h->pfb.produced;
            } else {
                auto sit = h->chans.find(c.src);
                if (sit == h->chans.end()) continue;
                auto dit = dry.find(c.src);
                p0 = dit == dry.end() ? sit->second->produced : dit->second.first;
                p1 = dit == dry.end() ? sit->second->produced : dit->second.second;
            }
            const int64_t k_lo = std::max(ceil_div(p0, c.D), c.k_abs0);
            const int64_t k_hi = floor_div(p1 - 1, c.D);
            if (p1 <= p0 || k_hi < k_lo) { dry[c.id] = {c.produced, c.produced}; continue; }
            const int64_t cnt = k_hi - k_lo + 1;
            if ((size_t)cnt + bp.reach(c.id) > h->out_cap) {
                set_error("block yields %lld outputs (+%zu of history its consumers need) > ring capacity %zu",
                          (long long)cnt, bp.reach(c.id), h->out_cap);
                return RCF_ECAP;
            }
            if (c.audio) {
                const Chan::Audio &au = *c.audio;
                const int64_t n_lo = k_lo - c.k_abs0;
                const int64_t a_lo = std::max<int64_t>(n_lo, au.from);
                const int64_t a_n = n_lo + cnt - a_lo;
                const size_t reach = (size_t)std::max(std::max(au.n_lpf, au.n_hpf), au.nt_rs);
                if (a_n > 0 && (size_t)a_n + reach > h->out_cap) {
                    set_error("block yields %lld channel samples: audio rings of %zu too small", (long long)a_n, h->out_cap);
                    return RCF_ECAP;
                }
            }
            dry[c.id] = {c.produced, k_hi - c.k_abs0 + 1};
        }
    return RCF_OK;
}

int process_block(rcf_t *h, size_t n)
{
    if (h->graveyard.size() > 512) drain_graveyard(h);     // bounded even if nobody ever syncs or reads
    BlockPlan bp;
    bp.S0 = h->total_in;
    bp.S1 = bp.S0 + (int64_t)n;
    bp.n = n;
    int rc = plan_arena(h, bp);
    if (rc == RCF_OK) rc = plan_pfb(h, bp);
    if (rc != RCF_OK) return rc;
    {
        // no ring can overflow when even the fastest channel's outputs of this block plus the longest reach fit
        size_t worst = bp.min_d0 ? n / (size_t)bp.min_d0 + 2 : 0;
        if (bp.run_pfb) worst = std::max(worst, (size_t)bp.pl.n_frames + 1);
        if (worst + bp.max_reach > h->out_cap && (rc = check_block_capacity(h, bp)) != RCF_OK) {
            if (h->pfb.open) h->pfb.produced = h->pfb.produced_before;      // plan_pfb had counted the block's frames
            return rc;
        }
    }
    // Planning advances every channel's counters and rotator model (plan_channel) and can still fail after that -- a
    // bank matrix, split-K slab or tap matrix that cannot be allocated, an exhausted launch arena.  Nothing has been
    // queued at that point: put the counters back, so that they never claim outputs nobody computed (and the exact
    // rotator's device state stays in step with them).
    struct Saved { Chan *c; int64_t produced, n_seg0, blk_before, blk_after; uint64_t blk_serial; long double angle0; double logmag0; };
    std::vector<Saved> saved;
    saved.reserve(h->chans.size());
    for (auto &kv : h->chans) {
        Chan *c = kv.second.get();
        saved.push_back(Saved{c, c->produced, c->n_seg0, c->blk_before, c->blk_after, c->blk_serial, c->angle0, c->logmag0});
    }
    const uint64_t serial_before = h->blk_serial;
    auto roll_back = [&](int code) {
        for (const Saved &s : saved) {
            s.c->produced = s.produced; s.c->n_seg0 = s.n_seg0; s.c->blk_before = s.blk_before; s.c->blk_after = s.blk_after;
            s.c->blk_serial = s.blk_serial; s.c->angle0 = s.angle0; s.c->logmag0 = s.logmag0;
        }
        if (h->pfb.open) h->pfb.produced = h->pfb.produced_before;
        h->blk_serial = serial_before;
        return code;
    };
    // channels, by depth then by (D, T) class
    bp.fir_by_depth.resize(bp.max_depth + 1);
    bp.serial = ++h->blk_serial;
    for (int depth = 0; depth <= bp.max_depth; ++depth) {
        std::map<std::pair<int, int>, std::vector<Chan *>> classes;
        for (auto &kv : h->chans)
            if (kv.second->depth == depth) classes[{kv.second->D, kv.second->T}].push_back(kv.second.get());
        for (auto &cls : classes) {
            ClassPlan cp;
            cp.launches.reserve(cls.second.size());
            cp.launched.reserve(cls.second.size());
            cp.discs.reserve(cls.second.size());
            for (Chan *c : cls.second)
                if ((rc = plan_channel(h, bp, cp, c, cls.first.first)) != RCF_OK) return roll_back(rc);
            if ((rc = plan_class_jobs(h, bp, cp, depth, cls.first)) != RCF_OK) return roll_back(rc);
        }
    }
    if ((rc = plan_tail(h, bp)) != RCF_OK) return roll_back(rc);
    {
        // RCF_FAIL_PLAN_AT=<k>: the k-th block of a handle fails here as an exhausted launch arena would (tests of the
        // roll-back above: tests/test_gpu_round4.py)
        static const long fail_at = [] { const char *e = getenv("RCF_FAIL_PLAN_AT"); return e ? atol(e) : 0L; }();
        if (fail_at > 0 && (long)++h->plan_calls == fail_at) {
            set_error("injected planning failure (RCF_FAIL_PLAN_AT)");
            return roll_back(RCF_ENOMEM);
        }
    }
    if ((rc = launch_plan(h, bp)) != RCF_OK) return rc;      // kernels may be queued: the handle's stream state is undefined now
    if ((rc = run_scan(h, bp)) != RCF_OK) return rc;
    return finish_block(h, bp);
}

// queue the copies of one ring read on the handle's stream; the caller synchronises and then advances *cursor by the
// count returned (ring_read does both for a single ring; rcf_chan_read_many batches many rings behind ONE sync)
int64_t ring_read_enqueue(rcf_t *h, const void *ring, size_t elem, int64_t produced, int64_t *cursor, void *out,
                          size_t max_items)
{
    int64_t avail = produced - *cursor;
    if (avail <= 0 || max_items == 0) return 0;
    if ((size_t)avail > h->out_cap) {           // reader lagged: oldest samples are gone
        *cursor = produced - (int64_t)h->out_cap;
        avail = (int64_t)h->out_cap;
    }
    const int64_t n = std::min<int64_t>(avail, (int64_t)max_items);
    const size_t pos = (size_t)((uint64_t)*cursor & h->ring_mask);
    const size_t first = std::min<size_t>((size_t)n, h->out_cap - pos);
    const unsigned char *r = static_cast<const unsigned char *>(ring);
    if (hipMemcpyAsync(out, r + pos * elem, first * elem, hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
        set_error("ring read failed");
        return RCF_EHIP;
    }
    if ((size_t)n > first &&
        hipMemcpyAsync(static_cast<unsigned char *>(out) + first * elem, r, ((size_t)n - first) * elem,
                       hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
        set_error("ring read failed");
        return RCF_EHIP;
    }
    return n;
}

int64_t ring_read(rcf_t *h, const void *ring, size_t elem, int64_t produced, int64_t *cursor, void *out,
                  size_t max_items)
{
    const int64_t n = ring_read_enqueue(h, ring, elem, produced, cursor, out, max_items);
    if (n <= 0) return n;
    if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("stream sync failed"); return RCF_EHIP; }
    free_graveyard_idle(h);       // retuned / closed channels' old buffers: every read is a chance to release them
    *cursor += n;
    return n;
}

// ------------------------------------------------------------------ RCCL (peak-list all-gather over xGMI)
// librccl.so is loaded on first use: a single-GPU front-end never pays for it.  Types are restated from rccl.h
// (ncclUniqueId = 128 opaque bytes passed BY VALUE, ncclInt64 = 4, ncclFloat64 = 8, ncclMax = 2).
struct RcclId { char internal[128]; };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

RcclApi *rccl()
{
    static std::mutex mu;
    static RcclApi api;
    static bool tried = false;
    std::lock_guard<std::mutex> g(mu);
    if (!tried) {
        tried = true;
        void *l = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!l) l = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (l) {
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(l, "ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(l, "ncclCommInitRank"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(l, "ncclCommDestroy"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(l, "ncclAllGather"));
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(l, "ncclAllReduce"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(l, "ncclGetErrorString"));
            if (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.AllReduce) api.lib = l;
        }
    }
    return api.lib ? &api : nullptr;
}

bool rccl_ok(RcclApi *r, int rc, const char *what)
{
    if (rc == 0) return true;
    set_error("RCCL error %d (%s) in %s", rc, r && r->GetErrorString ? r->GetErrorString(rc) : "?", what);
    return false;
}

void comm_destroy(rcf_t *h)
{
    if (!h->comm) return;
    if (RcclApi *r = rccl()) (void)r->CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_rank = 0;
    h->comm_size = 1;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int rcf_comm_unique_id(void *id128)
{
    if (!id128) { set_error("bad arguments"); return RCF_EINVAL; }
    RcclApi *r = rccl();
    if (!r) { set_error("librccl.so not available"); return RCF_ESTATE; }
    RcclId id;
    if (!rccl_ok(r, r->GetUniqueId(&id), "ncclGetUniqueId")) return RCF_EHIP;
    std::memcpy(id128, id.internal, sizeof(id.internal));
    return RCF_OK;
}

int rcf_comm_init(rcf_t *h, int rank, int n_ranks, const void *id128)
{
    if (!h || n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !id128)) {
        set_error("bad communicator arguments");
        return RCF_EINVAL;
    }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    comm_destroy(h);
    if (n_ranks == 1 && !id128) return RCF_OK;      // one front-end, no id: the gather is a local copy
    // (n_ranks == 1 WITH an id builds a real one-rank communicator: the same RCCL calls, on one GPU)
    RcclApi *r = rccl();
    if (!r) { set_error("librccl.so not available"); return RCF_ESTATE; }
    RcclId id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    void *comm = nullptr;
    if (!rccl_ok(r, r->CommInitRank(&comm, n_ranks, id, rank), "ncclCommInitRank")) return RCF_EHIP;
    h->comm = comm;
    h->comm_rank = rank;
    h->comm_size = n_ranks;
    return RCF_OK;
}

int rcf_comm_destroy(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    comm_destroy(h);
    return RCF_OK;
}

int rcf_comm_size(rcf_t *h) { return h ? h->comm_size : RCF_EINVAL; }

int rcf_allgather_peaks(rcf_t *h, const int64_t *mine, int n, int64_t *all, int cap, int *counts)
{
    if (!h || n < 0 || cap < 1 || (n && !mine) || !all || !counts) { set_error("bad all-gather arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    const int W = h->comm_size, keep = std::min(n, cap);
    if (!h->comm) {                                 // single rank
        counts[0] = keep;
        std::memcpy(all, mine, sizeof(int64_t) * (size_t)keep);
        return RCF_OK;
    }
    RcclApi *r = rccl();
    if (!r) { set_error("librccl.so not available"); return RCF_ESTATE; }
    // fixed-size record per rank: [count, v_0 .. v_{cap-1}] int64 (8 KiB at cap = 1024: latency-bound)
    const size_t rec = (size_t)cap + 1, need = rec * (size_t)(W + 1);
    if (need > h->gather_cap) {
        bury(h, h->d_gather);
        h->d_gather = nullptr;
        h->gather_cap = 0;
        RCF_HIP(hipMalloc(&h->d_gather, sizeof(int64_t) * need));
        h->gather_cap = need;
    }
    std::vector<int64_t> host(need, -1);
    host[0] = keep;
    std::memcpy(host.data() + 1, mine, sizeof(int64_t) * (size_t)keep);
    int64_t *d_send = h->d_gather, *d_recv = h->d_gather + rec;
    RCF_HIP(hipMemcpyAsync(d_send, host.data(), sizeof(int64_t) * rec, hipMemcpyHostToDevice, h->stream));
    if (!rccl_ok(r, r->AllGather(d_send, d_recv, rec, 4 /* ncclInt64 */, h->comm, h->stream), "ncclAllGather")) return RCF_EHIP;
    RCF_HIP(hipMemcpyAsync(host.data() + rec, d_recv, sizeof(int64_t) * rec * (size_t)W, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    for (int w = 0; w < W; ++w) {
        const int64_t *rc = host.data() + rec * (size_t)(w + 1);
        const int c = (int)std::max<int64_t>(0, std::min<int64_t>(rc[0], cap));
        counts[w] = c;
        std::memcpy(all + (size_t)w * cap, rc + 1, sizeof(int64_t) * (size_t)c);
    }
    return RCF_OK;
}

int rcf_allreduce_max(rcf_t *h, double *value)
{
    if (!h || !value) { set_error("bad arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    RCF_HIP(hipStreamSynchronize(h->stream));
    if (!h->comm) return RCF_OK;
    RcclApi *r = rccl();
    if (!r) { set_error("librccl.so not available"); return RCF_ESTATE; }
    if (h->gather_cap < 2) {
        bury(h, h->d_gather);
        h->d_gather = nullptr;
        h->gather_cap = 0;
        RCF_HIP(hipMalloc(&h->d_gather, sizeof(int64_t) * 16));
        h->gather_cap = 16;
    }
    double *d = reinterpret_cast<double *>(h->d_gather);
    RCF_HIP(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (!rccl_ok(r, r->AllReduce(d, d, 1, 8 /* ncclFloat64 */, 2 /* ncclMax */, h->comm, h->stream), "ncclAllReduce")) return RCF_EHIP;
    RCF_HIP(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    return RCF_OK;
}


const char *rcf_version(void) { return "rcf-mi355x 0.1 (gfx950)"; }
const char *rcf_last_error(void) { return g_err; }

int rcf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rcf_design_low_pass_2(double gain, double fs, double fc, double tw, double att_db, int window, float *taps,
                          int cap)
{
    if (fs <= 0 || tw <= 0) { set_error("bad design arguments"); return RCF_EINVAL; }
    const int n = design_ntaps(fs, tw, att_db);
    if (!taps || cap < n) return -n;
    std::vector<float> t = design_low_pass_2(gain, fs, fc, tw, att_db, window);
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

int rcf_design_window(int window, int n, float *w)
{
    if (n < 2 || !w) { set_error("bad window arguments"); return RCF_EINVAL; }
    design_window(window, n, w);
    return RCF_OK;
}

int rcf_channel_params_ex(double samp_rate, int channel_rate, int decim_rule, int *decim, int *ntaps, double *out_rate)
{
    if (samp_rate <= 0 || channel_rate <= 0 || (decim_rule != RCF_DECIM_EXACT && decim_rule != RCF_DECIM_FLOOR)) {
        set_error("bad rates / decimation rule");
        return RCF_EINVAL;
    }
    const int q = (int)(samp_rate / channel_rate);
    if (q < 2 || ((q & 1) && decim_rule == RCF_DECIM_EXACT)) {
        set_error("int(fs/cr)/2 is not a positive integer for fs=%g cr=%d", samp_rate, channel_rate);
        return RCF_ERANGE;
    }
    if (decim) *decim = q / 2;
    if (ntaps) *ntaps = design_ntaps(samp_rate, channel_rate / 2.0, 20.0);
    if (out_rate) *out_rate = samp_rate / (q / 2);
    return RCF_OK;
}

int rcf_channel_params(double samp_rate, int channel_rate, int *decim, int *ntaps)
{
    return rcf_channel_params_ex(samp_rate, channel_rate, RCF_DECIM_EXACT, decim, ntaps, nullptr);
}

int rcf_open(int device, double samp_rate, double center_freq, rcf_t **out)
{
    return rcf_open_ex(device, samp_rate, center_freq, 0, 0, 0, out);
}

int rcf_open_ex(int device, double samp_rate, double center_freq, size_t block_capacity, size_t hist_capacity,
                size_t out_capacity, rcf_t **out)
{
    if (!out || samp_rate <= 0) { set_error("bad open arguments"); return RCF_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device: librcf has no CPU fallback");
        return RCF_EHIP;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return RCF_EINVAL; }
    std::unique_ptr<rcf> h(new rcf);
    h->device = device;
    h->fs = samp_rate;
    h->fc = center_freq;
    h->block_cap = block_capacity ? block_capacity : (size_t(1) << 22);
    h->hist_cap = hist_capacity ? hist_capacity : (size_t(1) << 16);
    h->out_cap = pow2_at_least(out_capacity ? out_capacity : (size_t(1) << 16));
    h->ring_mask = (uint64_t)h->out_cap - 1;
    {
        if (const char *nm = getenv("RCF_FIR_NOMFMA")) h->no_mfma = atoi(nm) != 0;
        if (const char *rm = getenv("RCF_ROTATOR")) h->exact_rot = std::strcmp(rm, "exact") == 0;
        if (const char *df = getenv("RCF_DECIM_FLOOR")) h->decim_rule = std::atoi(df) ? RCF_DECIM_FLOOR : RCF_DECIM_EXACT;
        if (const char *ck = getenv("RCF_COPY_KERNELS")) h->copy_kernels = h->copy_kernels && atoi(ck) != 0;
    if (const char *nm = getenv("RCF_FIR_MFMA_MIN")) h->mfma_min = std::max(1, atoi(nm));
        if (const char *nm = getenv("RCF_FIR_MFMA_NT")) h->mfma_nt = atoi(nm);
        if (const char *nm = getenv("RCF_FIR_MFMA_PARTS")) h->mfma_parts = atoi(nm);
    }
    RCF_HIP(hipSetDevice(device));
    RCF_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const size_t buf_samples = h->hist_cap + h->block_cap;
    for (int i = 0; i < 2; ++i) {
        RCF_HIP(hipMalloc(&h->d_buf[i], sizeof(float2) * buf_samples));
        RCF_HIP(hipMemsetAsync(h->d_buf[i], 0, sizeof(float2) * buf_samples, h->stream));
        RCF_HIP(hipHostMalloc(&h->h_arena[i], h->arena_cap, hipHostMallocDefault));
        RCF_HIP(hipMalloc(&h->d_arena[i], h->arena_cap));
        {
            void *dv = nullptr;
            h->h_arena_dev[i] = hipHostGetDevicePointer(&dv, h->h_arena[i], 0) == hipSuccess ? static_cast<unsigned char *>(dv) : nullptr;
            if (!h->h_arena_dev[i]) h->copy_kernels = false;
        }
        RCF_HIP(hipEventCreateWithFlags(&h->arena_ev[i], hipEventDisableTiming));
    }
    RCF_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) RCF_HIP(hipEventCreateWithFlags(&h->buf_done[i], hipEventDisableTiming));
    RCF_HIP(hipEventCreateWithFlags(&h->copy_ev, hipEventDisableTiming));
    RCF_HIP(hipEventCreateWithFlags(&h->raw_done, hipEventDisableTiming));
    RCF_HIP(hipMalloc(&h->d_atan, sizeof(float) * 257));
    RCF_HIP(hipMemcpy(h->d_atan, atan_table_host(), sizeof(float) * 257, hipMemcpyHostToDevice));
    RCF_HIP(hipStreamSynchronize(h->stream));
    *out = h.release();
    return RCF_OK;
}

int rcf_close(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (auto &kv : h->chans) free_channel(h, kv.second.get());
    h->chans.clear();
    Pfb &p = h->pfb;
    bury(h, p.d_ptaps); bury(h, p.d_tw); bury(h, p.d_bins); bury(h, p.d_stage);
    Scan &s = h->scan;
    bury(h, s.d_window); bury(h, s.d_vring); bury(h, s.d_sum); bury(h, s.d_out); bury(h, s.d_tw);
    bury(h, s.d_scratch); bury(h, s.d_peaks); bury(h, s.d_peak_ws);
    for (auto &kv : h->banks) bury(h, kv.second.d);
    h->banks.clear();
    bury(h, h->d_atan);
    bury(h, h->d_raw);
    bury(h, h->d_level);
    for (int i = 0; i < 2; ++i) {
        bury(h, h->d_buf[i]);
        bury(h, h->d_arena[i]);
        if (h->h_arena[i]) (void)hipHostFree(h->h_arena[i]);
        if (h->arena_ev[i]) (void)hipEventDestroy(h->arena_ev[i]);
    }
    bury(h, h->d_gather);
    if (h->h_many) (void)hipHostFree(h->h_many);
    bury(h, h->d_partial);
    bury(h, h->d_tapmat);
    drain_graveyard(h);
    for (auto &kv : h->pools)
        for (void *slab : kv.second.slabs) (void)hipFree(slab);
    h->pools.clear();
    time_collect(h);
    for (hipEvent_t e : h->time_pool) (void)hipEventDestroy(e);
    comm_destroy(h);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    for (int i = 0; i < 2; ++i) if (h->buf_done[i]) (void)hipEventDestroy(h->buf_done[i]);
    if (h->copy_ev) (void)hipEventDestroy(h->copy_ev);
    if (h->raw_done) (void)hipEventDestroy(h->raw_done);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return RCF_OK;
}

int rcf_sync(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    RCF_HIP(hipStreamSynchronize(h->stream));
    drain_graveyard(h);
    return RCF_OK;
}

int rcf_timing_enable(rcf_t *h, int on)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    RCF_HIP(hipStreamSynchronize(h->stream));
    time_collect(h);
    h->timing = on != 0;
    h->timing_mask = on == 1 ? ~0u : (unsigned)on >> 1;      // 1 = every class, else bit (class + 1)
    return RCF_OK;
}

int rcf_timing_stride(rcf_t *h, int every)
{
    if (!h || every < 1) { set_error("bad timing stride"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    h->timing_stride = (unsigned)every;
    for (unsigned &v : h->time_seen) v = 0;
    return RCF_OK;
}

int rcf_timing_read(rcf_t *h, int what, double *total_ms, int64_t *launches, int reset)
{
    if (!h || what < 0 || what >= RCF_T_COUNT) { set_error("bad timing class"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    RCF_HIP(hipStreamSynchronize(h->stream));
    time_collect(h);
    if (total_ms) *total_ms = h->time_ms[what];
    if (launches) *launches = h->time_n[what];
    if (reset) { h->time_ms[what] = 0; h->time_n[what] = 0; }
    return RCF_OK;
}

int rcf_set_rotator(rcf_t *h, int exact)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->chans.empty()) { set_error("rcf_set_rotator: channels are already open"); return RCF_ESTATE; }
    h->exact_rot = exact != 0;
    return RCF_OK;
}

int rcf_set_decim_rule(rcf_t *h, int decim_rule)
{
    if (!h || (decim_rule != RCF_DECIM_EXACT && decim_rule != RCF_DECIM_FLOOR)) { set_error("bad decimation rule"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    h->decim_rule = decim_rule;
    return RCF_OK;
}

void *rcf_stream(rcf_t *h) { return h ? (void *)h->stream : nullptr; }
int rcf_device(rcf_t *h) { return h ? h->device : RCF_EINVAL; }
int64_t rcf_samples_in(rcf_t *h) { return h ? h->total_in : RCF_EINVAL; }

// The H2D copy of a block runs on the copy stream: it waits only for the kernels that last read the target buffer
// (the block before the previous one), so it overlaps the previous block's kernels; the compute stream waits for
// the copy.  The call returns when the copy has been read from the caller's buffer (pageable copies are staged by
// the runtime, pinned ones -- rcf_host_alloc -- are DMA'd in place), not when the kernels are done.
int rcf_push_iq(rcf_t *h, const float *iq, size_t n)
{
    if (!h || (!iq && n)) { set_error("bad push arguments"); return RCF_EINVAL; }
    if (n == 0) return RCF_OK;
    if (n > h->block_cap) { set_error("push of %zu samples exceeds block capacity %zu", n, h->block_cap); return RCF_ECAP; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    {
        // a real-time-sized block in pinned memory (see rcf_push_raw): copied by a kernel on the compute stream straight
        // out of host memory -- no second stream, no cross-stream waits.  In order behind every kernel that read this
        // buffer, so no buf_done bookkeeping either.
        static const int direct = [] { const char *e = getenv("RCF_RAW_DIRECT"); return e ? atoi(e) : (4 << 20); }();
        void *dv = nullptr;
        if (direct && n * sizeof(float2) <= (size_t)direct && h->copy_kernels &&
            hipHostGetDevicePointer(&dv, const_cast<float *>(iq), 0) == hipSuccess && dv) {
            launch_copy8(h->d_buf[h->cur] + h->hist_cap, dv, sizeof(float2) * n, h->stream);
            RCF_HIP(hipEventRecord(h->copy_ev, h->stream));
            int rc = process_block(h, n);
            (void)hipEventSynchronize(h->copy_ev);
            return rc;
        }
        (void)hipGetLastError();                              // (pageable memory: not an error)
    }
    h->eager_buf_done = true;
    if (h->buf_dirty[h->cur]) {                               // blocks committed in place read this buffer since
        RCF_HIP(hipEventRecord(h->buf_done[h->cur], h->stream));
        h->buf_done_set[h->cur] = true;
        h->buf_dirty[h->cur] = false;
    }
    if (h->buf_done_set[h->cur]) RCF_HIP(hipStreamWaitEvent(h->copy_stream, h->buf_done[h->cur], 0));
    RCF_HIP(hipMemcpyAsync(h->d_buf[h->cur] + h->hist_cap, iq, sizeof(float2) * n, hipMemcpyHostToDevice,
                           h->copy_stream));
    RCF_HIP(hipEventRecord(h->copy_ev, h->copy_stream));
    RCF_HIP(hipStreamWaitEvent(h->stream, h->copy_ev, 0));
    int rc = process_block(h, n);
    (void)hipEventSynchronize(h->copy_ev);
    return rc;
}

int rcf_push_raw(rcf_t *h, const void *iq_raw, size_t n, int fmt, float scale, float offset)
{
    const size_t bps = raw_sample_bytes(fmt);
    if (!h || (!iq_raw && n) || bps == 0) { set_error("bad raw push arguments"); return RCF_EINVAL; }
    if (n == 0) return RCF_OK;
    if (n > h->block_cap) { set_error("push of %zu samples exceeds block capacity %zu", n, h->block_cap); return RCF_ECAP; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    {
        // Pinned caller memory (rcf_host_alloc): the conversion kernel reads the wire-format block straight out of host
        // memory across PCIe -- no staging copy, no second stream, no cross-stream event waits: one event instead of
        // two and two barrier packets fewer per block, which is what a real-time-sized block (a handful of ~5 us
        // kernels) is made of: 256 front-ends of the bench's real-time leg p99 1.9 -> 0.2 ms, 384 sustained instead
        // of missing.  Blocks above RCF_RAW_DIRECT bytes (default 4 MiB; 0 = never) keep the staged copy: in a bulk
        // replay the copy of block n + 1 then overlaps the kernels of block n, which a PCIe-bound kernel on the
        // compute stream would not.
        static const int direct = [] { const char *e = getenv("RCF_RAW_DIRECT"); return e ? atoi(e) : (4 << 20); }();
        void *dv = nullptr;
        if (direct && n * bps <= (size_t)direct && hipHostGetDevicePointer(&dv, const_cast<void *>(iq_raw), 0) == hipSuccess && dv) {
            launch_convert(fmt, dv, h->d_buf[h->cur] + h->hist_cap, n, scale, offset, h->stream);
            RCF_HIP(hipEventRecord(h->copy_ev, h->stream));
            int rc = process_block(h, n);
            (void)hipEventSynchronize(h->copy_ev);  // the caller may reuse its buffer once the conversion has read it
            return rc;
        }
        (void)hipGetLastError();                    // (pageable memory: not an error)
    }
    if (!h->d_raw) RCF_HIP(hipMalloc(&h->d_raw, h->block_cap * 4));       // staging for the widest format
    if (h->raw_done_set) RCF_HIP(hipStreamWaitEvent(h->copy_stream, h->raw_done, 0));   // previous conversion read it
    RCF_HIP(hipMemcpyAsync(h->d_raw, iq_raw, n * bps, hipMemcpyHostToDevice, h->copy_stream));
    RCF_HIP(hipEventRecord(h->copy_ev, h->copy_stream));
    RCF_HIP(hipStreamWaitEvent(h->stream, h->copy_ev, 0));
    launch_convert(fmt, h->d_raw, h->d_buf[h->cur] + h->hist_cap, n, scale, offset, h->stream);
    RCF_HIP(hipEventRecord(h->raw_done, h->stream));
    h->raw_done_set = true;
    int rc = process_block(h, n);
    (void)hipEventSynchronize(h->copy_ev);      // the caller may reuse its buffer once the H2D copy has been read
    return rc;
}

void *rcf_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        set_error("pinned host allocation of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}

void rcf_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int rcf_ingest_ptr(rcf_t *h, float **dev_ptr, size_t *max_samples)
{
    if (!h || !dev_ptr) { set_error("bad ingest arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    *dev_ptr = reinterpret_cast<float *>(h->d_buf[h->cur] + h->hist_cap);
    if (max_samples) *max_samples = h->block_cap;
    return RCF_OK;
}

int rcf_ingest_write(rcf_t *h, const float *iq, size_t n, size_t at)
{
    if (!h || (!iq && n)) { set_error("bad ingest arguments"); return RCF_EINVAL; }
    if (at + n > h->block_cap) { set_error("ingest write past block capacity"); return RCF_ECAP; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    RCF_HIP(hipMemcpyAsync(h->d_buf[h->cur] + h->hist_cap + at, iq, sizeof(float2) * n, hipMemcpyHostToDevice,
                           h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    return RCF_OK;
}

int rcf_commit(rcf_t *h, size_t n)
{
    if (!h) return RCF_EINVAL;
    if (n == 0) return RCF_OK;
    if (n > h->block_cap) { set_error("commit of %zu samples exceeds block capacity %zu", n, h->block_cap); return RCF_ECAP; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    return process_block(h, n);
}

// ------------------------------------------------------------------ channels
int rcf_chan_open(rcf_t *h, int channel_rate, double offset_hz, int *chan_id)
{
    if (!h || !chan_id) { set_error("bad channel arguments"); return RCF_EINVAL; }
    int D = 0, T = 0;
    int rc = rcf_channel_params_ex(h->fs, channel_rate, h->decim_rule, &D, &T, nullptr);
    if (rc != RCF_OK) return rc;
    if (!(std::fabs(offset_hz) < h->fs / 2)) { set_error("offset %g Hz outside +-fs/2", offset_hz); return RCF_ERANGE; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    std::vector<float> &taps = h->proto_cache[channel_rate];
    if (taps.empty())
        taps = design_low_pass_2(1.0, h->fs, channel_rate / 2.0, channel_rate / 2.0, 20.0, RCF_WIN_HAMMING);
    return new_channel(h, -1, D, taps.data(), (int)taps.size(), offset_hz, chan_id);
}

int rcf_chan_open_taps(rcf_t *h, int src_chan, int decim, const float *taps, int ntaps, double offset_hz,
                       int *chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    return new_channel(h, src_chan < 0 ? -1 : src_chan, decim, taps, ntaps, offset_hz, chan_id);
}

int rcf_pfb_chan_open(rcf_t *h, int bin, int channel_rate, double delta_hz, int *chan_id)
{
    if (!h || !chan_id) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    if (!h->pfb.open || bin < 0 || bin >= h->pfb.NB) { set_error("no such PFB bin %d", bin); return RCF_EINVAL; }
    const double rate = h->fs / h->pfb.D;
    int D = 0, T = 0;
    int rc = rcf_channel_params_ex(rate, channel_rate, h->decim_rule, &D, &T, nullptr);
    if (rc != RCF_OK) return rc;
    std::vector<float> taps = design_low_pass_2(1.0, rate, channel_rate / 2.0, channel_rate / 2.0, 20.0,
                                                RCF_WIN_HAMMING);
    return new_channel(h, RCF_SRC_PFB_BIN0 + bin, D, taps.data(), (int)taps.size(), delta_hz, chan_id);
}

int rcf_pfb_tap_open(rcf_t *h, int bin, int gr_phase, int *chan_id)
{
    if (!h || !chan_id) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open || bin < 0 || bin >= p.NB) { set_error("no such PFB bin %d", bin); return RCF_EINVAL; }
    const float one = 1.0f;
    int rc = new_channel(h, RCF_SRC_PFB_BIN0 + bin, 1, &one, 1, 0.0, chan_id);
    if (rc != RCF_OK) return rc;
    h->chans[*chan_id]->is_tap = p.frame_major;     // power-of-two banks: an ordinary D = 1, T = 1 channel on the bin's ring
    if (gr_phase) {
        // What GNU Radio's freq_xlating_fir_filter_ccc(D, h, f_k, fs) would have done differently from the bank's
        // exact phases: its rotator advances by a = float32(-float32(2 pi f_k / fs) * D) per output instead of
        // -2 pi k D / NB, and the float32 increment (cosf a, sinf a) is not exactly of unit length.  Both are
        // per-output factors: this channel's own rotator carries them.
        Chan *c = h->chans[*chan_id].get();
        const int ks = bin < p.NB / 2 ? bin : bin - p.NB;
        const double f_k = (double)ks * h->fs / p.NB;
        const float fwT0 = (float)(kTwoPi * f_k / h->fs);
        const float a = -fwT0 * (float)p.D;
        const long double exact = -2.0L * 3.14159265358979323846264338327950288L *
                                  (long double)(((int64_t)ks * p.D) % p.NB) / (long double)p.NB;
        long double d = (long double)a - exact;
        d = remainderl(d, 2.0L * 3.14159265358979323846264338327950288L);
        c->extra_dangle = (double)d;
        c->extra_dlogmag = std::log(std::hypot((double)std::cos(a), (double)std::sin(a)));
        // ... and GNU Radio's float32 tap phases float32(i * fwT0) differ from the bank's 2 pi k i / NB by a constant
        // (their filter-weighted mean, up to ~3e-4 rad) plus rounding noise (rcf_pfb_tap_leakage): the constant is a
        // rotation of the whole output and goes into the rotator's start phase
        double cphase = 0.0;
        design_tap_leakage(h->fs, p.NB, p.proto.data(), (int)p.proto.size(), bin, nullptr, &cphase);
        // ... and GNU Radio's rotator stands at 1 when the channel emits its FIRST output, whereas the bank's bin carries
        // e^{-j 2 pi k D n / NB} counted from the stream's first sample: a channel that starts at the bank's frame n0
        // is the bin times e^{+j 2 pi k D n0 / NB} (exact: integers mod NB; a sign for the 12.5 kHz grid's OS = 2 banks)
        const int64_t n0 = p.n_abs0 + c->k_abs0;
        const int64_t kd = (((int64_t)ks * p.D) % p.NB + p.NB) % p.NB;
        const int64_t turn = (kd * (n0 % p.NB)) % p.NB;
        c->angle0 = (long double)cphase +
                    remainderl(2.0L * 3.14159265358979323846264338327950288L * (long double)turn / (long double)p.NB,
                               2.0L * 3.14159265358979323846264338327950288L);
        rc = upload_composite(h, c);
    }
    {
        // plan_channel never iterates the exact rotator for a frame-major bank's tap or a channel that carries GNU
        // Radio's phase corrections: give the phase ring (8 out_cap bytes, half a megabyte at 2^16) back -- a
        // receiver in 'pfb' mode opens hundreds of these.  Nothing has been queued on it yet.
        Chan *c = h->chans[*chan_id].get();
        if (c->d_rot && (c->is_tap || c->extra_dangle != 0.0 || c->extra_dlogmag != 0.0)) {
            h->pools[slice_round(sizeof(float2) * h->out_cap + 256)].free_.push_back(c->d_rot);
            c->d_rot = nullptr;
        }
    }
    return rc;
}

#define FIND_CHAN(h, id, c)                                                 \
    auto it_ = (h)->chans.find(id);                                         \
    if (it_ == (h)->chans.end()) { set_error("no such channel %d", id); return RCF_ENOCHAN; } \
    Chan *c = it_->second.get()

int rcf_chan_set_offset(rcf_t *h, int chan_id, double offset_hz)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    c->offset_hz = offset_hz;
    return upload_composite(h, c);
}

int rcf_chan_close(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    free_channel(h, c);
    h->chans.erase(chan_id);
    return RCF_OK;
}

int rcf_chan_info(rcf_t *h, int chan_id, int *decim, int *ntaps, double *out_rate, double *offset_hz)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    FIND_CHAN(h, chan_id, c);
    if (decim) *decim = c->D;
    if (ntaps) *ntaps = c->T;
    if (out_rate) *out_rate = c->src_rate / c->D;
    if (offset_hz) *offset_hz = c->offset_hz;
    return RCF_OK;
}

int64_t rcf_chan_produced(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    FIND_CHAN(h, chan_id, c);
    return c->produced;
}

int64_t rcf_chan_start(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    FIND_CHAN(h, chan_id, c);
    return c->start_sample;
}

int64_t rcf_chan_read_iq(rcf_t *h, int chan_id, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    return ring_read(h, c->d_iq, sizeof(float2), c->produced, &c->rd_iq, out, max_samples);
}

int64_t rcf_chan_read_fm(rcf_t *h, int chan_id, float gain, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    const int64_t n = ring_read(h, c->d_fm, sizeof(float), c->produced, &c->rd_fm, out, max_samples);
    // quadrature_demod_cf: out = gain * fast_atan2f(...), one float32 multiply per sample
    for (int64_t i = 0; i < n; ++i) out[i] = gain * out[i];
    return n;
}

int rcf_chan_read_many(rcf_t *h, int what, const int *chan_ids, int n_chans, float gain, void *out, size_t cap_each,
                       int64_t *counts)
{
    if (!h || !chan_ids || !out || !counts || n_chans < 0 || (what != RCF_READ_IQ && what != RCF_READ_FM)) {
        set_error("bad batched read arguments");
        return RCF_EINVAL;
    }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    const size_t elem = what == RCF_READ_IQ ? sizeof(float2) : sizeof(float);
    const uint32_t ew = (uint32_t)(elem / 4);
    // what every channel has to give, and where its reader stands
    struct Item { Chan *c; int64_t *cur; const void *ring; int64_t n; size_t pos; };
    std::vector<Item> items((size_t)n_chans);
    size_t total = 0;
    uint32_t max_w = 0;
    const uint64_t stamp = ++h->many_stamp;
    for (int i = 0; i < n_chans; ++i) {
        Item &it = items[(size_t)i];
        it = Item{nullptr, nullptr, nullptr, 0, 0};
        auto f = h->chans.find(chan_ids[i]);
        if (f == h->chans.end()) { counts[i] = RCF_ENOCHAN; continue; }
        Chan *c = f->second.get();
        if (c->many_stamp == stamp) { counts[i] = RCF_EINVAL; continue; }   // listed twice: one reader position per channel
        c->many_stamp = stamp;
        it.c = c;
        it.cur = what == RCF_READ_IQ ? &c->rd_iq : &c->rd_fm;
        it.ring = what == RCF_READ_IQ ? (const void *)c->d_iq : (const void *)c->d_fm;
        int64_t avail = c->produced - *it.cur;
        if (avail > 0 && (size_t)avail > h->out_cap) {          // reader lagged: oldest samples are gone
            *it.cur = c->produced - (int64_t)h->out_cap;
            avail = (int64_t)h->out_cap;
        }
        it.n = avail <= 0 ? 0 : std::min<int64_t>(avail, (int64_t)cap_each);
        it.pos = (size_t)((uint64_t)*it.cur & h->ring_mask);
        counts[i] = it.n;
        total += (size_t)it.n;
        max_w = std::max<uint32_t>(max_w, (uint32_t)it.n * ew);
    }
    if (total == 0) return RCF_OK;
    // One gather launch packs every ring segment back to back into pinned host memory, one synchronisation, then the
    // rows are handed out.  (A device round trip per channel -- rcf_chan_read_iq in a loop -- costs ~10 us each: 256
    // tapped bins of ten front-ends are 25 ms per pass.)
    const size_t rec_bytes = ((size_t)n_chans * sizeof(GatherRec) + 255) & ~(size_t)255;
    const size_t need = rec_bytes + total * elem;
    if (need > h->many_cap && (uint64_t)total * ew <= 0xffffffffull) {
        if (h->h_many) { (void)hipStreamSynchronize(h->stream); (void)hipHostFree(h->h_many); h->h_many = nullptr; h->many_cap = 0; }
        size_t cap = 1 << 16;
        while (cap < need) cap <<= 1;
        void *p = nullptr, *dv = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer(&dv, p, 0) == hipSuccess) {
            h->h_many = static_cast<unsigned char *>(p);
            h->h_many_dev = static_cast<unsigned char *>(dv);
            h->many_cap = cap;
        } else if (p) {
            (void)hipHostFree(p);
        }
    }
    if (h->h_many && need <= h->many_cap && (uint64_t)total * ew <= 0xffffffffull) {     // (GatherRec counts 32-bit words)
        GatherRec *recs = reinterpret_cast<GatherRec *>(h->h_many);
        uint32_t at_w = 0;
        int n_recs = 0;
        for (int i = 0; i < n_chans; ++i) {
            const Item &it = items[(size_t)i];
            if (it.n <= 0) continue;
            recs[n_recs++] = GatherRec{static_cast<const uint32_t *>(it.ring), (uint32_t)(it.pos * ew), (uint32_t)it.n * ew,
                                       (uint32_t)(h->out_cap * ew - 1), at_w};
            at_w += (uint32_t)it.n * ew;
        }
        launch_gather_rings(reinterpret_cast<const GatherRec *>(h->h_many_dev), n_recs,
                            reinterpret_cast<uint32_t *>(h->h_many_dev + rec_bytes), max_w, h->stream);
        if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("stream sync failed"); return RCF_EHIP; }
        const unsigned char *src = h->h_many + rec_bytes;
        for (int i = 0; i < n_chans; ++i) {
            const Item &it = items[(size_t)i];
            if (it.n <= 0) continue;
            std::memcpy(static_cast<unsigned char *>(out) + (size_t)i * cap_each * elem, src, (size_t)it.n * elem);
            src += (size_t)it.n * elem;
        }
    } else {
        // no mapped pinned memory: ring by ring, still behind one synchronisation
        for (int i = 0; i < n_chans; ++i) {
            const Item &it = items[(size_t)i];
            if (it.n <= 0) continue;
            int64_t cur = *it.cur;
            const int64_t n = ring_read_enqueue(h, it.ring, elem, it.c->produced, &cur,
                                                static_cast<unsigned char *>(out) + (size_t)i * cap_each * elem, (size_t)it.n);
            if (n < 0) { (void)hipStreamSynchronize(h->stream); return (int)n; }
        }
        if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("stream sync failed"); return RCF_EHIP; }
    }
    free_graveyard_idle(h);
    for (int i = 0; i < n_chans; ++i) {
        const Item &it = items[(size_t)i];
        if (it.n <= 0) continue;
        *it.cur += it.n;
        if (what == RCF_READ_FM) {
            float *o = static_cast<float *>(out) + (size_t)i * cap_each;
            for (int64_t k = 0; k < it.n; ++k) o[k] = gain * o[k];
        }
    }
    return RCF_OK;
}

int rcf_chan_fm_filter(rcf_t *h, int chan_id, float gain, const float *taps, int ntaps)
{
    if (!h || !taps || ntaps < 1 || ntaps > 4096) { set_error("bad fm filter arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if ((size_t)ntaps * 2 > h->out_cap) { set_error("ring too small for %d taps", ntaps); return RCF_ECAP; }
    float *fresh = nullptr;
    RCF_HIP(hipMalloc(&fresh, sizeof(float) * (size_t)ntaps));
    RCF_HIP(hipMemcpy(fresh, taps, sizeof(float) * (size_t)ntaps, hipMemcpyHostToDevice));
    bury(h, c->d_symtaps);
    c->d_symtaps = fresh;
    c->sym_ntaps = ntaps;
    c->sym_gain = gain;
    if (!c->d_sym) {
        RCF_HIP(hipMalloc(&c->d_sym, sizeof(float) * h->out_cap));
        RCF_HIP(hipMemsetAsync(c->d_sym, 0, sizeof(float) * h->out_cap, h->stream));
        c->sym_from = c->produced;      // a new GR block starts with zero history
        c->rd_sym = c->produced;
    }
    return RCF_OK;
}

int64_t rcf_chan_read_sym(rcf_t *h, int chan_id, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (!c->d_sym) { set_error("channel %d has no fm filter", chan_id); return RCF_ESTATE; }
    return ring_read(h, c->d_sym, sizeof(float), c->produced, &c->rd_sym, out, max_samples);
}

int rcf_chan_audio_open(rcf_t *h, int chan_id, const rcf_audio_params_t *p)
{
    if (!h || !p || !p->lpf_taps || !p->hpf_taps || !p->rs_taps || p->n_lpf < 1 || p->n_hpf < 1 || p->n_rs < 1 ||
        p->interpolation < 1 || p->decimation < 1 || p->deemph_a[0] == 0.0) {
        set_error("bad audio chain arguments");
        return RCF_EINVAL;
    }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    const int I = p->interpolation;
    const int n_rs_pad = (p->n_rs + I - 1) / I * I;              // rational_resampler_base: pad to a multiple of I
    const size_t reach = (size_t)std::max(std::max(p->n_lpf, p->n_hpf), n_rs_pad / I);
    if (reach * 2 > h->out_cap) { set_error("ring of %zu too small for %zu-tap audio filters", h->out_cap, reach); return RCF_ECAP; }
    std::unique_ptr<Chan::Audio> au(new Chan::Audio);
    au->n_lpf = p->n_lpf; au->n_hpf = p->n_hpf; au->nt_rs = n_rs_pad / I;
    au->interp = I; au->decim = p->decimation;
    au->gain = p->quad_gain;
    au->thr = std::pow(10.0, p->squelch_db / 10);                // pwr_squelch_cc::set_threshold
    au->alpha = p->squelch_alpha;
    // iir_filter(fftaps, fbtaps, oldstyle = false): feedback taps are negated, a[0] must be 1
    au->b0 = p->deemph_b[0]; au->b1 = p->deemph_b[1]; au->fb1 = -p->deemph_a[1];
    std::vector<float> taps((size_t)p->n_lpf + p->n_hpf + n_rs_pad, 0.0f);
    std::memcpy(taps.data(), p->lpf_taps, sizeof(float) * (size_t)p->n_lpf);
    std::memcpy(taps.data() + p->n_lpf, p->hpf_taps, sizeof(float) * (size_t)p->n_hpf);
    std::memcpy(taps.data() + p->n_lpf + p->n_hpf, p->rs_taps, sizeof(float) * (size_t)p->n_rs);
    RCF_HIP(hipMalloc(&au->d_taps, sizeof(float) * taps.size()));
    RCF_HIP(hipMemcpy(au->d_taps, taps.data(), sizeof(float) * taps.size(), hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&au->d_rings, sizeof(float) * 6 * h->out_cap));
    RCF_HIP(hipMemsetAsync(au->d_rings, 0, sizeof(float) * 6 * h->out_cap, h->stream));
    AudioState st0{};
    st0.muted = 1;                                               // squelch_base_cc starts in ST_MUTED
    RCF_HIP(hipMalloc(&au->d_state, sizeof(AudioState)));
    RCF_HIP(hipMemcpy(au->d_state, &st0, sizeof(st0), hipMemcpyHostToDevice));
    au->from = c->produced;                                      // a new flowgraph: zero state from here on
    if (c->audio) { bury(h, c->audio->d_state); bury(h, c->audio->d_rings); bury(h, c->audio->d_taps); }
    c->audio = std::move(au);
    return RCF_OK;
}

int rcf_chan_audio_close(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (c->audio) { bury(h, c->audio->d_state); bury(h, c->audio->d_rings); bury(h, c->audio->d_taps); c->audio.reset(); }
    return RCF_OK;
}

static int audio_counts(rcf_t *h, Chan *c, int64_t *n_audio, int64_t *n_ungated)
{
    AudioState st{};
    RCF_HIP(hipMemcpyAsync(&st, c->audio->d_state, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    const int64_t I = c->audio->interp, D = c->audio->decim;
    *n_ungated = st.n_a;
    *n_audio = (st.n_a * I + D - 1) / D;
    return RCF_OK;
}

int rcf_chan_audio_produced(rcf_t *h, int chan_id, int64_t *n_audio, int64_t *n_ungated)
{
    if (!h || !n_audio) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (!c->audio) { set_error("channel %d has no audio chain", chan_id); return RCF_ESTATE; }
    int64_t a = 0, u = 0;
    const int rc = audio_counts(h, c, &a, &u);
    if (rc != RCF_OK) return rc;
    *n_audio = a;
    if (n_ungated) *n_ungated = u;
    return RCF_OK;
}

int64_t rcf_chan_read_audio(rcf_t *h, int chan_id, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (!c->audio) { set_error("channel %d has no audio chain", chan_id); return RCF_ESTATE; }
    int64_t a = 0, u = 0;
    const int rc = audio_counts(h, c, &a, &u);
    if (rc != RCF_OK) return rc;
    return ring_read(h, c->audio->d_rings + 3 * h->out_cap, sizeof(float), a, &c->audio->rd, out, max_samples);
}

int rcf_design_firdes(int kind, double gain, double fs, double fc, double tw, int window, double beta, float *taps,
                      int cap)
{
    if (fs <= 0 || tw <= 0 || (kind != RCF_FIR_LOW_PASS && kind != RCF_FIR_HIGH_PASS) ||
        design_max_attenuation(window, beta) <= 0) {
        set_error("bad design arguments");
        return RCF_EINVAL;
    }
    const int n = design_ntaps(fs, tw, design_max_attenuation(window, beta));
    if (!taps || cap < n) return -n;
    std::vector<float> t = design_firdes(kind, gain, fs, fc, tw, window, beta);
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

int rcf_design_optfir_low_pass(double gain, double fs, double freq1, double freq2, double passband_ripple_db,
                               double stopband_atten_db, float *taps, int cap)
{
    std::vector<float> t;
    if (!design_optfir_low_pass(gain, fs, freq1, freq2, passband_ripple_db, stopband_atten_db, 2, t)) {
        set_error("equiripple design failed (bad band edges, or the exchange did not find its extrema)");
        return RCF_EINVAL;
    }
    const int n = (int)t.size();
    if (!taps || cap < n) return -n;
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

int rcf_design_fm_deemph(double fs, double tau, double btaps[2], double ataps[2])
{
    if (fs <= 0 || tau <= 0 || !btaps || !ataps) { set_error("bad de-emphasis arguments"); return RCF_EINVAL; }
    design_fm_deemph(fs, tau, btaps, ataps);
    return RCF_OK;
}

int rcf_design_resampler(int interpolation, int decimation, int *interp_out, int *decim_out, float *taps, int cap)
{
    if (interpolation < 1 || decimation < 1) { set_error("bad resampler ratio"); return RCF_EINVAL; }
    int a = interpolation, b = decimation;
    while (b) { const int t = a % b; a = b; b = t; }
    const int I = interpolation / a, D = decimation / a;
    if (interp_out) *interp_out = I;
    if (decim_out) *decim_out = D;
    std::vector<float> t = design_resampler(I, D);
    const int n = (int)t.size();
    if (!taps || cap < n) return -n;
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

int rcf_chan_fm_level(rcf_t *h, int chan_id, float gain, int window, float *level)
{
    if (!h || !level || window < 1) { set_error("bad fm level arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if ((size_t)window > h->out_cap) { set_error("window %d exceeds the ring", window); return RCF_ECAP; }
    if (!h->d_level) RCF_HIP(hipMalloc(&h->d_level, sizeof(float)));
    launch_fm_level(c->d_fm, c->produced, window, gain, h->ring_mask, h->d_level, h->stream);
    RCF_HIP(hipMemcpyAsync(level, h->d_level, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    return RCF_OK;
}

int rcf_chan_rings(rcf_t *h, int chan_id, void **iq_ring, void **fm_ring, size_t *capacity)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    FIND_CHAN(h, chan_id, c);
    if (iq_ring) *iq_ring = c->d_iq;
    if (fm_ring) *fm_ring = c->d_fm;
    if (capacity) *capacity = h->out_cap;
    return RCF_OK;
}

int rcf_source_shift(rcf_t *h, double delta_hz)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    h->shift_hz += delta_hz;
    for (auto &kv : h->chans)
        if (kv.second->src < 0 || kv.second->src >= RCF_SRC_PFB_BIN0) {
            int rc = upload_composite(h, kv.second.get());
            if (rc != RCF_OK) return rc;
        }
    return RCF_OK;
}

// ------------------------------------------------------------------ PFB
int rcf_pfb_open(rcf_t *h, int n_bins, int decim, const float *taps, int ntaps)
{
    if (!h || !taps || ntaps < 1 || n_bins < 1 || decim < 1) { set_error("bad PFB arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    if (h->pfb.open) { set_error("PFB already open"); return RCF_ESTATE; }
    const int P = (ntaps + n_bins - 1) / n_bins;
    if (n_bins % decim || !pfb_supported(n_bins, decim, P)) {
        set_error("unsupported PFB shape: bins=%d decim=%d taps/branch=%d", n_bins, decim, P);
        return RCF_EINVAL;
    }
    const bool fm = pfb_frame_major(n_bins);
    if (!fm && h->out_cap < (size_t(1) << kPfbTileLog2)) { set_error("output capacity %zu < one ring tile", h->out_cap); return RCF_ECAP; }
    const size_t ring_samples = fm ? (size_t)n_bins * h->out_cap : (size_t)(h->out_cap >> kPfbTileLog2) * (size_t)pfb_tile_pitch(n_bins);
    // 32-bit buffer offsets: the wideband buffer, and the tiled ring of the power-of-two banks (one descriptor for the
    // whole ring).  The frame-major banks address their ring through one descriptor per frame row: no limit there.
    if ((!fm && (uint64_t)ring_samples * sizeof(float2) >= (1ull << 31)) ||
        (uint64_t)(h->hist_cap + h->block_cap) * sizeof(float2) >= (1ull << 31)) {
        set_error("PFB rings / wideband buffer exceed the 2 GiB range of 32-bit buffer offsets");
        return RCF_ECAP;
    }
    if ((size_t)P * n_bins + (size_t)decim > h->hist_cap) { set_error("history capacity %zu < P*bins", h->hist_cap); return RCF_ECAP; }
    Pfb &p = h->pfb;
    p.NB = n_bins; p.D = decim; p.T = ntaps; p.P = P;
    p.proto.assign(taps, taps + ntaps);
    p.Ppad = pfb_padded_p(n_bins, decim, P);
    std::vector<float> pt((size_t)p.Ppad * n_bins, 0.f);
    for (int i = 0; i < ntaps; ++i) pt[i] = taps[i];          // pt[p*NB + rho] = h[NB p + rho]
    std::vector<float> tw(2 * (size_t)n_bins);
    for (int i = 0; i < n_bins; ++i) {
        const double a = kTwoPi * i / n_bins;
        tw[2 * i] = (float)std::cos(a);
        tw[2 * i + 1] = (float)std::sin(a);
    }
    RCF_HIP(hipMalloc(&p.d_ptaps, sizeof(float) * pt.size()));
    RCF_HIP(hipMemcpy(p.d_ptaps, pt.data(), sizeof(float) * pt.size(), hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&p.d_tw, sizeof(float2) * (size_t)n_bins));
    RCF_HIP(hipMemcpy(p.d_tw, tw.data(), sizeof(float2) * (size_t)n_bins, hipMemcpyHostToDevice));
    p.frame_major = fm;
    RCF_HIP(hipMalloc(&p.d_bins, sizeof(float2) * ring_samples));
    RCF_HIP(hipMemsetAsync(p.d_bins, 0, sizeof(float2) * ring_samples, h->stream));
    p.rd.assign(n_bins, 0);
    p.start_sample = h->total_in;
    p.n_abs0 = ceil_div(p.start_sample, decim);
    p.produced = p.produced_before = 0;
    p.open = true;
    return RCF_OK;
}

int rcf_pfb_close(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open) return RCF_OK;
    for (auto it = h->chans.begin(); it != h->chans.end();) {
        if (it->second->src >= RCF_SRC_PFB_BIN0) { free_channel(h, it->second.get()); it = h->chans.erase(it); }
        else ++it;
    }
    bury(h, p.d_ptaps); bury(h, p.d_tw); bury(h, p.d_bins); bury(h, p.d_stage);
    p = Pfb();
    return RCF_OK;
}

int rcf_pfb_tap_leakage(double samp_rate, int n_bins, const float *taps, int ntaps, int bin, double *leak_l2,
                        double *const_phase)
{
    if (!(samp_rate > 0) || n_bins < 1 || !taps || ntaps < 1 || bin < 0 || bin >= n_bins) {
        set_error("bad tap-leakage arguments");
        return RCF_EINVAL;
    }
    design_tap_leakage(samp_rate, n_bins, taps, ntaps, bin, leak_l2, const_phase);
    return RCF_OK;
}

int rcf_pfb_shape_supported(int n_bins, int decim, int ntaps)
{
    if (n_bins < 1 || decim < 1 || ntaps < 1 || n_bins % decim) return 0;
    return pfb_supported(n_bins, decim, (ntaps + n_bins - 1) / n_bins) ? 1 : 0;
}

int64_t rcf_pfb_produced(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    return h->pfb.open ? h->pfb.produced : RCF_ESTATE;
}

int64_t rcf_pfb_read_bin(rcf_t *h, int bin, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open || bin < 0 || bin >= p.NB) { set_error("no such PFB bin %d", bin); return RCF_EINVAL; }
    // one bin out of the bank's ring (tiled or frame-major): gather its unread samples into a contiguous staging buffer
    int64_t avail = p.produced - p.rd[bin];
    if (avail <= 0 || max_samples == 0) return 0;
    if ((size_t)avail > h->out_cap) { p.rd[bin] = p.produced - (int64_t)h->out_cap; avail = (int64_t)h->out_cap; }
    const int64_t n = std::min<int64_t>(avail, (int64_t)max_samples);
    if (!p.d_stage) RCF_HIP(hipMalloc(&p.d_stage, sizeof(float2) * h->out_cap));
    SrcRange sr{};
    if (!source_range(h, RCF_SRC_PFB_BIN0 + bin, 0, 0, &sr)) return RCF_ESTATE;
    launch_gather_view(sr.view, p.rd[bin], p.d_stage, (size_t)n, h->stream);
    RCF_HIP(hipMemcpyAsync(out, p.d_stage, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    free_graveyard_idle(h);
    p.rd[bin] += n;
    return n;
}

int rcf_pfb_rings(rcf_t *h, void **bins_ring, size_t *capacity, size_t *pitch)
{
    if (!h || !h->pfb.open) return RCF_ESTATE;
    if (bins_ring) *bins_ring = h->pfb.d_bins;
    if (capacity) *capacity = h->out_cap;
    if (pitch) *pitch = h->pfb.frame_major ? 0 : (size_t(1) << kPfbTileLog2);   // frames per tile (0: frame-major)
    return RCF_OK;
}

// ------------------------------------------------------------------ scan
int rcf_scan_start(rcf_t *h, int fft_len, int n_frames, int avg_len)
{
    if (!h || n_frames < 1 || avg_len < 1) { set_error("bad scan arguments"); return RCF_EINVAL; }
    if (!scan_supported(fft_len)) { set_error("unsupported scan FFT length %d", fft_len); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    if ((size_t)fft_len > h->hist_cap) { set_error("history capacity %zu < fft_len %d", h->hist_cap, fft_len); return RCF_ECAP; }
    Scan &s = h->scan;
    // frames per launch: enough workgroups to fill 256 CUs (2^25 samples per launch), bounded so that
    // the log-magnitude ring ((avg_len + chunk) x N floats) and the four-step scratch stay modest
    static const int chunk_log2 = [] { const char *e = getenv("RCF_SCAN_CHUNK_LOG2"); return e ? atoi(e) : 25; }();
    int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(512, (int64_t(1) << chunk_log2) / fft_len));
    chunk = std::min(chunk, n_frames);
    if (s.d_vring && s.N == fft_len && s.L == avg_len && s.chunk == chunk) {
        // same geometry as the previous scan: keep every buffer (fresh device allocations cost tens of
        // milliseconds of first-touch page faults), just reset the running state
        s.n_frames = n_frames;
        s.frames_done = 0;
        s.done = false;
        s.start_sample = h->total_in;
        RCF_HIP(hipMemsetAsync(s.d_sum, 0, sizeof(float) * (size_t)fft_len, h->stream));
        RCF_HIP(hipMemsetAsync(s.d_out, 0, sizeof(float) * (size_t)fft_len, h->stream));
        s.armed = true;
        return RCF_OK;
    }
    bury(h, s.d_window); bury(h, s.d_vring); bury(h, s.d_sum); bury(h, s.d_out); bury(h, s.d_tw);
    bury(h, s.d_scratch); bury(h, s.d_peaks); bury(h, s.d_peak_ws);
    s = Scan();
    s.N = fft_len; s.n_frames = n_frames; s.L = avg_len;
    s.chunk = chunk;
    s.R = avg_len + s.chunk;
    std::vector<float> win(fft_len), tw(2 * (size_t)fft_len);
    design_window(RCF_WIN_BLACKMAN_HARRIS, fft_len, win.data());
    auto fill = [&](size_t at, int count, double step) {      // tw[at + i] = e^{-j step i}
        for (int i = 0; i < count; ++i) {
            tw[2 * (at + i)] = (float)std::cos(-step * i);
            tw[2 * (at + i) + 1] = (float)std::sin(-step * i);
        }
    };
    int n1 = 0, n2 = 0;
    if (fft_len > 16384 && scan4_split(fft_len, &n1, &n2)) {
        // four-step tables: [e^{-2 pi i n/N1} | e^{-2 pi i n/N2} | W_N^i, i<1024 | W_N^{1024 j}]
        fill(0, n1, kTwoPi / n1);
        fill((size_t)n1, n2, kTwoPi / n2);
        fill((size_t)n1 + n2, 1024, kTwoPi / fft_len);
        fill((size_t)n1 + n2 + 1024, fft_len / 1024, kTwoPi * 1024.0 / fft_len);
    } else {
        fill(0, fft_len, kTwoPi / fft_len);
    }
    RCF_HIP(hipMalloc(&s.d_window, sizeof(float) * (size_t)fft_len));
    RCF_HIP(hipMemcpy(s.d_window, win.data(), sizeof(float) * (size_t)fft_len, hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&s.d_tw, sizeof(float2) * (size_t)fft_len));
    RCF_HIP(hipMemcpy(s.d_tw, tw.data(), sizeof(float2) * (size_t)fft_len, hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&s.d_vring, sizeof(float) * (size_t)fft_len * s.R));
    RCF_HIP(hipMalloc(&s.d_sum, sizeof(float) * (size_t)fft_len));
    RCF_HIP(hipMalloc(&s.d_out, sizeof(float) * (size_t)fft_len));
    RCF_HIP(hipMemsetAsync(s.d_sum, 0, sizeof(float) * (size_t)fft_len, h->stream));
    RCF_HIP(hipMemsetAsync(s.d_out, 0, sizeof(float) * (size_t)fft_len, h->stream));
    if (fft_len > 16384) RCF_HIP(hipMalloc(&s.d_scratch, sizeof(float2) * (size_t)fft_len * s.chunk));
    s.start_sample = h->total_in;
    s.armed = true;
    return RCF_OK;
}

int rcf_scan_frames_done(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    return h->scan.armed ? h->scan.frames_done : RCF_ESTATE;
}

int rcf_scan_result(rcf_t *h, float *out)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Scan &s = h->scan;
    if (!s.armed) { set_error("scan not armed"); return RCF_ESTATE; }
    if (!s.done) return RCF_EAGAIN;
    RCF_HIP(hipMemcpyAsync(out, s.d_out, sizeof(float) * (size_t)s.N, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    return RCF_OK;
}

int rcf_scan_result_device(rcf_t *h, void **dev_spectrum)
{
    if (!h || !dev_spectrum) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->scan.armed) return RCF_ESTATE;
    if (!h->scan.done) return RCF_EAGAIN;
    *dev_spectrum = h->scan.d_out;
    return RCF_OK;
}

int rcf_find_peaks(const float *spectrum, int64_t n, double min_w, double max_w, double prominence, int64_t *idx,
                   int64_t cap, int64_t *count, double *mean_out)
{
    if (!spectrum || n < 0 || cap < 0 || (cap > 0 && !idx)) { set_error("bad find_peaks arguments"); return RCF_EINVAL; }
    const int64_t c = find_peaks_host(spectrum, n, min_w, max_w, prominence, idx, cap, mean_out);
    if (count) *count = c;
    return RCF_OK;
}

int64_t rcf_peak_frequency(int64_t line, double samp_rate, int64_t fft_len, double center_freq)
{
    const double hz_per_bin = samp_rate / (double)fft_len;
    return (int64_t)(((double)line * hz_per_bin) - (samp_rate / 2) + center_freq);
}

int rcf_scan_find_peaks(rcf_t *h, double prominence, int64_t *idx, int64_t cap, int64_t *count, double *mean_out,
                        void **dev_idx)
{
    if (!h || cap < 1) { set_error("bad arguments"); return RCF_EINVAL; }
    if (cap > 4096) cap = 4096;                       // device sort capacity
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Scan &s = h->scan;
    if (!s.armed) return RCF_ESTATE;
    if (!s.done) return RCF_EAGAIN;
    const int N = s.N;
    const double hz_per_bin = h->fs / N;              // fft_peak_detection.py:46-52
    if (!s.d_peak_ws) RCF_HIP(hipMalloc(&s.d_peak_ws, peaks_workspace_bytes(N)));
    if (!s.d_peaks) RCF_HIP(hipMalloc(&s.d_peaks, sizeof(int64_t) * 4096));
    int *d_count = nullptr;
    double *d_mean = nullptr;
    launch_find_peaks(s.d_out, N, 3000 / hz_per_bin, 30000 / hz_per_bin, prominence, s.d_peak_ws, s.d_peaks,
                      (int)cap, &d_count, &d_mean, h->stream);
    int c = 0;
    double mean = 0.0;
    RCF_HIP(hipMemcpyAsync(&c, d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipMemcpyAsync(&mean, d_mean, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (idx) RCF_HIP(hipMemcpyAsync(idx, s.d_peaks, sizeof(int64_t) * (size_t)cap, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    if (count) *count = c;
    if (mean_out) *mean_out = mean;
    if (dev_idx) *dev_idx = s.d_peaks;                // sorted ascending, -1 padded to `cap`
    return RCF_OK;
}

}  // extern "C"
