// rcf_state.h -- host-side state of librcf.so (the kernels never see it): the front-end handle with its channels,
// filterbank, scanner, slab pools and launch arenas, and the helpers the host modules share.
//   rcf_handle.cpp   open / close / sync, pools, wideband ingest        rcf_plan.cpp    the per-block schedule (host)
//   rcf_launch.cpp   the block's launches in dependency order           rcf_chan.cpp    channels: lifecycle, reads, voice chain
//   rcf_bank.cpp     filterbank + scanner ABI                           rcf_timing.cpp  HIP-event timing
//   rcf_comm.cpp     RCCL peak-list exchange                            rcf_group.cpp   grouped launches over front-ends
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "rcf_internal.h"

namespace rcfx {

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
    // ---- what planning a block reads and writes, together at the front of the object (three cache lines; plan_block
    // prefetches them): with a thousand front-ends x 256 channels the planner's time is the cache misses of this walk
    int id = -1;
    int src = -1;                 // -1 wideband; RCF_SRC_PFB_BIN0 + bin; else source channel id
    int D = 0, T = 0;
    int depth = 0;
    bool fm_only = false;         // rcf_chan_set_fm_only: tap_finalize writes the discriminator ring only (a tap's IQ ring keeps
                                  // just the launch's last output, for the next launch's first discriminator sample)
    bool is_tap = false;          // a bin of a frame-major filterbank open as a channel: the bank's kernel copies it
                                  // into the launch's tap matrix, tap_finalize_kernel fills the rings
    float2 *d_ctaps = nullptr;
    float2 *d_iq = nullptr;
    float *d_fm = nullptr;
    int64_t start_sample = 0;     // in source index space
    int64_t k_abs0 = 0;
    int64_t produced = 0;         // relative output count
    // exact rotator (rcf_set_rotator): phase ring + {phase, counter} state, one pool slice
    float2 *d_rot = nullptr;
    float *d_sym = nullptr;       // optional real FIR over gain * fm (P25 symbol filter), below
    // rotator model
    double extra_dangle = 0, extra_dlogmag = 0;   // added to the increment's own angle / log magnitude (filterbank taps)
    double dangle = 0, dlogmag = 0;
    long double angle0 = 0;
    double logmag0 = 0;
    int64_t n_seg0 = 0;
    // output range [blk_before, blk_after) the block with serial blk_serial gave this channel (its derived channels'
    // input range; process_block)
    uint64_t blk_serial = 0;
    int64_t blk_before = 0, blk_after = 0;
    int64_t rd_iq = 0, rd_fm = 0;
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
    // ---- the rest
    float incr[2] = {1.f, 0.f};   // exact rotator: what GNU Radio iterates
    float *d_symtaps = nullptr;
    int sym_ntaps = 0;
    float sym_gain = 1.f;
    int64_t sym_from = 0;         // first relative output index the filter is defined for
    int64_t rd_sym = 0;
    uint64_t many_stamp = 0;      // the rcf_chan_read_many call that last listed this channel
    double src_rate = 0, offset_hz = 0;
    uint64_t taps_version = 0;    // bumped whenever d_ctaps changes (bank-matrix cache key)
    std::vector<float> proto;     // prototype taps (host)
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
    // the discriminator fused into a frame-major bank (rcf_pfb_fm_enable): d_fm[(i & ring_mask) NB + k], i = frame - n_abs0
    int fm_mode = 0;               // 0 off, 1 beside the bins ring, 2 instead of it
    int fm_gr_phase = 0;
    float *d_fm = nullptr;
    float2 *d_fm_inc = nullptr;    // [NB] per-bin rotator increment as a phasor
    float *d_fm_stage = nullptr;   // contiguous staging for rcf_pfb_read_fm
    int64_t fm_from = 0;           // first relative frame the discriminator ring holds
    int64_t fm_until = 0;          // (fm_mode == 0) the frame the discriminator was switched off at
    // look-back form (pfb5_fmlb_kernel): edge rows + flags the chunks' workgroups hand their last frames over through
    unsigned long long *d_fm_edge = nullptr, *d_fm_flag = nullptr;
    int *d_fm_err = nullptr;
    int fm_slots = 0;
    int fm_local = 0;              // the hand-over stays in one XCD's L2 (pfb5_xcd_map_ok said so; RCF_PFB5_FM_LOCAL=0: never)
    uint64_t fm_serial = 0;        // launches so far (the flags' tags)
    std::vector<int64_t> rd_fm;    // per-bin read cursors
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

namespace rcfx {
// Launch-parameter arenas: two pinned host buffers with device twins.  The records of successive blocks are APPENDED to the
// current one; only when it is full is an event recorded (every hipEventRecord costs ~6 us of queue gap: rocprof trace
// of the timed configuration) and the other one taken, once the kernels that read it have finished.
struct ArenaSet {
    size_t cap = 8u << 20;
    unsigned char *h[2] = {nullptr, nullptr};
    unsigned char *h_dev[2] = {nullptr, nullptr};     // the same pinned memory as the device sees it
    unsigned char *d[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    int cur = 0;
    size_t fill = 0;              // bytes of the current arena taken by earlier commits
    bool mapped = true;           // both pinned arenas are visible to the device (copy kernels can read them)
    int create();
    void destroy();               // the stream that read them is idle
    int reserve(size_t need, hipStream_t stream);   // room for `need` more bytes in arena `cur` from `fill` on
};
}  // namespace rcfx

using rcfx::Chan;
using rcfx::Pfb;
using rcfx::Scan;

struct rcf {
    int device = 0;
    int fault = 0;                 // sticky: a group block failed after its launches had been queued (set_dev refuses from then on)
    char fault_text[160] = "";
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
    rcfx::ArenaSet arenas;
    bool copy_kernels = true;     // RCF_COPY_KERNELS=0: hipMemcpyAsync for the launch records and the history (A/B)
    // Stage-2 lag (rcf_launch.cpp): the small-T FIR + discriminator launch of the last block has NOT been queued -- it rides
    // in the next block's filterbank launch (S2Rider), or goes out on its own as soon as anybody could look at its outputs
    // (every entry point that touches the stream flushes it: set_dev).  RCF_S2_LAG=0 / rcf_set_stage2_lag(h, 0): off.
    struct Lag {
        bool pending = false;
        rcfx::FirLaunchDims dims{};
        const rcfx::ChanLaunch *dev = nullptr;
        int64_t frames = 0;           // bank frames of the block it belongs to (ring room: the next block must not overwrite them)
    } lag;
    bool lag_enabled = true;
    // a handle that belongs to a group (rcf_group_open) runs on the group's stream; its own comes back at rcf_group_close
    struct rcf_group *group = nullptr;
    hipStream_t own_stream = nullptr;
    std::map<int, std::unique_ptr<Chan>> chans;
    uint64_t chans_epoch = 0;     // bumped whenever a channel is opened or closed or gains / loses a symbol filter or voice
                                  // chain (cached Chan pointers: the pump's; the cached arena need below)
    // What planning a block needs to know about the channel SET (not their counters), valid while epoch == chans_epoch:
    // the summary plan_arena() used to rebuild from a walk over every channel, and the (depth, D, T) classes plan_block()
    // used to re-bucket -- three passes of pointer chasing per block (20 us of a 30 us plan for a front-end with 256
    // tapped bins once a thousand front-ends no longer fit the host's caches, RCF_PLAN_PROF).
    struct PlanCache {
        uint64_t epoch = ~0ull;
        std::unordered_map<int, size_t> reach_x;
        int max_depth = 0, min_d0 = 0;
        size_t max_reach = 1, arena_need = 0;
        typedef std::pair<std::pair<int, int>, std::vector<Chan *>> ClassBucket;
        std::vector<std::vector<ClassBucket>> by_depth;     // classes sorted by (D, T); channels in id order
    } plan_cache;
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

namespace rcfx {

// ---------------------------------------------------------------- rcf_handle.cpp
int set_dev(rcf_t *h);               // hipSetDevice + flush_lagged: what every entry point that touches the stream calls
int set_dev_ingest(rcf_t *h);        // hipSetDevice only: push / commit decide themselves what becomes of a lagging launch
void flush_lagged(rcf_t *h);         // rcf_launch.cpp
void bury(rcf_t *h, void *p, size_t slice = 0);
void free_graveyard_idle(rcf_t *h);      // the stream is known to be idle (the caller just synchronised it)
void drain_graveyard(rcf_t *h);
size_t slice_round(size_t bytes);
void *pool_get(rcf_t *h, size_t bytes);  // one slice of `bytes` (a multiple of 256) from the handle's pools

// ---------------------------------------------------------------- rcf_timing.cpp
hipEvent_t time_event(rcf_t *h);
void time_collect(rcf_t *h);

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

// ---------------------------------------------------------------- rcf_chan.cpp
// source description for one commit
struct SrcRange {
    StreamView view;
    int64_t p0, p1;      // new samples [p0, p1) in the source's index space
};
bool source_range(rcf_t *h, int src, int64_t S0, int64_t S1, SrcRange *out);
int upload_composite(rcf_t *h, Chan *c);
double pfb_tap_gr_dangle(const rcf_t *h, int bin);
int pfb_fm_upload_increments(rcf_t *h);
int new_channel(rcf_t *h, int src, int D, const float *taps, int T, double offset_hz, int *chan_id);
void free_channel(rcf_t *h, Chan *c);
int64_t ring_read_enqueue(rcf_t *h, const void *ring, size_t elem, int64_t produced, int64_t *cursor, void *out,
                          size_t max_items);
int64_t ring_read(rcf_t *h, const void *ring, size_t elem, int64_t produced, int64_t *cursor, void *out,
                  size_t max_items);

// ---------------------------------------------------------------- rcf_plan.cpp / rcf_launch.cpp
int process_block(rcf_t *h, size_t n);

// ---------------------------------------------------------------- rcf_comm.cpp
void comm_destroy(rcf_t *h);

}  // namespace rcfx
