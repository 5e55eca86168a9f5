// rcf_pump.cpp -- the native real-time pump: one thread per group of front-ends that takes whichever input blocks are
// complete, issues them as one group block (rcf_group.cpp), keeps two group blocks in flight and delivers every channel's
// output into its pinned host ring.  It replaces the interpreter between "the block's last sample exists" and "its outputs
// are in host memory" (the per-source work loop of rc_frontend/receiver.py:477-700 as GNU Radio's scheduler threads run it).
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/resource.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>

#include "rcf_group.h"

using namespace rcfx;

// =================================================================== the pump
struct rcf_pump {
    rcf_group *g = nullptr;
    rcf_pump_config_t cfg{};
    std::vector<const unsigned char *> rings, rings_dev;     // host address / the same memory as the device sees it (or nullptr)
    std::vector<size_t> ring_blocks;
    std::vector<double> phase;
    std::vector<const volatile uint64_t *> written;
    // subscription slots (guarded by g->mu): slot e delivers channel rd_chan[e] of member rd_member[e] (-1: free) as
    // rd_what[e] (RCF_READ_IQ / RCF_READ_FM x rd_gain[e]).  Slots are fixed at start (cfg.n_read of them subscribed, room
    // for max_read): rcf_pump_subscribe / rcf_pump_unsubscribe move channels in and out while the thread runs
    std::vector<int> rd_member, rd_chan, rd_what;
    std::vector<float> rd_gain;
    std::vector<std::vector<int>> entries_of;      // member -> indices into rd_*
    std::vector<int64_t> bin_rd;                   // slots on a BIN of a member's fused discriminator ring (rd_chan >= RCF_SRC_PFB_BIN0): frames delivered
    std::vector<int64_t> queued;                   // items ever queued for the host ring of each slot
    std::vector<Chan *> chan_of;                   // subscribed channels, resolved once per channel-set epoch
    std::vector<uint64_t> epoch_of;                // per member: the epoch chan_of was resolved at (~0: resolve again)
    size_t out_cap = 0;                            // host ring length per slot (items, power of two)
    static constexpr size_t kSlotItemBytes = 8;    // every slot is out_cap x 8 bytes (a discriminator slot uses half)
    unsigned char *h_out = nullptr, *h_out_dev = nullptr;   // the slots' rings
    std::unique_ptr<std::atomic<int64_t>[]> out_written;    // items delivered per slot
    std::unique_ptr<std::atomic<int64_t>[]> out_queued;     // items whose gather has been (or is about to be) launched
    // gather records of the two group blocks in flight (pinned)
    unsigned char *h_recs = nullptr, *h_recs_dev = nullptr;
    size_t recs_cap = 0;                           // records per slot
    hipEvent_t slot_ev[2] = {nullptr, nullptr};
    std::thread th;
    std::atomic<bool> stop{false}, running{false};
    std::atomic<int> error{0}, rt_granted{0};
    std::mutex st_mu;                              // the statistics below
    std::vector<float> lat_ms;
    int64_t blocks_done = 0, judged = 0, late = 0, overruns = 0, group_blocks = 0, max_batch = 0, samples_out = 0;
    double plan_ms = 0, wait_ms = 0;
    double max_plan_ms = 0, max_wait_ms = 0, max_idle_gap_ms = 0;   // longest single planning / device wait / sleep overshoot
    int64_t slow_plans = 0, slow_waits = 0, slow_sleeps = 0;        // ... and how many of them exceeded 5 / 5 / 2 ms (after the warm-up)
    bool warm_done = false;
    // why late: run-queue delay of the thread (schedstat), see rcf_pump_stats_t
    double runq_total_ms = -1, slow_wait_total_ms = 0, runq_in_slow_waits_ms = 0, slow_sleep_total_ms = 0, runq_in_slow_sleeps_ms = 0;
    int64_t nivcsw = 0;
    std::chrono::steady_clock::time_point t_start, t_end;
    char err_text[256] = "";
    size_t item_bytes(int e) const { return rd_what[(size_t)e] == RCF_READ_IQ ? sizeof(float2) : sizeof(float); }
    ~rcf_pump()
    {
        // (also what an rcf_pump_start that fails half-way leaves through: pinned rings, records, events)
        for (int i = 0; i < 2; ++i) if (slot_ev[i]) (void)hipEventDestroy(slot_ev[i]);
        if (h_recs) (void)hipHostFree(h_recs);
        if (h_out) (void)hipHostFree(h_out);
    }
};

namespace {

using Clock = std::chrono::steady_clock;
inline double secs(Clock::duration d) { return std::chrono::duration<double>(d).count(); }

struct InFlight {
    bool busy = false;
    std::vector<int> members;          // member indices of the batch
    std::vector<double> due;           // per member: when its block was complete (seconds since t0)
    std::vector<int64_t> kidx;         // per member: which of its blocks
    std::vector<std::pair<int, int64_t>> delivered;   // (entry, items) to publish once the gather has run
};

// nanoseconds this thread has spent runnable without a CPU (second field of /proc/thread-self/schedstat); -1: unreadable
long long run_delay_ns(int fd)
{
    if (fd < 0) return -1;
    char b[96];
    const ssize_t n = pread(fd, b, sizeof b - 1, 0);
    if (n <= 0) return -1;
    b[n] = 0;
    long long run = 0, delay = 0;
    return std::sscanf(b, "%lld %lld", &run, &delay) == 2 ? delay : -1;
}

long thread_nivcsw()
{
    rusage ru{};
    return getrusage(RUSAGE_THREAD, &ru) == 0 ? ru.ru_nivcsw : 0;
}

void pump_fail(rcf_pump *p, int code)
{
    p->error.store(code);
    std::snprintf(p->err_text, sizeof p->err_text, "%s", rcf_last_error());
}

void pump_main(rcf_pump *p)
{
    rcf_group *g = p->g;
    const rcf_pump_config_t &cfg = p->cfg;
    const size_t G = g->members.size();
    if (cfg.cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cfg.cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    if (cfg.rt_priority > 0) {
        sched_param sp{};
        sp.sched_priority = cfg.rt_priority;
        p->rt_granted.store(pthread_setschedparam(pthread_self(), SCHED_FIFO, &sp) == 0 ? 1 : 0);
    }
    (void)hipSetDevice(g->device);
    const int sfd = open("/proc/thread-self/schedstat", O_RDONLY | O_CLOEXEC);
    long long rq_warm = -1;                        // run_delay / involuntary switches when the judged region began
    long nv_warm = 0;
    const double period = (double)cfg.block_samples / cfg.samp_rate;
    const size_t blk_bytes = cfg.block_samples * group_sample_bytes(cfg.fmt);
    const Clock::time_point t0 = Clock::now() + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(cfg.start_delay_s));
    { std::lock_guard<std::mutex> l(p->st_mu); p->t_start = t0; }
    std::vector<int64_t> next_k(G, 0);             // next block of each member
    std::vector<double> seen_at(G, -1.0);          // counter-fed members: when the pump first saw block next_k complete
    InFlight slots[2];
    int head = 0, in_flight = 0;                   // slots[head] is the oldest busy one
    std::vector<GroupItem> items;
    auto &queued = p->queued;
    auto &chan_of = p->chan_of;
    auto &epoch_of = p->epoch_of;
    constexpr uint32_t slot_w = (uint32_t)(rcf_pump::kSlotItemBytes / 4);   // words per item a slot is laid out for

    auto complete_oldest = [&](bool block) -> bool {
        InFlight &s = slots[head];
        if (!s.busy) return false;
        if (!block && hipEventQuery(p->slot_ev[head]) != hipSuccess) { (void)hipGetLastError(); return false; }
        const long long rq0 = run_delay_ns(sfd);
        const Clock::time_point w0 = Clock::now();
        if (hipEventSynchronize(p->slot_ev[head]) != hipSuccess) { set_error("pump: event wait failed"); pump_fail(p, RCF_EHIP); return false; }
        const Clock::time_point now = Clock::now();
        const long long rq1 = secs(now - w0) > 5e-3 ? run_delay_ns(sfd) : -1;
        const double t_done = secs(now - t0);
        int64_t items_out = 0;
        for (auto &d : s.delivered) { p->out_written[(size_t)d.first].fetch_add(d.second, std::memory_order_release); items_out += d.second; }
        {
            std::lock_guard<std::mutex> l(p->st_mu);
            p->wait_ms += secs(now - w0) * 1e3;
            if (p->warm_done) {
                p->max_wait_ms = std::max(p->max_wait_ms, secs(now - w0) * 1e3);
                if (secs(now - w0) > 5e-3) {
                    ++p->slow_waits;
                    p->slow_wait_total_ms += secs(now - w0) * 1e3;
                    if (rq0 >= 0 && rq1 >= 0) p->runq_in_slow_waits_ms += (double)(rq1 - rq0) * 1e-6;
                }
            }
            p->samples_out += items_out;
            for (size_t i = 0; i < s.members.size(); ++i) {
                ++p->blocks_done;
                if (s.kidx[i] >= cfg.warm_blocks) {
                    const double lat = t_done - s.due[i];
                    p->lat_ms.push_back((float)(lat * 1e3));
                    ++p->judged;
                    if (lat > period) ++p->late;
                }
            }
        }
        s.busy = false;
        head ^= 1;
        --in_flight;
        return true;
    };

    while (!p->stop.load(std::memory_order_relaxed) && !p->error.load()) {
        // members whose next block is complete
        const double now_s = secs(Clock::now() - t0);
        items.clear();
        std::vector<double> due;
        double next_due = 1e300;
        bool all_finished = true;
        for (size_t m = 0; m < G; ++m) {
            const int64_t k = next_k[m];
            if (cfg.n_blocks > 0 && k >= cfg.n_blocks) continue;
            all_finished = false;
            if (cfg.max_batch > 0 && (int)items.size() >= cfg.max_batch) { next_due = std::min(next_due, now_s); continue; }
            double d;
            if (p->written[m]) {
                if (*p->written[m] <= (uint64_t)k) { next_due = std::min(next_due, now_s + 50e-6); continue; }
                if (seen_at[m] < 0) seen_at[m] = now_s;
                d = seen_at[m];
            } else {
                d = (double)(k + 1) * period + p->phase[m];
                if (d > now_s) { next_due = std::min(next_due, d); continue; }
            }
            const size_t at = (size_t)(k % (int64_t)p->ring_blocks[m]) * blk_bytes;
            items.push_back(GroupItem{(int)m, cfg.block_samples, p->rings[m] + at, p->rings_dev[m] ? p->rings_dev[m] + at : nullptr});
            due.push_back(d);
        }
        if (all_finished && in_flight == 0) break;
        // batching window: the first block that is complete waits up to batch_window_s for company -- every block that
        // completes meanwhile rides in the same launches (at K front-ends a block completes every period / K)
        bool hold = false;
        if (!items.empty() && cfg.batch_window_s > 0 && !(cfg.max_batch > 0 && (int)items.size() >= cfg.max_batch)) {
            const double oldest = *std::min_element(due.begin(), due.end());
            if (now_s - oldest < cfg.batch_window_s) { hold = true; next_due = std::min(next_due, oldest + cfg.batch_window_s); }
        }
        if (!items.empty() && !hold && in_flight < 2) {
            const Clock::time_point p0 = Clock::now();
            const int slot = head ^ (in_flight & 1);
            InFlight &s = slots[slot];
            s.members.clear();
            s.kidx.clear();
            s.due = due;
            s.delivered.clear();
            int n_over = 0;
            for (size_t i = 0; i < items.size(); ++i) {
                s.members.push_back(items[i].m);
                s.kidx.push_back(next_k[(size_t)items[i].m]);
                if (now_s - due[i] > period && next_k[(size_t)items[i].m] >= cfg.warm_blocks) ++n_over;
            }
            int rc;
            {
                std::lock_guard<std::mutex> gl(g->mu);
                MemberLocks ml(g->members);
                rc = group_process(g, items, cfg.fmt, cfg.scale, cfg.offset, false);
                if (rc == RCF_OK) {
                    // the read of this batch: every subscribed channel of its members, straight into the host rings
                    GatherRec *recs = reinterpret_cast<GatherRec *>(p->h_recs + (size_t)slot * p->recs_cap * sizeof(GatherRec));
                    int n_recs = 0;
                    uint32_t max_w = 0;
                    for (const GroupItem &it : items) {
                        rcf_t *h = g->members[(size_t)it.m];
                        if (epoch_of[(size_t)it.m] != h->chans_epoch) {            // channels were opened / closed: look them up again
                            for (int e : p->entries_of[(size_t)it.m]) {
                                if (p->rd_chan[(size_t)e] >= RCF_SRC_PFB_BIN0) continue;
                                auto f = h->chans.find(p->rd_chan[(size_t)e]);
                                chan_of[(size_t)e] = f == h->chans.end() ? nullptr : f->second.get();
                            }
                            epoch_of[(size_t)it.m] = h->chans_epoch;
                        }
                        for (int e : p->entries_of[(size_t)it.m]) {
                            if (p->rd_chan[(size_t)e] >= RCF_SRC_PFB_BIN0) {
                                // one bin of the bank's fused discriminator ring (rcf_pfb_fm_enable): frame-major floats,
                                // read with a stride of n_bins words
                                Pfb &pf = h->pfb;
                                const int bin = p->rd_chan[(size_t)e] - RCF_SRC_PFB_BIN0;
                                if (!pf.open || !pf.d_fm || bin >= pf.NB) continue;
                                const int64_t end = pf.fm_mode ? pf.produced : pf.fm_until;
                                int64_t &cur = p->bin_rd[(size_t)e];
                                int64_t avail = end - cur;
                                if (avail <= 0) continue;
                                if ((size_t)avail > h->out_cap) { cur = end - (int64_t)h->out_cap; avail = (int64_t)h->out_cap; }
                                if ((size_t)avail > p->out_cap) { cur += avail - (int64_t)p->out_cap; avail = (int64_t)p->out_cap; }
                                const float gain = p->rd_gain[(size_t)e];
                                const uint64_t dst_pos = (uint64_t)queued[(size_t)e] & (p->out_cap - 1);
                                queued[(size_t)e] += avail;
                                p->out_queued[(size_t)e].store(queued[(size_t)e], std::memory_order_release);
                                recs[n_recs++] = GatherRec{reinterpret_cast<const uint32_t *>(pf.d_fm + bin),
                                                           (uint32_t)((uint64_t)cur & h->ring_mask), (uint32_t)avail,
                                                           (uint32_t)(h->out_cap - 1), (uint32_t)((size_t)e * p->out_cap * slot_w),
                                                           (uint32_t)dst_pos, (uint32_t)(p->out_cap - 1), gain, gain != 1.0f ? 1u : 0u,
                                                           (uint32_t)pf.NB};
                                max_w = std::max<uint32_t>(max_w, (uint32_t)avail);
                                cur += avail;
                                s.delivered.push_back({e, avail});
                                continue;
                            }
                            Chan *c = chan_of[(size_t)e];
                            if (!c) continue;                                      // closed under the pump: starves
                            const int what = p->rd_what[(size_t)e];
                            const float gain = p->rd_gain[(size_t)e];
                            const uint32_t ew = what == RCF_READ_IQ ? 2u : 1u;     // words per item of this slot's stream
                            if (what == RCF_READ_IQ && c->fm_only) continue;       // discriminator only: no IQ stream to hand out
                            if (what == RCF_READ_FM && !c->d_fm) continue;
                            int64_t *cur = what == RCF_READ_IQ ? &c->rd_iq : &c->rd_fm;
                            int64_t avail = c->produced - *cur;
                            if (avail <= 0) continue;
                            if ((size_t)avail > h->out_cap) { *cur = c->produced - (int64_t)h->out_cap; avail = (int64_t)h->out_cap; }
                            if ((size_t)avail > p->out_cap) { *cur += avail - (int64_t)p->out_cap; avail = (int64_t)p->out_cap; }
                            const uint64_t dst_pos = (uint64_t)queued[(size_t)e] & (p->out_cap - 1);
                            queued[(size_t)e] += avail;
                            // (published BEFORE the launch: a reader that copied positions this gather overwrites finds out)
                            p->out_queued[(size_t)e].store(queued[(size_t)e], std::memory_order_release);
                            recs[n_recs++] = GatherRec{static_cast<const uint32_t *>(what == RCF_READ_IQ ? (const void *)c->d_iq : (const void *)c->d_fm),
                                                       (uint32_t)(((uint64_t)*cur & h->ring_mask) * ew), (uint32_t)avail * ew,
                                                       (uint32_t)(h->out_cap * ew - 1), (uint32_t)((size_t)e * p->out_cap * slot_w),
                                                       (uint32_t)(dst_pos * ew), (uint32_t)(p->out_cap * ew - 1), gain,
                                                       (what == RCF_READ_FM && gain != 1.0f) ? 1u : 0u};
                            max_w = std::max<uint32_t>(max_w, (uint32_t)avail * ew);
                            *cur += avail;
                            s.delivered.push_back({e, avail});
                        }
                    }
                    if (n_recs)
                        launch_gather_rings(reinterpret_cast<const GatherRec *>(p->h_recs_dev + (size_t)slot * p->recs_cap * sizeof(GatherRec)),
                                            n_recs, reinterpret_cast<uint32_t *>(p->h_out_dev), max_w, g->stream);
                    if (hipEventRecord(p->slot_ev[slot], g->stream) != hipSuccess) { set_error("pump: event record failed"); rc = RCF_EHIP; }
                }
            }
            if (rc != RCF_OK) { pump_fail(p, rc); break; }
            for (const GroupItem &it : items) { ++next_k[(size_t)it.m]; seen_at[(size_t)it.m] = -1.0; }
            s.busy = true;
            ++in_flight;
            {
                std::lock_guard<std::mutex> l(p->st_mu);
                p->plan_ms += secs(Clock::now() - p0) * 1e3;
                if (!p->warm_done && *std::min_element(s.kidx.begin(), s.kidx.end()) >= cfg.warm_blocks) {
                    p->warm_done = true;
                    rq_warm = run_delay_ns(sfd);
                    nv_warm = thread_nivcsw();
                }
                if (p->warm_done) {
                    p->max_plan_ms = std::max(p->max_plan_ms, secs(Clock::now() - p0) * 1e3);
                    if (secs(Clock::now() - p0) > 5e-3) ++p->slow_plans;
                }
                ++p->group_blocks;
                p->max_batch = std::max<int64_t>(p->max_batch, (int64_t)items.size());
                p->overruns += n_over;
            }
            (void)complete_oldest(false);          // (usually the previous batch has finished by now)
            continue;
        }
        if (in_flight > 0) {
            // nothing to queue (or both slots taken): the oldest batch's outputs are what the host waits for
            if ((!items.empty() && !hold) || next_due - now_s > 200e-6) { (void)complete_oldest(true); continue; }
            if (complete_oldest(false)) continue;
        }
        const double wait_s = next_due - secs(Clock::now() - t0);
        if (wait_s > 0) {
            const double want = std::min(wait_s, 1e-3);
            const long long rq0 = run_delay_ns(sfd);
            const Clock::time_point s0 = Clock::now();
            if (cfg.spin_us > 0 && want <= cfg.spin_us * 1e-6) {
                while (secs(Clock::now() - s0) < want && !p->stop.load(std::memory_order_relaxed)) __builtin_ia32_pause();
            } else {
                std::this_thread::sleep_for(std::chrono::duration<double>(cfg.spin_us > 0 ? want - cfg.spin_us * 0.5e-6 : want));
                while (secs(Clock::now() - s0) < want && cfg.spin_us > 0) __builtin_ia32_pause();
            }
            const double over = (secs(Clock::now() - s0) - want) * 1e3;      // how much later than asked the thread came back
            if (p->warm_done && over > 2.0) {
                const long long rq1 = run_delay_ns(sfd);
                std::lock_guard<std::mutex> l(p->st_mu);
                ++p->slow_sleeps;
                p->max_idle_gap_ms = std::max(p->max_idle_gap_ms, over);
                p->slow_sleep_total_ms += over;
                if (rq0 >= 0 && rq1 >= 0) p->runq_in_slow_sleeps_ms += (double)(rq1 - rq0) * 1e-6;
            }
        }
    }
    while (in_flight > 0 && !p->error.load()) (void)complete_oldest(true);
    (void)hipStreamSynchronize(g->stream);
    {
        const long long rq_end = run_delay_ns(sfd);
        std::lock_guard<std::mutex> l(p->st_mu);
        p->t_end = Clock::now();
        if (rq_warm >= 0 && rq_end >= 0) p->runq_total_ms = (double)(rq_end - rq_warm) * 1e-6;
        p->nivcsw = thread_nivcsw() - nv_warm;
    }
    if (sfd >= 0) close(sfd);
    p->running.store(false);
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ pump
int rcf_pump_start(rcf_group_t *g, const rcf_pump_config_t *cfg, rcf_pump_t **out)
{
    if (!g || !cfg || !out || cfg->block_samples == 0 || !(cfg->samp_rate > 0) || !cfg->rings || !cfg->ring_blocks ||
        cfg->n_read < 0 || (cfg->n_read && (!cfg->read_members || !cfg->read_chans)) ||
        (cfg->what != RCF_READ_IQ && cfg->what != RCF_READ_FM) || group_sample_bytes(cfg->fmt) == 0) {
        set_error("bad pump configuration");
        return RCF_EINVAL;
    }
    std::lock_guard<std::mutex> gl(g->mu);
    if (g->pump) { set_error("the group already has a pump"); return RCF_ESTATE; }
    RCF_HIP(hipSetDevice(g->device));
    const size_t G = g->members.size();
    std::unique_ptr<rcf_pump> p(new rcf_pump);
    p->g = g;
    p->cfg = *cfg;
    for (size_t m = 0; m < G; ++m) {
        if (!cfg->rings[m] || cfg->ring_blocks[m] == 0) { set_error("member %zu has no source ring", m); return RCF_EINVAL; }
        if (cfg->block_samples > g->members[m]->block_cap) { set_error("block of %zu samples exceeds member %zu's capacity", cfg->block_samples, m); return RCF_ECAP; }
        p->rings.push_back(static_cast<const unsigned char *>(cfg->rings[m]));
        void *rdv = nullptr;
        if (hipHostGetDevicePointer(&rdv, const_cast<void *>(cfg->rings[m]), 0) != hipSuccess) { rdv = nullptr; (void)hipGetLastError(); }
        p->rings_dev.push_back(static_cast<const unsigned char *>(rdv));
        p->ring_blocks.push_back(cfg->ring_blocks[m]);
        p->phase.push_back(cfg->phase_s ? cfg->phase_s[m] : 0.0);
        p->written.push_back(cfg->written ? cfg->written[m] : nullptr);
    }
    const int n_slots = std::max(std::max(cfg->n_read, cfg->max_read), 1);
    p->entries_of.resize(G);
    p->rd_member.assign((size_t)n_slots, -1);
    p->rd_chan.assign((size_t)n_slots, -1);
    p->rd_what.assign((size_t)n_slots, cfg->what);
    p->rd_gain.assign((size_t)n_slots, cfg->gain);
    p->queued.assign((size_t)n_slots, 0);
    p->bin_rd.assign((size_t)n_slots, 0);
    p->chan_of.assign((size_t)n_slots, nullptr);
    p->epoch_of.assign(G, ~0ull);
    for (int e = 0; e < cfg->n_read; ++e) {
        if (cfg->read_members[e] < 0 || cfg->read_members[e] >= (int)G) { set_error("subscribed channel %d: no such member", e); return RCF_EINVAL; }
        p->rd_member[(size_t)e] = cfg->read_members[e];
        p->rd_chan[(size_t)e] = cfg->read_chans[e];
        p->entries_of[(size_t)cfg->read_members[e]].push_back(e);
    }
    // the configuration's arrays belong to the caller: from here on the pump's own copies are used
    p->cfg.rings = nullptr; p->cfg.ring_blocks = nullptr; p->cfg.phase_s = nullptr; p->cfg.written = nullptr;
    p->cfg.read_members = nullptr; p->cfg.read_chans = nullptr;
    p->out_cap = pow2_at_least(cfg->out_ring_samples ? cfg->out_ring_samples : 4096);
    const size_t out_bytes = std::max<size_t>(64, (size_t)n_slots * p->out_cap * rcf_pump::kSlotItemBytes);
    if ((uint64_t)out_bytes / 4 > 0xffffffffull) { set_error("host rings of %zu bytes exceed the 32-bit word range", out_bytes); return RCF_ECAP; }
    void *hp = nullptr, *dv = nullptr;
    if (hipHostMalloc(&hp, out_bytes, hipHostMallocDefault) != hipSuccess || hipHostGetDevicePointer(&dv, hp, 0) != hipSuccess) {
        if (hp) (void)hipHostFree(hp);
        (void)hipGetLastError();
        set_error("pinned host rings of %zu bytes failed", out_bytes);
        return RCF_ENOMEM;
    }
    p->h_out = static_cast<unsigned char *>(hp);
    p->h_out_dev = static_cast<unsigned char *>(dv);
    p->out_written.reset(new std::atomic<int64_t>[(size_t)n_slots]);
    p->out_queued.reset(new std::atomic<int64_t>[(size_t)n_slots]);
    for (int e = 0; e < n_slots; ++e) { p->out_written[(size_t)e].store(0); p->out_queued[(size_t)e].store(0); }
    p->recs_cap = (size_t)n_slots;
    hp = dv = nullptr;
    if (hipHostMalloc(&hp, 2 * p->recs_cap * sizeof(GatherRec), hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer(&dv, hp, 0) != hipSuccess) {
        if (hp) (void)hipHostFree(hp);
        (void)hipGetLastError();
        set_error("pinned gather records failed");
        return RCF_ENOMEM;                            // (~rcf_pump releases the rings)
    }
    p->h_recs = static_cast<unsigned char *>(hp);
    p->h_recs_dev = static_cast<unsigned char *>(dv);
    {
        // RCF_PUMP_BLOCKING=1: the pump sleeps in the driver while it waits for a group block instead of spinning on the event
        static const bool blocking = [] { const char *e = getenv("RCF_PUMP_BLOCKING"); return e && atoi(e) != 0; }();
        for (int i = 0; i < 2; ++i)
            RCF_HIP(hipEventCreateWithFlags(&p->slot_ev[i], hipEventDisableTiming | (blocking ? hipEventBlockingSync : 0)));
    }
    // room in the group's arena for a group block of ALL members at once (after a hiccup everything that is complete goes
    // out together): growing the arena means a stream synchronisation and pinned allocations -- not in the middle of a run
    {
        MemberLocks ml(g->members);
        size_t all = 65536;
        for (rcf_t *h : g->members) all += arena_need_bound(h) + 1024;
        if (!g->arenas.h[0] && g->arenas.cap < all) { size_t c_ = g->arenas.cap; while (c_ < all) c_ *= 2; g->arenas.cap = c_; }
        if (g->arenas.reserve(all, g->stream) != RCF_OK) return RCF_EHIP;
    }
    // the subscribed channels' readers start at what has been produced so far
    {
        MemberLocks ml(g->members);
        for (int e = 0; e < cfg->n_read; ++e) {
            rcf_t *h = g->members[(size_t)p->rd_member[(size_t)e]];
            if (p->rd_chan[(size_t)e] >= RCF_SRC_PFB_BIN0) {              // a bin of the fused discriminator ring
                p->rd_what[(size_t)e] = RCF_READ_FM;
                p->bin_rd[(size_t)e] = h->pfb.produced;
                continue;
            }
            auto f = h->chans.find(p->rd_chan[(size_t)e]);
            if (f == h->chans.end()) continue;
            (cfg->what == RCF_READ_IQ ? f->second->rd_iq : f->second->rd_fm) = f->second->produced;
        }
    }
    p->running.store(true);
    g->pump = p.get();
    rcf_pump *raw = p.release();
    raw->th = std::thread(pump_main, raw);
    *out = raw;
    return RCF_OK;
}

int rcf_pump_stats(rcf_pump_t *p, rcf_pump_stats_t *st)
{
    if (!p || !st) return RCF_EINVAL;
    std::vector<float> lat;
    {
        std::lock_guard<std::mutex> l(p->st_mu);
        st->blocks_done = p->blocks_done;
        st->blocks_judged = p->judged;
        st->late = p->late;
        st->overruns = p->overruns;
        st->group_blocks = p->group_blocks;
        st->max_batch = p->max_batch;
        st->samples_out = p->samples_out;
        st->host_plan_ms = p->plan_ms;
        st->host_wait_ms = p->wait_ms;
        st->max_plan_ms = p->max_plan_ms;
        st->max_wait_ms = p->max_wait_ms;
        st->max_sleep_overshoot_ms = p->max_idle_gap_ms;
        st->slow_plans = p->slow_plans;
        st->slow_waits = p->slow_waits;
        st->slow_sleeps = p->slow_sleeps;
        st->cpu = p->cfg.cpu;
        st->runq_ms_total = p->runq_total_ms;
        st->slow_wait_ms_total = p->slow_wait_total_ms;
        st->runq_ms_in_slow_waits = p->runq_in_slow_waits_ms;
        st->slow_sleep_ms_total = p->slow_sleep_total_ms;
        st->runq_ms_in_slow_sleeps = p->runq_in_slow_sleeps_ms;
        st->involuntary_switches = p->nivcsw;
        const bool run = p->running.load();
        st->elapsed_s = secs((run ? Clock::now() : p->t_end) - p->t_start);
        st->running = run ? 1 : 0;
        st->rt_priority_granted = p->rt_granted.load();
        st->error = p->error.load();
        lat = p->lat_ms;
    }
    st->latency_ms_p50 = st->latency_ms_p99 = st->latency_ms_max = 0.0;
    if (!lat.empty()) {
        std::sort(lat.begin(), lat.end());
        st->latency_ms_p50 = lat[lat.size() / 2];
        st->latency_ms_p99 = lat[std::min(lat.size() - 1, (size_t)(0.99 * (double)lat.size()))];
        st->latency_ms_max = lat.back();
    }
    if (st->error) set_error("pump stopped: %s", p->err_text);
    return RCF_OK;
}

int64_t rcf_pump_written(rcf_pump_t *p, int entry)
{
    if (!p || entry < 0 || entry >= (int)p->rd_member.size()) return RCF_EINVAL;
    return p->out_written[(size_t)entry].load(std::memory_order_acquire);
}

// one slot's new items from *cursor on -> out; the items the pump has meanwhile queued over (up to two gathers are in flight and
// write [written, queued) of the ring -- the oldest out_cap region a lagging reader may be copying) are dropped from the
// front AFTER the copy, so that what is returned was whole when it was read
static int64_t pump_read_slot(rcf_pump *p, int entry, int64_t *cursor, unsigned char *out, size_t max_items)
{
    const size_t elem = p->item_bytes(entry);
    const int64_t w = p->out_written[(size_t)entry].load(std::memory_order_acquire);
    int64_t avail = w - *cursor;
    if (avail <= 0 || max_items == 0) return 0;
    const int64_t q0 = p->out_queued[(size_t)entry].load(std::memory_order_acquire);
    if (*cursor < q0 - (int64_t)p->out_cap) { *cursor = q0 - (int64_t)p->out_cap; avail = w - *cursor; if (avail <= 0) return 0; }
    int64_t n = std::min<int64_t>(avail, (int64_t)max_items);
    const unsigned char *ring = p->h_out + (size_t)entry * p->out_cap * rcf_pump::kSlotItemBytes;
    const size_t pos = (size_t)((uint64_t)*cursor & (p->out_cap - 1));
    const size_t first = std::min<size_t>((size_t)n, p->out_cap - pos);
    std::memcpy(out, ring + pos * elem, first * elem);
    if ((size_t)n > first) std::memcpy(out + first * elem, ring, ((size_t)n - first) * elem);
    std::atomic_thread_fence(std::memory_order_acquire);
    const int64_t lo = p->out_queued[(size_t)entry].load(std::memory_order_acquire) - (int64_t)p->out_cap;
    if (lo > *cursor) {                               // overwritten while it was copied: what is left starts at lo
        const int64_t drop = std::min<int64_t>(n, lo - *cursor);
        if (drop < n) std::memmove(out, out + (size_t)drop * elem, (size_t)(n - drop) * elem);
        *cursor += drop;
        n -= drop;
    }
    *cursor += n;
    return n;
}

int64_t rcf_pump_read(rcf_pump_t *p, int entry, int64_t *cursor, void *out, size_t max_items)
{
    if (!p || !cursor || !out || entry < 0 || entry >= (int)p->rd_member.size()) { set_error("bad pump read arguments"); return RCF_EINVAL; }
    return pump_read_slot(p, entry, cursor, static_cast<unsigned char *>(out), max_items);
}

int rcf_pump_read_many(rcf_pump_t *p, const int *entries, int64_t *cursors, int n, void *out, size_t cap_each, int64_t *counts)
{
    if (!p || n < 0 || (n && (!entries || !cursors || !out || !counts))) { set_error("bad pump read arguments"); return RCF_EINVAL; }
    for (int i = 0; i < n; ++i) {
        if (entries[i] < 0 || entries[i] >= (int)p->rd_member.size()) { counts[i] = -1; continue; }
        counts[i] = pump_read_slot(p, entries[i], &cursors[i], static_cast<unsigned char *>(out) + (size_t)i * cap_each * rcf_pump::kSlotItemBytes, cap_each);
    }
    return RCF_OK;
}

int rcf_pump_subscribe(rcf_pump_t *p, int member, int chan_id, int what, float gain, int64_t *cursor)
{
    if (!p || (what != RCF_READ_IQ && what != RCF_READ_FM)) { set_error("bad subscription"); return RCF_EINVAL; }
    rcf_group *g = p->g;
    std::lock_guard<std::mutex> gl(g->mu);
    if (member < 0 || member >= (int)g->members.size()) { set_error("no such member %d", member); return RCF_EINVAL; }
    int e = -1;
    for (size_t i = 0; i < p->rd_member.size(); ++i) if (p->rd_member[i] < 0) { e = (int)i; break; }
    if (e < 0) { set_error("all %zu subscription slots of the pump are taken (rcf_pump_config_t.max_read)", p->rd_member.size()); return RCF_ECAP; }
    rcf_t *h = g->members[(size_t)member];
    {
        // (the channel's own reader position is left where it is: a channel nobody has read yet is delivered from its
        // first output on -- what rcf_chan_read_iq would have handed out -- as far as its device ring still holds it)
        std::lock_guard<std::mutex> l(h->mu);
        if (chan_id >= RCF_SRC_PFB_BIN0) {
            // a bin of the member's fused discriminator ring (rcf_pfb_fm_enable): delivered from the bank's next frame on
            const int bin = chan_id - RCF_SRC_PFB_BIN0;
            if (what != RCF_READ_FM || !h->pfb.open || !h->pfb.d_fm || bin >= h->pfb.NB) {
                set_error("member %d has no fused discriminator ring with a bin %d (rcf_pfb_fm_enable; what = RCF_READ_FM)", member, bin);
                return RCF_EINVAL;
            }
            p->bin_rd[(size_t)e] = h->pfb.fm_mode ? h->pfb.produced : h->pfb.fm_until;
        } else if (h->chans.find(chan_id) == h->chans.end()) {
            set_error("no such channel %d on member %d", chan_id, member);
            return RCF_EINVAL;
        }
    }
    p->rd_member[(size_t)e] = member;
    p->rd_chan[(size_t)e] = chan_id;
    p->rd_what[(size_t)e] = what;
    p->rd_gain[(size_t)e] = gain;
    p->chan_of[(size_t)e] = nullptr;
    p->entries_of[(size_t)member].push_back(e);
    p->epoch_of[(size_t)member] = ~0ull;               // resolved at the member's next block
    // the slot's counters go on from where its previous tenant left them: the new stream starts at `queued`, which the
    // reader's cursor is set to (the gathers still in flight for the previous tenant end below it)
    if (cursor) *cursor = p->queued[(size_t)e];
    return e;
}

int rcf_pump_unsubscribe(rcf_pump_t *p, int entry)
{
    if (!p) return RCF_EINVAL;
    rcf_group *g = p->g;
    std::lock_guard<std::mutex> gl(g->mu);
    if (entry < 0 || entry >= (int)p->rd_member.size() || p->rd_member[(size_t)entry] < 0) { set_error("no such subscription %d", entry); return RCF_EINVAL; }
    auto &v = p->entries_of[(size_t)p->rd_member[(size_t)entry]];
    v.erase(std::remove(v.begin(), v.end(), entry), v.end());
    p->rd_member[(size_t)entry] = -1;
    p->rd_chan[(size_t)entry] = -1;
    p->chan_of[(size_t)entry] = nullptr;
    return RCF_OK;
}

int rcf_pump_stop(rcf_pump_t *p)
{
    if (!p) return RCF_EINVAL;
    p->stop.store(true);
    if (p->th.joinable()) p->th.join();
    rcf_group *g = p->g;
    if (getenv("RCF_PUMP_DEBUG"))
        fprintf(stderr, "pump: longest group block by part, ms: reserve %.2f plan %.2f merge %.2f prep-launch %.2f launches %.2f\n",
                g->dbg_ms[0], g->dbg_ms[1], g->dbg_ms[2], g->dbg_ms[3], g->dbg_ms[4]);
    {
        std::lock_guard<std::mutex> gl(g->mu);
        (void)hipSetDevice(g->device);
        (void)hipStreamSynchronize(g->stream);
        g->pump = nullptr;
    }
    delete p;                                          // (~rcf_pump: events, records, rings)
    return RCF_OK;
}

}  // extern "C"
