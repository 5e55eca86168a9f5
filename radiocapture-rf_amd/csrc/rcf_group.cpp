// rcf_group.cpp -- grouped launches over front-ends, and the native real-time pump that drives them.
//
// The reference's receiver holds every configured SDR source in one top block (rc_frontend/receiver.py:67-70,170-204;
// ten sources per host in configs/config_denver_dev_den817.py:25-118).  Here a source is an rcf_t with its own buffers
// and channels; run one by one, a real-time block (20 ms of 20 Msps) is a handful of ~5 us kernels per front-end and one
// MI355X is LAUNCH-bound at a few hundred of them while its memory system idles.  A group plans the blocks of G
// front-ends with the per-front-end planner (rcf_plan.cpp, unchanged arithmetic), concatenates what is concatenable and
// launches ONE kernel per stage:
//   group_prep_kernel          wire format -> cf32 for every member (pinned host memory read in place), history tails
//                              dual-written, the group's launch records host -> device        (ingest.hip)
//   pfb_group_kernel_* / pfb5_group_kernel   the chunks of every member of one bank shape   (pfb.hip, pfb5.hip)
//   fir_small_kernel / fir_bank_kernel       stage-2 channels of all members of one (D, T) class (records concatenated)
//   tap_finalize_group_kernel  the tapped bins of every member                               (fir.hip)
//   disc / fm_fir / rot_fill   records concatenated
//   gather_rings_kernel        the read: new output of any channels of any members -> pinned host memory, one launch
// What is not concatenable (matrix-core banks with their per-handle tap slabs, voice chains, scans, banks that still see
// zero history) follows per member on the same stream, in dependency order.  The bits are those of the members run alone.
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <tuple>

#include "rcf_plan.h"

using namespace rcfx;

struct rcf_pump;

struct rcf_group {
    int device = 0;
    std::vector<rcf_t *> members;
    hipStream_t stream = nullptr;
    ArenaSet arenas;
    hipEvent_t ingest_ev = nullptr;            // the callers' buffers of the last push have been read
    std::vector<void *> d_stage;               // per member: staging for pageable source buffers
    std::vector<size_t> stage_cap;
    // rcf_group_read_many: pinned staging the gather kernel writes (and reads its records from) across PCIe
    unsigned char *h_many = nullptr, *h_many_dev = nullptr;
    size_t many_cap = 0;
    rcf_pump *pump = nullptr;
    std::mutex mu;
    // RCF_PUMP_DEBUG=1: the longest time one group block spent in each part of group_process (printed by rcf_pump_stop)
    double dbg_ms[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

struct GroupItem {
    int m;                 // member index
    size_t n;              // samples
    const void *src;       // host samples (nullptr: already resident -- commit)
    const void *dsrc;      // the same memory as the device sees it, if the caller knows (the pump resolves its rings once)
};

size_t group_sample_bytes(int fmt) { return fmt == RCF_FMT_CF32 ? sizeof(float2) : raw_sample_bytes(fmt); }

struct MemberLocks {       // every member's mutex, in index order (a group call owns all of its members)
    std::vector<rcf_t *> &ms;
    explicit MemberLocks(std::vector<rcf_t *> &m) : ms(m) { for (rcf_t *h : ms) h->mu.lock(); }
    ~MemberLocks() { for (auto it = ms.rbegin(); it != ms.rend(); ++it) (*it)->mu.unlock(); }
};

struct MergedFir {
    FirLaunchDims dims{};
    std::vector<ChanLaunch> recs;
    const ChanLaunch *dev = nullptr;
};

// one block of each listed member.  g->mu and the members' mutexes are held.  wait: return only once the sources have
// been read (the caller may reuse its buffers); the pump passes false -- its rings are not overwritten for many periods.
int group_process(rcf_group *g, const std::vector<GroupItem> &items, int fmt, float scale, float offset, bool wait)
{
    if (items.empty()) return RCF_OK;
    const auto dbg_t0 = std::chrono::steady_clock::now();
    auto dbg_mark = [&](int i, std::chrono::steady_clock::time_point from) {
        const double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - from).count() * 1e3;
        if (ms > g->dbg_ms[i]) g->dbg_ms[i] = ms;
        return std::chrono::steady_clock::now();
    };
    RCF_HIP(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    const size_t bps = group_sample_bytes(fmt);
    const size_t NI = items.size();
    rcf_t *h0 = g->members[0];                  // merged launches are timed on the first member (rcf_timing_* of that handle)

    // ---- 1. room in the group's arena for every member's records and the group's own
    size_t need = 16384 + NI * (3 * sizeof(PrepRec) + sizeof(PfbLaunch) + sizeof(TapFinArgs) + 4 * sizeof(int32_t) + 512);
    for (const GroupItem &it : items) {
        rcf_t *h = g->members[(size_t)it.m];
        flush_lagged(h);                                       // (a member that was fed on its own before: nothing lags inside a group)
        if (h->graveyard.size() > 512) drain_graveyard(h);
        need += arena_need_bound(h);
    }
    if (g->arenas.reserve(need, st) != RCF_OK) return RCF_EHIP;
    if (!g->arenas.mapped) { set_error("group launches need device-mapped pinned memory for their records"); return RCF_ESTATE; }
    auto dbg_t1 = dbg_mark(0, dbg_t0);                          // set device + arena reserve
    const int a = g->arenas.cur;
    const size_t base = g->arenas.fill;
    Arena ga{g->arenas.h[a], g->arenas.d[a], base, g->arenas.cap};

    // ---- 2. where the device reads each source: pinned memory in place, pageable memory through a staging copy
    std::vector<const void *> dsrc(NI, nullptr);
    for (size_t i = 0; i < NI; ++i) {
        const GroupItem &it = items[i];
        if (!it.src) continue;
        if (it.dsrc) { dsrc[i] = it.dsrc; continue; }
        void *dv = nullptr;
        if (hipHostGetDevicePointer(&dv, const_cast<void *>(it.src), 0) == hipSuccess && dv) { dsrc[i] = dv; continue; }
        (void)hipGetLastError();                               // (pageable memory: not an error)
        const size_t bytes = it.n * bps;
        if (g->stage_cap[(size_t)it.m] < bytes) {
            void *nd = nullptr;
            RCF_HIP(hipMalloc(&nd, bytes));
            if (g->d_stage[(size_t)it.m]) { RCF_HIP(hipStreamSynchronize(st)); (void)hipFree(g->d_stage[(size_t)it.m]); }
            g->d_stage[(size_t)it.m] = nd;
            g->stage_cap[(size_t)it.m] = bytes;
        }
        RCF_HIP(hipMemcpyAsync(g->d_stage[(size_t)it.m], it.src, bytes, hipMemcpyHostToDevice, st));
        dsrc[i] = g->d_stage[(size_t)it.m];
    }

    // ---- 3. every member's block planned by the per-front-end planner into the group's arena; all or none
    std::vector<std::unique_ptr<BlockPlan>> plans(NI);
    std::vector<BlockUndo> undo(NI);
    for (size_t i = 0; i < NI; ++i) {
        rcf_t *h = g->members[(size_t)items[i].m];
        plans[i].reset(new BlockPlan);
        plans[i]->defer = true;
        plans[i]->ar = &ga;
        const int rc = plan_block(h, items[i].n, *plans[i], undo[i]);
        if (rc != RCF_OK) {
            for (size_t j = 0; j < i; ++j) undo_block(g->members[(size_t)items[j].m], undo[j]);
            return rc;
        }
    }
    dbg_t1 = dbg_mark(1, dbg_t1);                               // planning
    auto fail_all = [&](int code) {
        for (size_t j = 0; j < NI; ++j) undo_block(g->members[(size_t)items[j].m], undo[j]);
        return code;
    };
    auto oom = [&]() { set_error("launch arena exhausted"); return fail_all(RCF_ENOMEM); };

    // ---- 4. what goes out together
    // filterbanks: members of one shape in steady state share a launch
    struct BankGroup { std::vector<size_t> idx; const PfbLaunch *d_pls = nullptr; GroupMap gm{}; };
    std::map<std::tuple<int, int, int>, BankGroup> banks;
    std::vector<size_t> bank_singles;
    for (size_t i = 0; i < NI; ++i) {
        BlockPlan &bp = *plans[i];
        if (!bp.run_pfb) continue;
        bp.pl.ev_start = bp.pl.ev_stop = nullptr;
        if (pfb_sees_zero_history(bp.pl)) { bank_singles.push_back(i); continue; }
        banks[std::make_tuple(bp.pl.NB, bp.pl.D, pfb_padded_p(bp.pl.NB, bp.pl.D, bp.pl.P))].idx.push_back(i);
    }
    for (auto it = banks.begin(); it != banks.end();) {
        BankGroup &bg = it->second;
        if (bg.idx.size() < 2) { bank_singles.push_back(bg.idx[0]); it = banks.erase(it); continue; }
        const int F = pfb_chunk_frames(std::get<0>(it->first));
        std::vector<PfbLaunch> pls;
        std::vector<int32_t> first;
        pls.reserve(bg.idx.size());
        first.reserve(bg.idx.size() + 1);
        int32_t total = 0, uniform = -1;
        for (size_t i : bg.idx) {
            const PfbLaunch &pl = plans[i]->pl;
            const int32_t nwg = (pl.n_frames + F - 1) / F;
            uniform = uniform < 0 ? nwg : (uniform == nwg ? uniform : 0);
            first.push_back(total);
            total += nwg;
            pls.push_back(pl);
        }
        first.push_back(total);
        if (!ga.put(pls, &bg.d_pls) || !ga.put(first, &bg.gm.wg_first)) return oom();
        bg.gm.n_fe = (int32_t)bg.idx.size();
        bg.gm.total_wg = total;
        bg.gm.uniform_nwg = uniform > 0 ? uniform : 0;
        ++it;
    }
    std::sort(bank_singles.begin(), bank_singles.end());
    // taps
    std::vector<TapFinArgs> tap_args;
    int tap_max_taps = 0, tap_max_rows = 0;
    for (size_t i = 0; i < NI; ++i) {
        const BlockPlan &bp = *plans[i];
        if (!bp.run_pfb || bp.pl.n_taps <= 0) continue;
        const PfbLaunch &pl = bp.pl;
        tap_args.push_back(TapFinArgs{bp.d_tap_list, pl.tap_mat, bp.d_group_bin0, pl.bins_ring, pl.n_lo - pl.n_abs0, pl.n_taps,
                                      pl.tap_pitch, pl.n_frames, pl.tap_first, pl.NB, 0});
        tap_max_taps = std::max(tap_max_taps, (int)pl.n_taps);
        tap_max_rows = std::max(tap_max_rows, (int)pl.n_frames);
    }
    const TapFinArgs *d_tap_args = nullptr;
    if (!tap_args.empty() && !ga.put(tap_args, &d_tap_args)) return oom();
    // FIR jobs whose records are self-contained: one launch per depth and (D, T) class
    int max_depth = 0;
    for (auto &bp : plans) max_depth = std::max(max_depth, (int)bp->fir_by_depth.size() - 1);
    std::vector<std::map<std::tuple<int, int, int, int>, MergedFir>> merged((size_t)max_depth + 1);
    for (auto &bp : plans)
        for (size_t d = 0; d < bp->fir_by_depth.size(); ++d)
            for (FirJob &j : bp->fir_by_depth[d]) {
                if (j.host.empty()) continue;
                MergedFir &mf = merged[d][std::make_tuple(j.dims.D, j.dims.T, j.dims.small, j.dims.KT)];
                if (mf.recs.empty()) { mf.dims = j.dims; mf.dims.max_n_k = 0; mf.dims.atan_tab = h0->d_atan; }
                mf.dims.max_n_k = std::max(mf.dims.max_n_k, j.dims.max_n_k);
                mf.recs.insert(mf.recs.end(), j.host.begin(), j.host.end());
            }
    for (auto &lvl : merged)
        for (auto &kv : lvl) {
            kv.second.dims.n_chans = (int)kv.second.recs.size();
            if (!ga.put(kv.second.recs, &kv.second.dev)) return oom();
        }
    // discriminators, symbol filters, exact-rotator fills
    std::vector<DiscLaunch> discs;
    std::vector<FmFirLaunch> symf;
    std::vector<RotFill> rots;
    int disc_max_n = 0, symf_max_n = 0;
    for (auto &bp : plans) {
        for (DiscJob &dj : bp->disc_jobs) {
            discs.insert(discs.end(), dj.host.begin(), dj.host.end());
            disc_max_n = std::max(disc_max_n, dj.max_n);
        }
        symf.insert(symf.end(), bp->symf.begin(), bp->symf.end());
        symf_max_n = std::max(symf_max_n, bp->symf_max_n);
        rots.insert(rots.end(), bp->rot_fills.begin(), bp->rot_fills.end());
    }
    const DiscLaunch *d_discs = nullptr;
    const FmFirLaunch *d_symf = nullptr;
    const RotFill *d_rots = nullptr;
    if ((!discs.empty() && !ga.put(discs, &d_discs)) || (!symf.empty() && !ga.put(symf, &d_symf)) ||
        (!rots.empty() && !ga.put(rots, &d_rots)))
        return oom();

    // ---- 5. the prep launch's records, last: one of them uploads everything put so far
    std::vector<PrepRec> prep;
    prep.reserve(2 * NI + 1);
    uint32_t prep_max = 0;
    for (size_t i = 0; i < NI; ++i) {
        rcf_t *h = g->members[(size_t)items[i].m];
        const size_t n = items[i].n, H = h->hist_cap;
        float2 *curb = h->d_buf[h->cur], *oth = h->d_buf[h->cur ^ 1];
        PrepRec r{};
        if (dsrc[i]) {
            // block sample i sits at buffer index H + i; the next block's history is buffer [n, n + H): sample i lands at
            // other[H + i - n] once that is >= 0
            r.src = dsrc[i];
            r.dst = curb + H;
            r.n = (uint32_t)n;
            r.hist_from = n >= H ? (uint32_t)(n - H) : 0u;
            r.hist_dst = n >= H ? oth : oth + (H - n);
            r.fmt = fmt;
            r.scale = scale;
            r.offset = offset;
            const size_t item = fmt == RCF_FMT_CF32 ? 4 : bps / 2;      // bytes per raw value
            r.aligned = ((uintptr_t)r.src % (4 * item)) == 0 ? 1 : 0;
            r.dst_aligned = ((uintptr_t)r.dst % 16) == 0 ? 1 : 0;
            prep.push_back(r);
            prep_max = std::max(prep_max, r.n);
            if (n < H) {                                       // the part of the history that is older than this block
                PrepRec c{};
                c.src = curb + n;
                c.dst = oth;
                c.n = (uint32_t)(H - n);
                c.fmt = -1;
                prep.push_back(c);
                prep_max = std::max(prep_max, c.n);
            }
        } else {                                               // resident data: only the history moves
            r.src = curb + n;
            r.dst = oth;
            r.n = (uint32_t)H;
            r.fmt = -1;
            prep.push_back(r);
            prep_max = std::max(prep_max, r.n);
        }
        plans[i]->history_done = true;
    }
    {
        const size_t from = base & ~size_t(63);
        const size_t bytes = ga.used > base ? ((ga.used + 63) & ~size_t(63)) - from : 0;
        if (bytes) {
            PrepRec c{};
            c.src = g->arenas.h_dev[a] + from;
            c.dst = reinterpret_cast<float2 *>(ga.d + from);
            c.n = (uint32_t)(bytes / 8);
            c.fmt = -1;
            prep.push_back(c);
            prep_max = std::max(prep_max, c.n);
        }
    }
    // (at most kPrepMaxRecs records per launch: more go out as further launches)
    std::vector<uint32_t> prep_tiles;
    for (size_t at = 0; at < prep.size(); at += kPrepMaxRecs)
        prep_tiles.push_back(fill_prep_tiles(prep.data() + at, (int)std::min<size_t>(kPrepMaxRecs, prep.size() - at)));
    const PrepRec *d_prep = nullptr;
    if (!ga.put(prep, &d_prep)) return oom();
    // (the kernel reads ITS records where the host wrote them: the pinned arena as the device sees it)
    const PrepRec *prep_mapped = reinterpret_cast<const PrepRec *>(
        g->arenas.h_dev[a] + (reinterpret_cast<const unsigned char *>(d_prep) - ga.d));
    g->arenas.fill = (ga.used + 63) & ~size_t(63);

    dbg_t1 = dbg_mark(2, dbg_t1);                               // merging + records
    // ---- 6. launches, in dependency order.  From here on a failure leaves queued work behind: no roll-back.
    for (size_t at = 0, li = 0; at < prep.size(); at += kPrepMaxRecs, ++li)
        launch_group_prep(prep_mapped + at, (int)std::min<size_t>(kPrepMaxRecs, prep.size() - at), prep_tiles[li], st);
    dbg_t1 = dbg_mark(3, dbg_t1);                               // the prep launch
    if (wait) RCF_HIP(hipEventRecord(g->ingest_ev, st));
    if (d_rots) launch_rot_fill(d_rots, (int)rots.size(), h0->ring_mask, st);
    auto launch_depth = [&](size_t d, int timing_class_default) {
        for (size_t i = 0; i < NI; ++i) {
            rcf_t *h = g->members[(size_t)items[i].m];
            BlockPlan &bp = *plans[i];
            if (d >= bp.fir_by_depth.size()) continue;
            for (FirJob &j : bp.fir_by_depth[d]) {
                if (!j.host.empty()) continue;                 // merged below
                if (j.repack) {
                    launch_fir_pack(j.dev, j.dims.n_chans, j.dims.T, const_cast<float *>(j.dims.bank), j.dirty, st);
                    if (j.bc) j.bc->key = std::move(j.key);
                }
                Timed t(h, d == 0 ? (j.dims.mfma ? RCF_T_FIR_MFMA : RCF_T_FIR) : timing_class_default);
                launch_fir_bank(j.dev, j.dims, st);
            }
        }
        if (d < merged.size())
            for (auto &kv : merged[d]) {
                Timed t(h0, d == 0 ? RCF_T_FIR : timing_class_default);
                launch_fir_bank(kv.second.dev, kv.second.dims, st);
            }
    };
    launch_depth(0, RCF_T_FIR);
    for (auto &kv : banks) {
        BankGroup &bg = kv.second;
        bool done;
        { Timed t(h0, RCF_T_PFB); done = launch_pfb_group(plans[bg.idx[0]]->pl, bg.d_pls, bg.gm, st); }
        if (!done)                                             // a shape without a grouped kernel: one by one
            for (size_t i : bg.idx) { Timed t(g->members[(size_t)items[i].m], RCF_T_PFB); launch_pfb(plans[i]->pl, st); }
    }
    for (size_t i : bank_singles) { Timed t(g->members[(size_t)items[i].m], RCF_T_PFB); launch_pfb(plans[i]->pl, st); }
    if (d_tap_args) {
        Timed t(h0, RCF_T_TAPS);
        launch_tap_finalize_group(d_tap_args, (int)tap_args.size(), tap_max_taps, tap_max_rows, h0->ring_mask, h0->d_atan, st);
    }
    for (size_t d = 1; d <= (size_t)max_depth; ++d) launch_depth(d, RCF_T_FIR_DERIVED);
    if (d_discs) { Timed t(h0, RCF_T_DISC); launch_discriminator(d_discs, (int)discs.size(), disc_max_n, h0->ring_mask, h0->d_atan, st); }
    if (d_symf) { Timed t(h0, RCF_T_DISC); launch_fm_fir(d_symf, (int)symf.size(), symf_max_n, h0->ring_mask, st); }
    for (size_t i = 0; i < NI; ++i) {
        rcf_t *h = g->members[(size_t)items[i].m];
        BlockPlan &bp = *plans[i];
        if (bp.d_audf) {
            Timed t(h, RCF_T_AUDIO);
            launch_audio(bp.d_audf, (int)bp.audf.size(), bp.audf_max_n, bp.audf_num, bp.audf_den, h->ring_mask, h->d_atan, st);
        }
        int rc = run_scan(h, bp);
        if (rc != RCF_OK) return rc;
        h->buf_dirty[h->cur] = true;                           // (rcf_push_iq on this member later orders its copy behind these reads)
        h->cur ^= 1;
        h->total_in = bp.S1;
    }
    RCF_HIP(hipGetLastError());
    dbg_t1 = dbg_mark(4, dbg_t1);                               // the other launches
    if (wait) (void)hipEventSynchronize(g->ingest_ev);
    return RCF_OK;
}

// ---------------------------------------------------------------- batched read over members
struct ReadItem { rcf_t *h; Chan *c; int64_t *cur; const void *ring; int64_t n; size_t pos; };

// what entry (member, chan) has to give; marks duplicates through the handle's many_stamp (stamps[] holds one fresh stamp
// per member for this call)
int resolve_read(rcf_group *g, int what, int member, int chan_id, size_t cap_each, const std::vector<uint64_t> &stamps,
                 ReadItem &it, int64_t &count)
{
    it = ReadItem{nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (member < 0 || member >= (int)g->members.size()) { count = RCF_EINVAL; return 0; }
    rcf_t *h = g->members[(size_t)member];
    auto f = h->chans.find(chan_id);
    if (f == h->chans.end()) { count = RCF_ENOCHAN; return 0; }
    Chan *c = f->second.get();
    if (c->many_stamp == stamps[(size_t)member]) { count = RCF_EINVAL; return 0; }     // listed twice
    c->many_stamp = stamps[(size_t)member];
    it.h = h;
    it.c = c;
    it.cur = what == RCF_READ_IQ ? &c->rd_iq : &c->rd_fm;
    it.ring = what == RCF_READ_IQ ? (const void *)c->d_iq : (const void *)c->d_fm;
    int64_t avail = c->produced - *it.cur;
    if (avail > 0 && (size_t)avail > h->out_cap) {              // reader lagged: oldest samples are gone
        *it.cur = c->produced - (int64_t)h->out_cap;
        avail = (int64_t)h->out_cap;
    }
    it.n = avail <= 0 ? 0 : std::min<int64_t>(avail, (int64_t)cap_each);
    it.pos = (size_t)((uint64_t)*it.cur & h->ring_mask);
    count = it.n;
    return 1;
}

int ensure_many(rcf_group *g, size_t need)
{
    if (need <= g->many_cap) return RCF_OK;
    if (g->h_many) { RCF_HIP(hipStreamSynchronize(g->stream)); (void)hipHostFree(g->h_many); g->h_many = nullptr; g->many_cap = 0; }
    size_t cap = 1 << 16;
    while (cap < need) cap <<= 1;
    void *p = nullptr, *dv = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess || hipHostGetDevicePointer(&dv, p, 0) != hipSuccess) {
        if (p) (void)hipHostFree(p);
        set_error("pinned staging of %zu bytes for the batched read failed", cap);
        return RCF_ENOMEM;
    }
    g->h_many = static_cast<unsigned char *>(p);
    g->h_many_dev = static_cast<unsigned char *>(dv);
    g->many_cap = cap;
    return RCF_OK;
}

}  // namespace

// =================================================================== the pump
struct rcf_pump {
    rcf_group *g = nullptr;
    rcf_pump_config_t cfg{};
    std::vector<const unsigned char *> rings, rings_dev;     // host address / the same memory as the device sees it (or nullptr)
    std::vector<size_t> ring_blocks;
    std::vector<double> phase;
    std::vector<const volatile uint64_t *> written;
    std::vector<int> rd_member, rd_chan;
    std::vector<std::vector<int>> entries_of;      // member -> indices into rd_*
    size_t out_cap = 0;                            // host ring length per entry (items, power of two)
    size_t elem = 4;                               // bytes per item
    unsigned char *h_out = nullptr, *h_out_dev = nullptr;   // n_read rings of out_cap items
    std::unique_ptr<std::atomic<int64_t>[]> out_written;    // items delivered per entry
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
    std::chrono::steady_clock::time_point t_start, t_end;
    char err_text[256] = "";
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
    const double period = (double)cfg.block_samples / cfg.samp_rate;
    const size_t blk_bytes = cfg.block_samples * group_sample_bytes(cfg.fmt);
    const Clock::time_point t0 = Clock::now() + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(cfg.start_delay_s));
    { std::lock_guard<std::mutex> l(p->st_mu); p->t_start = t0; }
    std::vector<int64_t> next_k(G, 0);             // next block of each member
    std::vector<double> seen_at(G, -1.0);          // counter-fed members: when the pump first saw block next_k complete
    InFlight slots[2];
    int head = 0, in_flight = 0;                   // slots[head] is the oldest busy one
    std::vector<GroupItem> items;
    std::vector<int64_t> queued(p->rd_member.size(), 0);   // items ever queued for the host ring of each subscribed channel
    std::vector<Chan *> chan_of(p->rd_member.size(), nullptr);      // subscribed channels, resolved once per channel-set epoch
    std::vector<uint64_t> epoch_of(G, ~0ull);
    const uint32_t ew = (uint32_t)(p->elem / 4);

    auto complete_oldest = [&](bool block) -> bool {
        InFlight &s = slots[head];
        if (!s.busy) return false;
        if (!block && hipEventQuery(p->slot_ev[head]) != hipSuccess) { (void)hipGetLastError(); return false; }
        const Clock::time_point w0 = Clock::now();
        if (hipEventSynchronize(p->slot_ev[head]) != hipSuccess) { set_error("pump: event wait failed"); pump_fail(p, RCF_EHIP); return false; }
        const Clock::time_point now = Clock::now();
        const double t_done = secs(now - t0);
        int64_t items_out = 0;
        for (auto &d : s.delivered) { p->out_written[(size_t)d.first].fetch_add(d.second, std::memory_order_release); items_out += d.second; }
        {
            std::lock_guard<std::mutex> l(p->st_mu);
            p->wait_ms += secs(now - w0) * 1e3;
            if (p->warm_done) {
                p->max_wait_ms = std::max(p->max_wait_ms, secs(now - w0) * 1e3);
                if (secs(now - w0) > 5e-3) ++p->slow_waits;
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
                                auto f = h->chans.find(p->rd_chan[(size_t)e]);
                                chan_of[(size_t)e] = f == h->chans.end() ? nullptr : f->second.get();
                            }
                            epoch_of[(size_t)it.m] = h->chans_epoch;
                        }
                        for (int e : p->entries_of[(size_t)it.m]) {
                            Chan *c = chan_of[(size_t)e];
                            if (!c) continue;                                      // closed under the pump: starves
                            int64_t *cur = cfg.what == RCF_READ_IQ ? &c->rd_iq : &c->rd_fm;
                            int64_t avail = c->produced - *cur;
                            if (avail <= 0) continue;
                            if ((size_t)avail > h->out_cap) { *cur = c->produced - (int64_t)h->out_cap; avail = (int64_t)h->out_cap; }
                            if ((size_t)avail > p->out_cap) { *cur += avail - (int64_t)p->out_cap; avail = (int64_t)p->out_cap; }
                            const uint64_t dst_pos = (uint64_t)queued[(size_t)e] & (p->out_cap - 1);
                            queued[(size_t)e] += avail;
                            recs[n_recs++] = GatherRec{static_cast<const uint32_t *>(cfg.what == RCF_READ_IQ ? (const void *)c->d_iq : (const void *)c->d_fm),
                                                       (uint32_t)(((uint64_t)*cur & h->ring_mask) * ew), (uint32_t)avail * ew,
                                                       (uint32_t)(h->out_cap * ew - 1), (uint32_t)((size_t)e * p->out_cap * ew),
                                                       (uint32_t)(dst_pos * ew), (uint32_t)(p->out_cap * ew - 1), cfg.gain,
                                                       (cfg.what == RCF_READ_FM && cfg.gain != 1.0f) ? 1u : 0u};
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
                if (!p->warm_done && *std::min_element(s.kidx.begin(), s.kidx.end()) >= cfg.warm_blocks) p->warm_done = true;
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
            const Clock::time_point s0 = Clock::now();
            if (cfg.spin_us > 0 && want <= cfg.spin_us * 1e-6) {
                while (secs(Clock::now() - s0) < want && !p->stop.load(std::memory_order_relaxed)) __builtin_ia32_pause();
            } else {
                std::this_thread::sleep_for(std::chrono::duration<double>(cfg.spin_us > 0 ? want - cfg.spin_us * 0.5e-6 : want));
                while (secs(Clock::now() - s0) < want && cfg.spin_us > 0) __builtin_ia32_pause();
            }
            const double over = (secs(Clock::now() - s0) - want) * 1e3;      // how much later than asked the thread came back
            if (p->warm_done && over > 2.0) { std::lock_guard<std::mutex> l(p->st_mu); ++p->slow_sleeps; p->max_idle_gap_ms = std::max(p->max_idle_gap_ms, over); }
        }
    }
    while (in_flight > 0 && !p->error.load()) (void)complete_oldest(true);
    (void)hipStreamSynchronize(g->stream);
    { std::lock_guard<std::mutex> l(p->st_mu); p->t_end = Clock::now(); }
    p->running.store(false);
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int rcf_group_open(rcf_t *const *handles, int n, rcf_group_t **out)
{
    if (!handles || n < 1 || !out) { set_error("bad group arguments"); return RCF_EINVAL; }
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) { set_error("group member %d is NULL", i); return RCF_EINVAL; }
        for (int j = 0; j < i; ++j)
            if (handles[j] == handles[i]) { set_error("group member %d listed twice", i); return RCF_EINVAL; }
        if (handles[i]->device != handles[0]->device || handles[i]->out_cap != handles[0]->out_cap) {
            set_error("group members must share the device and the output capacity (member %d differs)", i);
            return RCF_EINVAL;
        }
        if (handles[i]->group) { set_error("group member %d already belongs to a group", i); return RCF_ESTATE; }
    }
    std::unique_ptr<rcf_group> g(new rcf_group);
    g->device = handles[0]->device;
    RCF_HIP(hipSetDevice(g->device));
    RCF_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    RCF_HIP(hipEventCreateWithFlags(&g->ingest_ev, hipEventDisableTiming));
    g->members.assign(handles, handles + n);
    g->d_stage.assign((size_t)n, nullptr);
    g->stage_cap.assign((size_t)n, 0);
    for (rcf_t *h : g->members) {
        std::lock_guard<std::mutex> l(h->mu);
        flush_lagged(h);
        RCF_HIP(hipStreamSynchronize(h->stream));          // whatever it queued on its own stream comes first
        time_collect(h);
        h->own_stream = h->stream;
        h->stream = g->stream;
        h->group = g.get();
    }
    *out = g.release();
    return RCF_OK;
}

int rcf_group_close(rcf_group_t *g)
{
    if (!g) return RCF_EINVAL;
    if (g->pump) { set_error("stop the group's pump first"); return RCF_ESTATE; }
    (void)hipSetDevice(g->device);
    {
        std::lock_guard<std::mutex> gl(g->mu);
        MemberLocks ml(g->members);
        (void)hipStreamSynchronize(g->stream);
        for (rcf_t *h : g->members) {
            time_collect(h);
            free_graveyard_idle(h);
            h->stream = h->own_stream;
            h->own_stream = nullptr;
            h->group = nullptr;
        }
    }
    for (void *p : g->d_stage) if (p) (void)hipFree(p);
    if (g->h_many) (void)hipHostFree(g->h_many);
    g->arenas.destroy();
    if (g->ingest_ev) (void)hipEventDestroy(g->ingest_ev);
    (void)hipStreamDestroy(g->stream);
    delete g;
    return RCF_OK;
}

int rcf_group_size(rcf_group_t *g) { return g ? (int)g->members.size() : RCF_EINVAL; }

static int group_call(rcf_group_t *g, const void *const *blocks, const size_t *n_samples, int fmt, float scale, float offset,
                      bool commit)
{
    if (!g || !n_samples || (!commit && !blocks)) { set_error("bad group push arguments"); return RCF_EINVAL; }
    if (!commit && group_sample_bytes(fmt) == 0) { set_error("unknown sample format %d", fmt); return RCF_EINVAL; }
    std::vector<GroupItem> items;
    for (size_t i = 0; i < g->members.size(); ++i) {
        const size_t n = n_samples[i];
        if (n == 0 || (!commit && !blocks[i])) continue;
        if (n > g->members[i]->block_cap) {
            set_error("member %zu: %zu samples exceed its block capacity %zu", i, n, g->members[i]->block_cap);
            return RCF_ECAP;
        }
        items.push_back(GroupItem{(int)i, n, commit ? nullptr : blocks[i], nullptr});
    }
    std::lock_guard<std::mutex> gl(g->mu);
    if (g->pump) { set_error("the group is fed by its pump"); return RCF_ESTATE; }
    MemberLocks ml(g->members);
    return group_process(g, items, fmt, scale, offset, !commit);
}

int rcf_group_push(rcf_group_t *g, const void *const *blocks, const size_t *n_samples, int fmt, float scale, float offset)
{
    return group_call(g, blocks, n_samples, fmt, scale, offset, false);
}

int rcf_group_commit(rcf_group_t *g, const size_t *n_samples)
{
    return group_call(g, nullptr, n_samples, RCF_FMT_CF32, 1.0f, 0.0f, true);
}

int rcf_group_sync(rcf_group_t *g)
{
    if (!g) return RCF_EINVAL;
    std::lock_guard<std::mutex> gl(g->mu);
    MemberLocks ml(g->members);
    RCF_HIP(hipSetDevice(g->device));
    for (rcf_t *h : g->members) flush_lagged(h);
    RCF_HIP(hipStreamSynchronize(g->stream));
    for (rcf_t *h : g->members) free_graveyard_idle(h);
    return RCF_OK;
}

int rcf_group_read_many(rcf_group_t *g, int what, const int *members, const int *chan_ids, int n, float gain, void *out,
                        size_t cap_each, int64_t *counts)
{
    if (!g || !members || !chan_ids || !out || !counts || n < 0 || (what != RCF_READ_IQ && what != RCF_READ_FM)) {
        set_error("bad batched read arguments");
        return RCF_EINVAL;
    }
    std::lock_guard<std::mutex> gl(g->mu);
    MemberLocks ml(g->members);
    RCF_HIP(hipSetDevice(g->device));
    for (rcf_t *h : g->members) flush_lagged(h);            // (a member that was fed on its own meanwhile)
    const size_t elem = what == RCF_READ_IQ ? sizeof(float2) : sizeof(float);
    const uint32_t ew = (uint32_t)(elem / 4);
    std::vector<uint64_t> stamps(g->members.size());
    for (size_t m = 0; m < g->members.size(); ++m) stamps[m] = ++g->members[m]->many_stamp;
    std::vector<ReadItem> items((size_t)n);
    size_t total = 0;
    uint32_t max_w = 0;
    for (int i = 0; i < n; ++i) {
        resolve_read(g, what, members[i], chan_ids[i], cap_each, stamps, items[(size_t)i], counts[i]);
        total += (size_t)items[(size_t)i].n;
        max_w = std::max<uint32_t>(max_w, (uint32_t)items[(size_t)i].n * ew);
    }
    if (total == 0) return RCF_OK;
    if ((uint64_t)total * ew > 0xffffffffull) { set_error("batched read of %zu items exceeds the 32-bit word range", total); return RCF_ECAP; }
    const size_t rec_bytes = ((size_t)n * sizeof(GatherRec) + 255) & ~(size_t)255;
    int rc = ensure_many(g, rec_bytes + total * elem);
    if (rc != RCF_OK) return rc;
    GatherRec *recs = reinterpret_cast<GatherRec *>(g->h_many);
    uint32_t at_w = 0;
    int n_recs = 0;
    for (int i = 0; i < n; ++i) {
        const ReadItem &it = items[(size_t)i];
        if (it.n <= 0) continue;
        recs[n_recs++] = GatherRec{static_cast<const uint32_t *>(it.ring), (uint32_t)(it.pos * ew), (uint32_t)it.n * ew,
                                   (uint32_t)(it.h->out_cap * ew - 1), at_w, 0u, ~0u, 1.0f, 0u};
        at_w += (uint32_t)it.n * ew;
    }
    launch_gather_rings(reinterpret_cast<const GatherRec *>(g->h_many_dev), n_recs,
                        reinterpret_cast<uint32_t *>(g->h_many_dev + rec_bytes), max_w, g->stream);
    RCF_HIP(hipStreamSynchronize(g->stream));
    for (rcf_t *h : g->members) free_graveyard_idle(h);
    const unsigned char *src = g->h_many + rec_bytes;
    for (int i = 0; i < n; ++i) {
        const ReadItem &it = items[(size_t)i];
        if (it.n <= 0) continue;
        unsigned char *o = static_cast<unsigned char *>(out) + (size_t)i * cap_each * elem;
        std::memcpy(o, src, (size_t)it.n * elem);
        src += (size_t)it.n * elem;
        *it.cur += it.n;
        if (what == RCF_READ_FM) {
            float *f = reinterpret_cast<float *>(o);
            for (int64_t k = 0; k < it.n; ++k) f[k] = gain * f[k];
        }
    }
    return RCF_OK;
}

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
    p->entries_of.resize(G);
    for (int e = 0; e < cfg->n_read; ++e) {
        if (cfg->read_members[e] < 0 || cfg->read_members[e] >= (int)G) { set_error("subscribed channel %d: no such member", e); return RCF_EINVAL; }
        p->rd_member.push_back(cfg->read_members[e]);
        p->rd_chan.push_back(cfg->read_chans[e]);
        p->entries_of[(size_t)cfg->read_members[e]].push_back(e);
    }
    // the configuration's arrays belong to the caller: from here on the pump's own copies are used
    p->cfg.rings = nullptr; p->cfg.ring_blocks = nullptr; p->cfg.phase_s = nullptr; p->cfg.written = nullptr;
    p->cfg.read_members = nullptr; p->cfg.read_chans = nullptr;
    p->elem = cfg->what == RCF_READ_IQ ? sizeof(float2) : sizeof(float);
    p->out_cap = pow2_at_least(cfg->out_ring_samples ? cfg->out_ring_samples : 4096);
    const size_t out_bytes = std::max<size_t>(64, (size_t)cfg->n_read * p->out_cap * p->elem);
    if ((uint64_t)out_bytes / 4 > 0xffffffffull) { set_error("host rings of %zu bytes exceed the 32-bit word range", out_bytes); return RCF_ECAP; }
    void *hp = nullptr, *dv = nullptr;
    if (hipHostMalloc(&hp, out_bytes, hipHostMallocDefault) != hipSuccess || hipHostGetDevicePointer(&dv, hp, 0) != hipSuccess) {
        if (hp) (void)hipHostFree(hp);
        set_error("pinned host rings of %zu bytes failed", out_bytes);
        return RCF_ENOMEM;
    }
    p->h_out = static_cast<unsigned char *>(hp);
    p->h_out_dev = static_cast<unsigned char *>(dv);
    p->out_written.reset(new std::atomic<int64_t>[(size_t)std::max(1, cfg->n_read)]);
    for (int e = 0; e < std::max(1, cfg->n_read); ++e) p->out_written[(size_t)e].store(0);
    p->recs_cap = (size_t)std::max(1, cfg->n_read);
    hp = dv = nullptr;
    if (hipHostMalloc(&hp, 2 * p->recs_cap * sizeof(GatherRec), hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer(&dv, hp, 0) != hipSuccess) {
        if (hp) (void)hipHostFree(hp);
        (void)hipHostFree(p->h_out);
        set_error("pinned gather records failed");
        return RCF_ENOMEM;
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

int64_t rcf_pump_read(rcf_pump_t *p, int entry, int64_t *cursor, void *out, size_t max_items)
{
    if (!p || !cursor || !out || entry < 0 || entry >= (int)p->rd_member.size()) { set_error("bad pump read arguments"); return RCF_EINVAL; }
    const int64_t w = p->out_written[(size_t)entry].load(std::memory_order_acquire);
    int64_t avail = w - *cursor;
    if (avail <= 0 || max_items == 0) return 0;
    if ((size_t)avail > p->out_cap) { *cursor = w - (int64_t)p->out_cap; avail = (int64_t)p->out_cap; }
    const int64_t n = std::min<int64_t>(avail, (int64_t)max_items);
    const unsigned char *ring = p->h_out + (size_t)entry * p->out_cap * p->elem;
    const size_t pos = (size_t)((uint64_t)*cursor & (p->out_cap - 1));
    const size_t first = std::min<size_t>((size_t)n, p->out_cap - pos);
    std::memcpy(out, ring + pos * p->elem, first * p->elem);
    if ((size_t)n > first) std::memcpy(static_cast<unsigned char *>(out) + first * p->elem, ring, ((size_t)n - first) * p->elem);
    *cursor += n;
    return n;
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
    for (int i = 0; i < 2; ++i) if (p->slot_ev[i]) (void)hipEventDestroy(p->slot_ev[i]);
    if (p->h_recs) (void)hipHostFree(p->h_recs);
    if (p->h_out) (void)hipHostFree(p->h_out);
    delete p;
    return RCF_OK;
}

}  // extern "C"
