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
//   tap_finalize_group_kernel  the tapped bins of every member                               (tapfin.hip)
//   disc / fm_fir / rot_fill   records concatenated
//   gather_rings_kernel        the read: new output of any channels of any members -> pinned host memory, one launch
// What is not concatenable (matrix-core banks with their per-handle tap slabs, voice chains, scans, banks that still see
// zero history) follows per member on the same stream, in dependency order.  The bits are those of the members run alone.
#include <tuple>

#include "rcf_group.h"

using namespace rcfx;


namespace {

struct MergedFir {
    FirLaunchDims dims{};
    std::vector<ChanLaunch> recs;
    const ChanLaunch *dev = nullptr;
};

}  // namespace

// (declared in rcf_group.h: the pump calls it)
int rcfx::group_process(rcf_group *g, const std::vector<GroupItem> &items, int fmt, float scale, float offset, bool wait)
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
    auto tp = dbg_t0;
    RCF_PROF(8, "group: arena reserve", tp);
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
    RCF_PROF(9, "group: planning (all)", tp);
    auto fail_all = [&](int code) {
        for (size_t j = 0; j < NI; ++j) undo_block(g->members[(size_t)items[j].m], undo[j]);
        return code;
    };
    auto oom = [&]() { set_error("launch arena exhausted"); return fail_all(RCF_ENOMEM); };

    // ---- 4. what goes out together
    // filterbanks: members of one shape in steady state share a launch
    struct BankGroup { std::vector<size_t> idx; const PfbLaunch *d_pls = nullptr; GroupMap gm{}; };
    std::map<std::tuple<int, int, int, int>, BankGroup> banks;       // (bins, decimation, taps per branch, fused-discriminator mode)
    std::vector<size_t> bank_singles;
    for (size_t i = 0; i < NI; ++i) {
        BlockPlan &bp = *plans[i];
        if (!bp.run_pfb) continue;
        bp.pl.ev_start = bp.pl.ev_stop = nullptr;
        // (a fused-discriminator bank joins a grouped launch in its look-back form only, and not while the chunk before
        // its first frame -- which its first workgroup recomputes -- reaches before the stream's start)
        if (pfb_sees_zero_history(bp.pl) || (bp.pl.fm_ring && (!bp.pl.fm_edge || pfb5_fm_sees_zero_history(bp.pl)))) {
            bank_singles.push_back(i);
            continue;
        }
        banks[std::make_tuple(bp.pl.NB, bp.pl.D, pfb_padded_p(bp.pl.NB, bp.pl.D, bp.pl.P), bp.pl.fm_ring ? bp.pl.fm_mode : 0)].idx.push_back(i);
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
    RCF_PROF(10, "group: merging + records", tp);
    // ---- 6. launches, in dependency order.  From here on a failure leaves queued work behind: no roll-back.
    for (size_t at = 0, li = 0; at < prep.size(); at += kPrepMaxRecs, ++li)
        launch_group_prep(prep_mapped + at, (int)std::min<size_t>(kPrepMaxRecs, prep.size() - at), prep_tiles[li], st);
    dbg_t1 = dbg_mark(3, dbg_t1);                               // the prep launch
    RCF_PROF(11, "group: prep launch", tp);
    // (past this point every listed member's channel counters have been advanced by the planning and the prep kernel has
    // written its buffers: a failure no longer returns at once -- the bookkeeping of every member is finished, the members
    // are marked faulted, and the error is returned at the end)
    int rc_late = RCF_OK;
    if (wait && hipEventRecord(g->ingest_ev, st) != hipSuccess) { set_error("group: event record failed"); rc_late = RCF_EHIP; }
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
        const int rc = run_scan(h, bp);
        if (rc != RCF_OK && rc_late == RCF_OK) rc_late = rc;
        h->buf_dirty[h->cur] = true;                           // (rcf_push_iq on this member later orders its copy behind these reads)
        h->cur ^= 1;
        h->total_in = bp.S1;
    }
    if (rc_late == RCF_OK) {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("group launch failed: %s", hipGetErrorString(e)); rc_late = RCF_EHIP; }
    }
    if (rc_late != RCF_OK) {
        for (size_t i = 0; i < NI; ++i) {
            rcf_t *h = g->members[(size_t)items[i].m];
            h->fault = rc_late;
            std::snprintf(h->fault_text, sizeof h->fault_text, "group block failed: %s", rcf_last_error());
        }
        return rc_late;
    }
    dbg_t1 = dbg_mark(4, dbg_t1);                               // the other launches
    RCF_PROF(12, "group: other launches", tp);
    if (wait) (void)hipEventSynchronize(g->ingest_ev);
    return RCF_OK;
}

namespace {

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
    if (what == RCF_READ_IQ && c->fm_only) { count = RCF_ESTATE; return 0; }           // discriminator only
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
    g->members.assign(handles, handles + n);
    g->d_stage.assign((size_t)n, nullptr);
    g->stage_cap.assign((size_t)n, 0);
    // everything that can fail comes first and touches no member's state: a failure below leaves the handles as they were
    // (no dangling group / stream pointers on them, rcf_close still works), the stream and the event are destroyed
    MemberLocks ml(g->members);
    for (rcf_t *h : g->members)
        if (h->group) { set_error("a group member already belongs to a group"); return RCF_ESTATE; }   // (checked again under its lock)
    int rc = RCF_OK;
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) rc = RCF_EHIP;
    if (rc == RCF_OK && hipEventCreateWithFlags(&g->ingest_ev, hipEventDisableTiming) != hipSuccess) rc = RCF_EHIP;
    for (rcf_t *h : g->members) {
        if (rc != RCF_OK) break;
        flush_lagged(h);
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = RCF_EHIP;   // whatever it queued on its own stream comes first
    }
    if (rc != RCF_OK) {
        set_error("group open: %s", hipGetErrorString(hipGetLastError()));
        if (g->ingest_ev) (void)hipEventDestroy(g->ingest_ev);
        if (g->stream) (void)hipStreamDestroy(g->stream);
        return rc;
    }
    for (rcf_t *h : g->members) {                          // (cannot fail)
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

}  // extern "C"
