// rcf_plan.cpp -- the per-block schedule.  Everything one commit schedules is built on the host first (a BlockPlan:
// launch records in the pinned arena, jobs per dependency depth), then uploaded with one copy and launched in
// dependency order (rcf_launch.cpp).  process_block() is the sequence; the plan_*() functions each build one part.
#include <atomic>
#include <chrono>

#include "rcf_plan.h"

namespace rcfx {

namespace {
std::atomic<uint64_t> g_prof_ns[PlanProf::N];
std::atomic<uint64_t> g_prof_cnt[PlanProf::N];
const char *g_prof_name[PlanProf::N];
void prof_dump()
{
    for (int i = 0; i < PlanProf::N; ++i)
        if (g_prof_cnt[i].load())
            fprintf(stderr, "RCF_PLAN_PROF %-28s %10.3f ms  %9llu calls  %8.3f us/call\n", g_prof_name[i] ? g_prof_name[i] : "?",
                    g_prof_ns[i].load() * 1e-6, (unsigned long long)g_prof_cnt[i].load(),
                    g_prof_ns[i].load() * 1e-3 / (double)g_prof_cnt[i].load());
}
}  // namespace

bool PlanProf::on()
{
    static const bool v = [] {
        const char *e = getenv("RCF_PLAN_PROF");
        const bool o = e && atoi(e) != 0;
        if (o) atexit(prof_dump);
        return o;
    }();
    return v;
}

void PlanProf::add(int slot, const char *name, std::chrono::steady_clock::time_point &from)
{
    const auto now = std::chrono::steady_clock::now();
    g_prof_ns[slot] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - from).count();
    g_prof_cnt[slot] += 1;
    g_prof_name[slot] = name;
    from = now;
}

// the channel set's planning summary (see rcf_t::PlanCache), rebuilt when a channel was opened or closed or changed kind
const rcf_t::PlanCache &plan_cache(rcf_t *h)
{
    rcf_t::PlanCache &pc = h->plan_cache;
    if (pc.epoch == h->chans_epoch) return pc;
    pc.reach_x.clear();
    pc.max_depth = 0; pc.min_d0 = 0; pc.max_reach = 1;
    // (one pass over the channel map for everything that needs one: at 196608 channels each pass is ~8 ms of
    // pointer chasing)
    size_t need = 4096, pfb_reach = 0, n_fir = 0;
    for (auto &kv : h->chans) {
        const Chan &c = *kv.second;
        // (a filterbank tap has one short record; a FIR channel up to two launch records -- matrix-core launch and
        // zero-history fix-up --, a discriminator record, an exact-rotator fill, its bank-matrix dirty flag)
        need += c.is_tap ? sizeof(TapLaunch) + 8 : 2 * sizeof(ChanLaunch) + sizeof(DiscLaunch) + sizeof(RotFill) + 12 + 128;
        n_fir += c.is_tap ? 0 : 1;
        if (c.d_sym) need += sizeof(FmFirLaunch);
        if (c.audio) need += sizeof(AudioLaunch);
        pc.max_depth = std::max(pc.max_depth, c.depth);
        if (c.src < 0 && (pc.min_d0 == 0 || c.D < pc.min_d0)) pc.min_d0 = c.D;
        if (c.audio) pc.max_reach = std::max<size_t>(pc.max_reach, (size_t)std::max(std::max(c.audio->n_lpf, c.audio->n_hpf), c.audio->nt_rs));
        if (c.d_sym) { size_t &own = pc.reach_x[c.id]; own = std::max<size_t>(own, std::max<size_t>(1, (size_t)c.sym_ntaps)); }
        if (c.src >= RCF_SRC_PFB_BIN0) {              // (the bank's ring: one entry for all of its consumers, set after the loop)
            pfb_reach = std::max<size_t>(pfb_reach, (size_t)(c.T - 1 + c.D));
        } else if (c.src >= 0) {
            size_t &r = pc.reach_x[c.src];
            r = std::max<size_t>(r, (size_t)(c.T - 1 + c.D));
            pc.max_reach = std::max(pc.max_reach, r);
        }
        if (c.d_sym) pc.max_reach = std::max<size_t>(pc.max_reach, (size_t)c.sym_ntaps);
    }
    if (pfb_reach) {
        size_t &r = pc.reach_x[RCF_SRC_PFB_BIN0];
        r = std::max(r, pfb_reach);
        pc.max_reach = std::max(pc.max_reach, r);
    }
    need += 64 * (n_fir / 4 + 64);                        // per-class alignment slack
    pc.arena_need = need;
    // channels by depth, then by (D, T) class (a front-end has a handful of classes: linear search, then sorted -- the
    // order classes are launched in is the key order, channels inside a class in id order, as it always was)
    pc.by_depth.assign((size_t)pc.max_depth + 1, {});
    for (auto &kv : h->chans) {
        Chan *c = kv.second.get();
        auto &lvl = pc.by_depth[(size_t)c->depth];
        const std::pair<int, int> key{c->D, c->T};
        size_t q = 0;
        while (q < lvl.size() && lvl[q].first != key) ++q;
        if (q == lvl.size()) lvl.push_back(rcf_t::PlanCache::ClassBucket{key, {}});
        lvl[q].second.push_back(c);
    }
    for (auto &classes : pc.by_depth)
        std::sort(classes.begin(), classes.end(), [](const rcf_t::PlanCache::ClassBucket &a, const rcf_t::PlanCache::ClassBucket &b) { return a.first < b.first; });
    pc.epoch = h->chans_epoch;
    return pc;
}

// an upper bound of what plan_arena() will ask for, without planning anything (a group sizes its arena for all of its
// members before it plans any of them)
size_t arena_need_bound(rcf_t *h) { return plan_cache(h).arena_need; }

// arena for this commit: sized for every channel's launch records before anything is scheduled, so the schedule
// cannot run out half way (it mutates channel state as it goes); also the consumers' reach and the deepest chain
int plan_arena(rcf_t *h, BlockPlan &bp)
{
    const rcf_t::PlanCache &pc = plan_cache(h);
    bp.reach_x = &pc.reach_x;
    bp.max_depth = pc.max_depth;
    bp.min_d0 = pc.min_d0;
    bp.max_reach = pc.max_reach;
    bp.arena_need = pc.arena_need;
    const size_t arena_need = pc.arena_need;
    if (bp.ar != &bp.own_ar) return RCF_OK;        // a group's block: the group reserved its arena for all members
    // (a lagging stage-2 launch reads its records in the current arena: it goes out before the arenas are re-allocated or
    // the one it lives in can come round again)
    if (h->lag.pending && (arena_need > h->arenas.cap || h->arenas.fill + arena_need > h->arenas.cap)) flush_lagged(h);
    if (h->arenas.reserve(arena_need, h->stream) != RCF_OK) return RCF_EHIP;
    if (!h->arenas.mapped) h->copy_kernels = false;
    bp.a = h->arenas.cur;
    bp.arena_base = h->arenas.fill;
    bp.own_ar = Arena{h->arenas.h[bp.a], h->arenas.d[bp.a], bp.arena_base, h->arenas.cap};
    return RCF_OK;
}

// the filterbank's share of the block (derived channels need its new range)
int plan_pfb(rcf_t *h, BlockPlan &bp)
{
    const int64_t S0 = bp.S0, S1 = bp.S1;
    const size_t n = bp.n;
    PfbLaunch &pl = bp.pl;
    bool &run_pfb = bp.run_pfb;

    if (h->pfb.open) {
        Pfb &p = h->pfb;
        const int64_t n_lo = std::max(ceil_div(S0, p.D), p.n_abs0);
        const int64_t n_hi = floor_div(S1 - 1, p.D);
        p.produced_before = p.produced;
        if (n_hi >= n_lo) {
            const int64_t cnt = n_hi - n_lo + 1;
            if ((size_t)cnt + (bp.reach_x && bp.reach_x->count(RCF_SRC_PFB_BIN0) ? bp.reach_x->at(RCF_SRC_PFB_BIN0) : 0) > h->out_cap) {
                set_error("block yields %lld PFB frames (+%zu of history its stage-2 channels need) > ring capacity %zu",
                          (long long)cnt, bp.reach_x && bp.reach_x->count(RCF_SRC_PFB_BIN0) ? bp.reach_x->at(RCF_SRC_PFB_BIN0) : (size_t)0, h->out_cap);
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
            if (p.fm_mode) {                    // the discriminator of every bin, in the bank's own kernel (rcf_pfb_fm_enable)
                pl.fm_ring = p.d_fm;
                pl.fm_inc = p.d_fm_inc;
                pl.atan_tab = h->d_atan;
                pl.fm_mode = p.fm_mode;
                pl.fm_span = 0;                 // chosen at the launch (pfb5_fm_span_for)
                if (p.d_fm_edge) {              // look-back form: the chunks' workgroups hand their last frames over
                    pl.fm_edge = p.d_fm_edge;
                    pl.fm_flag = p.d_fm_flag;
                    pl.fm_err = p.d_fm_err;
                    pl.fm_slots = p.fm_slots;
                    pl.fm_local = p.fm_local;
                    pl.fm_tag = (unsigned long long)(++p.fm_serial) << 32;
                }
            } else {
                pl.fm_ring = nullptr;
                pl.fm_edge = nullptr;
                pl.fm_mode = 0;
            }
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
    if (c->src < RCF_SRC_PFB_BIN0) cp.all_bank_src = false;
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
    if (c->is_tap) {                            // served through the tap matrix, not by a FIR launch: its own short record
        TapLaunch tl{};
        tl.iq_ring = c->d_iq;
        tl.fm_ring = c->d_fm;
        tl.k_lo = k_lo; tl.k_abs0 = c->k_abs0; tl.n_seg0 = c->n_seg0;
        tl.angle0 = (double)c->angle0; tl.dangle = c->dangle; tl.logmag0 = c->logmag0; tl.dlogmag = c->dlogmag;
        tl.n_k = (int32_t)cnt;
        tl.bin = c->src - RCF_SRC_PFB_BIN0;
        tl.fm_only = c->fm_only ? 1 : 0;
        tap_list.push_back(tl);
        tap_bins.push_back(tl.bin);
    }
    ChanLaunch L{};
    if (!c->is_tap) {
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
    }
    DiscLaunch dl{};
    dl.iq_ring = c->d_iq;
    dl.fm_ring = c->d_fm;
    dl.n_lo = k_lo - c->k_abs0;
    dl.n_k = (int32_t)cnt;
    if (!c->is_tap) {
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
    Arena &ar = *bp.ar;
    auto &fir_by_depth = bp.fir_by_depth;
    auto &disc_jobs = bp.disc_jobs;
    auto &launches = cp.launches;
    auto &launched = cp.launched;
    auto &discs = cp.discs;
    const int max_n = cp.max_n;
    const bool shared_src = cp.shared_src;

    if (launches.empty()) return RCF_OK;
    FirJob job{};
    job.bank_src = cp.all_bank_src;
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
        // a group's block: launches whose records are self-contained (every channel its own source view) are merged
        // with the same class of the other front-ends -- the records stay on the host until the group has them all
        if (bp.defer && job.dims.chans_per_wg == 1) job.host.swap(rest);
        else if (!ar.put(rest, &job.dev)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
        fir_by_depth[depth].push_back(std::move(job));
    }
    if (!(job.dims.small && clean.empty())) {   // the small-T kernel writes the discriminator ring itself
        DiscJob dj{};
        dj.n = (int)discs.size(); dj.max_n = max_n;
        if (bp.defer) dj.host.swap(discs);
        else if (!ar.put(discs, &dj.dev)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
        disc_jobs.push_back(std::move(dj));
    }
    return RCF_OK;
}

// filterbank taps (matrix + records) and the records that go out as one launch each
int plan_tail(rcf_t *h, BlockPlan &bp)
{
    Arena &ar = *bp.ar;
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
        for (int b0 = 0; b0 + 16 <= NB && pl.fm_mode != 2; b0 += 16) {     // (fm_mode 2: the bank writes no bins ring to read runs from)
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
    // (a group's block: the exact-rotator fills and the symbol filters of all members go out as one launch each)
    if (!bp.defer && !rot_fills.empty() && !ar.put(rot_fills, &d_rot_fills)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
    if (!bp.defer && !symf.empty() && !ar.put(symf, &d_symf)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
    if (!audf.empty() && !ar.put(audf, &d_audf)) { set_error("launch arena exhausted"); return RCF_ENOMEM; }
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

void undo_block(rcf_t *h, BlockUndo &u)
{
    if (!u.armed) return;
    for (const BlockUndo::Saved &s : u.saved) {
        s.c->produced = s.produced; s.c->n_seg0 = s.n_seg0; s.c->blk_before = s.blk_before; s.c->blk_after = s.blk_after;
        s.c->blk_serial = s.blk_serial; s.c->angle0 = s.angle0; s.c->logmag0 = s.logmag0;
    }
    if (h->pfb.open) h->pfb.produced = h->pfb.produced_before;
    h->blk_serial = u.serial_before;
    u.armed = false;
}

// Everything of one block up to (not including) its launches.  On failure the handle is as it was before the call; on
// success the channels' counters have advanced and `undo` can still take that back (a group whose LATER member fails).
int plan_block(rcf_t *h, size_t n, BlockPlan &bp, BlockUndo &undo)
{
    bp.S0 = h->total_in;
    bp.S1 = bp.S0 + (int64_t)n;
    bp.n = n;
    auto tp = std::chrono::steady_clock::now();
    int rc = plan_arena(h, bp);
    if (rc == RCF_OK) rc = plan_pfb(h, bp);
    if (rc != RCF_OK) return rc;
    RCF_PROF(0, "plan_arena+pfb", tp);
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
    // (every channel is saved right before plan_channel() touches it: one pass over the channels instead of two)
    undo.saved.clear();
    undo.saved.reserve(h->chans.size());
    undo.serial_before = h->blk_serial;
    undo.armed = true;
    auto roll_back = [&](int code) { undo_block(h, undo); return code; };
    // channels, by depth then by (D, T) class (the handle's PlanCache holds the buckets)
    bp.fir_by_depth.resize(bp.max_depth + 1);
    bp.serial = ++h->blk_serial;
    const auto &by_depth = h->plan_cache.by_depth;
    RCF_PROF(1, "undo + buckets", tp);
    for (int depth = 0; depth <= bp.max_depth; ++depth) {
        const auto &classes = by_depth[(size_t)depth];
        for (const auto &cls : classes) {
            ClassPlan cp;
            cp.launches.reserve(cls.second.size());
            cp.launched.reserve(cls.second.size());
            cp.discs.reserve(cls.second.size());
            const size_t nc = cls.second.size();
            for (size_t ci = 0; ci < nc; ++ci) {
                Chan *c = cls.second[ci];
                if (ci + 6 < nc) {                    // (a thousand front-ends' channels do not fit the host's caches)
                    const char *nx = reinterpret_cast<const char *>(cls.second[ci + 6]);
                    __builtin_prefetch(nx); __builtin_prefetch(nx + 64); __builtin_prefetch(nx + 128); __builtin_prefetch(nx + 192);
                }
                undo.saved.push_back(BlockUndo::Saved{c, c->produced, c->n_seg0, c->blk_before, c->blk_after, c->blk_serial, c->angle0, c->logmag0});
                if ((rc = plan_channel(h, bp, cp, c, cls.first.first)) != RCF_OK) return roll_back(rc);
            }
            RCF_PROF(3, "plan_channel (class)", tp);
            if ((rc = plan_class_jobs(h, bp, cp, depth, cls.first)) != RCF_OK) return roll_back(rc);
            RCF_PROF(4, "plan_class_jobs", tp);
        }
    }
    if ((rc = plan_tail(h, bp)) != RCF_OK) return roll_back(rc);
    RCF_PROF(5, "plan_tail", tp);
    {
        // RCF_FAIL_PLAN_AT=<k>: the k-th block of a handle fails here as an exhausted launch arena would (tests of the
        // roll-back above: tests/test_gpu_round4.py)
        static const long fail_at = [] { const char *e = getenv("RCF_FAIL_PLAN_AT"); return e ? atol(e) : 0L; }();
        if (fail_at > 0 && (long)++h->plan_calls == fail_at) {
            set_error("injected planning failure (RCF_FAIL_PLAN_AT)");
            return roll_back(RCF_ENOMEM);
        }
    }
    return RCF_OK;
}

int process_block(rcf_t *h, size_t n)
{
    if (h->graveyard.size() > 512) drain_graveyard(h);     // bounded even if nobody ever syncs or reads
    BlockPlan bp;
    BlockUndo undo;
    int rc = plan_block(h, n, bp, undo);
    if (rc != RCF_OK) return rc;
    if ((rc = launch_plan(h, bp)) != RCF_OK) return rc;      // kernels may be queued: the handle's stream state is undefined now
    if ((rc = run_scan(h, bp)) != RCF_OK) return rc;
    return finish_block(h, bp);
}

}  // namespace rcfx
