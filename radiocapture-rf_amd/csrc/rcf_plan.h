// rcf_plan.h -- the per-block schedule: what one commit of one front-end launches, built on the host first.
#pragma once
#include <chrono>

#include "rcf_state.h"

namespace rcfx {

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

inline int choose_kt(int D, int T)
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
    std::vector<ChanLaunch> host;      // BlockPlan::defer: records not yet in the arena (the group merges them first)
    bool bank_src = false;             // every channel of the job reads a bin of the filterbank (what the stage-2 lag may defer)
};
struct DiscJob { const DiscLaunch *dev; int n; int max_n; std::vector<DiscLaunch> host; };

struct BlockPlan {
    int64_t S0 = 0, S1 = 0;            // the block's samples [S0, S1)
    size_t n = 0;

    // how far back every consumer of a ring reaches beyond the block's new samples (a block's writes must not
    // overwrite what the same block's readers still need): derived channels T - 1 + D of source output, the
    // discriminator one sample, the symbol filter its taps.  Every ring reaches back 1 (the discriminator); only
    // sources of other channels and channels with a symbol filter reach further -- the map holds just those (with
    // 131072 plain wideband channels it stays empty: a std::map entry per channel and block was a tenth of the
    // host's schedule time).
    const std::unordered_map<int, size_t> *reach_x = nullptr;   // source id (channel id / RCF_SRC_PFB_BIN0) -> samples (the handle's PlanCache)
    int max_depth = 0;
    int min_d0 = 0;                    // smallest decimation among the channels on the wideband stream (0: none)
    size_t max_reach = 1;              // largest consumer reach of any ring (see reach_x) / voice-chain filter
    size_t arena_need = 0;
    int a = 0;                         // arena in use, and where this commit's records start in it
    size_t arena_base = 0;
    Arena own_ar{nullptr, nullptr, 0, 0};
    Arena *ar = &own_ar;               // a member of a group plans into the group's arena instead
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
    bool defer = false;                // a member of a group: mergeable records stay on the host (FirJob::host, DiscJob::host)
    bool history_done = false;         // launch_plan copied the history tail together with the launch records

    size_t reach(int id) const
    {
        if (!reach_x || reach_x->empty()) return 1;
        auto it = reach_x->find(id);
        return it == reach_x->end() ? (size_t)1 : std::max<size_t>(1, it->second);
    }
};

// channels of one (depth, D, T) class collected for launching
struct ClassPlan {
    std::vector<ChanLaunch> launches;
    std::vector<Chan *> launched;
    std::vector<DiscLaunch> discs;
    int max_n = 0;
    bool shared_src = true;
    bool all_bank_src = true;          // every launched channel's source is a filterbank bin
};

// what planning a block changed in the handle, so that it can be taken back while nothing has been queued
struct BlockUndo {
    struct Saved { Chan *c; int64_t produced, n_seg0, blk_before, blk_after; uint64_t blk_serial; long double angle0; double logmag0; };
    std::vector<Saved> saved;
    uint64_t serial_before = 0;
    bool armed = false;
};

// RCF_PLAN_PROF=1: cumulative host time per planning section, printed at exit (tools/rt_probe.py runs with it)
struct PlanProf {
    static constexpr int N = 16;
    static bool on();
    static void add(int slot, const char *name, std::chrono::steady_clock::time_point &from);
};
#define RCF_PROF(slot, name, tp) do { if (PlanProf::on()) PlanProf::add(slot, name, tp); } while (0)

// rcf_plan.cpp
int plan_block(rcf_t *h, size_t n, BlockPlan &bp, BlockUndo &undo);
void undo_block(rcf_t *h, BlockUndo &undo);
size_t arena_need_bound(rcf_t *h);
const rcf_t::PlanCache &plan_cache(rcf_t *h);
int plan_arena(rcf_t *h, BlockPlan &bp);
int plan_pfb(rcf_t *h, BlockPlan &bp);
int plan_channel(rcf_t *h, BlockPlan &bp, ClassPlan &cp, Chan *c, int D);
int plan_class_jobs(rcf_t *h, BlockPlan &bp, ClassPlan &cp, int depth, std::pair<int, int> cls_key);
int plan_tail(rcf_t *h, BlockPlan &bp);
int check_block_capacity(rcf_t *h, const BlockPlan &bp);
// rcf_launch.cpp
int launch_plan(rcf_t *h, BlockPlan &bp);
int run_scan(rcf_t *h, const BlockPlan &bp);
int finish_block(rcf_t *h, const BlockPlan &bp);

}  // namespace rcfx
