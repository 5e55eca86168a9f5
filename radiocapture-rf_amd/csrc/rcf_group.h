// rcf_group.h -- what rcf_group.cpp (grouped launches) and rcf_pump.cpp (the real-time pump that drives them) share.
#pragma once
#include <mutex>
#include <vector>

#include "rcf_plan.h"

struct rcf_pump;

struct rcf_group {
    int device = 0;
    std::vector<rcf_t *> members;
    hipStream_t stream = nullptr;
    rcfx::ArenaSet arenas;
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

namespace rcfx {

struct GroupItem {
    int m;                 // member index
    size_t n;              // samples
    const void *src;       // host samples (nullptr: already resident -- commit)
    const void *dsrc;      // the same memory as the device sees it, if the caller knows (the pump resolves its rings once)
};

inline size_t group_sample_bytes(int fmt) { return fmt == RCF_FMT_CF32 ? sizeof(float2) : raw_sample_bytes(fmt); }

struct MemberLocks {       // every member's mutex, in index order (a group call owns all of its members)
    std::vector<rcf_t *> &ms;
    explicit MemberLocks(std::vector<rcf_t *> &m) : ms(m) { for (rcf_t *h : ms) h->mu.lock(); }
    ~MemberLocks() { for (auto it = ms.rbegin(); it != ms.rend(); ++it) (*it)->mu.unlock(); }
};

// one block of each listed member.  g->mu and the members' mutexes are held.  wait: return only once the sources have
// been read (the caller may reuse its buffers); the pump passes false -- its rings are not overwritten for many periods.
int group_process(rcf_group *g, const std::vector<GroupItem> &items, int fmt, float scale, float offset, bool wait);

}  // namespace rcfx
