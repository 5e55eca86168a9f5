// rcf_timing.cpp -- per-kernel-class timing with HIP events on the handle's own stream (rcf_timing_*).
#include "rcf_plan.h"

namespace rcfx {

hipEvent_t time_event(rcf_t *h)
{
    hipEvent_t e = nullptr;
    if (!h->time_pool.empty()) { e = h->time_pool.back(); h->time_pool.pop_back(); return e; }
    (void)hipEventCreate(&e);
    return e;
}

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

}  // namespace rcfx

using namespace rcfx;

// =================================================================== C ABI
extern "C" {

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

}  // extern "C"
