// rcf_comm.cpp -- the one collective of the path: the all-gather of detected-peak lists (RCCL over xGMI).
#include <dlfcn.h>
#include "rcf_plan.h"

namespace rcfx {

namespace {
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
}  // namespace

void comm_destroy(rcf_t *h)
{
    if (!h->comm) return;
    if (RcclApi *r = rccl()) (void)r->CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_rank = 0;
    h->comm_size = 1;
}

}  // namespace rcfx

using namespace rcfx;

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

}  // extern "C"
