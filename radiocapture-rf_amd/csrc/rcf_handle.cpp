// rcf_handle.cpp -- the front-end handle of librcf.so (include/rcf.h): error plumbing, slab pools and deferred frees,
// open / close / sync, and the wideband ingest (push / commit).  One rcf_t == one SDR source of
// /root/reference/rc_frontend/receiver.py; every push/commit runs all open channels, the filterbank, the
// discriminators and an armed scan over the new block on the handle's HIP stream (rcf_plan.cpp, rcf_launch.cpp).
//
// Memory plan (sized for 288 GB of HBM3E): two wideband buffers [hist | block] ping-pong so the
// producer can fill the next block while kernels chew on the current one; the tail of every block is
// copied behind the other buffer's block as its history, which lets every kernel address the stream
// linearly (no wrap handling on the hot loads).  Narrowband outputs live in per-channel power-of-two
// rings addressed by the absolute output index.
#include "rcf_plan.h"

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

// a front-end whose group block failed AFTER launches had been queued (rcf_group.cpp) has channel counters ahead of what the
// device holds: every later call that would push, commit or read says so instead of going on with that state
static int faulted(rcf_t *h)
{
    if (!h->fault) return RCF_OK;
    set_error("the front-end is faulted (%s): close it and open a new one", h->fault_text);
    return RCF_ESTATE;
}

int set_dev_ingest(rcf_t *h)
{
    if (faulted(h)) return RCF_ESTATE;
    RCF_HIP(hipSetDevice(h->device));
    return RCF_OK;
}

int set_dev(rcf_t *h)
{
    if (faulted(h)) return RCF_ESTATE;
    RCF_HIP(hipSetDevice(h->device));
    flush_lagged(h);
    return RCF_OK;
}

void bury(rcf_t *h, void *p, size_t slice)
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

int ArenaSet::create()
{
    for (int i = 0; i < 2; ++i) {
        RCF_HIP(hipHostMalloc(&h[i], cap, hipHostMallocDefault));
        RCF_HIP(hipMalloc(&d[i], cap));
        void *dv = nullptr;
        h_dev[i] = hipHostGetDevicePointer(&dv, h[i], 0) == hipSuccess ? static_cast<unsigned char *>(dv) : nullptr;
        if (!h_dev[i]) mapped = false;
        RCF_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    return RCF_OK;
}

void ArenaSet::destroy()
{
    for (int i = 0; i < 2; ++i) {
        if (d[i]) (void)hipFree(d[i]);
        if (h[i]) (void)hipHostFree(h[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        d[i] = h[i] = h_dev[i] = nullptr;
        ev[i] = nullptr;
    }
}

int ArenaSet::reserve(size_t need, hipStream_t stream)
{
    if (!h[0] && create() != RCF_OK) return RCF_EHIP;
    if (need > cap) {
        RCF_HIP(hipStreamSynchronize(stream));
        size_t ncap = cap;
        while (ncap < need) ncap *= 2;
        for (int i = 0; i < 2; ++i) {
            unsigned char *nh = nullptr, *nd = nullptr;
            RCF_HIP(hipHostMalloc(&nh, ncap, hipHostMallocDefault));
            RCF_HIP(hipMalloc(&nd, ncap));
            (void)hipHostFree(h[i]);
            (void)hipFree(d[i]);
            h[i] = nh;
            d[i] = nd;
            used[i] = false;
            void *dv = nullptr;
            h_dev[i] = hipHostGetDevicePointer(&dv, nh, 0) == hipSuccess ? static_cast<unsigned char *>(dv) : nullptr;
            if (!h_dev[i]) mapped = false;
        }
        cap = ncap;
        fill = 0;
    }
    if (fill + need > cap) {
        // this arena is full: everything queued so far may still read it -- one event now guards its reuse -- and the
        // other one must have been drained
        RCF_HIP(hipEventRecord(ev[cur], stream));
        used[cur] = true;
        cur ^= 1;
        fill = 0;
        if (used[cur]) RCF_HIP(hipEventSynchronize(ev[cur]));
    }
    return RCF_OK;
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

}  // namespace rcfx

using namespace rcfx;

// =================================================================== C ABI
extern "C" {

const char *rcf_version(void) { return "rcf-mi355x 0.1 (gfx950)"; }
const char *rcf_last_error(void) { return g_err; }

int rcf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rcf_device_pci_bus_id(int device, char *out, size_t cap)
{
    if (!out || cap < 13 || device < 0 || device >= rcf_device_count()) { set_error("no such device / buffer too small"); return RCF_EINVAL; }
    if (hipDeviceGetPCIBusId(out, (int)cap, device) != hipSuccess) { (void)hipGetLastError(); set_error("hipDeviceGetPCIBusId failed"); return RCF_EHIP; }
    for (char *c = out; *c; ++c) *c = (char)tolower((unsigned char)*c);          // sysfs spells it in lower case
    return RCF_OK;
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
        if (const char *lg = getenv("RCF_S2_LAG")) h->lag_enabled = atoi(lg) != 0;
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
    }
    // (the launch-record arenas -- 2 x 8 MiB pinned + 2 x 8 MiB device -- come with the first block the handle processes on
    // its own: a member of a group plans into the group's arena and never needs them)
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
    if (h->group) { set_error("the handle belongs to a group: rcf_group_close first"); return RCF_ESTATE; }
    (void)hipSetDevice(h->device);
    flush_lagged(h);
    (void)hipStreamSynchronize(h->stream);
    for (auto &kv : h->chans) free_channel(h, kv.second.get());
    h->chans.clear();
    Pfb &p = h->pfb;
    bury(h, p.d_ptaps); bury(h, p.d_tw); bury(h, p.d_bins); bury(h, p.d_stage);
    bury(h, p.d_fm); bury(h, p.d_fm_inc); bury(h, p.d_fm_stage); bury(h, p.d_fm_edge); bury(h, p.d_fm_flag); bury(h, p.d_fm_err);
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
    }
    bury(h, h->d_gather);
    if (h->h_many) (void)hipHostFree(h->h_many);
    bury(h, h->d_partial);
    bury(h, h->d_tapmat);
    drain_graveyard(h);
    h->arenas.destroy();
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

int rcf_set_rotator(rcf_t *h, int exact)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->chans.empty()) { set_error("rcf_set_rotator: channels are already open"); return RCF_ESTATE; }
    h->exact_rot = exact != 0;
    return RCF_OK;
}

int rcf_set_stage2_lag(rcf_t *h, int on)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;            // (flushes a launch that is lagging now)
    h->lag_enabled = on != 0;
    return RCF_OK;
}

int rcf_set_decim_rule(rcf_t *h, int decim_rule)
{
    if (!h || (decim_rule != RCF_DECIM_EXACT && decim_rule != RCF_DECIM_FLOOR)) { set_error("bad decimation rule"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    h->decim_rule = decim_rule;
    return RCF_OK;
}

void *rcf_stream(rcf_t *h)
{
    if (!h) return nullptr;
    // (a consumer that orders itself on this stream must find every launch it could observe queued on it: the deferred
    // stage-2 launch of the previous block goes out first)
    std::lock_guard<std::mutex> g(h->mu);
    (void)set_dev(h);
    return (void *)h->stream;
}
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
    if (set_dev_ingest(h)) return RCF_EHIP;
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
    if (set_dev_ingest(h)) return RCF_EHIP;
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
    if (set_dev_ingest(h)) return RCF_EHIP;
    return process_block(h, n);
}

}  // extern "C"
