// rcf_bank.cpp -- C ABI of the polyphase filterbank and the scanner (fft_vector.py + fft_peak_detection.py).
#include "rcf_plan.h"

namespace rcfx {

// The rotator increment's angle a discriminator-only channel on bin k of the bank carries (Chan::dangle of a tap opened
// with rcf_pfb_tap_open(bin, gr_phase): rcf_chan.cpp) -- for every bin at once, uploaded and turned into float32 phasors
// on the device (launch_pfb5_fm_inc: tap_finalize's own expression)
int pfb_fm_upload_increments(rcf_t *h)
{
    Pfb &p = h->pfb;
    std::vector<double> dangle((size_t)p.NB, 0.0);
    // what rcf_source_shift adds to every bin-fed channel's NCO (upload_composite: a D = 1, T = 1 channel at the bin rate)
    double shift_angle = 0.0;
    if (h->shift_hz != 0.0) {
        const float one = 1.0f;
        std::vector<float> ct;
        float incr[2];
        design_composite(&one, 1, 1, h->shift_hz, h->fs / p.D, ct, incr);
        shift_angle = std::atan2((double)incr[1], (double)incr[0]);
    }
    for (int bin = 0; bin < p.NB; ++bin) {
        double d = shift_angle;
        if (p.fm_gr_phase) d += pfb_tap_gr_dangle(h, bin);
        dangle[(size_t)bin] = d;
    }
    double *d_dangle = nullptr;
    RCF_HIP(hipMalloc(&d_dangle, sizeof(double) * dangle.size()));
    if (!hip_ok(hipMemcpyAsync(d_dangle, dangle.data(), sizeof(double) * dangle.size(), hipMemcpyHostToDevice, h->stream), "hipMemcpy(fm increments)")) {
        (void)hipFree(d_dangle);
        return RCF_EHIP;
    }
    launch_pfb5_fm_inc(d_dangle, p.d_fm_inc, p.NB, h->stream);
    RCF_HIP(hipStreamSynchronize(h->stream));          // (dangle is a local vector; the stream also orders the table before the next block)
    (void)hipFree(d_dangle);
    return RCF_OK;
}

}  // namespace rcfx

using namespace rcfx;

// =================================================================== C ABI
extern "C" {

// ------------------------------------------------------------------ PFB
int rcf_pfb_open(rcf_t *h, int n_bins, int decim, const float *taps, int ntaps)
{
    if (!h || !taps || ntaps < 1 || n_bins < 1 || decim < 1) { set_error("bad PFB arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    if (h->pfb.open) { set_error("PFB already open"); return RCF_ESTATE; }
    const int P = (ntaps + n_bins - 1) / n_bins;
    if (n_bins % decim || !pfb_supported(n_bins, decim, P)) {
        set_error("unsupported PFB shape: bins=%d decim=%d taps/branch=%d", n_bins, decim, P);
        return RCF_EINVAL;
    }
    const bool fm = pfb_frame_major(n_bins);
    if (!fm && h->out_cap < (size_t(1) << kPfbTileLog2)) { set_error("output capacity %zu < one ring tile", h->out_cap); return RCF_ECAP; }
    const size_t ring_samples = fm ? (size_t)n_bins * h->out_cap : (size_t)(h->out_cap >> kPfbTileLog2) * (size_t)pfb_tile_pitch(n_bins);
    // 32-bit buffer offsets: the wideband buffer, and the tiled ring of the power-of-two banks (one descriptor for the
    // whole ring).  The frame-major banks address their ring through one descriptor per frame row: no limit there.
    if ((!fm && (uint64_t)ring_samples * sizeof(float2) >= (1ull << 31)) ||
        (uint64_t)(h->hist_cap + h->block_cap) * sizeof(float2) >= (1ull << 31)) {
        set_error("PFB rings / wideband buffer exceed the 2 GiB range of 32-bit buffer offsets");
        return RCF_ECAP;
    }
    if ((size_t)P * n_bins + (size_t)decim > h->hist_cap) { set_error("history capacity %zu < P*bins", h->hist_cap); return RCF_ECAP; }
    Pfb &p = h->pfb;
    p.NB = n_bins; p.D = decim; p.T = ntaps; p.P = P;
    p.proto.assign(taps, taps + ntaps);
    p.Ppad = pfb_padded_p(n_bins, decim, P);
    std::vector<float> pt((size_t)p.Ppad * n_bins, 0.f);
    for (int i = 0; i < ntaps; ++i) pt[i] = taps[i];          // pt[p*NB + rho] = h[NB p + rho]
    std::vector<float> tw(2 * (size_t)n_bins);
    for (int i = 0; i < n_bins; ++i) {
        const double a = kTwoPi * i / n_bins;
        tw[2 * i] = (float)std::cos(a);
        tw[2 * i + 1] = (float)std::sin(a);
    }
    RCF_HIP(hipMalloc(&p.d_ptaps, sizeof(float) * pt.size()));
    RCF_HIP(hipMemcpy(p.d_ptaps, pt.data(), sizeof(float) * pt.size(), hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&p.d_tw, sizeof(float2) * (size_t)n_bins));
    RCF_HIP(hipMemcpy(p.d_tw, tw.data(), sizeof(float2) * (size_t)n_bins, hipMemcpyHostToDevice));
    p.frame_major = fm;
    RCF_HIP(hipMalloc(&p.d_bins, sizeof(float2) * ring_samples));
    RCF_HIP(hipMemsetAsync(p.d_bins, 0, sizeof(float2) * ring_samples, h->stream));
    p.rd.assign(n_bins, 0);
    p.start_sample = h->total_in;
    p.n_abs0 = ceil_div(p.start_sample, decim);
    p.produced = p.produced_before = 0;
    p.open = true;
    return RCF_OK;
}

int rcf_pfb_close(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open) return RCF_OK;
    for (auto it = h->chans.begin(); it != h->chans.end();) {
        if (it->second->src >= RCF_SRC_PFB_BIN0) { free_channel(h, it->second.get()); it = h->chans.erase(it); }
        else ++it;
    }
    bury(h, p.d_ptaps); bury(h, p.d_tw); bury(h, p.d_bins); bury(h, p.d_stage);
    bury(h, p.d_fm); bury(h, p.d_fm_inc); bury(h, p.d_fm_stage); bury(h, p.d_fm_edge); bury(h, p.d_fm_flag); bury(h, p.d_fm_err);
    p = Pfb();
    ++h->chans_epoch;
    return RCF_OK;
}

int rcf_pfb_tap_leakage(double samp_rate, int n_bins, const float *taps, int ntaps, int bin, double *leak_l2,
                        double *const_phase)
{
    if (!(samp_rate > 0) || n_bins < 1 || !taps || ntaps < 1 || bin < 0 || bin >= n_bins) {
        set_error("bad tap-leakage arguments");
        return RCF_EINVAL;
    }
    design_tap_leakage(samp_rate, n_bins, taps, ntaps, bin, leak_l2, const_phase);
    return RCF_OK;
}

int rcf_pfb_shape_supported(int n_bins, int decim, int ntaps)
{
    if (n_bins < 1 || decim < 1 || ntaps < 1 || n_bins % decim) return 0;
    return pfb_supported(n_bins, decim, (ntaps + n_bins - 1) / n_bins) ? 1 : 0;
}

int64_t rcf_pfb_produced(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    return h->pfb.open ? h->pfb.produced : RCF_ESTATE;
}

int64_t rcf_pfb_read_bin(rcf_t *h, int bin, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open || bin < 0 || bin >= p.NB) { set_error("no such PFB bin %d", bin); return RCF_EINVAL; }
    if (p.fm_mode == 2) { set_error("the bank writes its discriminator ring only (rcf_pfb_fm_enable mode 2): no bins to read"); return RCF_ESTATE; }
    // one bin out of the bank's ring (tiled or frame-major): gather its unread samples into a contiguous staging buffer
    int64_t avail = p.produced - p.rd[bin];
    if (avail <= 0 || max_samples == 0) return 0;
    if ((size_t)avail > h->out_cap) { p.rd[bin] = p.produced - (int64_t)h->out_cap; avail = (int64_t)h->out_cap; }
    const int64_t n = std::min<int64_t>(avail, (int64_t)max_samples);
    if (!p.d_stage) RCF_HIP(hipMalloc(&p.d_stage, sizeof(float2) * h->out_cap));
    SrcRange sr{};
    if (!source_range(h, RCF_SRC_PFB_BIN0 + bin, 0, 0, &sr)) return RCF_ESTATE;
    launch_gather_view(sr.view, p.rd[bin], p.d_stage, (size_t)n, h->stream);
    RCF_HIP(hipMemcpyAsync(out, p.d_stage, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    free_graveyard_idle(h);
    p.rd[bin] += n;
    return n;
}

int rcf_pfb_fm_enable(rcf_t *h, int mode, int gr_phase)
{
    if (!h || mode < 0 || mode > 2) { set_error("bad fm mode"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open) { set_error("no filterbank open"); return RCF_ESTATE; }
    if (mode == 0) {
        if (p.fm_mode) p.fm_until = p.produced;          // frames from here on are not demodulated: readers stop here
        p.fm_mode = 0;
        return RCF_OK;
    }
    if (!p.frame_major || !pfb5_fm_supported(p.NB, p.D, p.P)) {
        set_error("no fused-discriminator kernel for bins=%d decim=%d taps/branch=%d", p.NB, p.D, p.P);
        return RCF_EINVAL;
    }
    const size_t need = pfb5_fm_history(p.NB, p.D, p.P);
    if (need > h->hist_cap) { set_error("history capacity %zu < %zu (the fused discriminator's halo chunk)", h->hist_cap, need); return RCF_ECAP; }
    if (!p.d_fm) {
        const size_t ring = (size_t)p.NB * h->out_cap;
        RCF_HIP(hipMalloc(&p.d_fm, sizeof(float) * ring));
        RCF_HIP(hipMemsetAsync(p.d_fm, 0, sizeof(float) * ring, h->stream));
        RCF_HIP(hipMalloc(&p.d_fm_inc, sizeof(float2) * (size_t)p.NB));
        // the look-back form's hand-over rows (RCF_PFB5_FM_LOOKBACK=0: the span form, which needs none): more slots than
        // workgroups can be resident at once, so a row is never rewritten while the chunk behind it still wants it
        static const bool lookback = [] { const char *e = getenv("RCF_PFB5_FM_LOOKBACK"); return !e || atoi(e) != 0; }();
        if (lookback) {
            p.fm_slots = 4096;
            RCF_HIP(hipMalloc(&p.d_fm_edge, sizeof(unsigned long long) * (size_t)(p.fm_slots + 9) * (size_t)p.NB));
            RCF_HIP(hipMalloc(&p.d_fm_flag, sizeof(unsigned long long) * (size_t)p.fm_slots * 8));      // one flag per wave of a chunk's workgroup
            RCF_HIP(hipMalloc(&p.d_fm_err, sizeof(int)));
            RCF_HIP(hipMemsetAsync(p.d_fm_flag, 0, sizeof(unsigned long long) * (size_t)p.fm_slots * 8, h->stream));
            RCF_HIP(hipMemsetAsync(p.d_fm_err, 0, sizeof(int), h->stream));
            static const bool want_local = [] { const char *e = getenv("RCF_PFB5_FM_LOCAL"); return !e || atoi(e) != 0; }();
            p.fm_local = want_local && pfb5_xcd_map_ok(h->device, h->stream) ? 1 : 0;
        }
        p.rd_fm.assign((size_t)p.NB, p.produced);
        p.fm_from = p.produced;
    } else if (p.fm_mode == 0) {                       // switched on again: the frames in between were not demodulated
        p.fm_from = p.produced;
        for (auto &r : p.rd_fm) r = std::max(r, p.produced);
    }
    p.fm_gr_phase = gr_phase ? 1 : 0;
    const int rc = pfb_fm_upload_increments(h);
    if (rc != RCF_OK) return rc;
    p.fm_mode = mode;
    return RCF_OK;
}

int64_t rcf_pfb_read_fm(rcf_t *h, int bin, float gain, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open || !p.d_fm || bin < 0 || bin >= p.NB) { set_error("no discriminator ring / no such bin %d", bin); return RCF_EINVAL; }
    int64_t &rd = p.rd_fm[(size_t)bin];
    const int64_t end = p.fm_mode ? p.produced : p.fm_until;
    int64_t avail = end - rd;
    if (avail <= 0 || max_samples == 0) return 0;
    if (p.produced - rd > (int64_t)h->out_cap) {        // overwritten since: skip to the oldest frame the ring still holds
        rd = p.produced - (int64_t)h->out_cap;
        avail = end - rd;
        if (avail <= 0) return 0;
    }
    const int64_t n = std::min<int64_t>(avail, (int64_t)max_samples);
    if (!p.d_fm_stage) RCF_HIP(hipMalloc(&p.d_fm_stage, sizeof(float) * h->out_cap));
    launch_gather_f32(p.d_fm + bin, h->ring_mask, p.NB, rd, gain, p.d_fm_stage, (size_t)n, h->stream);
    RCF_HIP(hipMemcpyAsync(out, p.d_fm_stage, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    free_graveyard_idle(h);
    rd += n;
    return n;
}

int64_t rcf_pfb_fm_lost(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open || !p.d_fm) { set_error("no discriminator ring"); return RCF_ESTATE; }
    if (!p.d_fm_err) return 0;
    int v = 0;
    RCF_HIP(hipMemcpyAsync(&v, p.d_fm_err, sizeof v, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    return v;
}

int rcf_pfb_fm_ring(rcf_t *h, void **fm_ring, size_t *capacity_frames, int64_t *first_frame)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    if (!h->pfb.open || !h->pfb.d_fm) return RCF_ESTATE;
    if (fm_ring) *fm_ring = h->pfb.d_fm;
    if (capacity_frames) *capacity_frames = h->out_cap;
    if (first_frame) *first_frame = h->pfb.fm_from;
    return RCF_OK;
}

int rcf_pfb_rings(rcf_t *h, void **bins_ring, size_t *capacity, size_t *pitch)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;                  // (queues the deferred stage-2 launch: see rcf_chan_rings)
    if (!h->pfb.open) return RCF_ESTATE;
    if (bins_ring) *bins_ring = h->pfb.d_bins;
    if (capacity) *capacity = h->out_cap;
    if (pitch) *pitch = h->pfb.frame_major ? 0 : (size_t(1) << kPfbTileLog2);   // frames per tile (0: frame-major)
    return RCF_OK;
}

// ------------------------------------------------------------------ scan
int rcf_scan_start(rcf_t *h, int fft_len, int n_frames, int avg_len)
{
    if (!h || n_frames < 1 || avg_len < 1) { set_error("bad scan arguments"); return RCF_EINVAL; }
    if (!scan_supported(fft_len)) { set_error("unsupported scan FFT length %d", fft_len); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    if ((size_t)fft_len > h->hist_cap) { set_error("history capacity %zu < fft_len %d", h->hist_cap, fft_len); return RCF_ECAP; }
    Scan &s = h->scan;
    // frames per launch: enough workgroups to fill 256 CUs (2^25 samples per launch), bounded so that
    // the log-magnitude ring ((avg_len + chunk) x N floats) and the four-step scratch stay modest
    static const int chunk_log2 = [] { const char *e = getenv("RCF_SCAN_CHUNK_LOG2"); return e ? atoi(e) : 25; }();
    int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(512, (int64_t(1) << chunk_log2) / fft_len));
    chunk = std::min(chunk, n_frames);
    if (s.d_vring && s.N == fft_len && s.L == avg_len && s.chunk == chunk) {
        // same geometry as the previous scan: keep every buffer (fresh device allocations cost tens of
        // milliseconds of first-touch page faults), just reset the running state
        s.n_frames = n_frames;
        s.frames_done = 0;
        s.done = false;
        s.start_sample = h->total_in;
        RCF_HIP(hipMemsetAsync(s.d_sum, 0, sizeof(float) * (size_t)fft_len, h->stream));
        RCF_HIP(hipMemsetAsync(s.d_out, 0, sizeof(float) * (size_t)fft_len, h->stream));
        s.armed = true;
        return RCF_OK;
    }
    bury(h, s.d_window); bury(h, s.d_vring); bury(h, s.d_sum); bury(h, s.d_out); bury(h, s.d_tw);
    bury(h, s.d_scratch); bury(h, s.d_peaks); bury(h, s.d_peak_ws);
    s = Scan();
    s.N = fft_len; s.n_frames = n_frames; s.L = avg_len;
    s.chunk = chunk;
    s.R = avg_len + s.chunk;
    std::vector<float> win(fft_len), tw(2 * (size_t)fft_len);
    design_window(RCF_WIN_BLACKMAN_HARRIS, fft_len, win.data());
    auto fill = [&](size_t at, int count, double step) {      // tw[at + i] = e^{-j step i}
        for (int i = 0; i < count; ++i) {
            tw[2 * (at + i)] = (float)std::cos(-step * i);
            tw[2 * (at + i) + 1] = (float)std::sin(-step * i);
        }
    };
    int n1 = 0, n2 = 0;
    if (fft_len > 16384 && scan4_split(fft_len, &n1, &n2)) {
        // four-step tables: [e^{-2 pi i n/N1} | e^{-2 pi i n/N2} | W_N^i, i<1024 | W_N^{1024 j}]
        fill(0, n1, kTwoPi / n1);
        fill((size_t)n1, n2, kTwoPi / n2);
        fill((size_t)n1 + n2, 1024, kTwoPi / fft_len);
        fill((size_t)n1 + n2 + 1024, fft_len / 1024, kTwoPi * 1024.0 / fft_len);
    } else {
        fill(0, fft_len, kTwoPi / fft_len);
    }
    RCF_HIP(hipMalloc(&s.d_window, sizeof(float) * (size_t)fft_len));
    RCF_HIP(hipMemcpy(s.d_window, win.data(), sizeof(float) * (size_t)fft_len, hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&s.d_tw, sizeof(float2) * (size_t)fft_len));
    RCF_HIP(hipMemcpy(s.d_tw, tw.data(), sizeof(float2) * (size_t)fft_len, hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&s.d_vring, sizeof(float) * (size_t)fft_len * s.R));
    RCF_HIP(hipMalloc(&s.d_sum, sizeof(float) * (size_t)fft_len));
    RCF_HIP(hipMalloc(&s.d_out, sizeof(float) * (size_t)fft_len));
    RCF_HIP(hipMemsetAsync(s.d_sum, 0, sizeof(float) * (size_t)fft_len, h->stream));
    RCF_HIP(hipMemsetAsync(s.d_out, 0, sizeof(float) * (size_t)fft_len, h->stream));
    if (fft_len > 16384) RCF_HIP(hipMalloc(&s.d_scratch, sizeof(float2) * (size_t)fft_len * s.chunk));
    s.start_sample = h->total_in;
    s.armed = true;
    return RCF_OK;
}

int rcf_scan_frames_done(rcf_t *h)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    return h->scan.armed ? h->scan.frames_done : RCF_ESTATE;
}

int rcf_scan_result(rcf_t *h, float *out)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Scan &s = h->scan;
    if (!s.armed) { set_error("scan not armed"); return RCF_ESTATE; }
    if (!s.done) return RCF_EAGAIN;
    RCF_HIP(hipMemcpyAsync(out, s.d_out, sizeof(float) * (size_t)s.N, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    return RCF_OK;
}

int rcf_scan_result_device(rcf_t *h, void **dev_spectrum)
{
    if (!h || !dev_spectrum) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->scan.armed) return RCF_ESTATE;
    if (!h->scan.done) return RCF_EAGAIN;
    *dev_spectrum = h->scan.d_out;
    return RCF_OK;
}

int rcf_find_peaks(const float *spectrum, int64_t n, double min_w, double max_w, double prominence, int64_t *idx,
                   int64_t cap, int64_t *count, double *mean_out)
{
    if (!spectrum || n < 0 || cap < 0 || (cap > 0 && !idx)) { set_error("bad find_peaks arguments"); return RCF_EINVAL; }
    const int64_t c = find_peaks_host(spectrum, n, min_w, max_w, prominence, idx, cap, mean_out);
    if (count) *count = c;
    return RCF_OK;
}

int64_t rcf_peak_frequency(int64_t line, double samp_rate, int64_t fft_len, double center_freq)
{
    const double hz_per_bin = samp_rate / (double)fft_len;
    return (int64_t)(((double)line * hz_per_bin) - (samp_rate / 2) + center_freq);
}

int rcf_scan_find_peaks(rcf_t *h, double prominence, int64_t *idx, int64_t cap, int64_t *count, double *mean_out,
                        void **dev_idx)
{
    if (!h || cap < 1) { set_error("bad arguments"); return RCF_EINVAL; }
    if (cap > 4096) cap = 4096;                       // device sort capacity
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Scan &s = h->scan;
    if (!s.armed) return RCF_ESTATE;
    if (!s.done) return RCF_EAGAIN;
    const int N = s.N;
    const double hz_per_bin = h->fs / N;              // fft_peak_detection.py:46-52
    if (!s.d_peak_ws) RCF_HIP(hipMalloc(&s.d_peak_ws, peaks_workspace_bytes(N)));
    if (!s.d_peaks) RCF_HIP(hipMalloc(&s.d_peaks, sizeof(int64_t) * 4096));
    int *d_count = nullptr;
    double *d_mean = nullptr;
    launch_find_peaks(s.d_out, N, 3000 / hz_per_bin, 30000 / hz_per_bin, prominence, s.d_peak_ws, s.d_peaks,
                      (int)cap, &d_count, &d_mean, h->stream);
    int c = 0;
    double mean = 0.0;
    RCF_HIP(hipMemcpyAsync(&c, d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipMemcpyAsync(&mean, d_mean, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (idx) RCF_HIP(hipMemcpyAsync(idx, s.d_peaks, sizeof(int64_t) * (size_t)cap, hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    if (count) *count = c;
    if (mean_out) *mean_out = mean;
    if (dev_idx) *dev_idx = s.d_peaks;                // sorted ascending, -1 padded to `cap`
    return RCF_OK;
}

}  // extern "C"
