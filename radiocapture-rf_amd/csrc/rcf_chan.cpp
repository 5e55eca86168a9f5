// rcf_chan.cpp -- channels: lifecycle (channel.channel / set_offset / destroy of /root/reference/rc_frontend/channel.py),
// ring reads, the P25 symbol filter, the analog voice chain, source shift.
#include "rcf_plan.h"

namespace rcfx {

bool source_range(rcf_t *h, int src, int64_t S0, int64_t S1, SrcRange *out)
{
    if (src < 0) {
        out->view.base = h->d_buf[h->cur];
        out->view.mask = ~0ull;
        out->view.origin = S0 - (int64_t)h->hist_cap;
        out->view.stride = 1;
        out->p0 = S0;
        out->p1 = S1;
        return true;
    }
    if (src >= RCF_SRC_PFB_BIN0) {
        if (!h->pfb.open) return false;
        const int bin = src - RCF_SRC_PFB_BIN0;
        if (h->pfb.frame_major) {                           // bins_ring[i NB + bin]
            out->view.base = h->pfb.d_bins + bin;
            out->view.stride = h->pfb.NB;
            out->view.tshift = 0;
        } else {                                            // bins_ring[(i >> 4) tile_pitch + 16 bin + (i & 15)]
            out->view.base = h->pfb.d_bins + ((size_t)bin << kPfbTileLog2);
            out->view.stride = pfb_tile_pitch(h->pfb.NB);
            out->view.tshift = kPfbTileLog2;
        }
        out->view.mask = h->ring_mask;
        out->view.origin = 0;
        out->p0 = h->pfb.produced_before;
        out->p1 = h->pfb.produced;
        return true;
    }
    return false;   // channel-sourced: resolved by the caller (needs per-commit bookkeeping)
}

// What GNU Radio's freq_xlating_fir_filter_ccc(D, h, f_k, fs) on bin k's frequency would have done differently from the
// bank's exact phases, per output: its rotator advances by a = float32(-float32(2 pi f_k / fs) * D) instead of
// -2 pi k D / NB.  The difference, folded into (-pi, pi]: a tap's Chan::extra_dangle.
double pfb_tap_gr_dangle(const rcf_t *h, int bin)
{
    const Pfb &p = h->pfb;
    const int ks = bin < p.NB / 2 ? bin : bin - p.NB;
    const double f_k = (double)ks * h->fs / p.NB;
    const float fwT0 = (float)(kTwoPi * f_k / h->fs);
    const float a = -fwT0 * (float)p.D;
    const long double exact = -2.0L * 3.14159265358979323846264338327950288L *
                              (long double)(((int64_t)ks * p.D) % p.NB) / (long double)p.NB;
    long double d = (long double)a - exact;
    d = remainderl(d, 2.0L * 3.14159265358979323846264338327950288L);
    return (double)d;
}

int upload_composite(rcf_t *h, Chan *c)
{
    std::vector<float> ct;
    float incr[2];
    // rcf_source_shift moves every signal of the source by -shift at baseband: the wideband channels' NCOs follow,
    // and so do the channels fed by filterbank bins (same Hz, at the bin rate)
    const bool shifted = c->src < 0 || c->src >= RCF_SRC_PFB_BIN0;
    design_composite(c->proto.data(), c->T, c->D, c->offset_hz + (shifted ? h->shift_hz : 0.0), c->src_rate,
                     ct, incr);
    const size_t slice = slice_round(sizeof(float2) * (size_t)c->T);
    float2 *fresh = static_cast<float2 *>(pool_get(h, slice));
    if (!fresh) return RCF_ENOMEM;
    if (!hip_ok(hipMemcpy(fresh, ct.data(), sizeof(float2) * (size_t)c->T, hipMemcpyHostToDevice), "hipMemcpy(taps)")) {
        h->pools[slice].free_.push_back(fresh);
        return RCF_EHIP;
    }
    bury(h, c->d_ctaps, slice);
    c->d_ctaps = fresh;
    c->taps_version = ++h->taps_clock;
    // GR iterates phase *= incr in float32; model it by the increment's actual angle and magnitude
    c->incr[0] = incr[0];
    c->incr[1] = incr[1];
    c->dangle = std::atan2((double)incr[1], (double)incr[0]) + c->extra_dangle;
    c->dlogmag = std::log(std::hypot((double)incr[0], (double)incr[1])) + c->extra_dlogmag;
    return RCF_OK;
}

int new_channel(rcf_t *h, int src, int D, const float *taps, int T, double offset_hz, int *chan_id)
{
    if (D < 1 || T < 1 || !taps || !chan_id) { set_error("bad channel arguments"); return RCF_EINVAL; }
    std::unique_ptr<Chan> c(new Chan);
    c->src = src;
    c->D = D;
    c->T = T;
    c->offset_hz = offset_hz;
    c->proto.assign(taps, taps + T);
    if (src < 0) {
        c->src_rate = h->fs;
        c->start_sample = h->total_in;
        c->depth = 0;
        if ((size_t)(T - 1 + D) > h->hist_cap) { set_error("history capacity %zu < T-1+D", h->hist_cap); return RCF_ECAP; }
    } else if (src >= RCF_SRC_PFB_BIN0) {
        if (!h->pfb.open || src - RCF_SRC_PFB_BIN0 >= h->pfb.NB) { set_error("no such PFB bin"); return RCF_EINVAL; }
        c->src_rate = h->fs / h->pfb.D;
        c->start_sample = h->pfb.produced;
        c->depth = 1;
    } else {
        auto it = h->chans.find(src);
        if (it == h->chans.end()) { set_error("no such source channel %d", src); return RCF_ENOCHAN; }
        if (it->second->fm_only) { set_error("channel %d exposes its discriminator only: no IQ stream to chain on", src); return RCF_ESTATE; }
        c->src_rate = it->second->src_rate / it->second->D;
        c->start_sample = it->second->produced;
        c->depth = it->second->depth + 1;
    }
    if (src >= 0 && (size_t)(T + D) * 2 > h->out_cap) { set_error("source ring too small for T=%d", T); return RCF_ECAP; }
    c->k_abs0 = ceil_div(c->start_sample, D);
    // iq ring + discriminator ring in one slice.  Not cleared: readers never go past `produced`, and every
    // kernel masks what lies before a channel's first output (GR zero history)
    const size_t ring_slice = slice_round(12 * h->out_cap);
    c->d_iq = static_cast<float2 *>(pool_get(h, ring_slice));
    if (!c->d_iq) return RCF_ENOMEM;
    c->d_fm = reinterpret_cast<float *>(c->d_iq + h->out_cap);
    if (h->exact_rot) {
        const size_t rot_slice = slice_round(sizeof(float2) * h->out_cap + 256);
        c->d_rot = static_cast<float2 *>(pool_get(h, rot_slice));
        if (!c->d_rot) { h->pools[ring_slice].free_.push_back(c->d_iq); return RCF_ENOMEM; }
        const float st0[4] = {1.0f, 0.0f, 0.0f, 0.0f};       // phase 1 + 0j, counter 0 (bit pattern of 0.0f)
        if (!hip_ok(hipMemcpy(c->d_rot + h->out_cap, st0, sizeof(st0), hipMemcpyHostToDevice), "hipMemcpy(rotator state)")) {
            h->pools[rot_slice].free_.push_back(c->d_rot);
            h->pools[ring_slice].free_.push_back(c->d_iq);
            return RCF_EHIP;
        }
    }
    int rc = upload_composite(h, c.get());
    if (rc != RCF_OK) {
        h->pools[ring_slice].free_.push_back(c->d_iq);      // never seen by the stream: straight back
        if (c->d_rot) h->pools[slice_round(sizeof(float2) * h->out_cap + 256)].free_.push_back(c->d_rot);
        return rc;
    }
    c->id = h->next_id++;
    *chan_id = c->id;
    h->chans[c->id] = std::move(c);
    ++h->chans_epoch;
    return RCF_OK;
}

void free_channel(rcf_t *h, Chan *c)
{
    bury(h, c->d_ctaps, slice_round(sizeof(float2) * (size_t)c->T));
    bury(h, c->d_iq, slice_round(12 * h->out_cap));       // d_fm lives in the same slice
    bury(h, c->d_rot, slice_round(sizeof(float2) * h->out_cap + 256));
    c->d_rot = nullptr;
    bury(h, c->d_sym);
    bury(h, c->d_symtaps);
    if (c->audio) { bury(h, c->audio->d_state); bury(h, c->audio->d_rings); bury(h, c->audio->d_taps); c->audio.reset(); }
    c->d_sym = nullptr;
    c->d_symtaps = nullptr;
    c->d_ctaps = nullptr;
    c->d_iq = nullptr;
    c->d_fm = nullptr;
    (void)h;
}

// queue the copies of one ring read on the handle's stream; the caller synchronises and then advances *cursor by the
// count returned (ring_read does both for a single ring; rcf_chan_read_many batches many rings behind ONE sync)
int64_t ring_read_enqueue(rcf_t *h, const void *ring, size_t elem, int64_t produced, int64_t *cursor, void *out,
                          size_t max_items)
{
    int64_t avail = produced - *cursor;
    if (avail <= 0 || max_items == 0) return 0;
    if ((size_t)avail > h->out_cap) {           // reader lagged: oldest samples are gone
        *cursor = produced - (int64_t)h->out_cap;
        avail = (int64_t)h->out_cap;
    }
    const int64_t n = std::min<int64_t>(avail, (int64_t)max_items);
    const size_t pos = (size_t)((uint64_t)*cursor & h->ring_mask);
    const size_t first = std::min<size_t>((size_t)n, h->out_cap - pos);
    const unsigned char *r = static_cast<const unsigned char *>(ring);
    if (hipMemcpyAsync(out, r + pos * elem, first * elem, hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
        set_error("ring read failed");
        return RCF_EHIP;
    }
    if ((size_t)n > first &&
        hipMemcpyAsync(static_cast<unsigned char *>(out) + first * elem, r, ((size_t)n - first) * elem,
                       hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
        set_error("ring read failed");
        return RCF_EHIP;
    }
    return n;
}

int64_t ring_read(rcf_t *h, const void *ring, size_t elem, int64_t produced, int64_t *cursor, void *out,
                  size_t max_items)
{
    const int64_t n = ring_read_enqueue(h, ring, elem, produced, cursor, out, max_items);
    if (n <= 0) return n;
    if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("stream sync failed"); return RCF_EHIP; }
    free_graveyard_idle(h);       // retuned / closed channels' old buffers: every read is a chance to release them
    *cursor += n;
    return n;
}

}  // namespace rcfx

using namespace rcfx;

// =================================================================== C ABI
extern "C" {

// ------------------------------------------------------------------ channels
int rcf_chan_open(rcf_t *h, int channel_rate, double offset_hz, int *chan_id)
{
    if (!h || !chan_id) { set_error("bad channel arguments"); return RCF_EINVAL; }
    int D = 0, T = 0;
    int rc = rcf_channel_params_ex(h->fs, channel_rate, h->decim_rule, &D, &T, nullptr);
    if (rc != RCF_OK) return rc;
    if (!(std::fabs(offset_hz) < h->fs / 2)) { set_error("offset %g Hz outside +-fs/2", offset_hz); return RCF_ERANGE; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    std::vector<float> &taps = h->proto_cache[channel_rate];
    if (taps.empty())
        taps = design_low_pass_2(1.0, h->fs, channel_rate / 2.0, channel_rate / 2.0, 20.0, RCF_WIN_HAMMING);
    return new_channel(h, -1, D, taps.data(), (int)taps.size(), offset_hz, chan_id);
}

int rcf_chan_open_taps(rcf_t *h, int src_chan, int decim, const float *taps, int ntaps, double offset_hz,
                       int *chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    return new_channel(h, src_chan < 0 ? -1 : src_chan, decim, taps, ntaps, offset_hz, chan_id);
}

int rcf_pfb_chan_open(rcf_t *h, int bin, int channel_rate, double delta_hz, int *chan_id)
{
    if (!h || !chan_id) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    if (!h->pfb.open || bin < 0 || bin >= h->pfb.NB) { set_error("no such PFB bin %d", bin); return RCF_EINVAL; }
    const double rate = h->fs / h->pfb.D;
    int D = 0, T = 0;
    int rc = rcf_channel_params_ex(rate, channel_rate, h->decim_rule, &D, &T, nullptr);
    if (rc != RCF_OK) return rc;
    std::vector<float> taps = design_low_pass_2(1.0, rate, channel_rate / 2.0, channel_rate / 2.0, 20.0,
                                                RCF_WIN_HAMMING);
    return new_channel(h, RCF_SRC_PFB_BIN0 + bin, D, taps.data(), (int)taps.size(), delta_hz, chan_id);
}

int rcf_pfb_tap_open(rcf_t *h, int bin, int gr_phase, int *chan_id)
{
    if (!h || !chan_id) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    Pfb &p = h->pfb;
    if (!p.open || bin < 0 || bin >= p.NB) { set_error("no such PFB bin %d", bin); return RCF_EINVAL; }
    const float one = 1.0f;
    int rc = new_channel(h, RCF_SRC_PFB_BIN0 + bin, 1, &one, 1, 0.0, chan_id);
    if (rc != RCF_OK) return rc;
    h->chans[*chan_id]->is_tap = p.frame_major;     // power-of-two banks: an ordinary D = 1, T = 1 channel on the bin's ring
    ++h->chans_epoch;                                // (the planning summary counts taps and FIR channels differently)
    if (gr_phase) {
        // What GNU Radio's freq_xlating_fir_filter_ccc(D, h, f_k, fs) would have done differently from the bank's
        // exact phases: its rotator advances by a = float32(-float32(2 pi f_k / fs) * D) per output instead of
        // -2 pi k D / NB, and the float32 increment (cosf a, sinf a) is not exactly of unit length.  Both are
        // per-output factors: this channel's own rotator carries them.
        Chan *c = h->chans[*chan_id].get();
        const int ks = bin < p.NB / 2 ? bin : bin - p.NB;
        const double f_k = (double)ks * h->fs / p.NB;
        const float fwT0 = (float)(kTwoPi * f_k / h->fs);
        const float a = -fwT0 * (float)p.D;
        c->extra_dangle = pfb_tap_gr_dangle(h, bin);
        c->extra_dlogmag = std::log(std::hypot((double)std::cos(a), (double)std::sin(a)));
        // ... and GNU Radio's float32 tap phases float32(i * fwT0) differ from the bank's 2 pi k i / NB by a constant
        // (their filter-weighted mean, up to ~3e-4 rad) plus rounding noise (rcf_pfb_tap_leakage): the constant is a
        // rotation of the whole output and goes into the rotator's start phase
        double cphase = 0.0;
        design_tap_leakage(h->fs, p.NB, p.proto.data(), (int)p.proto.size(), bin, nullptr, &cphase);
        // ... and GNU Radio's rotator stands at 1 when the channel emits its FIRST output, whereas the bank's bin carries
        // e^{-j 2 pi k D n / NB} counted from the stream's first sample: a channel that starts at the bank's frame n0
        // is the bin times e^{+j 2 pi k D n0 / NB} (exact: integers mod NB; a sign for the 12.5 kHz grid's OS = 2 banks)
        const int64_t n0 = p.n_abs0 + c->k_abs0;
        const int64_t kd = (((int64_t)ks * p.D) % p.NB + p.NB) % p.NB;
        const int64_t turn = (kd * (n0 % p.NB)) % p.NB;
        c->angle0 = (long double)cphase +
                    remainderl(2.0L * 3.14159265358979323846264338327950288L * (long double)turn / (long double)p.NB,
                               2.0L * 3.14159265358979323846264338327950288L);
        rc = upload_composite(h, c);
    }
    {
        // plan_channel never iterates the exact rotator for a frame-major bank's tap or a channel that carries GNU
        // Radio's phase corrections: give the phase ring (8 out_cap bytes, half a megabyte at 2^16) back -- a
        // receiver in 'pfb' mode opens hundreds of these.  Nothing has been queued on it yet.
        Chan *c = h->chans[*chan_id].get();
        if (c->d_rot && (c->is_tap || c->extra_dangle != 0.0 || c->extra_dlogmag != 0.0)) {
            h->pools[slice_round(sizeof(float2) * h->out_cap + 256)].free_.push_back(c->d_rot);
            c->d_rot = nullptr;
        }
    }
    return rc;
}

#define FIND_CHAN(h, id, c)                                                 \
    auto it_ = (h)->chans.find(id);                                         \
    if (it_ == (h)->chans.end()) { set_error("no such channel %d", id); return RCF_ENOCHAN; } \
    Chan *c = it_->second.get()

int rcf_chan_set_offset(rcf_t *h, int chan_id, double offset_hz)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    c->offset_hz = offset_hz;
    return upload_composite(h, c);
}

int rcf_chan_close(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    free_channel(h, c);
    h->chans.erase(chan_id);
    ++h->chans_epoch;
    return RCF_OK;
}

int rcf_chan_info(rcf_t *h, int chan_id, int *decim, int *ntaps, double *out_rate, double *offset_hz)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    FIND_CHAN(h, chan_id, c);
    if (decim) *decim = c->D;
    if (ntaps) *ntaps = c->T;
    if (out_rate) *out_rate = c->src_rate / c->D;
    if (offset_hz) *offset_hz = c->offset_hz;
    return RCF_OK;
}

int64_t rcf_chan_produced(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;                  // (the count includes the deferred stage-2 block: it is queued first)
    FIND_CHAN(h, chan_id, c);
    return c->produced;
}

int64_t rcf_chan_start(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    FIND_CHAN(h, chan_id, c);
    return c->start_sample;
}

int64_t rcf_chan_read_iq(rcf_t *h, int chan_id, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (c->fm_only) { set_error("channel %d exposes its discriminator only (rcf_chan_set_fm_only)", chan_id); return RCF_ESTATE; }
    return ring_read(h, c->d_iq, sizeof(float2), c->produced, &c->rd_iq, out, max_samples);
}

int64_t rcf_chan_read_fm(rcf_t *h, int chan_id, float gain, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    const int64_t n = ring_read(h, c->d_fm, sizeof(float), c->produced, &c->rd_fm, out, max_samples);
    // quadrature_demod_cf: out = gain * fast_atan2f(...), one float32 multiply per sample
    for (int64_t i = 0; i < n; ++i) out[i] = gain * out[i];
    return n;
}

int rcf_chan_read_many(rcf_t *h, int what, const int *chan_ids, int n_chans, float gain, void *out, size_t cap_each,
                       int64_t *counts)
{
    if (!h || !chan_ids || !out || !counts || n_chans < 0 || (what != RCF_READ_IQ && what != RCF_READ_FM)) {
        set_error("bad batched read arguments");
        return RCF_EINVAL;
    }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    const size_t elem = what == RCF_READ_IQ ? sizeof(float2) : sizeof(float);
    const uint32_t ew = (uint32_t)(elem / 4);
    // what every channel has to give, and where its reader stands
    struct Item { Chan *c; int64_t *cur; const void *ring; int64_t n; size_t pos; };
    std::vector<Item> items((size_t)n_chans);
    size_t total = 0;
    uint32_t max_w = 0;
    const uint64_t stamp = ++h->many_stamp;
    for (int i = 0; i < n_chans; ++i) {
        Item &it = items[(size_t)i];
        it = Item{nullptr, nullptr, nullptr, 0, 0};
        auto f = h->chans.find(chan_ids[i]);
        if (f == h->chans.end()) { counts[i] = RCF_ENOCHAN; continue; }
        Chan *c = f->second.get();
        if (c->many_stamp == stamp) { counts[i] = RCF_EINVAL; continue; }   // listed twice: one reader position per channel
        c->many_stamp = stamp;
        if (what == RCF_READ_IQ && c->fm_only) { counts[i] = RCF_ESTATE; continue; }   // discriminator only
        it.c = c;
        it.cur = what == RCF_READ_IQ ? &c->rd_iq : &c->rd_fm;
        it.ring = what == RCF_READ_IQ ? (const void *)c->d_iq : (const void *)c->d_fm;
        int64_t avail = c->produced - *it.cur;
        if (avail > 0 && (size_t)avail > h->out_cap) {          // reader lagged: oldest samples are gone
            *it.cur = c->produced - (int64_t)h->out_cap;
            avail = (int64_t)h->out_cap;
        }
        it.n = avail <= 0 ? 0 : std::min<int64_t>(avail, (int64_t)cap_each);
        it.pos = (size_t)((uint64_t)*it.cur & h->ring_mask);
        counts[i] = it.n;
        total += (size_t)it.n;
        max_w = std::max<uint32_t>(max_w, (uint32_t)it.n * ew);
    }
    if (total == 0) return RCF_OK;
    // One gather launch packs every ring segment back to back into pinned host memory, one synchronisation, then the
    // rows are handed out.  (A device round trip per channel -- rcf_chan_read_iq in a loop -- costs ~10 us each: 256
    // tapped bins of ten front-ends are 25 ms per pass.)
    const size_t rec_bytes = ((size_t)n_chans * sizeof(GatherRec) + 255) & ~(size_t)255;
    const size_t need = rec_bytes + total * elem;
    if (need > h->many_cap && (uint64_t)total * ew <= 0xffffffffull) {
        if (h->h_many) { (void)hipStreamSynchronize(h->stream); (void)hipHostFree(h->h_many); h->h_many = nullptr; h->many_cap = 0; }
        size_t cap = 1 << 16;
        while (cap < need) cap <<= 1;
        void *p = nullptr, *dv = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer(&dv, p, 0) == hipSuccess) {
            h->h_many = static_cast<unsigned char *>(p);
            h->h_many_dev = static_cast<unsigned char *>(dv);
            h->many_cap = cap;
        } else if (p) {
            (void)hipHostFree(p);
        }
    }
    if (h->h_many && need <= h->many_cap && (uint64_t)total * ew <= 0xffffffffull) {     // (GatherRec counts 32-bit words)
        GatherRec *recs = reinterpret_cast<GatherRec *>(h->h_many);
        uint32_t at_w = 0;
        int n_recs = 0;
        for (int i = 0; i < n_chans; ++i) {
            const Item &it = items[(size_t)i];
            if (it.n <= 0) continue;
            recs[n_recs++] = GatherRec{static_cast<const uint32_t *>(it.ring), (uint32_t)(it.pos * ew), (uint32_t)it.n * ew,
                                       (uint32_t)(h->out_cap * ew - 1), at_w, 0u, ~0u, 1.0f, 0u};
            at_w += (uint32_t)it.n * ew;
        }
        launch_gather_rings(reinterpret_cast<const GatherRec *>(h->h_many_dev), n_recs,
                            reinterpret_cast<uint32_t *>(h->h_many_dev + rec_bytes), max_w, h->stream);
        if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("stream sync failed"); return RCF_EHIP; }
        const unsigned char *src = h->h_many + rec_bytes;
        for (int i = 0; i < n_chans; ++i) {
            const Item &it = items[(size_t)i];
            if (it.n <= 0) continue;
            std::memcpy(static_cast<unsigned char *>(out) + (size_t)i * cap_each * elem, src, (size_t)it.n * elem);
            src += (size_t)it.n * elem;
        }
    } else {
        // no mapped pinned memory: ring by ring, still behind one synchronisation
        for (int i = 0; i < n_chans; ++i) {
            const Item &it = items[(size_t)i];
            if (it.n <= 0) continue;
            int64_t cur = *it.cur;
            const int64_t n = ring_read_enqueue(h, it.ring, elem, it.c->produced, &cur,
                                                static_cast<unsigned char *>(out) + (size_t)i * cap_each * elem, (size_t)it.n);
            if (n < 0) { (void)hipStreamSynchronize(h->stream); return (int)n; }
        }
        if (hipStreamSynchronize(h->stream) != hipSuccess) { set_error("stream sync failed"); return RCF_EHIP; }
    }
    free_graveyard_idle(h);
    for (int i = 0; i < n_chans; ++i) {
        const Item &it = items[(size_t)i];
        if (it.n <= 0) continue;
        *it.cur += it.n;
        if (what == RCF_READ_FM) {
            float *o = static_cast<float *>(out) + (size_t)i * cap_each;
            for (int64_t k = 0; k < it.n; ++k) o[k] = gain * o[k];
        }
    }
    return RCF_OK;
}

int rcf_chan_fm_filter(rcf_t *h, int chan_id, float gain, const float *taps, int ntaps)
{
    if (!h || !taps || ntaps < 1 || ntaps > 4096) { set_error("bad fm filter arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if ((size_t)ntaps * 2 > h->out_cap) { set_error("ring too small for %d taps", ntaps); return RCF_ECAP; }
    float *fresh = nullptr;
    RCF_HIP(hipMalloc(&fresh, sizeof(float) * (size_t)ntaps));
    RCF_HIP(hipMemcpy(fresh, taps, sizeof(float) * (size_t)ntaps, hipMemcpyHostToDevice));
    bury(h, c->d_symtaps);
    c->d_symtaps = fresh;
    c->sym_ntaps = ntaps;
    c->sym_gain = gain;
    ++h->chans_epoch;
    if (!c->d_sym) {
        RCF_HIP(hipMalloc(&c->d_sym, sizeof(float) * h->out_cap));
        RCF_HIP(hipMemsetAsync(c->d_sym, 0, sizeof(float) * h->out_cap, h->stream));
        c->sym_from = c->produced;      // a new GR block starts with zero history
        c->rd_sym = c->produced;
    }
    return RCF_OK;
}

int64_t rcf_chan_read_sym(rcf_t *h, int chan_id, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (!c->d_sym) { set_error("channel %d has no fm filter", chan_id); return RCF_ESTATE; }
    return ring_read(h, c->d_sym, sizeof(float), c->produced, &c->rd_sym, out, max_samples);
}

int rcf_chan_audio_open(rcf_t *h, int chan_id, const rcf_audio_params_t *p)
{
    if (!h || !p || !p->lpf_taps || !p->hpf_taps || !p->rs_taps || p->n_lpf < 1 || p->n_hpf < 1 || p->n_rs < 1 ||
        p->interpolation < 1 || p->decimation < 1 || p->deemph_a[0] == 0.0) {
        set_error("bad audio chain arguments");
        return RCF_EINVAL;
    }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (c->fm_only) { set_error("channel %d exposes its discriminator only: the voice chain reads IQ", chan_id); return RCF_ESTATE; }
    const int I = p->interpolation;
    const int n_rs_pad = (p->n_rs + I - 1) / I * I;              // rational_resampler_base: pad to a multiple of I
    const size_t reach = (size_t)std::max(std::max(p->n_lpf, p->n_hpf), n_rs_pad / I);
    if (reach * 2 > h->out_cap) { set_error("ring of %zu too small for %zu-tap audio filters", h->out_cap, reach); return RCF_ECAP; }
    std::unique_ptr<Chan::Audio> au(new Chan::Audio);
    au->n_lpf = p->n_lpf; au->n_hpf = p->n_hpf; au->nt_rs = n_rs_pad / I;
    au->interp = I; au->decim = p->decimation;
    au->gain = p->quad_gain;
    au->thr = std::pow(10.0, p->squelch_db / 10);                // pwr_squelch_cc::set_threshold
    au->alpha = p->squelch_alpha;
    // iir_filter(fftaps, fbtaps, oldstyle = false): feedback taps are negated, a[0] must be 1
    au->b0 = p->deemph_b[0]; au->b1 = p->deemph_b[1]; au->fb1 = -p->deemph_a[1];
    std::vector<float> taps((size_t)p->n_lpf + p->n_hpf + n_rs_pad, 0.0f);
    std::memcpy(taps.data(), p->lpf_taps, sizeof(float) * (size_t)p->n_lpf);
    std::memcpy(taps.data() + p->n_lpf, p->hpf_taps, sizeof(float) * (size_t)p->n_hpf);
    std::memcpy(taps.data() + p->n_lpf + p->n_hpf, p->rs_taps, sizeof(float) * (size_t)p->n_rs);
    RCF_HIP(hipMalloc(&au->d_taps, sizeof(float) * taps.size()));
    RCF_HIP(hipMemcpy(au->d_taps, taps.data(), sizeof(float) * taps.size(), hipMemcpyHostToDevice));
    RCF_HIP(hipMalloc(&au->d_rings, sizeof(float) * 6 * h->out_cap));
    RCF_HIP(hipMemsetAsync(au->d_rings, 0, sizeof(float) * 6 * h->out_cap, h->stream));
    AudioState st0{};
    st0.muted = 1;                                               // squelch_base_cc starts in ST_MUTED
    RCF_HIP(hipMalloc(&au->d_state, sizeof(AudioState)));
    RCF_HIP(hipMemcpy(au->d_state, &st0, sizeof(st0), hipMemcpyHostToDevice));
    au->from = c->produced;                                      // a new flowgraph: zero state from here on
    if (c->audio) { bury(h, c->audio->d_state); bury(h, c->audio->d_rings); bury(h, c->audio->d_taps); }
    c->audio = std::move(au);
    ++h->chans_epoch;
    return RCF_OK;
}

int rcf_chan_audio_close(rcf_t *h, int chan_id)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (c->audio) { bury(h, c->audio->d_state); bury(h, c->audio->d_rings); bury(h, c->audio->d_taps); c->audio.reset(); }
    ++h->chans_epoch;
    return RCF_OK;
}

static int audio_counts(rcf_t *h, Chan *c, int64_t *n_audio, int64_t *n_ungated)
{
    AudioState st{};
    RCF_HIP(hipMemcpyAsync(&st, c->audio->d_state, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    const int64_t I = c->audio->interp, D = c->audio->decim;
    *n_ungated = st.n_a;
    *n_audio = (st.n_a * I + D - 1) / D;
    return RCF_OK;
}

int rcf_chan_audio_produced(rcf_t *h, int chan_id, int64_t *n_audio, int64_t *n_ungated)
{
    if (!h || !n_audio) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (!c->audio) { set_error("channel %d has no audio chain", chan_id); return RCF_ESTATE; }
    int64_t a = 0, u = 0;
    const int rc = audio_counts(h, c, &a, &u);
    if (rc != RCF_OK) return rc;
    *n_audio = a;
    if (n_ungated) *n_ungated = u;
    return RCF_OK;
}

int64_t rcf_chan_read_audio(rcf_t *h, int chan_id, float *out, size_t max_samples)
{
    if (!h || !out) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (!c->audio) { set_error("channel %d has no audio chain", chan_id); return RCF_ESTATE; }
    int64_t a = 0, u = 0;
    const int rc = audio_counts(h, c, &a, &u);
    if (rc != RCF_OK) return rc;
    return ring_read(h, c->audio->d_rings + 3 * h->out_cap, sizeof(float), a, &c->audio->rd, out, max_samples);
}

int rcf_chan_fm_level(rcf_t *h, int chan_id, float gain, int window, float *level)
{
    if (!h || !level || window < 1) { set_error("bad fm level arguments"); return RCF_EINVAL; }
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if ((size_t)window > h->out_cap) { set_error("window %d exceeds the ring", window); return RCF_ECAP; }
    if (!h->d_level) RCF_HIP(hipMalloc(&h->d_level, sizeof(float)));
    launch_fm_level(c->d_fm, c->produced, window, gain, h->ring_mask, h->d_level, h->stream);
    RCF_HIP(hipMemcpyAsync(level, h->d_level, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    RCF_HIP(hipStreamSynchronize(h->stream));
    return RCF_OK;
}

int rcf_chan_set_fm_only(rcf_t *h, int chan_id, int on)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    FIND_CHAN(h, chan_id, c);
    if (!c->is_tap) { set_error("channel %d is not a tap of a frame-major filterbank", chan_id); return RCF_EINVAL; }
    if (on) {
        if (c->audio) { set_error("channel %d carries a voice chain, which reads its IQ stream", chan_id); return RCF_ESTATE; }
        for (auto &kv : h->chans)
            if (kv.second->src == chan_id) { set_error("channel %d reads channel %d's IQ stream", kv.first, chan_id); return RCF_ESTATE; }
    } else if (c->fm_only) {
        c->rd_iq = c->produced;                  // what lies behind was never written
    }
    if ((on != 0) != c->fm_only && c->produced > 0) {
        // The newest ring sample is the "output before" of the next block's first discriminator sample, and the two modes
        // keep it differently: rotated by the channel's rotator (ordinary tap) or as the bare bin (discriminator only: that
        // path never rotates, it turns the conjugate product by the rotator's increment instead).  Convert the one sample,
        // so that no discriminator sample straddles two conventions.  Only its angle matters to the discriminator.
        RCF_HIP(hipStreamSynchronize(h->stream));
        float2 *at = c->d_iq + ((uint64_t)(c->produced - 1) & h->ring_mask);
        float2 v;
        RCF_HIP(hipMemcpy(&v, at, sizeof(v), hipMemcpyDeviceToHost));
        const long double ang = c->angle0 + (long double)(c->produced - 1 - c->n_seg0) * (long double)c->dangle;
        const double pr = std::cos((double)ang), pi = (on ? -1.0 : 1.0) * std::sin((double)ang);
        const float2 w = make_float2((float)(v.x * pr - v.y * pi), (float)(v.x * pi + v.y * pr));
        RCF_HIP(hipMemcpy(at, &w, sizeof(w), hipMemcpyHostToDevice));
    }
    c->fm_only = on != 0;
    return RCF_OK;
}

int rcf_chan_rings(rcf_t *h, int chan_id, void **iq_ring, void **fm_ring, size_t *capacity)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;                  // (zero-copy readers order themselves on rcf_stream: nothing stays deferred)
    FIND_CHAN(h, chan_id, c);
    if (iq_ring && c->fm_only) { set_error("channel %d exposes its discriminator only (rcf_chan_set_fm_only)", chan_id); return RCF_ESTATE; }
    if (iq_ring) *iq_ring = c->d_iq;
    if (fm_ring) *fm_ring = c->d_fm;
    if (capacity) *capacity = h->out_cap;
    return RCF_OK;
}

int rcf_source_shift(rcf_t *h, double delta_hz)
{
    if (!h) return RCF_EINVAL;
    std::lock_guard<std::mutex> g(h->mu);
    if (set_dev(h)) return RCF_EHIP;
    h->shift_hz += delta_hz;
    for (auto &kv : h->chans)
        if (kv.second->src < 0 || kv.second->src >= RCF_SRC_PFB_BIN0) {
            int rc = upload_composite(h, kv.second.get());
            if (rc != RCF_OK) return rc;
        }
    if (h->pfb.open && h->pfb.d_fm_inc) return pfb_fm_upload_increments(h);     // the bank's own discriminator follows too
    return RCF_OK;
}

}  // extern "C"
