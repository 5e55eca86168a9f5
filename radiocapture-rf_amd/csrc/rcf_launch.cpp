// rcf_launch.cpp -- one front-end's block: upload the launch records, launch in dependency order, scan, history.
#include "rcf_plan.h"

namespace rcfx {

void flush_lagged(rcf_t *h)
{
    if (!h->lag.pending) return;
    h->lag.pending = false;
    Timed t(h, RCF_T_FIR_DERIVED);
    launch_fir_bank(h->lag.dev, h->lag.dims, h->stream);
}

// upload all launch parameters in one copy, then launch in dependency order
int launch_plan(rcf_t *h, BlockPlan &bp)
{
    hipStream_t st = h->stream;
    Arena &ar = *bp.ar;
    const int a = bp.a;
    const size_t arena_base = bp.arena_base;
    auto &fir_by_depth = bp.fir_by_depth;
    auto &disc_jobs = bp.disc_jobs;
    auto &symf = bp.symf;
    auto &audf = bp.audf;
    auto &rot_fills = bp.rot_fills;
    PfbLaunch &pl = bp.pl;
    const bool run_pfb = bp.run_pfb;
    const TapLaunch *d_tap_list = bp.d_tap_list;
    const RotFill *d_rot_fills = bp.d_rot_fills;
    const FmFirLaunch *d_symf = bp.d_symf;
    const AudioLaunch *d_audf = bp.d_audf;
    const int symf_max_n = bp.symf_max_n, audf_max_n = bp.audf_max_n, audf_num = bp.audf_num, audf_den = bp.audf_den;

    // ---- stage-2 lag.  The previous block's small-T launch, if it is still pending, rides in THIS block's filterbank launch
    // when that launch is the kernel that can carry it and the bank's ring has room for both blocks' frames; otherwise it
    // goes out now, ahead of everything of this block.
    const size_t pfb_reach = bp.reach(RCF_SRC_PFB_BIN0);
    const bool carry = run_pfb && pfb_can_carry_s2(pl);
    if (h->lag.pending && !(carry && (size_t)(pl.n_frames + h->lag.frames) + pfb_reach <= h->out_cap)) flush_lagged(h);
    S2Rider sr{};
    if (h->lag.pending) {
        const FirLaunchDims &ld = h->lag.dims;
        sr.chans = h->lag.dev;
        sr.atan_tab = ld.atan_tab;
        sr.ring_mask = ld.ring_mask;
        sr.D = ld.D; sr.T = ld.T; sr.KB = fir_small_outputs_rider(ld.D, ld.T);
        sr.n_chans = ld.n_chans;
        sr.n_tiles = (ld.max_n_k + sr.KB - 1) / sr.KB;
        sr.n_wgs = (sr.n_chans * sr.n_tiles + 7) & ~7;

        h->lag.pending = false;
    }
    // ... and this block's own: ONE small-T job on the bank's bins, nothing that consumes its outputs within the block
    FirJob *lag_job = nullptr;
    if (h->lag_enabled && carry && fir_by_depth.size() == 2 && fir_by_depth[1].size() == 1 && fir_by_depth[1][0].dims.small &&
        fir_by_depth[1][0].dev && fir_by_depth[1][0].bank_src && !d_symf && !d_audf && (size_t)pl.n_frames * 2 + pfb_reach <= h->out_cap)
        lag_job = &fir_by_depth[1][0];
    {
        // the block's launch records host -> device, and -- in the same launch -- its history tail behind the OTHER input
        // buffer's block (nothing in this block reads that place, and the kernels that did read it are earlier in
        // the stream): one small launch per block instead of two
        const size_t from = arena_base & ~size_t(63);
        const size_t bytes = ar.used > arena_base ? ((ar.used + 63) & ~size_t(63)) - from : 0;
        static const bool merge = [] { const char *e = getenv("RCF_COPY_MERGE"); return !e || atoi(e) != 0; }();   // A/B
        // ... or none at all: when the filterbank's launch is the first of the block that needs neither (no direct
        // channels, no exact-rotator fill before it, no tap matrix whose slot list the bank itself reads from the
        // arena), its first workgroups do both copies on the way in (PfbLaunch::rider_*)
        static const bool ride_env = [] { const char *e = getenv("RCF_COPY_RIDE"); return !e || atoi(e) != 0; }();    // A/B
        const bool ride = ride_env && merge && h->copy_kernels && run_pfb && !d_rot_fills &&
                          (fir_by_depth.empty() || fir_by_depth[0].empty()) && pl.n_taps == pl.tap_first &&
                          bytes / 8 < (1u << 31) && h->hist_cap < (1u << 28) && pfb_takes_rider(pl);
        if (ride) {
            pl.rider_dst[0] = reinterpret_cast<unsigned long long *>(ar.d + from);
            pl.rider_src[0] = reinterpret_cast<const unsigned long long *>(h->arenas.h_dev[a] + from);
            pl.rider_n8[0] = (uint32_t)((bytes + 7) / 8);
            pl.rider_dst[1] = reinterpret_cast<unsigned long long *>(h->d_buf[h->cur ^ 1]);
            pl.rider_src[1] = reinterpret_cast<const unsigned long long *>(h->d_buf[h->cur] + bp.n);
            pl.rider_n8[1] = (uint32_t)(sizeof(float2) * h->hist_cap / 8);
            bp.history_done = true;
        } else if (h->copy_kernels && !merge) {
            if (bytes) launch_copy8(ar.d + from, h->arenas.h_dev[a] + from, bytes, st);
        } else if (h->copy_kernels) {
            Timed t(h, RCF_T_HISTORY);
            launch_copy8x2(ar.d + from, h->arenas.h_dev[a] + from, bytes, h->d_buf[h->cur ^ 1], h->d_buf[h->cur] + bp.n,
                           sizeof(float2) * h->hist_cap, st);
            bp.history_done = true;
        } else if (bytes) {
            RCF_HIP(hipMemcpyAsync(ar.d + from, ar.h + from, bytes, hipMemcpyHostToDevice, st));
        }
        if (bytes) h->arenas.fill = (ar.used + 63) & ~size_t(63);
    }
    if (d_rot_fills) launch_rot_fill(d_rot_fills, (int)rot_fills.size(), h->ring_mask, st);
    if (!fir_by_depth.empty())
        for (auto &j : fir_by_depth[0]) {
            if (j.repack) {
                launch_fir_pack(j.dev, j.dims.n_chans, j.dims.T, const_cast<float *>(j.dims.bank), j.dirty, st);
                if (j.bc) j.bc->key = std::move(j.key);
            }
            Timed t(h, j.dims.mfma ? RCF_T_FIR_MFMA : RCF_T_FIR);
            launch_fir_bank(j.dev, j.dims, st);
        }
    if (run_pfb) { TimedAttached t(h, RCF_T_PFB, pl); launch_pfb(pl, st, sr.n_wgs ? &sr : nullptr); }
    if (run_pfb && pl.n_taps > 0) {
        Timed t(h, RCF_T_TAPS);
        launch_tap_finalize(d_tap_list, pl.n_taps, pl.tap_mat, pl.tap_pitch, pl.n_frames, pl.n_lo - pl.n_abs0,
                            h->ring_mask, h->d_atan, bp.d_group_bin0, pl.tap_first, pl.bins_ring, pl.NB, st);
    }
    for (size_t d = 1; d < fir_by_depth.size(); ++d)
        for (auto &j : fir_by_depth[d]) {
            if (&j == lag_job) {                            // not queued: it rides in the next block's filterbank launch
                h->lag.pending = true;
                h->lag.dims = j.dims;
                h->lag.dev = j.dev;
                h->lag.frames = pl.n_frames;
                continue;
            }
            Timed t(h, RCF_T_FIR_DERIVED);
            launch_fir_bank(j.dev, j.dims, st);
        }
    for (auto &dj : disc_jobs) {
        Timed t(h, RCF_T_DISC);
        launch_discriminator(dj.dev, dj.n, dj.max_n, h->ring_mask, h->d_atan, st);
    }
    if (d_symf) {
        Timed t(h, RCF_T_DISC);
        launch_fm_fir(d_symf, (int)symf.size(), symf_max_n, h->ring_mask, st);
    }
    if (d_audf) {
        Timed t(h, RCF_T_AUDIO);
        launch_audio(d_audf, (int)audf.size(), audf_max_n, audf_num, audf_den, h->ring_mask, h->d_atan, st);
    }
    return RCF_OK;
}

// the scan's share of the block: every frame that is complete now
int run_scan(rcf_t *h, const BlockPlan &bp)
{
    hipStream_t st = h->stream;
    const int64_t S0 = bp.S0, S1 = bp.S1;

    Scan &sc = h->scan;
    if (sc.armed && !sc.done) {
        int64_t avail = (S1 - sc.start_sample) / sc.N;
        if (avail > sc.n_frames) avail = sc.n_frames;
        while (sc.frames_done < avail) {
            const int cnt = (int)std::min<int64_t>(sc.chunk, avail - sc.frames_done);
            ScanLaunch sl{};
            sl.src.base = h->d_buf[h->cur];
            sl.src.mask = ~0ull;
            sl.src.origin = S0 - (int64_t)h->hist_cap;
            sl.src.stride = 1;
            sl.s0 = sc.start_sample + (int64_t)sc.frames_done * sc.N;
            sl.window = sc.d_window;
            sl.tw = sc.d_tw;
            sl.vring = sc.d_vring;
            sl.N = sc.N; sl.R = sc.R;
            sl.f0 = sc.frames_done; sl.n_frames = cnt;
            sl.scratch = sc.d_scratch;
            { Timed t(h, RCF_T_SCAN_FFT); launch_scan_fft(sl, st); }
            {
                Timed t(h, RCF_T_SCAN_MOVSUM);
                launch_scan_movsum(sc.d_vring, sc.N, sc.R, sc.L, sc.frames_done, cnt, sc.n_frames - 1, sc.d_sum,
                                   sc.d_out, st);
            }
            sc.frames_done += cnt;
        }
        if (sc.frames_done >= sc.n_frames) sc.done = true;
    }
    return RCF_OK;
}

// history for the next block, flip buffers
int finish_block(rcf_t *h, const BlockPlan &bp)
{
    hipStream_t st = h->stream;
    const size_t n = bp.n;
    const int64_t S1 = bp.S1;

    const int other = h->cur ^ 1;
    if (!bp.history_done) {
        Timed t(h, RCF_T_HISTORY);
        if (h->copy_kernels) launch_copy8(h->d_buf[other], h->d_buf[h->cur] + n, sizeof(float2) * h->hist_cap, st);
        else RCF_HIP(hipMemcpyAsync(h->d_buf[other], h->d_buf[h->cur] + n, sizeof(float2) * h->hist_cap,
                                    hipMemcpyDeviceToDevice, st));
    }
    if (h->eager_buf_done) {
        RCF_HIP(hipEventRecord(h->buf_done[h->cur], st));    // everything that reads this buffer is queued
        h->buf_done_set[h->cur] = true;
        h->buf_dirty[h->cur] = false;
    } else {
        h->buf_dirty[h->cur] = true;
    }
    h->cur = other;
    h->total_in = S1;
    RCF_HIP(hipGetLastError());
    return RCF_OK;
}

}  // namespace rcfx
