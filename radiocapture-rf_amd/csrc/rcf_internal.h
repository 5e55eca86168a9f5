// rcf_internal.h -- shared declarations between the C-ABI host layer and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/rcf.h"

namespace rcfx {

// ---------------------------------------------------------------- host design helpers (rcf_design.cpp)
void design_window(int type, int n, float *w, double beta = 6.76);
double design_max_attenuation(int window, double beta);
std::vector<float> design_firdes(int kind, double gain, double fs, double fc, double tw, int window, double beta);
void design_fm_deemph(double fs, double tau, double b[2], double a[2]);
std::vector<float> design_resampler(int interpolation, int decimation);
bool design_pm_remez(int numtaps, const std::vector<double> &bands, const std::vector<double> &des,
                     const std::vector<double> &weight, std::vector<double> &h);
bool design_optfir_low_pass(double gain, double fs, double f1, double f2, double ripple_db, double atten_db,
                            int nextra, std::vector<float> &taps);
int design_ntaps(double fs, double tw, double att_db);
std::vector<float> design_low_pass_2(double gain, double fs, double fc, double tw, double att_db, int window);
void design_composite(const float *taps, int T, int D, double f0, double fs,
                      std::vector<float> &ctaps_interleaved, float incr[2]);

void design_tap_leakage(double fs, int n_bins, const float *taps, int T, int bin, double *leak_l2, double *const_phase);

// ---------------------------------------------------------------- host peak picker (rcf_peaks.cpp)
int64_t find_peaks_host(const float *spectrum, int64_t n, double min_w, double max_w, double prominence,
                        int64_t *idx, int64_t cap, double *mean_out);

// ---------------------------------------------------------------- error plumbing
void set_error(const char *fmt, ...);
bool hip_ok(hipError_t e, const char *what);
#define RCF_HIP(call)                                 \
    do {                                              \
        if (!::rcfx::hip_ok((call), #call)) return RCF_EHIP; \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device): every launcher that needs more
// than 64 KB of dynamic LDS calls this with its own static DynLdsAttr (one per kernel instantiation), so that a
// second rcf_t on another GPU of the same process gets the attribute too and two handles' threads do not race
struct DynLdsAttr {
    std::mutex mu;
    size_t set[64] = {0};
    void ensure(const void *func, size_t lds)
    {
        if (lds <= 64 * 1024) return;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) dev = 0;
        std::lock_guard<std::mutex> g(mu);
        if (set[dev] >= lds) return;
        (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        set[dev] = lds;
    }
};

// ---------------------------------------------------------------- stream views
// sample s of a stream lives at base[at(i)], i = (s - origin) & mask, at(i) = (i >> tshift) * stride + (i & tmask):
//   linear buffers and channel rings      stride 1, tshift 0
//   one bin of a frame-major bank ring    stride n_bins, tshift 0       (pfb5.hip: bins_ring[i * NB + k])
//   one bin of a tiled bank ring          stride tile_pitch, tshift 4   (pfb.hip: bins_ring[(i >> 4) tile_pitch + 16 k + (i & 15)])
struct StreamView {
    const float2 *base;
    uint64_t mask;
    int64_t origin;
    int64_t stride;
    int32_t tshift;
    int32_t pad_;
    __host__ __device__ __forceinline__ uint64_t at(int64_t s) const
    {
        const uint64_t i = (uint64_t)(s - origin) & mask;
        return (i >> tshift) * (uint64_t)stride + (i & ((1ull << tshift) - 1));
    }
};
constexpr int kPfbTileLog2 = 4;          // frames per tile of a power-of-two bank's ring = pfb.hip's chunk (F = 16)
// A tile holds 16 frames of all NB bins (16 NB samples, contiguous for the writer).  Its pitch is NOT that power of two:
// a reader of ONE bin takes one 128-byte line per tile, and at a pitch of exactly 16 NB 8 bytes (32 KB for 256 bins)
// all of them would fall on the same one or two L2 / HBM channels.  Five lines of padding spread them.
inline int64_t pfb_tile_pitch(int NB) { return ((int64_t)NB << kPfbTileLog2) + 80; }

// ---------------------------------------------------------------- direct xlating-FIR bank
// One entry per channel per launch (device array).
struct ChanLaunch {
    const float2 *ctaps;     // T composite taps h[i] e^{j float32(i fwT0)}  (GR-faithful float32)
    float2 *iq_ring;         // output ring, index (k - k_abs0) & ring_mask
    StreamView src;          // input stream
    int64_t k_lo;            // first absolute output index produced by this launch
    int64_t k_abs0;          // absolute output index of the channel's first-ever output
    int64_t start_sample;    // samples before this index count as zero (GR zero history)
    int64_t n_seg0;          // relative output index at which (angle0, logmag0) hold
    double angle0, dangle;   // rotator phase angle model (radians)
    double logmag0, dlogmag; // rotator magnitude model (log |phase|)
    int32_t n_k;             // outputs to produce
    int32_t pad_;
    float *fm_ring;          // discriminator ring (written by the fused small-T kernel only)
    // exact rotator (rcf_set_rotator): GNU Radio's phase for output n = k - k_abs0, iterated in float32 by
    // rot_fill_kernel before the block's FIR launches; nullptr = the closed form above
    const float2 *rot_ring;
    uint64_t rot_mask;
    static constexpr bool kHasRotRing = true;
};
// one channel's share of rot_fill_kernel: phases of outputs [n_from, n_from + n_k) into the ring, state carried on
struct RotFill {
    float2 *ring;
    float *state;            // {phase.re, phase.im, call counter (uint32 bits)}: the phase the NEXT output gets
    int64_t n_from;
    int32_t n_k;
    float incr_re, incr_im;  // float32(cos a, sin a), a = float32(-fwT0 * D): what GNU Radio iterates
    int32_t pad_;
};
void launch_rot_fill(const RotFill *d_items, int n_items, uint64_t ring_mask, hipStream_t s);

// outputs per workgroup of the small-T kernel (tile of KB D + T samples within ~26 KB of LDS, two outputs per
// thread minus the recomputed predecessor); 0 = not applicable
inline int fir_small_outputs(int D, int T)
{
    if (T > 96) return 0;
    int kb = (3300 - T) / D;
    if (kb > 511) kb = 511;          // 512 slots (fir.hip kSmallPerThread = 2): one is the predecessor output the discriminator needs
    return kb >= 32 ? kb : 0;
}
// ... and of a RIDER workgroup (the same tile run as extra workgroups of the 256-bin filterbank launch, rcf_set_stage2_lag):
// there a tile's cost is the time it holds one of the launch's workgroup slots -- one memory round trip whatever its size --
// so the tile is as large as the filterbank workgroup's own 37120 bytes of LDS allow (four outputs per thread): the timed
// configuration's 32 x 86 tiles of 511 outputs become 32 x 43 of 1023, fused launch 114.5 -> 112.7 us (same box, three
// alternating runs).  As a launch of its own the large tile is SLOWER (18.6 -> 22.4 us): fir_small_kernel keeps 511.
constexpr int kSmallRiderPerThread = 4;
inline int fir_small_outputs_rider(int D, int T)
{
    const int base = fir_small_outputs(D, T);
    if (base == 0) return 0;
    // (KB D + T samples, KB + 1 outputs, T taps) x 8 bytes + the 264-float table <= 37120
    int kb = (4508 - 2 * T) / (D + 1);
    if (kb > 256 * kSmallRiderPerThread - 1) kb = 256 * kSmallRiderPerThread - 1;
    return kb > base ? kb : base;
}

// Matrix-core bank operand (fir_mfma_kernel, fir.hip): groups of 32 channels; the taps of a group are stored in
// PROCESSING order -- step p = 0 .. NS-1 handles tap pairs q = 4 (NS-1-p) + kap, pair q = taps (2q, 2q-1), i.e.
// the two samples x[kD-2q], x[kD-2q+1] one 16-byte load covers -- as four M-tiles of MFMA A operands:
//   bank2[((g * NS + p) * 4 + t) * 256 + lane * 4 + u],  lane = 16 kap + 2 c + r  (channel 32 g + 8 t + c, r: Re/Im y)
//   u = 0: x[kD-2q].re  1: x[kD-2q].im  2: x[kD-2q+1].re  3: x[kD-2q+1].im   times  [cr -ci; ci cr][r][re|im]
// zero for taps outside [0, T) and channels past the end.  Each group is its own slab (64-bit base), so a class
// has no size limit.
constexpr int kM2Group = 32;      // channels per group (4 M-tiles of 8)
constexpr int kM2ChunkSteps = 8;  // steps per LDS chunk (32 KB)
__host__ __device__ inline int bank2_steps(int T)
{
    const int s = ((T / 2 + 1) + 3) / 4;
    return (s + kM2ChunkSteps - 1) / kM2ChunkSteps * kM2ChunkSteps;
}
__host__ __device__ inline size_t bank2_group_floats(int T) { return (size_t)bank2_steps(T) * 1024; }
// buf_samples: hist_cap + block_cap of the wideband buffer -- the kernel addresses it with 32-bit byte offsets of one
// buffer descriptor, so it must stay under 2 GiB (larger handles keep the vector kernel, which indexes in 64 bits)
inline bool mfma2_applicable(int D, int T, size_t hist_cap, size_t buf_samples)
{
    return D >= 1 && T >= 64 && (size_t)(8 * bank2_steps(T) + 8 + D) <= hist_cap &&
           (uint64_t)buf_samples * sizeof(float2) < (1ull << 31);
}

// Work split of one matrix-core launch.  A workgroup is 4 waves x NT tiles of 16 outputs for one group of 32
// channels over a RANGE of the taps; 512 workgroups are resident (two per CU) and the dispatcher runs the grid in
// rounds of 512.  NT = 2 halves the operand traffic per MFMA; tap parts > 1 (split-K: every part writes its partial
// sums to its own slab, fir_mfma_finish_kernel adds the slabs in order and applies the rotator -- deterministic for
// any part count) shrink the unit when the whole launch is only a few units per CU: 256 channels x 5243 outputs are
// 2.56 units per CU, which unsplit run as two full rounds (0.59 of peak), in three parts as four rounds of thirds.
// The plan depends on the launch (channel count, outputs in the block): with parts > 1 a channel's taps are summed
// in `parts` float32 chains that are then added in order, so the LAST BITS of the bank's outputs depend on how the
// stream is cut into blocks (and freshly opened channels get their first outputs from the vector kernel's order).
// Deterministic for a given sequence of commits; cut-invariant to ~1e-6 relative (5e-6 on noise-only channels), not bit for bit
// (tests/test_gpu_round3.py::test_matrix_core_bank_uneven_cuts_agree_to_summation_order) -- unlike the filterbanks.
struct MfmaPlan { int nt, parts; };
inline MfmaPlan mfma_plan(int n_chans, int n_k, int T, int force_nt = 0, int force_parts = 0)
{
    const int groups = (n_chans + kM2Group - 1) / kM2Group;
    const int n_chunks = bank2_steps(T) / kM2ChunkSteps;
    MfmaPlan best{1, 1};
    double best_cost = 1e300;
    for (int nt = 1; nt <= 2; ++nt) {
        if (force_nt && nt != force_nt) continue;
        const int64_t wgs = (int64_t)groups * ((n_k + 64 * nt - 1) / (64 * nt));
        for (int parts = 1; parts <= 8 && parts <= n_chunks; ++parts) {
            if (force_parts && parts != force_parts) continue;
            const int64_t rounds = (wgs * parts + 511) / 512;
            // time in units of one NT = 1 full-K round; NT = 2 units are twice as long but ~6 % more efficient;
            // split launches pay the finishing pass (partials written + read) and a little per part
            double cost = (double)rounds * nt / parts * (nt == 2 ? 0.94 : 1.0);
            if (parts > 1) cost += 0.06 + 0.01 * parts;
            if (cost < best_cost - 1e-9) { best_cost = cost; best = MfmaPlan{nt, parts}; }
        }
    }
    return best;
}

struct FirLaunchDims {
    int D, T, KT;            // decimation, taps, outputs per workgroup tile
    int n_chans;             // entries in the ChanLaunch array
    int chans_per_wg;        // >1 only when every channel of the launch shares one source view
    int max_n_k;             // max over channels of n_k
    uint64_t ring_mask;
    int mfma;                // 1: every channel shares source, k_lo and n_k, and no zero-history masking is needed
    const float *bank;       // mfma: the class's tap slabs (bank2 layout above)
    int64_t src_len;         // mfma: samples addressable from the source view's base (buffer descriptor range)
    int mfma_nt, mfma_parts; // mfma: the launch's MfmaPlan
    float2 *partial;         // mfma, parts > 1: mfma_parts slabs of n_chans x max_n_k partial sums
    int small;               // 1: one-thread-per-output kernel with the discriminator fused in (no DiscLaunch)
    const float *atan_tab;   // small: gr::fast_atan2f table
};

void launch_fir_bank(const ChanLaunch *d_chans, const FirLaunchDims &dims, hipStream_t s);
// (re)build a bank matrix from the channels' composite taps
// dirty: device array of one byte per group of 32 (nullptr: every group), only flagged groups are rebuilt
void launch_fir_pack(const ChanLaunch *d_chans, int n_chans, int T, float *bank, const unsigned char *dirty,
                     hipStream_t s);

// discriminator: fm[n] = fast_atan2f(imag(y[n] conj(y[n-1])), real(.)) (unit gain), n relative index
struct DiscLaunch {
    const float2 *iq_ring;
    float *fm_ring;
    int64_t n_lo;            // first relative output index
    int32_t n_k;
    int32_t pad_;
};
void launch_discriminator(const DiscLaunch *d_items, int n_items, int max_n_k, uint64_t ring_mask,
                          const float *d_atan_table, hipStream_t s);

// real FIR on the (gain-scaled) discriminator stream: sym[n] = sum_i taps[i] * (gain * fm[n - i])
// (the P25 symbol filter fir_filter_fff(1, (1/sps,)*sps), p25_control_demod.py:129-133)
struct FmFirLaunch {
    const float *fm_ring;
    float *sym_ring;
    const float *taps;
    float gain;
    int32_t ntaps;
    int64_t n_lo;
    int64_t n_first;         // samples before this index count as zero (the filter's start)
    int32_t n_k;
    int32_t pad_;
};
void launch_fm_fir(const FmFirLaunch *d_items, int n_items, int max_n_k, uint64_t ring_mask, hipStream_t s);
// mean of gain * fm over the last `window` samples ending at n_end (exclusive), one workgroup
void launch_fm_level(const float *fm_ring, int64_t n_end, int window, float gain, uint64_t ring_mask, float *d_out,
                     hipStream_t s);

// ---------------------------------------------------------------- analog voice chain (audio.hip)
// per-channel running state, device resident, owned by audio_front_kernel
struct AudioState {
    double pwr;              // pwr_squelch_cc's single_pole_iir<double> output
    double iir_px, iir_py;   // iir_filter_ffd history: previous input, previous (double) output
    int32_t muted;           // squelch_base state: 1 = ST_MUTED
    int32_t pad_;
    int64_t n_a;             // samples that passed the squelch so far == outputs of demod / de-emphasis
    int64_t n_prev;          // n_a before the current block (the later stages work on [n_prev, n_a))
};
struct AudioLaunch {
    const float2 *iq_ring;
    AudioState *st;
    float2 *c_ring;                             // samples that passed the squelch, compacted
    float *a_ring, *l_ring, *h_ring, *o_ring;   // de-emphasised fm, after audio LPF, after HPF, 8 kHz audio
    const float *lpf, *hpf, *rs;                // rs: rational_resampler taps zero-padded to a multiple of interp
    int64_t n_lo;            // first relative channel output index to consume
    int32_t n_k;             // channel outputs to consume
    int32_t n_lpf, n_hpf, nt_rs;                // nt_rs = taps per polyphase arm
    int32_t interp, decim;
    float gain;
    int32_t pad_;
    double thr, alpha, b0, b1, fb1;
};
void launch_audio(const AudioLaunch *d_items, int n_items, int max_n_k, int max_interp_over_decim_num,
                  int max_interp_over_decim_den, uint64_t ring_mask, const float *d_atan_table, hipStream_t s);

// ---------------------------------------------------------------- polyphase filterbank
struct PfbLaunch {
    StreamView src;
    const float *ptaps;      // [P][NB] polyphase taps: ptaps[p*NB + rho] = h[NB p + rho] (0 beyond T)
    const float2 *tw;        // e^{+2 pi i n / NB}, n in [0, NB)
    float2 *bins_ring;       // layout below
    uint64_t ring_mask;
    int64_t tile_pitch;      // tiled layout: samples between consecutive tiles (>= 16 NB; see pfb_tile_pitch)
    int64_t n_lo;            // first absolute frame index of this launch
    int64_t n_abs0;          // absolute frame index of the PFB's first-ever frame
    int64_t start_sample;
    int64_t src_len;         // samples addressable from src.base (linear view), for the buffer descriptor
    int32_t n_frames;        // frames in this launch
    int32_t NB, D, P;
    // output layout, i = (n - n_abs0) & ring_mask:
    //   tiled (pfb.hip, power-of-two banks)   bins_ring[(i >> 4) tile_pitch + 16 k + (i & 15)]: a chunk of 16 frames is ONE
    //     contiguous run of 16 NB samples for the writer, and a bin's 16 frames are one 128-byte line for its readers
    //     (the first layout -- one ring per bin -- scattered a chunk over NB separate lines: 5 % slower)
    //   frame_major (pfb5.hip)                bins_ring[i NB + k]: a chunk of 2-4 frames cannot fill 128-byte lines
    //     per bin; whole frames leave as contiguous rows
    int32_t frame_major;
    int32_t n_taps;
    // frame-major banks: bins that are open as channels (rcf_pfb_tap_open).  The kernel has every bin of the chunk in
    // LDS when it writes the frames out, so it also copies the tapped bins -- and only those -- into a compact
    // frame-major matrix of THIS launch's frames, tap_mat[(frame - n_lo) tap_pitch + slot - tap_first] (slot = position in
    // tap_bins): whole rows of n_taps x 8 contiguous bytes, like the bins ring itself.  tap_finalize_kernel (tapfin.hip)
    // then transposes that matrix tile by tile through LDS into the channels' own rings, 128-byte lines, applying each
    // tap's rotator and the discriminator on the way.  (Round 2 stored a tap's F frames straight into its ring from
    // here: 32-byte pieces of lines a megabyte apart -- 0.18 -> 0.64 ms with all 1600 bins tapped; a ring TILED
    // [16 frames][bin][16] gets the same 32-byte pieces and measures 0.79 ms.)
    const int32_t *tap_bins;
    float2 *tap_mat;
    int32_t tap_pitch;       // row pitch of tap_mat in samples: the slots from tap_first on, rounded up to 16
    // slots [0, tap_first) -- a multiple of 16 -- are runs of 16 consecutive bins starting at a multiple of 16: row by
    // row such a run is one aligned 128-byte piece of the frame-major ring itself, tap_finalize reads it there, and
    // the bank copies only the slots from tap_first on.  With every bin tapped (the reference's intent,
    // receiver.py:343-383) the bank costs what it costs untapped.
    int32_t tap_first;
    // Rider (launch_plan): when nothing before the bank's launch needs them, the block's launch records (pinned host ->
    // device arena) and its history tail (behind the OTHER input buffer's block) are copied by the first kPfbRiderWgs
    // workgroups of the bank's kernel before they start on their chunks -- one launch less per block (4.7 us of the
    // timed configuration's 124).  8-byte words; rider_n8[0] + rider_n8[1] == 0: no rider.
    unsigned long long *rider_dst[2];
    const unsigned long long *rider_src[2];
    uint32_t rider_n8[2];
    // The discriminator fused into a frame-major bank (rcf_pfb_fm_enable; pfb5.hip): fm_ring[(i NB + k)] = fast_atan2f(bin_k[n]
    // conj(bin_k[n - 1]) x fm_inc[k]) for every bin of every frame, written by the bank's own kernel; fm_mode 1 = beside the
    // bins ring, 2 = INSTEAD of it.  fm_span = consecutive chunks one workgroup walks (it keeps the last frame in
    // registers from chunk to chunk and recomputes the ONE chunk before its span).  nullptr / 0: off.
    float *fm_ring;
    const float2 *fm_inc;    // [NB] the rotator increment a GNU Radio channel on bin k would carry, as a phasor (1 + 0j: none)
    const float *atan_tab;   // gr::fast_atan2f's 257-entry table
    int32_t fm_mode, fm_span;
    // ... or, instead of spans, ONE chunk per workgroup and the predecessor frame handed from workgroup to workgroup
    // through global memory (pfb5_fmlb_kernel): fm_edge[slot][NB] holds the last frame of chunk `slot mod fm_slots` (complex
    // bits as 64-bit words), fm_flag[8 slot + wave] = fm_tag + chunk once that wave's part of it is there; rows fm_slots .. fm_slots + 7 are the
    // predecessor frames the first workgroup of each XCD's range computes for itself, row fm_slots + 8 the one a front-end's
    // first chunk in a GROUPED launch computes.  fm_err counts predecessors that
    // never arrived (a bounded wait).  nullptr: the span form above.
    unsigned long long *fm_edge;
    unsigned long long *fm_flag;
    unsigned long long fm_tag;       // launch serial << 32
    int32_t fm_slots;
    int32_t fm_local;                // 1: the hand-over goes through ONE XCD's L2 (host-verified block -> XCD map), 0: agent scope
    int32_t *fm_err;
    // host side only (the kernels never look): events ATTACHED to the bank's dispatch (hipExtLaunchKernelGGL) instead of
    // a bracket of two event records around it (one barrier packet less inside the measured interval).  nullptr: plain launch.
    hipEvent_t ev_start, ev_stop;
};
// launch `kernel` with the launch's attached events if it has them
#define RCF_PFB_LAUNCH(p_, kernel, grid, block, lds, s, ...)                                                       \
    do {                                                                                                           \
        if ((p_).ev_start) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, (p_).ev_start, (p_).ev_stop, 0, __VA_ARGS__); \
        else               hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                            \
    } while (0)
constexpr int kPfbRiderWgs = 64;    // of thousands: the few microseconds the pinned-memory reads take are lost in the first round
// whether this launch's kernel takes the rider (pfb_kernel_os does: step of the timed configuration 128.5 -> 124.5 us,
// kernel unchanged): the persistent form of the 512 / 1024-bin banks does not -- its
// workgroups are ONE resident round, and 64 of them starting late set the whole launch back by what the copy launch
// cost (cfg5: kernel +2.8 us, step unchanged; spread over all 512 workgroups: every one waits for its pinned-memory
// word, 1600 bins +7 us)
bool pfb_takes_rider(const PfbLaunch &p);
#ifdef __HIPCC__
__device__ __forceinline__ void pfb_copy_rider(const PfbLaunch &p, int wg, int n_wgs, int tid, int n_threads)
{
    const size_t n0 = p.rider_n8[0], total = n0 + p.rider_n8[1], stride = (size_t)n_wgs * n_threads;
    for (size_t i0 = (size_t)wg * n_threads + tid; i0 < total; i0 += 4 * stride) {
        unsigned long long v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            v[u] = i < n0 ? p.rider_src[0][i] : i < total ? p.rider_src[1][i - n0] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            if (i < n0) p.rider_dst[0][i] = v[u];
            else if (i < total) p.rider_dst[1][i - n0] = v[u];
        }
    }
}
#endif
struct TapLaunch {           // one tapped bin, consumed by tap_finalize_kernel
    float2 *iq_ring;
    float *fm_ring;
    int64_t k_lo, k_abs0;    // as in ChanLaunch; a tap's output index = the bank's frame count
    int64_t n_seg0;          // rotator model, as in ChanLaunch (rotate_value reads these five)
    double angle0, dangle;
    double logmag0, dlogmag;
    int32_t n_k;
    int32_t bin;
    int32_t fm_only;         // 1: only the discriminator ring is written (and the launch's last IQ output, which the next launch's first discriminator sample needs)
    int32_t pad_;
    static constexpr bool kHasRotRing = false;
};
// what tap_finalize needs of ONE front-end (kernel argument of the single launch, arena record of the grouped one)
struct TapFinArgs {
    const TapLaunch *taps;
    const float2 *mat;
    const int32_t *group_bin0;
    const float2 *bins_ring;
    int64_t k_first;
    int32_t n_taps, pitch, n_rows, tap_first, n_bins, pad_;
};
void launch_tap_finalize_group(const TapFinArgs *d_args, int n_args, int max_taps, int max_rows, uint64_t ring_mask,
                               const float *d_atan_table, hipStream_t s);
// mat row r = the bank's frame k_first + r (tap output index); rows [0, n_rows)
// group_bin0[g] >= 0: the 16 taps of slot group g are the bins group_bin0[g] .. + 15 -- read from the bank's frame-major
// ring (bins_ring[((k_first + r) & ring_mask) n_bins + bin]) instead of the matrix
void launch_tap_finalize(const TapLaunch *d_taps, int n_taps, const float2 *tap_mat, int tap_pitch, int n_rows,
                         int64_t k_first, uint64_t ring_mask, const float *d_atan_table, const int32_t *d_group_bin0,
                         int tap_first, const float2 *bins_ring, int n_bins, hipStream_t s);
// ---- grouped filterbank launch: the chunks of G front-ends (same shape) in one grid
// virtual chunk v (after the XCD-aware map over the whole grid) belongs to front-end fe with wg_first[fe] <= v <
// wg_first[fe + 1]; uniform_nwg > 0: every front-end has that many chunks (the real-time case), no table walk
struct GroupMap {
    const int32_t *wg_first;     // n_fe + 1 entries (device)
    int32_t n_fe, total_wg, uniform_nwg;
};
#ifdef __HIPCC__
__device__ __forceinline__ void group_resolve(const GroupMap &m, int b, int &fe, int &wg)
{
    const int q = m.total_wg / 8, r = m.total_wg % 8, xcd = b % 8;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    if (m.uniform_nwg > 0) {
        fe = v / m.uniform_nwg;
        wg = v - fe * m.uniform_nwg;
    } else {
        int lo = 0, hi = m.n_fe;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (m.wg_first[mid] <= v) lo = mid; else hi = mid;
        }
        fe = lo;
        wg = v - m.wg_first[lo];
    }
    fe = __builtin_amdgcn_readfirstlane(fe);
    wg = __builtin_amdgcn_readfirstlane(wg);
}
#endif
// d_pls: the members' launch records (device); shape: any member's record (NB, D, P select the kernel).  false: this shape
// has no grouped kernel -- launch the members one by one
bool launch_pfb_group(const PfbLaunch &shape, const PfbLaunch *d_pls, const GroupMap &gm, hipStream_t s);
bool pfb5_dispatch_group(const PfbLaunch &p, const PfbLaunch *d_pls, const GroupMap &gm, hipStream_t s);
bool pfb_sees_zero_history(const PfbLaunch &p);
int pfb_chunk_frames(int NB);
// Stage-2 rider of a filterbank launch (pfb.hip): the small-T FIR + discriminator launch of the PREVIOUS block, run by the
// first n_wgs workgroups of this block's filterbank kernel (its records are already in the device arena)
struct S2Rider {
    const ChanLaunch *chans;
    const float *atan_tab;
    uint64_t ring_mask;
    int32_t n_chans, n_tiles;    // the deferred launch's grid: (channels, tiles of KB outputs)
    int32_t D, T, KB;
    int32_t n_wgs;               // rider workgroups in the grid (n_batches * batch_wgs); 0: no rider
    int32_t n_batches, batch_wgs, period;   // set by the launcher: batch q occupies blocks [q period, q period + batch_wgs)
};
bool pfb_can_carry_s2(const PfbLaunch &p);
bool pfb_supported(int NB, int D, int P);
int pfb_padded_p(int NB, int D, int P);   // rows the kernel instantiation reads from ptaps (zero padded)
void launch_pfb(const PfbLaunch &p, hipStream_t s, const S2Rider *sr = nullptr);
// bin counts with a factor 25 (pfb5.hip): 400, 800, 1600, 3200
bool pfb5_dispatch(const PfbLaunch &p, bool probe, hipStream_t s);
// the fused-discriminator form of a frame-major bank (rcf_pfb_fm_enable): whether the shape has one, the input history its
// halo chunk reaches back over, and the per-bin increment table inc[k] = (float)(cos, sin)(dangle[k]) -- computed on the
// device with tap_finalize's own sincos_fast, so that both paths turn the discriminator's product by the same bits
bool pfb5_fm_supported(int NB, int D, int P);
bool pfb5_fm_sees_zero_history(const PfbLaunch &p);
bool pfb5_xcd_map_ok(int device, hipStream_t s);
size_t pfb5_fm_history(int NB, int D, int P);
void launch_pfb5_fm_inc(const double *d_dangle, float2 *d_inc, int NB, hipStream_t s);
inline bool pfb_frame_major(int NB) { return NB % 25 == 0; }
// dst[i] = view sample (first + i), i < n (one bin's samples out of a bank ring; ingest.hip)
void launch_gather_view(const StreamView &v, int64_t first, float2 *dst, size_t n, hipStream_t s);
void launch_gather_f32(const float *base, uint64_t mask, int64_t stride, int64_t first, float gain, float *dst, size_t n, hipStream_t s);
// one ring segment of a batched read (rcf_chan_read_many), in 4-byte words: dst[dst_w + w] = ring[(pos_w + w) & mask_w], w < n_w
//   dst_mask_w = ~0u, dst_pos_w = 0: rows packed back to back; otherwise the destination is a ring of dst_mask_w + 1 words that
//   starts at word dst_w (the real-time pump's per-channel host rings), written from dst_pos_w on
struct GatherRec {
    const uint32_t *ring;
    uint32_t pos_w, n_w, mask_w, dst_w;
    uint32_t dst_pos_w, dst_mask_w;
    float gain;              // flags & 1: the words are float32 and leave multiplied by gain (quadrature_demod_cf's gain, one
    uint32_t flags;          // float32 multiply -- what rcf_chan_read_fm does on the host)
    uint32_t stride_w;       // words between consecutive source items (0 / 1: contiguous); > 1: one bin of a frame-major ring of
                             // floats (the bank's fused discriminator ring: ring = its bin, stride = bins)
};
void launch_gather_rings(const GatherRec *d_recs, int n_recs, uint32_t *d_dst, uint32_t max_words, hipStream_t s);
// one record of the grouped ingest launch (group_prep_kernel, ingest.hip): a block of one front-end, or a plain copy
struct PrepRec {
    const void *src;         // wire-format / cf32 samples (pinned host memory as the device sees it, or device memory)
    float2 *dst;             // the block's place in the front-end's wideband buffer (copy: destination words)
    float2 *hist_dst;        // where block sample hist_from lands in the OTHER buffer's history; nullptr: no dual write
    uint32_t n;              // samples (copy: 8-byte words)
    uint32_t hist_from;      // block samples i >= hist_from are also written to hist_dst[i - hist_from]
    int32_t fmt;             // RCF_FMT_CF32 / U8 / S8 / S16; < 0: plain 8-byte copy
    float scale, offset;
    int32_t aligned;         // src allows one vector load per two samples (4 / 8 / 16-byte aligned for 8 / 16 / 32-bit items)
    int32_t dst_aligned;     // dst is 16-byte aligned
    uint32_t tile_first;     // first tile of this record among the launch's tiles (fill_prep_tiles)
    int32_t pad_[2];
};
constexpr int kPrepMaxRecs = 1024;   // records per launch_group_prep
static_assert(sizeof(PrepRec) == 64, "one record = one 64-byte line of the pinned arena");
uint32_t fill_prep_tiles(PrepRec *recs, int n_recs);     // sets tile_first; returns the launch's tile count
void launch_group_prep(const PrepRec *d_recs, int n_recs, uint32_t total_tiles, hipStream_t s);
// dst[0, bytes) = src[0, bytes), both 8-byte aligned, bytes rounded up to 8; src may be pinned (device-mapped) host memory
void launch_copy8(void *dst, const void *src, size_t bytes, hipStream_t s);
void launch_copy8x2(void *d0, const void *s0, size_t bytes0, void *d1, const void *s1, size_t bytes1, hipStream_t s);
int pfb5_padded_p(int NB, int D, int P);

// ---------------------------------------------------------------- scan
struct ScanLaunch {
    StreamView src;
    int64_t s0;              // stream index of the first sample of frame `f0`
    const float *window;     // float32[N]
    const float2 *tw;        // e^{-2 pi i n / N}
    float *vring;            // [R][N] log-magnitude frames, frame f in slot f % R
    int32_t N, R;
    int32_t f0, n_frames;    // frames f0 .. f0+n_frames-1
    float2 *scratch;         // 4-step scratch (N >= 32768), n_frames * N complex
};
bool scan_supported(int N);
bool scan4_split(int N, int *N1, int *N2);   // four-step factorisation for N > 16384
void launch_scan_fft(const ScanLaunch &p, hipStream_t s);
// running sum: sum += v[f]; if (f == emit_frame) out = sum; if (f-(L-1) >= 0) sum -= v[f-(L-1)]
void launch_scan_movsum(float *vring, int N, int R, int L, int f0, int n_frames, int emit_frame,
                        float *sum, float *out, hipStream_t s);

// device peak picker (peaks.hip)
size_t peaks_workspace_bytes(int n);
void launch_find_peaks(const float *d_spec, int n, double min_w, double max_w, double prominence, void *ws,
                       int64_t *d_out, int cap, int **d_count_out, double **d_mean_out, hipStream_t s);

// wire-format ingest (ingest.hip)
size_t raw_sample_bytes(int fmt);
void launch_convert(int fmt, const void *d_raw, float2 *d_out, size_t n, float scale, float offset, hipStream_t s);

const float *atan_table_host();   // 257 floats: atan(i/255), i = 0..255, + pi/4

}  // namespace rcfx
