// audio.hip -- the reference's analog NBFM voice chain behind a channel (gfx950).
//
// /root/reference/logging_receiver.py:211-222 (file_to_wav.py:109-122) per analog call:
//   pwr_squelch_cc(-100, 0.01, 0, True) -> fm_demod_cf(rate, 1, 15000, 0.25 rate, 0.25 rate + 2000, 8, 75e-6)
//   -> fir_filter_fff(1, high_pass(1, rate, 300, 30, HAMMING)) -> rational_resampler_fff(8000, rate)
// with fm_demod_cf = quadrature_demod_cf(k) -> fm_deemph (iir_filter_ffd) -> fir_filter_fff(1, optfir taps).
//
// Two kinds of work.  The squelch (a one-pole power filter that GATES samples, so everything behind it runs
// on a data-dependent stream) and the de-emphasis IIR are recurrences in time: audio_gate_kernel and
// audio_deemph_kernel give each channel one lane that walks the block's new samples in order -- parallel over
// channels, exact in GNU Radio's operation order (double accumulators as in single_pole_iir<double> /
// iir_filter<float,float,double,double>), a few dependent double operations per step; their ring traffic is
// staged through LDS so that it stays coalesced.
// Everything else is a pure function of index and runs one thread per output on [n_prev, n_a), the range the
// gate published in device memory: the discriminator on the compacted stream, the two FIRs and the polyphase
// resampler (the host never learns how many samples passed the gate until it reads audio back).
// Volumes are tiny next to the channelizer (25 kS/s per channel): these kernels are written for exactness.
// Built with -ffp-contract=off (Makefile): the unfused float / double operations below are the specified result;
// where a fused multiply-add is meant it is written fmaf().
#include "rcf_internal.h"
#include "fast_atan2f_gr.hpp"

namespace rcfx {

namespace {

constexpr int kThreads = 256;

// gr::fast_atan2f: fast_atan2f_gr.hpp, the one every discriminator kernel runs (the table comes in through LDS)

// Stage 1, one lane per channel: pwr_squelch_cc over the channel's new samples, in order.  Survivors are compacted
// into c_ring; [n_prev, n_a) is published for the stages behind.
// A wave owns 64 channels.  Ring traffic goes through an LDS tile so that it is coalesced: for a chunk of 64
// samples the 64 lanes first fetch channel 0's 64 samples (one 512-byte run), then channel 1's, ... into
// xs[channel][sample]; each lane then walks ITS channel's row, parking the survivors in the same row, and the rows
// are written out channel by channel.  (A lane reading its own ring directly touches one cache line per lane per
// load instruction: measured 5x slower.)
constexpr int kSeqChunk = 64;
constexpr int kSeqRow = kSeqChunk + 1;               // row stride in elements: spreads a column over the banks

// broadcast lane `src`'s value (src wave-uniform) through the scalar unit
__device__ __forceinline__ int rl32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ long long rl64(long long v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), src);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void audio_gate_kernel(const AudioLaunch *__restrict__ items, int n_items,
                                                        uint64_t ring_mask)
{
    __shared__ float2 xs[64 * kSeqRow];
    const int lane = threadIdx.x;
    const int c0 = blockIdx.x * 64;
    const int nc = min(64, n_items - c0);
    const bool mine = lane < nc;
    const AudioLaunch *it = items + c0 + (mine ? lane : 0);
    AudioState s = *it->st;
    const double alpha = it->alpha, one_minus_alpha = 1.0 - it->alpha, thr = it->thr;
    const int my_nk = mine ? it->n_k : 0;
    // each lane keeps its own channel's ring pointers; the per-channel loops below broadcast them lane by lane
    const long long my_src = (long long)(uintptr_t)it->iq_ring, my_dst = (long long)(uintptr_t)it->c_ring;
    const long long my_lo = it->n_lo;
    s.n_prev = s.n_a;
    int max_nk = 0;
    for (int c = 0; c < nc; ++c) max_nk = max(max_nk, rl32(my_nk, c));
    // the whole next chunk (one coalesced 512-byte load per channel) is in flight while the current one is walked:
    // four waves carry a 256-channel launch, there is nothing else to hide the ring's latency behind
    float2 pre[64];
    auto prefetch = [&](int i0) {
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            const int cc = c < nc ? c : nc - 1;
            const float2 *src = reinterpret_cast<const float2 *>((uintptr_t)rl64(my_src, cc));
            const int i = i0 + lane < rl32(my_nk, cc) ? i0 + lane : 0;
            pre[c] = src[(uint64_t)(rl64(my_lo, cc) + i) & ring_mask];
        }
    };
    prefetch(0);
    for (int i0 = 0; i0 < max_nk; i0 += kSeqChunk) {
#pragma unroll
        for (int c = 0; c < 64; ++c) xs[c * kSeqRow + lane] = pre[c];
        wave_lds_sync();
        if (i0 + kSeqChunk < max_nk) prefetch(i0 + kSeqChunk);
        // Only two operations per sample sit on the recurrence (oma * pwr, + alpha * p); |x|^2 and alpha * p are done
        // 16 samples at a time ahead of it.  With ramp = 0 squelch_base_cc's state machine collapses to
        // "muted = mute()": MUTED leaves on !mute, UNMUTED leaves on mute.
        int kept = 0;
        const int n_here = min(kSeqChunk, my_nk - i0);
        for (int u0 = 0; u0 < n_here; u0 += 16) {
            float2 x[16];
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = xs[lane * kSeqRow + u0 + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // pwr_squelch_cc::update_state: float |x|^2, then single_pole_iir<double,double,double>
                const float p = __fadd_rn(__fmul_rn(x[u].x, x[u].x), __fmul_rn(x[u].y, x[u].y));
                t[u] = __dmul_rn(alpha, (double)p);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool live = u0 + u < n_here;
                const double pw = __dadd_rn(t[u], __dmul_rn(one_minus_alpha, s.pwr));
                s.pwr = live ? pw : s.pwr;
                const bool pass = live && !(pw < thr);              // gate = True: muted samples vanish
                if (live) s.muted = (pw < thr) ? 1 : 0;
                if (pass) xs[lane * kSeqRow + kept] = x[u];         // kept <= u0 + u: never ahead of the reads
                kept += pass ? 1 : 0;
            }
        }
        wave_lds_sync();
        for (int c = 0; c < nc; ++c) {                          // coalesced write-back of the survivors
            float2 *dst = reinterpret_cast<float2 *>((uintptr_t)rl64(my_dst, c));
            if (lane < rl32(kept, c)) dst[(uint64_t)(rl64(s.n_a, c) + lane) & ring_mask] = xs[c * kSeqRow + lane];
        }
        s.n_a += kept;
        wave_lds_sync();
    }
    if (mine) {
        it->st->pwr = s.pwr;
        it->st->muted = s.muted;
        it->st->n_a = s.n_a;
        it->st->n_prev = s.n_prev;
    }
}

// Stage 2, one thread per surviving sample: quadrature_demod_cf(gain) on the compacted stream
// (volk_32fc_x2_multiply_conjugate_32fc, fast_atan2f, gain).  Output parks in h_ring's slots [n_prev, n_a), which
// the high-pass overwrites later in the same block, after the de-emphasis has consumed them.
__global__ __launch_bounds__(kThreads) void audio_demod_kernel(const AudioLaunch *__restrict__ items,
                                                               uint64_t ring_mask, const float *__restrict__ atan_tab)
{
    __shared__ float tab[260];
    for (int i = threadIdx.x; i < 257; i += kThreads) tab[i] = atan_tab[i];
    __syncthreads();
    const AudioLaunch &it = items[blockIdx.y];
    const int64_t n = it.st->n_prev + (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= it.st->n_a) return;
    const float2 x = it.c_ring[(uint64_t)n & ring_mask];
    const float2 pv = n > 0 ? it.c_ring[(uint64_t)(n - 1) & ring_mask] : make_float2(0.f, 0.f);
    const float tr = __fadd_rn(__fmul_rn(x.x, pv.x), __fmul_rn(x.y, pv.y));
    const float ti = __fsub_rn(__fmul_rn(x.y, pv.x), __fmul_rn(x.x, pv.y));
    it.h_ring[(uint64_t)n & ring_mask] = __fmul_rn(it.gain, fast_atan2f_gr(ti, tr, tab));
}

// Stage 3, one lane per channel: fm_deemph = iir_filter<float,float,double,double>, two feed-forward taps and
// one feedback tap, in its operation order; ring traffic staged through LDS like the gate's
__global__ __launch_bounds__(64) void audio_deemph_kernel(const AudioLaunch *__restrict__ items, int n_items,
                                                          uint64_t ring_mask)
{
    __shared__ float gs[64 * kSeqRow];
    const int lane = threadIdx.x;
    const int c0 = blockIdx.x * 64;
    const int nc = min(64, n_items - c0);
    const bool mine = lane < nc;
    const AudioLaunch *it = items + c0 + (mine ? lane : 0);
    const double b0 = it->b0, b1 = it->b1, fb1 = it->fb1;
    double px = it->st->iir_px, py = it->st->iir_py;
    const long long my_p0 = it->st->n_prev, my_p1 = mine ? (long long)it->st->n_a : my_p0;
    const int my_n = (int)(my_p1 - my_p0);
    const long long my_src = (long long)(uintptr_t)it->h_ring, my_dst = (long long)(uintptr_t)it->a_ring;
    int max_n = 0;
    for (int c = 0; c < nc; ++c) max_n = max(max_n, rl32(my_n, c));
    float pre[64];
    auto prefetch = [&](int i0) {
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            const int cc = c < nc ? c : nc - 1;
            const float *src = reinterpret_cast<const float *>((uintptr_t)rl64(my_src, cc));
            const int i = i0 + lane < rl32(my_n, cc) ? i0 + lane : 0;
            pre[c] = src[(uint64_t)(rl64(my_p0, cc) + i) & ring_mask];
        }
    };
    prefetch(0);
    for (int i0 = 0; i0 < max_n; i0 += kSeqChunk) {
#pragma unroll
        for (int c = 0; c < 64; ++c) gs[c * kSeqRow + lane] = pre[c];
        wave_lds_sync();
        if (i0 + kSeqChunk < max_n) prefetch(i0 + kSeqChunk);
        // acc = b0 x[n] + b1 x[n-1] + fb1 y[n-1] in that order; only the last product and sum wait for y[n-1]
        const int n_here = min(kSeqChunk, my_n - i0);
        for (int u0 = 0; u0 < n_here; u0 += 16) {
            float v[16];
            double ff[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = gs[lane * kSeqRow + u0 + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double xm1 = u ? (double)v[u - 1] : px;
                ff[u] = __dadd_rn(__dmul_rn(b0, (double)v[u]), __dmul_rn(b1, xm1));
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool live = u0 + u < n_here;
                const double acc = __dadd_rn(ff[u], __dmul_rn(fb1, py));
                py = live ? acc : py;
                px = live ? (double)v[u] : px;
                if (live) gs[lane * kSeqRow + u0 + u] = (float)acc;
            }
        }
        wave_lds_sync();
        for (int c = 0; c < nc; ++c) {
            float *dst = reinterpret_cast<float *>((uintptr_t)rl64(my_dst, c));
            if (i0 + lane < rl32(my_n, c)) dst[(uint64_t)(rl64(my_p0, c) + i0 + lane) & ring_mask] = gs[c * kSeqRow + lane];
        }
        wave_lds_sync();
    }
    if (mine) {
        it->st->iir_px = px;
        it->st->iir_py = py;
    }
}

// fir_filter_fff(1, taps) on [n_prev, n_a): which = 0 a_ring -> l_ring (audio low-pass), 1 l_ring -> h_ring
__global__ __launch_bounds__(kThreads) void audio_fir_kernel(const AudioLaunch *__restrict__ items, int which,
                                                             uint64_t ring_mask)
{
    const AudioLaunch &it = items[blockIdx.y];
    const int64_t n0 = it.st->n_prev, n1 = it.st->n_a;
    const int64_t n = n0 + (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (n >= n1) return;
    const float *src = which ? it.l_ring : it.a_ring;
    float *dst = which ? it.h_ring : it.l_ring;
    const float *taps = which ? it.hpf : it.lpf;
    const int nt = which ? it.n_hpf : it.n_lpf;
    const int kmax = n + 1 < (int64_t)nt ? (int)(n + 1) : nt;        // x[< 0] = 0 (the filter's zero history)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = 0;
    for (; k + 4 <= kmax; k += 4) {
        a0 = fmaf(taps[k], src[(uint64_t)(n - k) & ring_mask], a0);
        a1 = fmaf(taps[k + 1], src[(uint64_t)(n - k - 1) & ring_mask], a1);
        a2 = fmaf(taps[k + 2], src[(uint64_t)(n - k - 2) & ring_mask], a2);
        a3 = fmaf(taps[k + 3], src[(uint64_t)(n - k - 3) & ring_mask], a3);
    }
    for (; k < kmax; ++k) a0 = fmaf(taps[k], src[(uint64_t)(n - k) & ring_mask], a0);
    dst[(uint64_t)n & ring_mask] = (a0 + a1) + (a2 + a3);
}

// rational_resampler_base_fff: output m reads input p = floor(m D / I) through arm ctr = (m D) mod I,
// out[m] = sum_k taps[ctr + I k] h[p - k]; it exists once input p does
__global__ __launch_bounds__(kThreads) void audio_resample_kernel(const AudioLaunch *__restrict__ items,
                                                                  uint64_t ring_mask)
{
    const AudioLaunch &it = items[blockIdx.y];
    const int64_t I = it.interp, D = it.decim;
    const int64_t m0 = (it.st->n_prev * I + D - 1) / D, m1 = (it.st->n_a * I + D - 1) / D;
    const int64_t m = m0 + (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (m >= m1) return;
    const int64_t p = (m * D) / I;
    const int ctr = (int)((m * D) - p * I);
    const int kmax = p + 1 < (int64_t)it.nt_rs ? (int)(p + 1) : it.nt_rs;
    float a0 = 0.f, a1 = 0.f;
    int k = 0;
    for (; k + 2 <= kmax; k += 2) {
        a0 = fmaf(it.rs[ctr + (int)I * k], it.h_ring[(uint64_t)(p - k) & ring_mask], a0);
        a1 = fmaf(it.rs[ctr + (int)I * (k + 1)], it.h_ring[(uint64_t)(p - k - 1) & ring_mask], a1);
    }
    if (k < kmax) a0 = fmaf(it.rs[ctr + (int)I * k], it.h_ring[(uint64_t)(p - k) & ring_mask], a0);
    it.o_ring[(uint64_t)m & ring_mask] = a0 + a1;
}

}  // namespace

// max_n_k: most channel samples any item consumes; (num, den): the largest interp/decim ratio among the items
void launch_audio(const AudioLaunch *d_items, int n_items, int max_n_k, int ratio_num, int ratio_den,
                  uint64_t ring_mask, const float *d_atan_table, hipStream_t s)
{
    if (n_items <= 0 || max_n_k <= 0) return;
    const dim3 grid((max_n_k + kThreads - 1) / kThreads, n_items);
    hipLaunchKernelGGL(audio_gate_kernel, dim3((n_items + 63) / 64), dim3(64), 0, s, d_items, n_items, ring_mask);
    hipLaunchKernelGGL(audio_demod_kernel, grid, dim3(kThreads), 0, s, d_items, ring_mask, d_atan_table);
    hipLaunchKernelGGL(audio_deemph_kernel, dim3((n_items + 63) / 64), dim3(64), 0, s, d_items, n_items, ring_mask);
    hipLaunchKernelGGL(audio_fir_kernel, grid, dim3(kThreads), 0, s, d_items, 0, ring_mask);
    hipLaunchKernelGGL(audio_fir_kernel, grid, dim3(kThreads), 0, s, d_items, 1, ring_mask);
    const int64_t max_out = ((int64_t)max_n_k * ratio_num + ratio_den - 1) / ratio_den + 1;
    hipLaunchKernelGGL(audio_resample_kernel, dim3((unsigned)((max_out + kThreads - 1) / kThreads), n_items),
                       dim3(kThreads), 0, s, d_items, ring_mask);
}

}  // namespace rcfx
