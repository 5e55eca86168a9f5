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
// iir_filter<float,float,double,double>), a few dependent double operations per step and nothing else.
// Everything else is a pure function of index and runs one thread per output on [n_prev, n_a), the range the
// gate published in device memory: the discriminator on the compacted stream, the two FIRs and the polyphase
// resampler (the host never learns how many samples passed the gate until it reads audio back).
// Volumes are tiny next to the channelizer (25 kS/s per channel): these kernels are written for exactness.
// Built with -ffp-contract=off (Makefile): the unfused float / double operations below are the specified result;
// where a fused multiply-add is meant it is written fmaf().
#include "rcf_internal.h"

namespace rcfx {

namespace {

constexpr int kThreads = 256;

// gr::fast_atan2f (same table and fix-ups as fir.hip's discriminator; the table comes in through LDS)
__device__ __forceinline__ float fast_atan2f_gr(float y, float x, const float *tab)
{
    const float TAN_MAP_RES = 0.003921569f;
    const float PI = 3.14159265358979323846f, PI_2 = 1.57079632679489661923f;
    const float ya = fabsf(y), xa = fabsf(x);
    if (!((ya > 0.0f) || (xa > 0.0f))) return 0.0f;
    const float z = (ya < xa) ? __fdiv_rn(ya, xa) : __fdiv_rn(xa, ya);
    float base;
    if (z < TAN_MAP_RES) {
        base = z;
    } else {
        float alpha = __fmul_rn(z, 255.0f);
        const int index = ((int)alpha) & 0xff;
        alpha = __fsub_rn(alpha, (float)index);
        const float t0 = tab[index], t1 = tab[index + 1];
        base = __fadd_rn(t0, __fmul_rn(__fsub_rn(t1, t0), alpha));
    }
    float angle;
    if (xa > ya) {
        if (x >= 0.0f) angle = (y >= 0.0f) ? base : -base;
        else           angle = (y >= 0.0f) ? __fsub_rn(PI, base) : __fsub_rn(base, PI);
    } else {
        if (y >= 0.0f) angle = (x >= 0.0f) ? __fsub_rn(PI_2, base) : __fadd_rn(PI_2, base);
        else           angle = (x >= 0.0f) ? __fadd_rn(-PI_2, base) : __fsub_rn(-PI_2, base);
    }
    return angle;
}

// Stage 1, one lane per channel: pwr_squelch_cc over the channel's new samples, in order.  Survivors are compacted
// into c_ring; [n_prev, n_a) is published for the stages behind.  Samples are fetched 16 at a time so the walk
// pays the ring's load latency once per 16 steps, not per step.
__global__ __launch_bounds__(64) void audio_gate_kernel(const AudioLaunch *__restrict__ items, int n_items,
                                                        uint64_t ring_mask)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_items) return;
    const AudioLaunch it = items[c];
    AudioState s = *it.st;
    s.n_prev = s.n_a;
    const double one_minus_alpha = 1.0 - it.alpha;
    // Only two operations per sample sit on the recurrence (oma * pwr, + alpha * p); everything else -- the loads,
    // |x|^2, alpha * p -- is done for 16 samples at once ahead of it.  With ramp = 0 squelch_base_cc's state machine
    // collapses to "muted = mute()": MUTED leaves on !mute, UNMUTED leaves on mute.
    for (int i0 = 0; i0 < it.n_k; i0 += 16) {
        float2 xs[16];
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + u < it.n_k ? i0 + u : it.n_k - 1;
            xs[u] = it.iq_ring[(uint64_t)(it.n_lo + i) & ring_mask];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // pwr_squelch_cc::update_state: float |x|^2, then single_pole_iir<double,double,double>
            const float p = __fadd_rn(__fmul_rn(xs[u].x, xs[u].x), __fmul_rn(xs[u].y, xs[u].y));
            t[u] = __dmul_rn(it.alpha, (double)p);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool live = i0 + u < it.n_k;
            const double pw = __dadd_rn(t[u], __dmul_rn(one_minus_alpha, s.pwr));
            s.pwr = live ? pw : s.pwr;
            const bool pass = live && !(pw < it.thr);               // gate = True: muted samples vanish
            if (live) s.muted = (pw < it.thr) ? 1 : 0;
            if (pass) it.c_ring[(uint64_t)s.n_a & ring_mask] = xs[u];   // in[i] * gr_complex(envelope = 1, 0)
            s.n_a += pass ? 1 : 0;
        }
    }
    it.st->pwr = s.pwr;
    it.st->muted = s.muted;
    it.st->n_a = s.n_a;
    it.st->n_prev = s.n_prev;
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
// one feedback tap, in its operation order
__global__ __launch_bounds__(64) void audio_deemph_kernel(const AudioLaunch *__restrict__ items, int n_items,
                                                          uint64_t ring_mask)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_items) return;
    const AudioLaunch it = items[c];
    const int64_t n0 = it.st->n_prev, n1 = it.st->n_a;
    double px = it.st->iir_px, py = it.st->iir_py;
    // acc = b0 x[n] + b1 x[n-1] + fb1 y[n-1] in that order: the first two terms do not depend on the recurrence
    for (int64_t j0 = n0; j0 < n1; j0 += 16) {
        float v[16];
        double ff[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = it.h_ring[(uint64_t)(j0 + u < n1 ? j0 + u : n1 - 1) & ring_mask];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double xm1 = u ? (double)v[u - 1] : px;
            ff[u] = __dadd_rn(__dmul_rn(it.b0, (double)v[u]), __dmul_rn(it.b1, xm1));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool live = j0 + u < n1;
            const double acc = __dadd_rn(ff[u], __dmul_rn(it.fb1, py));
            py = live ? acc : py;
            px = live ? (double)v[u] : px;
            if (live) it.a_ring[(uint64_t)(j0 + u) & ring_mask] = (float)acc;
        }
    }
    it.st->iir_px = px;
    it.st->iir_py = py;
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
