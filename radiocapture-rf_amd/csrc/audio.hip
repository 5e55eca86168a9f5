// audio.hip -- the reference's analog NBFM voice chain behind a channel (gfx950).
//
// /root/reference/logging_receiver.py:211-222 (file_to_wav.py:109-122) per analog call:
//   pwr_squelch_cc(-100, 0.01, 0, True) -> fm_demod_cf(rate, 1, 15000, 0.25 rate, 0.25 rate + 2000, 8, 75e-6)
//   -> fir_filter_fff(1, high_pass(1, rate, 300, 30, HAMMING)) -> rational_resampler_fff(8000, rate)
// with fm_demod_cf = quadrature_demod_cf(k) -> fm_deemph (iir_filter_ffd) -> fir_filter_fff(1, optfir taps).
//
// Two kinds of work.  The squelch (a one-pole power filter that GATES samples, so everything behind it runs
// on a data-dependent stream) and the de-emphasis IIR are recurrences in time: audio_front_kernel gives each
// channel one lane that walks the block's new samples in order -- parallel over channels, exact in GNU
// Radio's operation order (double accumulators as in single_pole_iir<double> / iir_filter<float,float,double,
// double>).  Everything after that is a pure function of index: the two FIRs and the polyphase resampler run
// one thread per output on [n_prev, n_a), the range the front kernel published in device memory (the host
// never learns how many samples passed the gate until it reads audio back).
// Volumes are tiny next to the channelizer (25 kS/s per channel): these kernels are written for exactness.
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

// one lane per channel: squelch -> quadrature demod -> de-emphasis over the channel's new samples, in order
__global__ __launch_bounds__(64) void audio_front_kernel(const AudioLaunch *__restrict__ items, int n_items,
                                                         uint64_t ring_mask, const float *__restrict__ atan_tab)
{
    __shared__ float tab[260];
    for (int i = threadIdx.x; i < 257; i += 64) tab[i] = atan_tab[i];
    __syncthreads();
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_items) return;
    const AudioLaunch it = items[c];
    AudioState s = *it.st;
    s.n_prev = s.n_a;
    for (int i = 0; i < it.n_k; ++i) {
        const float2 x = it.iq_ring[(uint64_t)(it.n_lo + i) & ring_mask];
        // pwr_squelch_cc::update_state: float |x|^2, then single_pole_iir<double,double,double>
        const float p = __fadd_rn(__fmul_rn(x.x, x.x), __fmul_rn(x.y, x.y));
        s.pwr = __dadd_rn(__dmul_rn(it.alpha, (double)p), __dmul_rn(1.0 - it.alpha, s.pwr));
        const bool mute = s.pwr < it.thr;
        // squelch_base_cc, ramp = 0: MUTED <-> UNMUTED on the spot; gate = True: muted samples vanish
        if (s.muted) { if (!mute) s.muted = 0; }
        else         { if (mute) s.muted = 1; }
        if (s.muted) continue;
        // quadrature_demod_cf: volk_32fc_x2_multiply_conjugate_32fc, fast_atan2f, gain
        const float tr = __fadd_rn(__fmul_rn(x.x, s.prev.x), __fmul_rn(x.y, s.prev.y));
        const float ti = __fsub_rn(__fmul_rn(x.y, s.prev.x), __fmul_rn(x.x, s.prev.y));
        const float fm = __fmul_rn(it.gain, fast_atan2f_gr(ti, tr, tab));
        s.prev = x;
        // iir_filter<float,float,double,double>::filter, two feed-forward taps, one feedback tap
        double acc = __dmul_rn(it.b0, (double)fm);
        acc = __dadd_rn(acc, __dmul_rn(it.b1, s.iir_px));
        acc = __dadd_rn(acc, __dmul_rn(it.fb1, s.iir_py));
        s.iir_py = acc;
        s.iir_px = (double)fm;
        it.a_ring[(uint64_t)s.n_a & ring_mask] = (float)acc;
        s.n_a += 1;
    }
    *it.st = s;
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
    hipLaunchKernelGGL(audio_front_kernel, dim3((n_items + 63) / 64), dim3(64), 0, s, d_items, n_items, ring_mask,
                       d_atan_table);
    const dim3 grid((max_n_k + kThreads - 1) / kThreads, n_items);
    hipLaunchKernelGGL(audio_fir_kernel, grid, dim3(kThreads), 0, s, d_items, 0, ring_mask);
    hipLaunchKernelGGL(audio_fir_kernel, grid, dim3(kThreads), 0, s, d_items, 1, ring_mask);
    const int64_t max_out = ((int64_t)max_n_k * ratio_num + ratio_den - 1) / ratio_den + 1;
    hipLaunchKernelGGL(audio_resample_kernel, dim3((unsigned)((max_out + kThreads - 1) / kThreads), n_items),
                       dim3(kThreads), 0, s, d_items, ring_mask);
}

}  // namespace rcfx
