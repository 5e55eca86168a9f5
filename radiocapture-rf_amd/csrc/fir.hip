// fir.hip -- batched direct frequency-translating decimating FIR bank + FM discriminator (gfx950).
//
// Replaces, for every open channel at once, the per-channel GNU Radio flowgraph of
// /root/reference/rc_frontend/channel.py:29-38 (sub_source -> freq_xlating_fir_filter_ccc -> pub_sink)
// and the consumers' analog.quadrature_demod_cf (/root/reference/p25_control_demod.py:120-121).
//
//   y_c[k] = rot_c[k] * sum_{i<T} ctaps_c[i] * x[k D - i]
//
// Workgroup = 256 threads = 4 wavefronts; one workgroup owns a tile of KT consecutive outputs for a
// group of channels that share one input stream.  The (KT-1) D + T input samples of the tile are
// staged ONCE in LDS (coalesced 8-byte reads of the interleaved cf32 stream) and re-used by every
// channel of the group and by all T/D overlapping windows.  Inside a wavefront the 64 lanes split
// the tap index (lane l takes taps l, l+64, ...): both the composite-tap reads (global, L2-resident)
// and the LDS reads are then unit-stride across lanes -- no bank conflicts for any D -- and each
// lane carries CT x KR complex accumulators that are reduced across the wave at the end with
// butterfly shuffles.  The rotator is evaluated in closed form in float64 from the float32
// increment GNU Radio would iterate (angle and the |incr|^n drift between its every-512 renormal-
// isations), so outputs do not depend on how the stream is cut into blocks.
#include "rcf_internal.h"

namespace rcfx {

namespace {

constexpr int kWave = 64;
constexpr int kThreads = 256;
constexpr int CT = 2;   // channels per wave item
constexpr int KR = 8;   // outputs per wave item

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// GNU Radio's rotator (phase *= incr in float32, renormalised every 512 calls) in closed form: the
// float32 increment's true angle and magnitude drive a float64 model, rebased by the host every block.
__device__ __forceinline__ void rotate_store(const ChanLaunch &L, int64_t k, float vr, float vi, uint64_t ring_mask)
{
    if (k < L.k_lo || k >= L.k_lo + L.n_k || k < L.k_abs0) return;
    const int64_t n = k - L.k_abs0;
    const int64_t dk = n - L.n_seg0;
    const int64_t r512 = n & ~(int64_t)511;
    const double ang = L.angle0 + (double)dk * L.dangle;
    const double lm = (r512 > L.n_seg0) ? (double)(n - r512) * L.dlogmag : L.logmag0 + (double)dk * L.dlogmag;
    double sn, cs;
    sincos(ang, &sn, &cs);
    const double mag = exp(lm);
    const float pr = (float)(mag * cs), pi = (float)(mag * sn);
    // rotator::rotate(): z = in * phase, float32 complex multiply, unfused
    float2 y;
    y.x = __fsub_rn(__fmul_rn(vr, pr), __fmul_rn(vi, pi));
    y.y = __fadd_rn(__fmul_rn(vr, pi), __fmul_rn(vi, pr));
    L.iq_ring[(uint64_t)n & ring_mask] = y;
}

template <bool MASK>
__device__ __forceinline__ void fir_item(const float2 *xs, const ChanLaunch *__restrict__ ch, const int (&cidx)[CT],
                                         int D, int T, int o_base, int kt_n, int64_t kt0, uint64_t ring_mask,
                                         int lane)
{
    float accr[CT][KR], acci[CT][KR];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < KR; ++r) accr[c][r] = acci[c][r] = 0.f;

    const float2 *tp[CT];
    int lim[CT][KR];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        tp[c] = cidx[c] >= 0 ? ch[cidx[c]].ctaps : nullptr;
        if (MASK) {
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                int64_t l = cidx[c] >= 0 ? (kt0 + o_base + r) * (int64_t)D - ch[cidx[c]].start_sample : -1;
                lim[c][r] = l > T ? T : (l < -1 ? -1 : (int)l);
            }
        }
    }
    int roff[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        int rr = o_base + r;
        roff[r] = (rr < kt_n ? rr : kt_n - 1) * D + (T - 1);
    }

    for (int m = 0; m < T; m += kWave) {
        const int i = m + lane;
        const bool valid = i < T;
        float2 tap[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) tap[c] = (valid && tp[c]) ? tp[c][i] : make_float2(0.f, 0.f);
        const int ii = valid ? i : T - 1;
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const float2 xv = xs[roff[r] - ii];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float2 t = tap[c];
                if (MASK) {
                    if (i > lim[c][r]) t = make_float2(0.f, 0.f);
                }
                accr[c][r] = fmaf(t.x, xv.x, accr[c][r]);
                accr[c][r] = fmaf(-t.y, xv.y, accr[c][r]);
                acci[c][r] = fmaf(t.x, xv.y, acci[c][r]);
                acci[c][r] = fmaf(t.y, xv.x, acci[c][r]);
            }
        }
    }
    // cross-lane reduction; every lane ends with the totals
    float vr = 0.f, vi = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            float sr = wave_sum(accr[c][r]);
            float si = wave_sum(acci[c][r]);
            if (lane == c * KR + r) { vr = sr; vi = si; }
        }
    if (lane < CT * KR) {
        const int c = lane / KR, r = lane % KR;
        int ci = -1;
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
            if (c == cc) ci = cidx[cc];
        const int rr = o_base + r;
        if (ci >= 0 && rr < kt_n) rotate_store(ch[ci], kt0 + rr, vr, vi, ring_mask);
    }
}

// Small-T path (stage-2 FIRs on narrowband rings, e.g. D = 3, T = 11; the P25 69-tap pre-filter):
// one thread per output, taps broadcast from the scalar cache, the few overlapping input reads served
// by L1.  The lanes-over-taps kernel above would idle 53 of 64 lanes at T = 11.
__global__ __launch_bounds__(kThreads) void fir_small_kernel(const ChanLaunch *__restrict__ chans, int D, int T,
                                                             uint64_t ring_mask)
{
    const ChanLaunch &L = chans[blockIdx.y];
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= L.n_k) return;
    const int64_t k = L.k_lo + j;
    const int64_t s0 = k * (int64_t)D;
    const int64_t lim = s0 - L.start_sample;             // taps i <= lim see real samples
    const int tmax = lim + 1 < (int64_t)T ? (int)(lim + 1) : T;
    const StreamView sv = L.src;
    float ar = 0.f, ai = 0.f;
    for (int i = 0; i < tmax; ++i) {
        const float2 t = L.ctaps[i];
        const float2 xv = sv.base[(uint64_t)(s0 - i - sv.origin) & sv.mask];
        ar = fmaf(t.x, xv.x, ar);
        ar = fmaf(-t.y, xv.y, ar);
        ai = fmaf(t.x, xv.y, ai);
        ai = fmaf(t.y, xv.x, ai);
    }
    rotate_store(L, k, ar, ai, ring_mask);
}

__global__ __launch_bounds__(kThreads) void fir_bank_kernel(const ChanLaunch *__restrict__ chans, FirLaunchDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.y * d.chans_per_wg;
    const int nc = min(d.chans_per_wg, d.n_chans - c0);
    const ChanLaunch &L0 = chans[c0];
    const int tile = blockIdx.x;
    if ((int64_t)tile * d.KT >= L0.n_k) return;
    const int kt_n = min(d.KT, L0.n_k - tile * d.KT);
    const int64_t kt0 = L0.k_lo + (int64_t)tile * d.KT;
    const int64_t s_tile0 = kt0 * d.D - (d.T - 1);
    const int len = (kt_n - 1) * d.D + d.T;

    const StreamView sv = L0.src;
    for (int p = tid; p < len; p += kThreads) {
        const uint64_t idx = (uint64_t)(s_tile0 + p - sv.origin) & sv.mask;
        xs[p] = sv.base[idx];
    }
    __syncthreads();

    const int n_sub = (kt_n + KR - 1) / KR;
    const int n_cg = (nc + CT - 1) / CT;
    for (int item = wave; item < n_sub * n_cg; item += kThreads / kWave) {
        const int cg = item / n_sub, o = item - cg * n_sub;
        int cidx[CT];
        bool need_mask = false;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int ci = cg * CT + c;
            cidx[c] = ci < nc ? c0 + ci : -1;
            if (cidx[c] >= 0) {
                const int64_t first = (kt0 + (int64_t)o * KR) * d.D - chans[cidx[c]].start_sample;
                need_mask |= first < (int64_t)(d.T - 1);
            }
        }
        if (need_mask) fir_item<true>(xs, chans, cidx, d.D, d.T, o * KR, kt_n, kt0, d.ring_mask, lane);
        else           fir_item<false>(xs, chans, cidx, d.D, d.T, o * KR, kt_n, kt0, d.ring_mask, lane);
    }
}

// ---------------------------------------------------------------- discriminator
// gr::fast_atan2f: 255-interval table + linear interpolation, octant fix-up (gr-runtime fast_atan2f.cc)
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

__global__ __launch_bounds__(kThreads) void disc_kernel(const DiscLaunch *__restrict__ items, uint64_t ring_mask,
                                                        const float *__restrict__ atan_tab)
{
    __shared__ float tab[260];
    for (int i = threadIdx.x; i < 257; i += kThreads) tab[i] = atan_tab[i];
    __syncthreads();
    const DiscLaunch it = items[blockIdx.y];
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= it.n_k) return;
    const int64_t n = it.n_lo + j;
    const float2 a = it.iq_ring[(uint64_t)n & ring_mask];
    const float2 b = n > 0 ? it.iq_ring[(uint64_t)(n - 1) & ring_mask] : make_float2(0.f, 0.f);
    // volk_32fc_x2_multiply_conjugate_32fc: a * conj(b), unfused
    const float tr = __fadd_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y));
    const float ti = __fsub_rn(__fmul_rn(a.y, b.x), __fmul_rn(a.x, b.y));
    it.fm_ring[(uint64_t)n & ring_mask] = fast_atan2f_gr(ti, tr, tab);
}

// P25 symbol filter and friends: a short real FIR over gain * fm (float32, taps in order)
__global__ __launch_bounds__(kThreads) void fm_fir_kernel(const FmFirLaunch *__restrict__ items, uint64_t ring_mask)
{
    const FmFirLaunch it = items[blockIdx.y];
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= it.n_k) return;
    const int64_t n = it.n_lo + j;
    float acc = 0.f;
    for (int i = 0; i < it.ntaps; ++i) {
        const int64_t m = n - i;
        const float v = m >= it.n_first ? __fmul_rn(it.gain, it.fm_ring[(uint64_t)m & ring_mask]) : 0.f;
        acc = fmaf(it.taps[i], v, acc);
    }
    it.sym_ring[(uint64_t)n & ring_mask] = acc;
}

// drift probe: mean of gain * fm over a window (p25_control_demod.py:123-127 computes it as a 10000-sample
// moving sum times 1e-4); float64 accumulation, one workgroup
__global__ __launch_bounds__(kThreads) void fm_level_kernel(const float *__restrict__ fm_ring, int64_t n_end, int window,
                                                            float gain, uint64_t ring_mask, float *out)
{
    __shared__ double red[kThreads];
    double s = 0.0;
    for (int i = threadIdx.x; i < window; i += kThreads) {
        const int64_t m = n_end - 1 - i;
        if (m >= 0) s += (double)__fmul_rn(gain, fm_ring[(uint64_t)m & ring_mask]);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = kThreads / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(red[0] / (double)window);
}

}  // namespace

void launch_fm_fir(const FmFirLaunch *d_items, int n_items, int max_n_k, uint64_t ring_mask, hipStream_t s)
{
    if (n_items <= 0 || max_n_k <= 0) return;
    hipLaunchKernelGGL(fm_fir_kernel, dim3((max_n_k + kThreads - 1) / kThreads, n_items), dim3(kThreads), 0, s,
                       d_items, ring_mask);
}

void launch_fm_level(const float *fm_ring, int64_t n_end, int window, float gain, uint64_t ring_mask, float *d_out,
                     hipStream_t s)
{
    hipLaunchKernelGGL(fm_level_kernel, dim3(1), dim3(kThreads), 0, s, fm_ring, n_end, window, gain, ring_mask, d_out);
}

void launch_fir_bank(const ChanLaunch *d_chans, const FirLaunchDims &dims, hipStream_t s)
{
    if (dims.n_chans <= 0 || dims.max_n_k <= 0) return;
    if (dims.T <= 96 && dims.chans_per_wg == 1) {
        hipLaunchKernelGGL(fir_small_kernel, dim3((dims.max_n_k + kThreads - 1) / kThreads, dims.n_chans),
                           dim3(kThreads), 0, s, d_chans, dims.D, dims.T, dims.ring_mask);
        return;
    }
    const int tiles = (dims.max_n_k + dims.KT - 1) / dims.KT;
    const int groups = (dims.n_chans + dims.chans_per_wg - 1) / dims.chans_per_wg;
    const size_t lds = (size_t)((dims.KT - 1) * dims.D + dims.T) * sizeof(float2);
    hipLaunchKernelGGL(fir_bank_kernel, dim3(tiles, groups), dim3(kThreads), lds, s, d_chans, dims);
}

void launch_discriminator(const DiscLaunch *d_items, int n_items, int max_n_k, uint64_t ring_mask,
                          const float *d_atan_table, hipStream_t s)
{
    if (n_items <= 0 || max_n_k <= 0) return;
    hipLaunchKernelGGL(disc_kernel, dim3((max_n_k + kThreads - 1) / kThreads, n_items), dim3(kThreads), 0, s,
                       d_items, ring_mask, d_atan_table);
}

static float g_atan_tab[257];
const float *atan_table_host()
{
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; ++i) g_atan_tab[i] = (float)atan((double)i / 255.0);
        g_atan_tab[256] = (float)(3.14159265358979323846 / 4.0);
        init = true;
    }
    return g_atan_tab;
}

}  // namespace rcfx
