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
constexpr int kSmallPerThread = 1;   // outputs per thread of the small-T kernel (fir_small_outputs <= 256 n - 1)

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// sin / cos of a float64 angle of moderate size (|a| < 1e6 rad) to ~1e-16: Cody-Waite reduction by pi/2 and the
// Taylor polynomials on |x| <= pi/4.  The library sincos() carries a Payne-Hanek path and costs ~10x as much;
// with eight of them per lane it was a tenth of the matrix-core bank's time.
__device__ __forceinline__ void sincos_fast(double a, double &sn, double &cs)
{
    const double k = rint(a * 0.63661977236758134308);
    double x = fma(-k, 1.57079632679489655800e+00, a);
    x = fma(-k, 6.12323399573676603587e-17, x);
    const double x2 = x * x;
    double ps = -1.0 / 1307674368000.0;
    ps = fma(ps, x2, 1.0 / 6227020800.0);
    ps = fma(ps, x2, -1.0 / 39916800.0);
    ps = fma(ps, x2, 1.0 / 362880.0);
    ps = fma(ps, x2, -1.0 / 5040.0);
    ps = fma(ps, x2, 1.0 / 120.0);
    ps = fma(ps, x2, -1.0 / 6.0);
    const double S = fma(x * x2, ps, x);
    double pc = 1.0 / 20922789888000.0;
    pc = fma(pc, x2, -1.0 / 87178291200.0);
    pc = fma(pc, x2, 1.0 / 479001600.0);
    pc = fma(pc, x2, -1.0 / 3628800.0);
    pc = fma(pc, x2, 1.0 / 40320.0);
    pc = fma(pc, x2, -1.0 / 720.0);
    pc = fma(pc, x2, 1.0 / 24.0);
    pc = fma(pc, x2, -0.5);
    const double Cc = fma(x2, pc, 1.0);
    const int nq = (int)k & 3;
    sn = (nq & 1) ? Cc : S;
    cs = (nq & 1) ? S : Cc;
    if (nq == 1 || nq == 2) cs = -cs;
    if (nq >= 2) sn = -sn;
}

// GNU Radio's rotator (phase *= incr in float32, renormalised every 512 calls) in closed form: the
// float32 increment's true angle and magnitude drive a float64 model, rebased by the host every block.
__device__ __forceinline__ float2 rotate_value(const ChanLaunch &L, int64_t n, float vr, float vi)
{
    const int64_t dk = n - L.n_seg0;
    const int64_t r512 = n & ~(int64_t)511;
    const double ang = L.angle0 + (double)dk * L.dangle;
    const double lm = (r512 > L.n_seg0) ? (double)(n - r512) * L.dlogmag : L.logmag0 + (double)dk * L.dlogmag;
    double sn, cs;
    sincos_fast(ang, sn, cs);
    // |lm| is a few hundred times log|incr| ~ 1e-7: four series terms are exact to double rounding
    const double mag = fabs(lm) < 1e-3 ? 1.0 + lm * (1.0 + lm * (0.5 + lm * (1.0 / 6.0))) : exp(lm);
    const float pr = (float)(mag * cs), pi = (float)(mag * sn);
    // rotator::rotate(): z = in * phase, float32 complex multiply, unfused
    float2 y;
    y.x = __fsub_rn(__fmul_rn(vr, pr), __fmul_rn(vi, pi));
    y.y = __fadd_rn(__fmul_rn(vr, pi), __fmul_rn(vi, pr));
    return y;
}

__device__ __forceinline__ void rotate_store(const ChanLaunch &L, int64_t k, float vr, float vi, uint64_t ring_mask)
{
    if (k < L.k_lo || k >= L.k_lo + L.n_k || k < L.k_abs0) return;
    const int64_t n = k - L.k_abs0;
    L.iq_ring[(uint64_t)n & ring_mask] = rotate_value(L, n, vr, vi);
}

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

// Small-T path (stage-2 FIRs on narrowband rings, e.g. D = 3, T = 11; the P25 69-tap pre-filter): one thread per
// output (the lanes-over-taps kernel above would idle 53 of 64 lanes at T = 11).  A workgroup stages the
// KB D + T input samples of its KB outputs (plus the output just before them) in LDS with coalesced loads,
// taps come from the scalar cache, and the FM discriminator is fused in: outputs meet their predecessor in LDS,
// thread 0 recomputes the one that belongs to the previous workgroup.  One launch and one pass over the channel
// stream instead of two (the stage-2 FIR + discriminator pair was 22 % of the bench step).
__global__ __launch_bounds__(kThreads) void fir_small_kernel(const ChanLaunch *__restrict__ chans, int D, int T, int KB,
                                                             uint64_t ring_mask, const float *__restrict__ atan_tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);          // KB D + T samples
    float2 *ys = xs + (size_t)KB * D + T;                        // KB + 1 outputs: ys[t] = y[k0 - 1 + t]
    float2 *cts = ys + KB + 1;                                   // T composite taps
    float *tab = reinterpret_cast<float *>(cts + T);             // 257 + pad
    const ChanLaunch &L = chans[blockIdx.y];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * KB;
    if (j0 >= L.n_k) return;
    const int nj = min(KB, L.n_k - j0);
    for (int i = tid; i < 257; i += kThreads) tab[i] = atan_tab[i];
    for (int i = tid; i < T; i += kThreads) cts[i] = L.ctaps[i];
    // samples (k0 - 1) D - (T - 1) .. (k0 + nj - 1) D, zero before the channel's start (GR zero history)
    const int64_t k0 = L.k_lo + j0;
    const int64_t s_first = (k0 - 1) * (int64_t)D - (T - 1);
    const int len = nj * D + T;
    const StreamView sv = L.src;
    // all of a thread's tile loads are issued before the first LDS store: a load -> store loop exposes the full
    // memory latency once per iteration (measured: that, not arithmetic, was this kernel's time)
    constexpr int LU = 12;
    for (int p0 = tid; p0 < len; p0 += kThreads * LU) {
        float2 v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = p0 + u * kThreads;
            const int64_t sidx = s_first + (p < len ? p : len - 1);
            v[u] = sidx >= L.start_sample ? sv.base[(uint64_t)(sidx - sv.origin) & sv.mask] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = p0 + u * kThreads;
            if (p < len) xs[p] = v[u];
        }
    }
    __syncthreads();
    // slot j = tid + 256 o holds y[k0 - 1 + j]: j = 0 is the predecessor the discriminator needs (it belongs to the
    // previous workgroup or launch and is only recomputed, not stored), j = 1 .. nj are this workgroup's outputs.
    // Up to kSmallPerThread outputs per thread: the whole launch then fits the GPU in one or two waves of
    // workgroups, and this latency-bound kernel costs about one load -> FIR -> store chain per wave.
    float2 y[kSmallPerThread];
#pragma unroll
    for (int o = 0; o < kSmallPerThread; ++o) {
        const int j = tid + o * kThreads;
        const int64_t n = k0 - 1 + j - L.k_abs0;               // relative output index
        y[o] = make_float2(0.f, 0.f);
        if (j <= nj && n >= 0) {
            const float2 *w = xs + (size_t)j * D + (T - 1);     // x[(k0 - 1 + j) D - i] = w[-i]
            float ar = 0.f, ai = 0.f;
            for (int i = 0; i < T; ++i) {
                const float2 c = cts[i];
                const float2 xv = w[-i];
                ar = fmaf(c.x, xv.x, ar);
                ar = fmaf(-c.y, xv.y, ar);
                ai = fmaf(c.x, xv.y, ai);
                ai = fmaf(c.y, xv.x, ai);
            }
            y[o] = rotate_value(L, n, ar, ai);
            if (j >= 1) L.iq_ring[(uint64_t)n & ring_mask] = y[o];
        }
        if (j <= nj) ys[j] = y[o];                               // n < 0: quadrature_demod's zero history
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < kSmallPerThread; ++o) {
        const int j = tid + o * kThreads;
        if (j >= 1 && j <= nj) {
            const float2 b = ys[j - 1];
            // volk_32fc_x2_multiply_conjugate_32fc: a * conj(b), unfused
            const float tr = __fadd_rn(__fmul_rn(y[o].x, b.x), __fmul_rn(y[o].y, b.y));
            const float ti = __fsub_rn(__fmul_rn(y[o].y, b.x), __fmul_rn(y[o].x, b.y));
            L.fm_ring[(uint64_t)(k0 - 1 + j - L.k_abs0) & ring_mask] = fast_atan2f_gr(ti, tr, tab);
        }
    }
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

// ---------------------------------------------------------------- matrix-core bank
// For >= 8 channels on one source the bank is a dense real contraction
//   [Re y_c ; Im y_c][k] = sum_tap [ cr  -ci ; ci  cr ]_c[tap] . [Re x ; Im x][k D - tap]
// and runs on the FP32 matrix cores (v_mfma_f32_16x16x4_f32: a sequential float32 FMA chain, i.e. the same
// arithmetic as the vector kernel above, without one LDS read and four VALU issues per complex MAC):
//   M = 16 rows   = 8 channels x {Re y, Im y}
//   N = 16 cols   = 16 consecutive outputs k
//   K = 4 per op  = 2 taps x {Re x, Im x}
// A (taps) streams from the launch's bank matrix (rcf_internal.h), stored in MFMA lane order with the
// [cr -ci; ci cr] signs applied: one fully coalesced 16-byte buffer load per lane covers four ops.  B (samples) is one ds_read_b32 per op from the LDS tile, whose rows of D
// samples are skewed by one sample when D is even so that the 16 columns (stride D) fall in 16 different banks.
// The 16-output tile is (15 D + T) samples = 119 KB at D = 800, T = 2909, so ONE workgroup owns a CU.  It runs
// 8 waves = (items of 32 channels: MT = 4 M-tiles x one N-tile, four accumulator sets per tile) x (parts of the tap range): 4 x 2 for a full 128-channel workgroup, 1 x 8 for a bank of up to 32 channels.
// A wave's own VALU / LDS / VMEM issue does not overlap its MFMAs (measured: additive), the second wave on each
// SIMD is what fills the matrix pipe meanwhile.  The parts add their partial sums through the (by then dead) tile
// memory; a lane holds complete (Re, Im) pairs of two channels for one output, so the rotator and the ring store
// need no cross-lane step.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1))) *gv4;
constexpr int MT = 4;
constexpr int kThreadsM = 512;
constexpr int kMfmaExchangeBytes = (kThreadsM / kWave) * MT * 4 * kWave * 4;   // 32 KB

__global__ __launch_bounds__(kThreadsM) void fir_mfma_kernel(const ChanLaunch *__restrict__ chans, FirLaunchDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *xf = reinterpret_cast<float *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.y * d.chans_per_wg;
    const int nc = min(d.chans_per_wg, d.n_chans - c0);
    const ChanLaunch &L0 = chans[c0];
    const int tile = blockIdx.x;
    const int KTM = d.KT;                               // outputs per tile: 16, or 8 when 16 do not fit the LDS
    if ((int64_t)tile * KTM >= L0.n_k) return;
    const int kt_n = min(KTM, L0.n_k - tile * KTM);
    const int64_t kt0 = L0.k_lo + (int64_t)tile * KTM;
    const int64_t s_tile0 = kt0 * d.D - (d.T - 1);
    const int len = (kt_n - 1) * d.D + d.T;
    const int delta = (d.D & 1) ? 0 : 1;
    const int Dp = d.D + delta;

    // tile load, 8 independent 8-byte loads in flight per thread (one workgroup per CU: nothing else hides HBM)
    const StreamView sv = L0.src;
    const unsigned magic = 0xffffffffu / (unsigned)d.D + 1;   // floor(p / D) = umulhi(p, magic) for p D < 2^32 / D
    constexpr int LU = 8;
    for (int p0 = tid; p0 < len; p0 += kThreadsM * LU) {
        float2 v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = p0 + u * kThreadsM;
            const uint64_t idx = (uint64_t)(s_tile0 + (p < len ? p : len - 1) - sv.origin) & sv.mask;
            v[u] = sv.base[idx];
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = p0 + u * kThreadsM;
            if (p < len) reinterpret_cast<float2 *>(xf)[p + (int)__umulhi((unsigned)p, magic) * delta] = v[u];
        }
    }
    __syncthreads();

    const int j = lane & 15, kap = lane >> 4;
    const int q = kap & 1, tp = kap >> 1;
    const int jj = j < kt_n ? j : kt_n - 1;
    const int lbase = jj * Dp * 2 + q;
    const int n_steps = bank_steps(d.T);
    // 8 waves = (items of 32 channels) x (tap parts): 4 x 2 for 97+ channels, 2 x 4 for 33..64, 1 x 8 up to 32 --
    // a small bank spreads its taps over all eight waves instead of leaving six of them idle
    const int items_p2 = nc > 64 ? 4 : (nc > 32 ? 2 : 1);
    const int n_parts = (kThreadsM / kWave) / items_p2;
    const int item = wave % items_p2, part = wave / items_p2;
    const int cw0 = c0 + item * 8 * MT;
    const bool active = item * 8 * MT < nc;

    // four accumulator sets per M-tile (one per op of a step): covers the MFMA latency and keeps each float32
    // summation chain a quarter of the part's taps long (rounding noise ~ sqrt(chain length))
    v4f acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = (v4f){0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int per = ((n_steps / 4 + n_parts - 1) / n_parts) * 4;     // steps per part, a multiple of the 4-step trip
        const int step0 = min(part * per, n_steps), step1 = min(step0 + per, n_steps);
        // A operand: one buffer descriptor over the bank matrix, lane offset in a VGPR, (tile, step) offset in an
        // SGPR -- no vector arithmetic per load
        const __amdgpu_buffer_rsrc_t bank_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(d.bank), 0, (int)(bank_floats(d.n_chans, d.T) * sizeof(float)), 0x00020000);
        int a_soff[MT];                                // byte offset of (tile, next step to fetch)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            int g = (cw0 >> 3) + t;
            if (g * 8 >= d.n_chans) g = (d.n_chans - 1) >> 3;   // dead tiles: computed, never stored
            a_soff[t] = (g * n_steps + step0) * 1024;
        }
        const int a_voff = lane * 16;
        // B operand: x[k D - tap] sits at LDS sample position rr + floor(rr / D) delta, rr = (T-1-tp) - 2 op.
        // Per step (4 ops, rr spans top-7 .. top with top = T-1-8 step) the quotient is one wave-uniform value
        // except in the few steps that straddle a multiple of D: track top's quotient / remainder in scalars.
        const int lconst = (lbase + 2 * (d.T - 1 - tp)) * 4;
        int top = d.T - 1 - 8 * step0;
        int topq = top >= 0 ? top / d.D : 0;
        int toprem = top >= 0 ? top - topq * d.D : 0;
        auto fetch_b = [&](float (&b)[4]) {
            if (top >= 7 && toprem >= 7) {                     // clean step: one add, four reads at fixed offsets
                const float *pb = reinterpret_cast<const float *>(
                    smem_raw + (lconst + 8 * (topq * delta + top - (d.T - 1)) - 48));
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = pb[(48 - 16 * u) / 4];
            } else if (top < 0) {                              // padding taps only (zero coefficients)
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = xf[lbase];
            } else {                                           // straddles a multiple of D, or runs into padding
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = top - tp - 2 * u;
                    const int qd = rr >= topq * d.D ? topq : topq - 1;
                    b[u] = xf[lbase + 2 * (rr < 0 ? 0 : rr + qd * delta)];
                }
            }
            top -= 8;
            toprem -= 8;
            if (toprem < 0) { toprem += d.D; topq -= 1; }      // D >= 8 (mfma_tile_bytes)
        };
        // Four steps per trip.  A sets rotate through four register groups and are requested three steps before
        // use, B comes from LDS one step ahead.  sched_barrier pins the issue order so the counter waits land on
        // the first use and not on the issue.  Everything that is not an MFMA costs its issue slot on this SIMD
        // (measured: VALU time adds to matrix time, also across the two waves of a SIMD), hence the effort above
        // to keep a step at 16 MFMAs + 4 loads + 4 LDS reads + one add.
        v4f a0[MT], a1[MT], a2[MT], a3[MT];
        float b0[4], b1[4];
        auto fetch_a = [&](v4f (&a)[MT]) {               // fetches past the end of the bank return zeros (descriptor)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                a[t] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(bank_rsrc, a_voff, a_soff[t], 0));
                a_soff[t] += 1024;
            }
        };
        auto mac = [&](const v4f (&a)[MT], const float (&b)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u], b[u], acc[t][u], 0, 0, 0);
        };
        __builtin_amdgcn_sched_barrier(0);            // same issue order as the loop body, or the loop-top waits
        fetch_a(a0);                                  // are sized for the worse of the two predecessors
        __builtin_amdgcn_sched_barrier(0);
        fetch_a(a1);
        __builtin_amdgcn_sched_barrier(0);
        fetch_a(a2);
        __builtin_amdgcn_sched_barrier(0);
        fetch_b(b0);
        __builtin_amdgcn_sched_barrier(0);
        for (int m4 = step0; m4 < step1; m4 += 4) {   // every part's step count is a multiple of 4
            fetch_a(a3); fetch_b(b1);
            __builtin_amdgcn_sched_barrier(0);
            mac(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(a0); fetch_b(b0);
            __builtin_amdgcn_sched_barrier(0);
            mac(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(a1); fetch_b(b1);
            __builtin_amdgcn_sched_barrier(0);
            mac(a2, b0);
            __builtin_amdgcn_sched_barrier(0);
            fetch_a(a2); fetch_b(b0);
            __builtin_amdgcn_sched_barrier(0);
            mac(a3, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // every wave parks its partial sums (4 M-tiles x 4 accumulator registers) in the now dead sample tile; M-tile t of
    // an item is finished by that item's tap part t % n_parts, which adds the parts in order 0 .. n_parts - 1
    __syncthreads();                                  // every wave is done with the sample tile
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const v4f sum = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[((wave * MT + t) * 4 + e) * kWave + lane] = sum[e];
    }
    __syncthreads();
    if (!active || j >= kt_n) return;
    for (int t = part % MT; t < MT; t += n_parts) {
        if (part >= MT) break;                        // 8 parts, 4 tiles: parts 4 .. 7 have nothing to finish
        v4f tot = (v4f){0.f, 0.f, 0.f, 0.f};
        for (int pp = 0; pp < n_parts; ++pp) {
            const int w2 = pp * items_p2 + item;
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[e] += xf[((w2 * MT + t) * 4 + e) * kWave + lane];
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int ci = cw0 + t * 8 + 2 * kap + hh;
            if (ci < c0 + nc) rotate_store(chans[ci], kt0 + j, tot[2 * hh], tot[2 * hh + 1], d.ring_mask);
        }
    }
}

__global__ __launch_bounds__(kThreads) void fir_pack_kernel(const ChanLaunch *__restrict__ chans, int n_chans, int T,
                                                            int n_steps, float *__restrict__ bank)
{
    const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= bank_floats(n_chans, T)) return;
    const int mm = e & 3, lane = (e >> 2) & 63;
    const size_t gs = e >> 8;
    const int step = (int)(gs % n_steps), g = (int)(gs / n_steps);
    const int j = lane & 15, kap = lane >> 4;
    const int c = j >> 1, r = j & 1, tp = kap >> 1, q = kap & 1;
    const int tap = 2 * (4 * step + mm) + tp, ci = g * 8 + c;
    float v = 0.f;
    if (ci < n_chans && tap < T) {
        const float2 ct = chans[ci].ctaps[tap];
        v = (r == q) ? ct.x : (r == 0 ? -ct.y : ct.y);       // [cr -ci; ci cr]
    }
    bank[e] = v;
}

// ---------------------------------------------------------------- discriminator
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

void launch_fir_pack(const ChanLaunch *d_chans, int n_chans, int T, float *bank, hipStream_t s)
{
    const size_t n = bank_floats(n_chans, T);
    hipLaunchKernelGGL(fir_pack_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, d_chans,
                       n_chans, T, bank_steps(T), bank);
}

void launch_fir_bank(const ChanLaunch *d_chans, const FirLaunchDims &dims, hipStream_t s)
{
    if (dims.n_chans <= 0 || dims.max_n_k <= 0) return;
    if (dims.small) {
        const int KB = fir_small_outputs(dims.D, dims.T);
        const size_t lds = ((size_t)KB * dims.D + 2 * dims.T + KB + 1) * sizeof(float2) + 264 * sizeof(float);
        hipLaunchKernelGGL(fir_small_kernel, dim3((dims.max_n_k + KB - 1) / KB, dims.n_chans), dim3(kThreads), lds, s,
                           d_chans, dims.D, dims.T, KB, dims.ring_mask, dims.atan_tab);
        return;
    }
    if (dims.mfma) {
        static size_t attr_lds = 0;
        size_t lds = mfma_tile_bytes(dims.D, dims.T);
        if (lds < (size_t)kMfmaExchangeBytes) lds = kMfmaExchangeBytes;
        if (lds > attr_lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fir_mfma_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_lds = lds;
        }
        const int groups = (dims.n_chans + dims.chans_per_wg - 1) / dims.chans_per_wg;
        FirLaunchDims md = dims;
        md.KT = mfma_tile_outputs(dims.D, dims.T);
        hipLaunchKernelGGL(fir_mfma_kernel, dim3((dims.max_n_k + md.KT - 1) / md.KT, groups), dim3(kThreadsM), lds, s,
                           d_chans, md);
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
