// fir_small.hpp -- the small-T decimating FIR tile with the discriminator fused in (stage-2 channels on narrowband rings),
// and gr::fast_atan2f.  A header because TWO kernels run it: fir_small_kernel (fir.hip) and the stage-2 rider of the
// power-of-two filterbank's launch (pfb.hip: the stage-2 work of block n rides in block n + 1's filterbank launch).
// fir.hip is built without implicit FMA contraction, pfb.hip is not: every function here (and in rotator.hpp) switches
// contraction off in its own body, so that both kernels round alike -- GNU Radio's unfused float32 arithmetic.
#pragma once
#include "rcf_internal.h"
#include "rotator.hpp"
#include "fast_atan2f_gr.hpp"

namespace rcfx {

namespace {

constexpr int kSmallThreads = 256;
constexpr int kSmallPerThread = 2;   // outputs per thread of the small-T kernel (fir_small_outputs <= 256 n - 1): 1 -> 2 took the stage-2 launch of the timed configuration from 0.030 to 0.025 ms, 3 and 4 are not faster

// gr::fast_atan2f: fast_atan2f_gr.hpp (a header of its own: tests/test_device_atan_cpu.py compiles it for the host and holds it
// bit for bit to the oracle's branch form)


// Small-T path (stage-2 FIRs on narrowband rings, e.g. D = 3, T = 11; the P25 69-tap pre-filter): one thread per
// output (the lanes-over-taps kernel above would idle 53 of 64 lanes at T = 11).  A workgroup stages the
// KB D + T input samples of its KB outputs (plus the output just before them) in LDS with coalesced loads,
// taps come from the scalar cache, and the FM discriminator is fused in: outputs meet their predecessor in LDS,
// thread 0 recomputes the one that belongs to the previous workgroup (the previous LAUNCH's is read back from the ring).  One launch and one pass over the channel
// stream instead of two (the stage-2 FIR + discriminator pair was 22 % of the bench step).
// PT: outputs per thread the tile may hold (KB <= 256 PT - 1)
template <int PT = kSmallPerThread>
__device__ __forceinline__ void fir_small_tile(const ChanLaunch *__restrict__ chans, const int chan_idx, const int tile_idx,
                                               const int D, const int T, const int KB, const uint64_t ring_mask,
                                               const float *__restrict__ atan_tab, unsigned char *smem_raw)
{
#pragma clang fp contract(off)
    constexpr int kThreads = kSmallThreads;
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);          // KB D + T samples
    float2 *ys = xs + (size_t)KB * D + T;                        // KB + 1 outputs: ys[t] = y[k0 - 1 + t]
    float2 *cts = ys + KB + 1;                                   // T composite taps
    float *tab = reinterpret_cast<float *>(cts + T);             // 257 + pad
    // grid = (channels, output tiles): workgroups that run together work on the SAME stretch of time of different
    // channels -- when the channels are bins of one filterbank ring (tiled or frame-major) their lines sit in the
    // same tiles, i.e. the same pages
    const ChanLaunch L = chans[chan_idx];     // by value: ONE batch of scalar loads (a reference is re-read after every global store -- 25 serialized scalar-memory waits per wave)
    const int tid = threadIdx.x;
    const int j0 = tile_idx * KB;
    if (j0 >= L.n_k) return;
    const int nj = min(KB, L.n_k - j0);
    // samples (k0 - 1) D - (T - 1) .. (k0 + nj - 1) D, zero before the channel's start (GR zero history)
    const int64_t k0 = L.k_lo + j0;
    const int64_t s_first = (k0 - 1) * (int64_t)D - (T - 1);
    const int len = nj * D + T;
    const StreamView sv = L.src;
    // EVERY load of the workgroup's prologue -- the arctangent table, the composite taps, the first LU samples per
    // thread -- is issued before the first LDS store: table -> taps -> samples as three load -> store loops were three
    // memory latencies in a row in a kernel that is nothing but a chain of them (and a load -> store loop over the
    // samples exposes the latency once per iteration: measured, that, not arithmetic, was this kernel's time)
    constexpr int LU = 4 * PT;  // the stage-2 shape of the timed configuration (D = 3, T = 11: 1544 samples at 511 outputs) needs 7 per thread
    static_assert(kThreads == 256, "one table entry per thread, the 257th on thread 0");
    const float tab_a = atan_tab[tid], tab_b = atan_tab[256];
    const float2 ct_a = tid < T ? L.ctaps[tid] : make_float2(0.f, 0.f);
    {
        float2 v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = tid + u * kThreads;
            const int64_t sidx = s_first + (p < len ? p : len - 1);
            v[u] = sidx >= L.start_sample ? sv.base[sv.at(sidx)] : make_float2(0.f, 0.f);
        }
        tab[tid] = tab_a;
        if (tid == 0) tab[256] = tab_b;
        if (tid < T) cts[tid] = ct_a;
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = tid + u * kThreads;
            if (p < len) xs[p] = v[u];
        }
    }
    for (int i = tid + kThreads; i < T; i += kThreads) cts[i] = L.ctaps[i];
    for (int p0 = tid + kThreads * LU; p0 < len; p0 += kThreads * LU) {
        float2 v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int p = p0 + u * kThreads;
            const int64_t sidx = s_first + (p < len ? p : len - 1);
            v[u] = sidx >= L.start_sample ? sv.base[sv.at(sidx)]
                                          : make_float2(0.f, 0.f);
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
    // Up to kSmallPerThread outputs per thread: the whole launch then fits the GPU in one or two rounds of
    // workgroups.  (Three per thread with the tile chosen so that the launch is exactly ONE round of resident
    // workgroups -- 32 x 64 of 683 outputs instead of 32 x 86 of 511 -- measured the same, 19.7 against 19.2 us:
    // it is the memory system's rate for 128-byte pieces, not the rounds.)
    float2 y[PT];
#pragma unroll
    for (int o = 0; o < PT; ++o) {
        const int j = tid + o * kThreads;
        const int64_t n = k0 - 1 + j - L.k_abs0;               // relative output index
        y[o] = make_float2(0.f, 0.f);
        if (j == 0 && j0 == 0) {
            // the launch's first output meets the output the PREVIOUS launch stored, not a recomputation of it: after a
            // retune the taps and the rotator increment in force now are not the ones that made it (quadrature_demod
            // sees the stream as it was emitted -- found by tests/test_gpu_fuzz.py: one discriminator sample per retune)
            if (n >= 0) y[o] = L.iq_ring[(uint64_t)n & ring_mask];
        } else if (j <= nj && n >= 0) {
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
            // (ring stores non-temporal: 20.6 -> 19.7 us for the timed configuration's 32 channels)
            if (j >= 1) { typedef float v2f_ __attribute__((ext_vector_type(2))); v2f_ o_; o_.x = y[o].x; o_.y = y[o].y;
                          __builtin_nontemporal_store(o_, reinterpret_cast<v2f_ *>(L.iq_ring + ((uint64_t)n & ring_mask))); }
        }
        if (j <= nj) ys[j] = y[o];                               // n < 0: quadrature_demod's zero history
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < PT; ++o) {
        const int j = tid + o * kThreads;
        if (j >= 1 && j <= nj) {
            const float2 b = ys[j - 1];
            // volk_32fc_x2_multiply_conjugate_32fc: a * conj(b), unfused
            const float tr = (y[o].x * b.x) + (y[o].y * b.y);
            const float ti = (y[o].y * b.x) - (y[o].x * b.y);
            __builtin_nontemporal_store(fast_atan2f_gr(ti, tr, tab), L.fm_ring + ((uint64_t)(k0 - 1 + j - L.k_abs0) & ring_mask));
        }
    }
}


}  // namespace

}  // namespace rcfx
