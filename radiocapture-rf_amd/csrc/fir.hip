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
#include <cstdlib>

#include "rcf_internal.h"
#include "rotator.hpp"
#include "fir_small.hpp"

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

// Small-T path: fir_small_tile (fir_small.hpp), one (channel, tile) per workgroup
__global__ __launch_bounds__(kThreads) void fir_small_kernel(const ChanLaunch *__restrict__ chans, int D, int T, int KB,
                                                             uint64_t ring_mask, const float *__restrict__ atan_tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    fir_small_tile(chans, blockIdx.x, blockIdx.y, D, T, KB, ring_mask, atan_tab, smem_raw);
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
        xs[p] = sv.base[sv.at(s_tile0 + p)];
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
// Operand plan (rcf_internal.h, "bank2"):
//   * workgroup = 4 waves = ONE group of 32 channels x (4 x NT) tiles of 16 consecutive outputs; two workgroups
//     per CU (64 KB of LDS each), so one's prologue / epilogue / barrier waits are covered by the other's MFMAs;
//   * the group's taps (A operand) stream through LDS in 32 KB chunks of 8 steps (64 taps), double buffered,
//     fetched ONE chunk ahead with coalesced 16-byte loads and shared by the four waves: 32 B of L2 traffic per
//     MFMA instead of 256;
//   * the samples (B operand) come straight from the wideband buffer (L2 / Infinity Cache): lane (kap, j) needs
//     x[(k0 + j) D - 2q], x[.. + 1] for pair q -- ONE 16-byte load feeds its four ops of a step, the four kap lanes
//     of an output read one 64-byte run, a step later the next 64 bytes.  Steps run from the highest taps down so
//     the addresses ascend and a chunk's eight steps are immediate offsets of one VGPR.
//   * a wave keeps 4 x NT accumulator tiles (32 channels x 16 NT outputs): each A register feeds NT MFMAs, each B
//     register four -- 6 operand registers per 32 MFMAs at NT = 2.
// There is no sample tile in LDS: no D-dependent LDS limit (6.25 kHz channels at 20 Msps run full 16-wide tiles),
// no serial tile-load phase, and no cross-wave reduction.  (The first version of this kernel kept a 16-output
// sample tile in LDS -- 119 KB at D = 800, T = 2909, one workgroup per CU -- and streamed the taps from L2 into
// registers: 256 B of L2 traffic per MFMA and a serial tile load + reduction per workgroup; 113 TF at 4096
// channels against 128 here.)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int MT = 4;
constexpr int kM2Threads = 256;

struct MfmaArgs {
    const float *bank;
    int64_t src_len;
    uint64_t ring_mask;
    float2 *partial;         // n_parts > 1: [part][channel][n_k] partial sums, finished by fir_mfma_finish_kernel
    int D, T, n_chans, n_groups, n_wt, n_parts, n_k;
};

template <int NT, int PD>
__global__ __launch_bounds__(kM2Threads, 2) void fir_mfma_kernel(const ChanLaunch *__restrict__ chans, MfmaArgs d)
{
    constexpr int CS = kM2ChunkSteps;
    constexpr int kM2ChunkBytes = CS * 4096;
    constexpr int NLD = CS * 4096 / (kM2Threads * 16);      // 16-byte loads per thread per chunk
    constexpr int NB = PD + 1;                              // B register stages: loads run PD steps ahead
    static_assert(CS % NB == 0, "stage rotation must close over a chunk");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // groups fastest: the workgroups resident together mostly share their output tiles, i.e. their samples, through
    // L2 (each streams its own taps once) -- measured 2-4 % better than tiles fastest
    const int g = blockIdx.x % d.n_groups;
    const int part = (blockIdx.x / d.n_groups) % d.n_parts, wt = blockIdx.x / (d.n_groups * d.n_parts);
    const ChanLaunch &L0 = chans[g * kM2Group];
    const int n_k = L0.n_k;
    const int k_rel0 = (wt * (kM2Threads / kWave) + wave) * (NT * 16);
    const int j = lane & 15, kap = lane >> 4;
    const int NS = bank2_steps(d.T);
    const int NC_all = NS / CS;
    const int c_first = (part * NC_all) / d.n_parts;         // this workgroup's range of tap chunks
    const int NC = ((part + 1) * NC_all) / d.n_parts - c_first;

    const StreamView sv = L0.src;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2 *>(sv.base), 0, (int)(d.src_len * (int64_t)sizeof(float2)), 0x00020000);
    int voff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int kr = k_rel0 + n * 16 + j;
        if (kr > n_k - 1) kr = n_k - 1;                     // past the block: recompute the last output, never stored
        if (kr < 0) kr = 0;
        const int64_t sidx = (L0.k_lo + kr) * (int64_t)d.D - sv.origin - 8 * (NS - 1) - 2 * kap + 8 * CS * c_first;
        voff[n] = (int)(sidx * (int64_t)sizeof(float2));
    }
    const float *bank_g = d.bank + (size_t)g * bank2_group_floats(d.T) + (size_t)c_first * (kM2ChunkBytes / 4);

    v4f acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = (v4f){0.f, 0.f, 0.f, 0.f};
    v4f a[2][MT], b[NB][NT], stg[NLD];

    auto ldA = [&](int c) {                                  // chunk c: 32 KB, 8 coalesced 16-byte loads per thread
        const v4f *src = reinterpret_cast<const v4f *>(bank_g + (size_t)c * (kM2ChunkBytes / 4)) + tid;
#pragma unroll
        for (int r = 0; r < NLD; ++r) stg[r] = src[r * kM2Threads];
    };
    auto stA = [&](int buf) {
        v4f *dst = reinterpret_cast<v4f *>(smem_raw + buf * kM2ChunkBytes) + tid;
#pragma unroll
        for (int r = 0; r < NLD; ++r) dst[r * kM2Threads] = stg[r];
    };
    auto rdA = [&](int buf, int i, v4f (&dst)[MT]) {
        const v4f *src = reinterpret_cast<const v4f *>(smem_raw + buf * kM2ChunkBytes + i * 4096) + lane;
#pragma unroll
        for (int t = 0; t < MT; ++t) dst[t] = src[t * kWave];
    };
    // step i of the current chunk (i >= CS: the next chunk's step i - CS -- the addresses simply continue)
    auto ldB = [&](int i, v4f (&dst)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
            dst[n] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, voff[n] + 64 * i, 0, 0));
    };
    auto mac = [&](const v4f (&aa)[MT], const v4f (&bb)[NT]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[t][u], bb[n][u], acc[t][n], 0, 0, 0);
    };

    // Pipeline.  Samples: PD steps ahead (they come from L2 / Infinity Cache / HBM -- a miss under load is several
    // thousand cycles, a step is ~1000-2000).  Taps: chunk c + 1 is requested at the top of chunk c, parked in
    // LDS at step CS - 3 (the other buffer: everybody left it at the previous chunk's barrier), published by ONE
    // barrier after step CS - 2, so that step CS - 1 prefetches its first operands like any other step.  The memory
    // counter is in order: a wave waiting for a step's samples also waits for every older load, so the tap loads'
    // latency is covered by the same PD steps.
    ldA(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) ldB(i, b[i]);
    stA(0);
    __syncthreads();
    rdA(0, 0, a[0]);
    for (int c = 0; c < NC; ++c) {
        const int buf = c & 1;
        __builtin_amdgcn_sched_barrier(0);
        ldA(c + 1);                       // past the last chunk: one chunk of slack behind the bank, never used
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CS; ++i) {
            ldB(i + PD, b[(i + PD) % NB]);   // past the last chunk: in range or zero (buffer descriptor), never used
            if (i + 1 < CS) rdA(buf, i + 1, a[(i + 1) & 1]);
            else            rdA(buf ^ 1, 0, a[0]);
            if (i == CS - 3) stA(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            mac(a[i & 1], b[i % NB]);
            __builtin_amdgcn_sched_barrier(0);
            if (i == CS - 2) __syncthreads();
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) voff[n] += 64 * CS;
    }

    // C/D layout: lane (kap, j) holds rows 4 kap .. 4 kap + 3 = (Re, Im) of channels 2 kap, 2 kap + 1 for output j
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int kr = k_rel0 + n * 16 + j;
        if (kr >= n_k) continue;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int ci = g * kM2Group + t * 8 + 2 * kap + hh;
                if (ci >= d.n_chans) continue;
                if (d.n_parts == 1)
                    rotate_store(chans[ci], L0.k_lo + kr, acc[t][n][2 * hh], acc[t][n][2 * hh + 1], d.ring_mask);
                else                                           // 16 lanes j = 128 contiguous bytes of the slab
                    d.partial[((size_t)part * d.n_chans + ci) * d.n_k + kr] =
                        make_float2(acc[t][n][2 * hh], acc[t][n][2 * hh + 1]);
            }
    }
}

// split-K finish: y = rotator * (part 0 + part 1 + ...), the parts added in order (deterministic)
__global__ __launch_bounds__(kThreads) void fir_mfma_finish_kernel(const ChanLaunch *__restrict__ chans,
                                                                   const float2 *__restrict__ partial, int n_chans,
                                                                   int n_k, int n_parts, uint64_t ring_mask)
{
    const int kr = blockIdx.x * kThreads + threadIdx.x;
    const int ci = blockIdx.y;
    if (kr >= n_k) return;
    const size_t at = (size_t)ci * n_k + kr, slab = (size_t)n_chans * n_k;
    float2 s = partial[at];
    for (int p = 1; p < n_parts; ++p) {
        const float2 v = partial[at + p * slab];
        s.x = __fadd_rn(s.x, v.x);
        s.y = __fadd_rn(s.y, v.y);
    }
    const ChanLaunch &L = chans[ci];
    rotate_store(L, L.k_lo + kr, s.x, s.y, ring_mask);
}

__global__ __launch_bounds__(kThreads) void fir_pack_kernel(const ChanLaunch *__restrict__ chans, int n_chans, int T,
                                                             int NS, float *__restrict__ bank,
                                                             const unsigned char *__restrict__ dirty)
{
    const int g = blockIdx.y;
    if (dirty && !dirty[g]) return;
    const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= (size_t)NS * 1024) return;
    const int u = e & 3, lane = (e >> 2) & 63, t = (e >> 8) & 3;
    const int p = (int)(e >> 10);
    const int kap = lane >> 4, row = lane & 15, c = row >> 1, r = row & 1;
    const int q = 4 * (NS - 1 - p) + kap;
    const int tap = (u < 2) ? 2 * q : 2 * q - 1, im = u & 1;
    const int ci = g * kM2Group + t * 8 + c;
    float v = 0.f;
    if (ci < n_chans && tap >= 0 && tap < T) {
        const float2 ct = chans[ci].ctaps[tap];
        v = im ? (r == 0 ? -ct.y : ct.x) : (r == 0 ? ct.x : ct.y);     // [cr -ci; ci cr][r][re|im]
    }
    bank[(size_t)g * NS * 1024 + e] = v;
}

// ---------------------------------------------------------------- discriminator
constexpr int kDiscPerThread = 4;      // outputs per thread: the 1 KB table a workgroup stages is then a quarter of
                                       // its traffic instead of as much as its data
__global__ __launch_bounds__(kThreads) void disc_kernel(const DiscLaunch *__restrict__ items, uint64_t ring_mask,
                                                        const float *__restrict__ atan_tab)
{
    __shared__ float tab[260];
    const DiscLaunch it = items[blockIdx.y];
    const int j0 = blockIdx.x * (kThreads * kDiscPerThread) + threadIdx.x;
    if (blockIdx.x * (kThreads * kDiscPerThread) >= it.n_k) return;
    float2 a[kDiscPerThread], b[kDiscPerThread];
#pragma unroll
    for (int u = 0; u < kDiscPerThread; ++u) {
        const int j = j0 + u * kThreads;
        const int64_t n = it.n_lo + (j < it.n_k ? j : it.n_k - 1);
        a[u] = it.iq_ring[(uint64_t)n & ring_mask];
        b[u] = n > 0 ? it.iq_ring[(uint64_t)(n - 1) & ring_mask] : make_float2(0.f, 0.f);
    }
    for (int i = threadIdx.x; i < 257; i += kThreads) tab[i] = atan_tab[i];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kDiscPerThread; ++u) {
        const int j = j0 + u * kThreads;
        if (j >= it.n_k) break;
        const int64_t n = it.n_lo + j;
        // volk_32fc_x2_multiply_conjugate_32fc: a * conj(b), unfused
        const float tr = __fadd_rn(__fmul_rn(a[u].x, b[u].x), __fmul_rn(a[u].y, b[u].y));
        const float ti = __fsub_rn(__fmul_rn(a[u].y, b[u].x), __fmul_rn(a[u].x, b[u].y));
        it.fm_ring[(uint64_t)n & ring_mask] = fast_atan2f_gr(ti, tr, tab);
    }
}

// ---------------------------------------------------------------- exact rotator (rcf_set_rotator)
// gr::blocks::rotator as freq_xlating_fir_filter_ccc drives it, one lane per channel, the block's outputs in order:
//   y = v * phase;  phase *= incr;  if (++counter % 512 == 0) phase /= |phase|       (float32, unfused)
// The kernel only TABULATES phase per output (the multiply happens in the FIR kernels' epilogues, rotate_value): the
// sequence does not depend on the data.  Sequential by nature -- measured ~30 ns per output per block (0.15 ms for 5243
// outputs, independent of the channel count: one lane per channel) -- which is why it is an
// option: at real-time block sizes (thousands of outputs) it hides behind the FIR launch it precedes; at the bench's
// 10^4 x real time it would not.  |phase| as glibc's hypotf computes it: sqrt of the exact double sum, rounded twice.
__global__ __launch_bounds__(64) void rot_fill_kernel(const RotFill *__restrict__ items, int n_items, uint64_t ring_mask)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_items) return;
    const RotFill it = items[i];
    float pr = it.state[0], pi = it.state[1];
    unsigned cnt = __float_as_uint(it.state[2]);
    const float ir = it.incr_re, ii = it.incr_im;
    // runs between two renormalisations (every 512th call) are plain loops: nothing but the recurrence and a store
    int j = 0;
    while (j < it.n_k) {
        const int run = min(it.n_k - j, 512 - (int)(cnt & 511u));
        uint64_t at = (uint64_t)(it.n_from + j);
#pragma unroll 8
        for (int q = 0; q < run; ++q) {
            it.ring[(at + q) & ring_mask] = make_float2(pr, pi);
            const float nr = __fsub_rn(__fmul_rn(pr, ir), __fmul_rn(pi, ii));
            const float ni = __fadd_rn(__fmul_rn(pr, ii), __fmul_rn(pi, ir));
            pr = nr;
            pi = ni;
        }
        cnt += (unsigned)run;
        j += run;
        if ((cnt & 511u) == 0) {
            const float mag = (float)sqrt((double)pr * (double)pr + (double)pi * (double)pi);
            pr = __fdiv_rn(pr, mag);
            pi = __fdiv_rn(pi, mag);
        }
    }
    it.state[0] = pr;
    it.state[1] = pi;
    it.state[2] = __uint_as_float(cnt);
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

void launch_rot_fill(const RotFill *d_items, int n_items, uint64_t ring_mask, hipStream_t s)
{
    if (n_items <= 0) return;
    hipLaunchKernelGGL(rot_fill_kernel, dim3((n_items + 63) / 64), dim3(64), 0, s, d_items, n_items, ring_mask);
}

void launch_fm_level(const float *fm_ring, int64_t n_end, int window, float gain, uint64_t ring_mask, float *d_out,
                     hipStream_t s)
{
    hipLaunchKernelGGL(fm_level_kernel, dim3(1), dim3(kThreads), 0, s, fm_ring, n_end, window, gain, ring_mask, d_out);
}

void launch_fir_pack(const ChanLaunch *d_chans, int n_chans, int T, float *bank, const unsigned char *dirty,
                      hipStream_t s)
{
    const int NS = bank2_steps(T);
    const int groups = (n_chans + kM2Group - 1) / kM2Group;
    hipLaunchKernelGGL(fir_pack_kernel, dim3((unsigned)((NS * 1024 + kThreads - 1) / kThreads), groups), dim3(kThreads),
                       0, s, d_chans, n_chans, T, NS, bank, dirty);
}

template <int NT, int PD>
static void launch_mfma_t(const ChanLaunch *d_chans, MfmaArgs a, int max_n_k, hipStream_t s)
{
    const size_t lds = 2 * kM2ChunkSteps * 4096;             // 64 KB: two workgroups per CU
    static DynLdsAttr attr;
    attr.ensure(reinterpret_cast<const void *>(fir_mfma_kernel<NT, PD>), lds);
    a.n_wt = (max_n_k + 64 * NT - 1) / (64 * NT);
    hipLaunchKernelGGL((fir_mfma_kernel<NT, PD>), dim3((unsigned)(a.n_wt * a.n_groups * a.n_parts)), dim3(kM2Threads), lds,
                       s, d_chans, a);
}

static void launch_mfma(const ChanLaunch *d_chans, const FirLaunchDims &dims, hipStream_t s)
{
    MfmaArgs a{};
    a.bank = dims.bank;
    a.src_len = dims.src_len;
    a.ring_mask = dims.ring_mask;
    a.D = dims.D; a.T = dims.T; a.n_chans = dims.n_chans;
    a.n_groups = (dims.n_chans + kM2Group - 1) / kM2Group;
    a.n_parts = dims.partial ? dims.mfma_parts : 1;
    a.partial = dims.partial;
    a.n_k = dims.max_n_k;
    if (dims.mfma_nt == 2) launch_mfma_t<2, 3>(d_chans, a, dims.max_n_k, s);
    else                   launch_mfma_t<1, 3>(d_chans, a, dims.max_n_k, s);
    if (a.n_parts > 1)
        hipLaunchKernelGGL(fir_mfma_finish_kernel, dim3((dims.max_n_k + kThreads - 1) / kThreads, dims.n_chans),
                           dim3(kThreads), 0, s, d_chans, a.partial, dims.n_chans, dims.max_n_k, a.n_parts, dims.ring_mask);
}

void launch_fir_bank(const ChanLaunch *d_chans, const FirLaunchDims &dims, hipStream_t s)
{
    if (dims.n_chans <= 0 || dims.max_n_k <= 0) return;
    if (dims.mfma) { launch_mfma(d_chans, dims, s); return; }
    if (dims.small) {
        const int KB = fir_small_outputs(dims.D, dims.T);
        const size_t lds = ((size_t)KB * dims.D + 2 * dims.T + KB + 1) * sizeof(float2) + 264 * sizeof(float);
        hipLaunchKernelGGL(fir_small_kernel, dim3(dims.n_chans, (dims.max_n_k + KB - 1) / KB), dim3(kThreads), lds, s,
                           d_chans, dims.D, dims.T, KB, dims.ring_mask, dims.atan_tab);
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
    const int per_wg = kThreads * kDiscPerThread;
    hipLaunchKernelGGL(disc_kernel, dim3((max_n_k + per_wg - 1) / per_wg, n_items), dim3(kThreads), 0, s,
                       d_items, ring_mask, d_atan_table);
}

static float g_atan_tab[257];
const float *atan_table_host()
{
    static bool init = false;
    if (!init) {
        // gr::fast_atan2f's table is 257 literals of 7 significant digits -- atan(i / 255) printed with %.6e, the last
        // one (pi / 4) twice -- not the floats nearest to atan(i / 255): restated the same way
        char buf[32];
        for (int i = 0; i < 257; ++i) {
            snprintf(buf, sizeof buf, "%.6e", i < 256 ? atan((double)i / 255.0) : 3.14159265358979323846 / 4.0);
            g_atan_tab[i] = (float)strtod(buf, nullptr);
        }
        init = true;
    }
    return g_atan_tab;
}

}  // namespace rcfx
