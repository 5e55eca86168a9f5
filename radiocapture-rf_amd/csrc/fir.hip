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

#include <type_traits>
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

// ---------------------------------------------------------------- filterbank taps: matrix -> channel rings
// The frame-major banks (pfb5.hip) leave the tapped bins of a launch as a compact frame-major matrix, row r = the
// bank's frame k_first + r, one column per tap.  A workgroup takes 16 taps x 128 outputs: it reads the rows the way
// they lie (16 taps = one 128-byte piece of a row), applies each tap's rotator (GNU Radio's increment and / or the
// source shift: rotate_value(), the FIR bank's epilogue) into LDS, then turns the tile round -- 16 lanes x 2
// consecutive outputs of ONE tap = two 128-byte lines of that channel's IQ ring and one of its discriminator ring per
// store -- and writes IQ and the discriminator (which needs the output before: the row above in LDS; rows before the
// launch come from the ring).  The 128 outputs of a tap are ALIGNED to 32 in the tap's own ring index (every tap has
// its own k_abs0), so that every store is whole lines: a partial line costs a read-modify-write in the memory system
// (32-byte pieces measured 4-5x slower than lines).  The price is the 32 extra rows a tile loads.
#ifndef RCF_TAPFIN_WGS
#define RCF_TAPFIN_WGS 5      // workgroups per CU the finalize kernels are compiled for (96 VGPRs; tools/variant_lib.sh for A/B)
#endif
constexpr int kTapOut = 128, kTapCols = 16, kTapAlign = 32, kTapLdsRows = kTapOut + kTapAlign + 1,
              kTapLdsPitch = kTapCols;
// LDS position of (row lr, tap column sl): the column is rotated by half the row number, so that BOTH phases are free of
// bank conflicts -- phase 1 writes 16 columns of one row (any rotation of 16 consecutive 8-byte words), phase 2 reads
// rows 2 q + c, q = 0 .. 15, of ONE column: 16 different rotations = 16 different bank pairs.  (A pitch of 17 had the
// second phase at stride 68 dwords: lanes q and q + 8 on the same banks, 8.9 M conflict cycles per 1600-tap launch.)
__device__ __forceinline__ int tap_lds_at(int lr, int sl) { return lr * kTapLdsPitch + ((sl + (lr >> 1)) & (kTapCols - 1)); }
// A tap's rows of a launch in 32 bits.  Matrix row r is the bank's frame k_first + r = the tap's output n = k - k_abs0:
//   [lo, hi]   rows that are outputs of THIS launch (0 <= r < n_rows, k_lo <= k < k_lo + n_k, n >= 0), hi < lo: none
//   old_lo     first row (< 0) whose output an earlier launch left in the tap's ring (n >= 0)
//   first_ever the row of the tap's output 0 -- it has no predecessor -- clamped far below the tile when it is long past
//   last       the row of the launch's last output of this tap (what a discriminator-only tap still stores as IQ)
struct TapRows { int lo, hi, old_lo, first_ever, last; };
// what the second phase needs of a tap, left in LDS by the first phase's lane of that tap (rr == 0): the second phase then
// loads no launch record, does no 64-bit row arithmetic and no sincos of its own (its preamble was ~220 of a lane's ~900
// instructions on the discriminator-only path, and the kernel is vector-issue bound)
struct __attribute__((aligned(16))) TapInfo {
    float2 *iq_ring;
    float *fm_ring;
    int lo, hi, first_ever, last;      // TapRows of the launch
    uint32_t o32;                      // ring position of matrix row 0 in the tap's own rings
    int a;                             // the tile's outputs start at row r0 - a
    float inc_r, inc_i;                // discriminator-only taps: the rotator's increment as a phasor
    int fm_only, pad_[3];
};
__device__ __forceinline__ TapRows tap_rows(const TapLaunch &L, int64_t k_first, int n_rows)
{
    constexpr int64_t FAR = (int64_t)1 << 30;
    auto clamp32 = [&](int64_t v) { return (int)(v < -FAR ? -FAR : (v > FAR ? FAR : v)); };
    const int64_t first = L.k_abs0 - k_first;                  // row of output 0
    const int64_t lo = max(max((int64_t)0, L.k_lo - k_first), first);
    const int64_t hi = min((int64_t)n_rows, L.k_lo + L.n_k - k_first) - 1;
    TapRows t;
    t.lo = clamp32(lo);
    t.hi = clamp32(hi);
    t.old_lo = clamp32(first);
    t.first_ever = clamp32(first);
    t.last = clamp32(L.k_lo + L.n_k - 1 - k_first);
    return t;
}

// one tile (16 taps x 128 outputs) of one front-end's taps; shared by the single-front-end kernel and the grouped one
__device__ __forceinline__ void tap_finalize_tile(const TapFinArgs &A, const int bx, const int by, const uint64_t ring_mask,
                                                  const float *__restrict__ atan_tab, float *tab, float2 *ys, TapInfo *info)
{
    static_assert(kThreads == kTapCols * 16, "16 x 16 lanes");
    const TapLaunch *__restrict__ taps = A.taps;
    const int n_taps = A.n_taps, pitch = A.pitch, n_rows = A.n_rows, tap_first = A.tap_first, n_bins = A.n_bins;
    const float2 *__restrict__ mat = A.mat;
    const int64_t k_first = A.k_first;
    const int32_t *__restrict__ group_bin0 = A.group_bin0;
    const float2 *__restrict__ bins_ring = A.bins_ring;
    const int tid = threadIdx.x;
    // grid: x = group of 16 taps (fastest), y = tile of rows -- workgroups dispatched together read neighbouring
    // 128-byte pieces of the SAME matrix rows (whole rows between them), not one piece from each of 161 rows apart
    const int s0 = bx * kTapCols, r0 = by * kTapOut;
    const int r_lds0 = r0 - kTapAlign - 1;                   // matrix row of LDS row 0
    for (int i = tid; i < 257; i += kThreads) tab[i] = atan_tab[i];
    {
        const int sl = tid & (kTapCols - 1), rr = tid >> 4;
        const int slot = s0 + sl;
        if (slot < n_taps) {
            const TapLaunch L = taps[slot];
            const bool idle = L.dangle == 0.0 && L.dlogmag == 0.0 && L.angle0 == 0.0 && L.logmag0 == 0.0;
            const int b0 = group_bin0[bx];                      // >= 0: this group's taps are 16 consecutive bins of the ring
            // all of a lane's rows are requested before the first is used (the loop below would otherwise pay one
            // memory round trip per row: 11 in a row)
            constexpr int NIT = (kTapLdsRows + 15) / 16;
            float2 z[NIT];
            // the rows THIS tap's 128 outputs (and the output before them) come from: its tile starts a_own rows before r0
            // (its ring index aligned to 32).  Rows outside are not requested: with the taps of a group opened together --
            // one offset for all sixteen -- the 32 alignment rows are then never fetched (they were a quarter more traffic)
            const int a_own = (int)((uint64_t)(k_first - L.k_abs0 + r0) & (kTapAlign - 1));
            const int r_need_lo = r0 - a_own - 1, r_need_hi = r0 - a_own + kTapOut - 1;
            // The per-row index arithmetic in 32 bits (the kernel is vector-issue bound and int64 compares / masks / multiplies
            // were a fifth of its instructions): a row r is FRESH (a frame of this launch inside the tap's range) for r in
            // [rf_lo, rf_hi], OLD (an output of an earlier launch, read back from the tap's ring) for r in [ro_lo, -1]; ring
            // positions wrap in uint32 (the ring is a power of two of at most 2^31 samples)
            const TapRows tr_ = tap_rows(L, k_first, n_rows);
            const int rf_lo = max(tr_.lo, r_need_lo), rf_hi = min(tr_.hi, r_need_hi);
            const int ro_lo = max(tr_.old_lo, r_need_lo);
            const uint32_t mask32 = (uint32_t)ring_mask;
            const uint32_t kf32 = (uint32_t)((uint64_t)k_first & ring_mask);           // ring position of matrix row 0 (bank ring)
            const uint32_t o32 = (uint32_t)((uint64_t)(k_first - L.k_abs0) & ring_mask);   // ... in the tap's own rings
            if (rr == 0) {
                TapInfo ti;
                ti.iq_ring = L.iq_ring;
                ti.fm_ring = L.fm_ring;
                ti.lo = tr_.lo; ti.hi = tr_.hi; ti.first_ever = tr_.first_ever; ti.last = tr_.last;
                ti.o32 = o32;
                ti.a = a_own;
                ti.inc_r = 1.f; ti.inc_i = 0.f;
                if (L.fm_only && L.dangle != 0.0) {
                    double sn_, cs_;
                    sincos_fast(L.dangle, sn_, cs_);
                    ti.inc_r = (float)cs_;
                    ti.inc_i = (float)sn_;
                }
                ti.fm_only = L.fm_only;
                ti.pad_[0] = ti.pad_[1] = ti.pad_[2] = 0;
                info[sl] = ti;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int lr = rr + 16 * it, r = r_lds0 + lr;
                z[it] = make_float2(0.f, 0.f);
                if (lr < kTapLdsRows) {
                    if (r >= rf_lo && r <= rf_hi)
                        z[it] = b0 >= 0 ? bins_ring[(uint64_t)((kf32 + (uint32_t)r) & mask32) * (uint32_t)n_bins + (unsigned)(b0 + sl)]
                                        : mat[(uint64_t)(uint32_t)r * (uint32_t)pitch + (unsigned)(slot - tap_first)];
                    else if (r < 0 && r >= ro_lo)
                        z[it] = L.iq_ring[(o32 + (uint32_t)r) & mask32];   // produced by an earlier launch: already rotated
                }
            }
            if (L.fm_only) {
                // discriminator only: no rotation at all.  The discriminator of the rotated stream, arg(y[n] conj(y[n-1])) with
                // y = bin x phase, phase[n] = phase[n-1] x incr, is arg(bin[n] conj(bin[n-1]) x incr): the second phase turns
                // the product by the tap's ONE angle instead of walking a float64 phasor along every row (a quarter of this
                // kernel's vector instructions, and the kernel is vector-issue bound)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int lr = rr + 16 * it;
                    if (lr >= kTapLdsRows) break;
                    ys[tap_lds_at(lr, sl)] = z[it];
                }
            } else {
            RotatorWalk<TapLaunch> walk(L, k_first + r_lds0 + rr - L.k_abs0, 16);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int lr = rr + 16 * it, r = r_lds0 + lr;
                if (lr >= kTapLdsRows) break;
                float2 v = z[it];
                // (rows outside the tap's own range hold zeros nobody reads: no rotation spent on them)
                if (!idle && r >= 0 && r >= r_need_lo && r <= r_need_hi) v = walk.rotate(L, v.x, v.y);
                walk.advance();
                ys[tap_lds_at(lr, sl)] = v;
            }
            }
        }
    }
    __syncthreads();
    {
        const int sl = tid >> 4, q = tid & 15;
        const int slot = s0 + sl;
        if (slot >= n_taps) return;
        const TapInfo L = info[sl];                                  // (the first phase's lane of this tap left it there)
        const int a = L.a;                                           // this tile's outputs start at row r0 - a
        struct { int lo, hi, last; } tr_ = {L.lo, L.hi, L.last};     // rows [lo, hi] are outputs of this launch
        const uint32_t mask32 = (uint32_t)ring_mask;
        const uint32_t o32 = L.o32;
        const int r_first = L.first_ever;                            // the row of the tap's very first output (no predecessor), or far below
        const float inc_r = L.inc_r, inc_i = L.inc_i;                // discriminator-only taps: the rotator's increment as a phasor
        auto fm_of = [&](float2 y1, float2 y0) {
            // volk_32fc_x2_multiply_conjugate_32fc: y1 * conj(y0), unfused (as disc_kernel)
            const float tr = __fadd_rn(__fmul_rn(y1.x, y0.x), __fmul_rn(y1.y, y0.y));
            const float ti = __fsub_rn(__fmul_rn(y1.y, y0.x), __fmul_rn(y1.x, y0.y));
            return fast_atan2f_gr(ti, tr, tab);
        };
        // one tile of one tap, four ways: INTERIOR = every one of the tile's 128 outputs is an output of this launch with a
        // predecessor in LDS and none of them the launch's last (the steady state: 326 of a 2^25-sample launch's 328 tiles) --
        // no per-pair range checks, no zero predecessors, whole-line stores only; FMO = discriminator-only tap.  The choice is
        // uniform over a tap's 16 lanes (a wavefront holds four taps).
        auto run = [&](auto interior_c, auto fmo_c) {
            constexpr bool INTERIOR = decltype(interior_c)::value, FMO = decltype(fmo_c)::value;
#pragma unroll
            for (int j = 0; j < kTapOut / 32; ++j) {
                const int ra = r0 - a + 2 * (q + 16 * j);               // rows ra, ra + 1 -> ring indices na (even), na + 1
                const int lr = ra - r_lds0;
                const bool va = INTERIOR || (ra >= tr_.lo && ra <= tr_.hi);
                const bool vb = INTERIOR || (ra + 1 >= tr_.lo && ra + 1 <= tr_.hi);
                if (!va && !vb) continue;
                const float2 ym = (INTERIOR || ra > r_first) ? ys[tap_lds_at(lr - 1, sl)] : make_float2(0.f, 0.f);   // (na > 0)
                const float2 ya = ys[tap_lds_at(lr, sl)], yb = ys[tap_lds_at(lr + 1, sl)];
                const float2 yb0 = (INTERIOR || ra + 1 > r_first) ? ya : make_float2(0.f, 0.f);
                const uint32_t ia = (o32 + (uint32_t)ra) & mask32;
                if constexpr (FMO) {
                    // discriminator only (rcf_chan_set_fm_only): 4 of the 12 bytes per output; the launch's LAST bin value still goes
                    // to the IQ ring (unrotated, as every row is here) -- it is the "output before" of the next launch's first
                    // discriminator sample.  fm_c: bin[n] conj(bin[n-1]) turned by the rotator's increment (cr, ci)
                    auto fm_c = [&](float2 y1, float2 y0) {
                        const float tr = __fadd_rn(__fmul_rn(y1.x, y0.x), __fmul_rn(y1.y, y0.y));
                        const float ti = __fsub_rn(__fmul_rn(y1.y, y0.x), __fmul_rn(y1.x, y0.y));
                        const float ur = __fsub_rn(__fmul_rn(tr, inc_r), __fmul_rn(ti, inc_i));
                        const float ui = __fadd_rn(__fmul_rn(tr, inc_i), __fmul_rn(ti, inc_r));
                        return fast_atan2f_gr(ui, ur, tab);
                    };
                    if (va && vb) {
                        typedef float v2f_ __attribute__((ext_vector_type(2)));
                        v2f_ b_; b_.x = fm_c(ya, ym); b_.y = fm_c(yb, yb0);
                        __builtin_nontemporal_store(b_, reinterpret_cast<v2f_ *>(L.fm_ring + ia));
                    } else if (va) {
                        L.fm_ring[ia] = fm_c(ya, ym);
                    } else {
                        L.fm_ring[(ia + 1) & mask32] = fm_c(yb, yb0);
                    }
                    if constexpr (!INTERIOR) {
                        if (va && ra == tr_.last) L.iq_ring[ia] = ya;
                        if (vb && ra + 1 == tr_.last) L.iq_ring[(ia + 1) & mask32] = yb;
                    }
                } else {
                    if (va && vb) {                                      // na is even and the ring a power of two: no wrap inside the pair
                        // (non-temporal: 256 taps 55.4 -> 53.1 us, 1600 taps 308 -> 302 us per 2^25-sample block)
                        typedef float v4f_ __attribute__((ext_vector_type(4))); typedef float v2f_ __attribute__((ext_vector_type(2)));
                        v4f_ a_; a_.x = ya.x; a_.y = ya.y; a_.z = yb.x; a_.w = yb.y;
                        v2f_ b_; b_.x = fm_of(ya, ym); b_.y = fm_of(yb, yb0);
                        __builtin_nontemporal_store(a_, reinterpret_cast<v4f_ *>(L.iq_ring + ia));
                        __builtin_nontemporal_store(b_, reinterpret_cast<v2f_ *>(L.fm_ring + ia));
                    } else if (va) {
                        L.iq_ring[ia] = ya;
                        L.fm_ring[ia] = fm_of(ya, ym);
                    } else {
                        const uint32_t ib = (ia + 1) & mask32;
                        L.iq_ring[ib] = yb;
                        L.fm_ring[ib] = fm_of(yb, yb0);
                    }
                }
            }
        };
        const int t_lo = r0 - a, t_hi = r0 - a + kTapOut - 1;          // the tile's rows
        const bool interior = t_lo > r_first && t_lo >= tr_.lo && t_hi <= tr_.hi && t_hi < tr_.last;
        if (L.fm_only) {
            if (interior) run(std::true_type{}, std::true_type{});
            else          run(std::false_type{}, std::true_type{});
        } else {
            if (interior) run(std::true_type{}, std::false_type{});
            else          run(std::false_type{}, std::false_type{});
        }
    }
}

__global__ __launch_bounds__(kThreads, RCF_TAPFIN_WGS) void tap_finalize_kernel(TapFinArgs A, uint64_t ring_mask,
                                                                const float *__restrict__ atan_tab)
{
    __shared__ float tab[260];
    __shared__ float2 ys[kTapLdsRows * kTapLdsPitch];
    __shared__ TapInfo info[kTapCols];
    tap_finalize_tile(A, blockIdx.x, blockIdx.y, ring_mask, atan_tab, tab, ys, info);
}

// the taps of G front-ends in one launch: grid.z = front-end, x / y sized for the largest of them
__global__ __launch_bounds__(kThreads, RCF_TAPFIN_WGS) void tap_finalize_group_kernel(const TapFinArgs *__restrict__ args, uint64_t ring_mask,
                                                                      const float *__restrict__ atan_tab)
{
    __shared__ float tab[260];
    __shared__ float2 ys[kTapLdsRows * kTapLdsPitch];
    __shared__ TapInfo info[kTapCols];
    const TapFinArgs A = args[blockIdx.z];
    if ((int)blockIdx.x * kTapCols >= A.n_taps || (int)blockIdx.y * kTapOut >= A.n_rows + kTapAlign - 1) return;
    tap_finalize_tile(A, blockIdx.x, blockIdx.y, ring_mask, atan_tab, tab, ys, info);
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

void launch_tap_finalize(const TapLaunch *d_taps, int n_taps, const float2 *tap_mat, int tap_pitch, int n_rows,
                         int64_t k_first, uint64_t ring_mask, const float *d_atan_table, const int32_t *d_group_bin0,
                         int tap_first, const float2 *bins_ring, int n_bins, hipStream_t s)
{
    if (n_taps <= 0 || n_rows <= 0) return;
    const TapFinArgs A{d_taps, tap_mat, d_group_bin0, bins_ring, k_first, n_taps, tap_pitch, n_rows, tap_first, n_bins, 0};
    hipLaunchKernelGGL(tap_finalize_kernel,
                       dim3((n_taps + kTapCols - 1) / kTapCols, (n_rows + kTapAlign - 1 + kTapOut - 1) / kTapOut),
                       dim3(kThreads), 0, s, A, ring_mask, d_atan_table);
}

void launch_tap_finalize_group(const TapFinArgs *d_args, int n_args, int max_taps, int max_rows, uint64_t ring_mask,
                               const float *d_atan_table, hipStream_t s)
{
    if (n_args <= 0 || max_taps <= 0 || max_rows <= 0) return;
    hipLaunchKernelGGL(tap_finalize_group_kernel,
                       dim3((max_taps + kTapCols - 1) / kTapCols, (max_rows + kTapAlign - 1 + kTapOut - 1) / kTapOut, n_args),
                       dim3(kThreads), 0, s, d_args, ring_mask, d_atan_table);
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
