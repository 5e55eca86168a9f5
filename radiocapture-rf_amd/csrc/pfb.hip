// pfb.hip -- polyphase filterbank channelizer for gfx950: polyphase FIR in registers fused with an
// in-LDS NB-point inverse-sign FFT and a channel-major epilogue.
//
// Math: with prototype h (T taps), NB bins, decimation D (OS = NB / D),
//   out_k[n] = e^{-j 2 pi k n D / NB} * sum_{rho<NB} e^{+j 2 pi k rho / NB} * u_rho[n]
//   u_rho[n] = sum_{p<P} h[NB p + rho] * x[n D - rho - NB p]
// which is exactly freq_xlating_fir_filter_ccc(D, h, k fs / NB, fs) -- the block the reference
// instantiates once PER CHANNEL at /root/reference/rc_frontend/channel.py:35 -- for all NB on-grid
// frequencies at once, with mathematically exact phases.
//
// Mapping (workgroup = NB threads, thread rho = branch rho):
//   * x[m D - rho] for consecutive rho is a reversed unit-stride run of the interleaved cf32 stream:
//     every wavefront load is one contiguous 512-byte segment; each input sample is read once per
//     workgroup (plus OS (P-1) halo frames at the head of the workgroup's frame range).
//   * the P-tap branch FIR slides over frames entirely in VGPRs (window of F + OS (P-1) samples,
//     real taps in registers): 2 FMA per tap per output, no LDS traffic.
//   * the NEXT chunk's F samples are requested before the current chunk enters its LDS/FFT phases, so
//     HBM latency hides under the butterflies (register prefetch; the phases are barrier-separated
//     and would otherwise serialise load -> compute at 2-3 workgroups per CU).
//   * F = 16 frames of u are parked in LDS ([frame][branch], rows padded 1-in-16 + 2), transformed by
//     radix-16/8/4/2 Stockham passes (fft_core.hpp) with exact twiddles read from an LDS table, then
//     read back transposed so that each bin's F consecutive outputs leave as one contiguous 128-byte
//     run of its ring (channel-major output: what the stage-2 FIR and the egress pump read).
// Bound: HBM.  Algorithmic bytes per input sample = 8 (read) + 8 NB / D (write) = 16 at OS = 1.
#include <cstdlib>

#include "fft_core.hpp"
#include "rcf_internal.h"

namespace rcfx {

namespace {

constexpr int F = 16;   // frames per LDS chunk
int env_int(const char *name, int dflt);
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NB> struct Plan;
template <> struct Plan<64>   { static constexpr int n = 2; static constexpr int r[3] = {16, 4, 1}; };
template <> struct Plan<128>  { static constexpr int n = 2; static constexpr int r[3] = {16, 8, 1}; };
template <> struct Plan<256>  { static constexpr int n = 2; static constexpr int r[3] = {16, 16, 1}; };
template <> struct Plan<512>  { static constexpr int n = 3; static constexpr int r[3] = {16, 16, 2}; };
template <> struct Plan<1024> { static constexpr int n = 3; static constexpr int r[3] = {16, 16, 4}; };

template <int NB> __host__ __device__ constexpr int row_stride() { return lds_padded_len(NB) + 2; }

// LDS hand-off inside ONE wavefront: DS operations of a wave execute in order, so ordering the
// compiler is enough -- no s_barrier.  Used where a pass's butterflies for a frame all live in one wave.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one Stockham pass over the F frames held in LDS; thread's butterflies all share j = tid % (NB/R).
// When NB/R divides 64 every frame's butterflies sit in a single wavefront (frame = b / (NB/R)), so
// the read->write hazard of the in-place pass is wave-local; END_WG says whether the NEXT consumer of
// the buffer uses a different frame->wave map (then the trailing barrier must be workgroup-wide).
template <int NB, int R, int NS, bool END_WG, bool NOWG = false>
__device__ __forceinline__ void pfb_pass(cf *buf, const cf *tw_lds, int tid)
{
    constexpr bool WAVE_LOCAL = (64 % (NB / R)) == 0;
    static_assert(!NOWG || WAVE_LOCAL, "role-split passes must be wave-local");
    constexpr int BPF = NB / R;          // butterflies per frame
    constexpr int CNT = F / R;           // butterflies per thread (F * BPF / NB)
    static_assert(F % R == 0, "F must be a multiple of every radix");
    constexpr int RS = row_stride<NB>();
    using Pass = StockhamPass<NB, R, +1>;
    const int j = tid % BPF;
    cf v[CNT][R];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int frame = (tid + i * NB) / BPF;
        Pass::load(buf + frame * RS, j, v[i]);
    }
    if (NS > 1) {
        const int k = j & (NS - 1);
#pragma unroll
        for (int t = 1; t < R; ++t) {
            const cf w = tw_lds[(k * t) * (NB / (NS * R))];   // exact table entry e^{+2 pi i k t / (NS R)}
#pragma unroll
            for (int i = 0; i < CNT; ++i) v[i][t] = cmul(v[i][t], w);
            // role-split build: stop the scheduler from issuing all R-1 table reads at once (2 VGPRs
            // each); a compiler-level memory barrier every 4 twiddles keeps the FFT role under 128 VGPRs
            if (NOWG && (t & 3) == 0) asm volatile("" ::: "memory");
        }
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i) Dft<R, +1>::run(v[i]);
    if (WAVE_LOCAL || NOWG) wave_sync(); else __syncthreads();   // every butterfly has read before any writes
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int frame = (tid + i * NB) / BPF;
        Pass::template store_t<NS>(buf + frame * RS, j, v[i]);
    }
    if ((WAVE_LOCAL && !END_WG) || NOWG) wave_sync(); else __syncthreads();
}

template <int NB, int OS, int P, int MINW, int FB, int POL, bool ZH, int ABL = 0>
__global__ __launch_bounds__(NB, MINW) void pfb_kernel(PfbLaunch p, int frames_per_wg, int n_wg)
{
    constexpr int D = NB / OS;
    constexpr int HALO = OS * (P - 1);
    constexpr int RS = row_stride<NB>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + F * RS;

    const int tid = threadIdx.x;
    // XCD-aware remap (bijective): consecutive frame ranges -- which share HALO input frames -- run
    // on the same XCD so the halo re-read hits that XCD's L2.
    int wg;
    {
        const int b = blockIdx.x, q = n_wg / 8, r = n_wg % 8, xcd = b % 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const int fb0 = wg * frames_per_wg;
    if (fb0 >= p.n_frames) return;
    const int nfr = min(frames_per_wg, p.n_frames - fb0);
    const int64_t n0 = p.n_lo + fb0;

    tw_lds[tid] = p.tw[tid];              // NB entries, NB threads
    float h[P];
#pragma unroll
    for (int q = 0; q < P; ++q) h[q] = p.ptaps[q * NB + tid];

    // Buffer addressing: one 32-bit per-lane byte offset + scalar frame offsets (f * D * 8) instead of
    // 16 64-bit per-lane pointers -- the address VGPRs and their v_add chains were the bulk of this
    // kernel's register pressure.  The source is the linear [hist | block] buffer (mask = ~0); every
    // sample the kernel touches lies inside it (hist >= P*NB + D), so offsets are never negative.
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cf *>(p.src.base), 0, (int)(p.src_len * (int64_t)sizeof(cf)), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.bins_ring, 0, (int)((int64_t)NB * p.ring_cap * (int64_t)sizeof(cf)), 0x00020000);
    // sample m*D - tid of the stream.  ZH (zero history) instantiation only: frames before m_min
    // predate the filterbank and count as zero (a freshly started GNU Radio block's history); the host
    // selects it for the first launches after rcf_pfb_open, every later launch runs predicate-free.
    // Rows past this workgroup's range are loaded unconditionally: they are inside the buffer (or
    // beyond src_len, where the buffer descriptor returns 0) and their outputs are never stored.
    const int64_t m_min = (p.start_sample + tid + D - 1) / D;
    auto frame_off = [&](int64_t m) -> int { return (int)((m * D - tid - p.src.origin) * (int64_t)sizeof(cf)); };
    auto ld = [&](int vo, int so, int64_t m) -> v2f {
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo, so, (POL & 1) ? 2 : 0);
        v2f x;
        x.x = __uint_as_float(r.x);
        x.y = __uint_as_float(r.y);
        if (ZH && m < m_min) x = (v2f)(0.f);
        return x;
    };

    // FB rows are fetched and filtered at a time (FB = 16: one batch per chunk; FB = 8: two half
    // batches -> 16 fewer live VGPRs, one more workgroup per CU).
    v2f w[FB + HALO];
    {
        const int64_t m0 = n0 - HALO;
        const int vo = frame_off(m0);
#pragma unroll
        for (int i = 0; i < HALO; ++i) w[i] = ld(vo, i * D * (int)sizeof(cf), m0 + i);
    }
    const int k0 = tid / F, f_lane = tid % F;                 // epilogue role: bin k0 + i*NB/F, frame f_lane
    const int out_so_step = (int)((int64_t)(NB / F) * p.ring_cap * (int64_t)sizeof(cf));

    for (int ch = 0; ch < nfr; ch += F) {
        const int nf = min(F, nfr - ch);
#pragma unroll
        for (int sb = 0; sb < F; sb += FB) {
            {
                const int64_t m0 = n0 + ch + sb;
                const int vo = frame_off(m0);
#pragma unroll
                for (int f = 0; f < FB; ++f) w[HALO + f] = ld(vo, f * D * (int)sizeof(cf), m0 + f);
            }
            // branch FIR, straight into the LDS chunk; four outputs advance together so that the
            // dependent FMA chains interleave
#pragma unroll
            for (int f0 = 0; f0 < FB; f0 += 4) {
                float ur[4], ui[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ur[j] = ui[j] = 0.f;
#pragma unroll
                for (int q = 0; q < ((ABL & 1) ? 1 : P); ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const v2f xv = w[f0 + j + HALO - OS * q];
                        ur[j] = fmaf(h[q], xv.x, ur[j]);
                        ui[j] = fmaf(h[q], xv.y, ui[j]);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) buf[(sb + f0 + j) * RS + lds_pad(tid)] = make_float2(ur[j], ui[j]);
            }
#pragma unroll
            for (int i = 0; i < HALO; ++i) w[i] = w[i + FB];
        }
        __syncthreads();

        if (!(ABL & 2)) {
            using PL = Plan<NB>;
            // a pass hands over wave-locally only to a pass with the same radix (same frame->wave map)
            pfb_pass<NB, PL::r[0], 1, (PL::n < 2 || PL::r[1] != PL::r[0])>(buf, tw_lds, tid);
            if constexpr (PL::n >= 2)
                pfb_pass<NB, PL::r[1], PL::r[0], (PL::n < 3 || PL::r[2] != PL::r[1])>(buf, tw_lds, tid);
            if constexpr (PL::n >= 3) pfb_pass<NB, PL::r[2], PL::r[0] * PL::r[1], true>(buf, tw_lds, tid);
        }

        // transposed epilogue: lanes run along the frame axis of one bin
        {
            const int64_t n = n0 + ch + f_lane;
            const int ridx = (int)((uint64_t)(n - p.n_abs0) & p.ring_mask);
            const int vo = (int)(((int64_t)k0 * p.ring_cap + ridx) * (int64_t)sizeof(cf));
            if (f_lane < nf) {
#pragma unroll
                for (int i = 0; i < F; ++i) {
                    const int k = k0 + i * (NB / F);
                    cf v = ((NB / F) % 16 == 0) ? buf[f_lane * RS + lds_pad(k0) + i * ((NB / F) + (NB / F) / 16)]
                                                : buf[f_lane * RS + lds_pad(k)];
                    if (OS == 2) {
                        if ((k & 1) && (n & 1)) v = make_float2(-v.x, -v.y);
                    } else if (OS == 4) {
                        const int q = (int)((k * n) & 3);                 // e^{-j pi q / 2}
                        if (q == 1) v = make_float2(v.y, -v.x);
                        else if (q == 2) v = make_float2(-v.x, -v.y);
                        else if (q == 3) v = make_float2(-v.y, v.x);
                    }
                    u32x2 o;
                    o.x = __float_as_uint(v.x);
                    o.y = __float_as_uint(v.y);
                    __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, i * out_so_step, (POL & 2) ? 2 : 0);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Output-stationary variant (the default): one 16-frame chunk per workgroup, and instead of holding a
// 16 + OS(P-1)-sample window in registers the thread keeps the 16 OUTPUT accumulators and streams the
// input rows through them in groups of G (row j feeds output f with tap q = (HALO + f - j) / OS; the
// schedule is fully unrolled, every index is a compile-time constant).  26+ fewer live VGPRs than the
// sliding window => 4 workgroups per CU (the LDS limit) instead of 3, which is what this
// latency-bound kernel responds to (measured: workgroups per CU, not instruction count, set its rate).
template <int NB, int OS, int P, int MINW, int POL, bool ZH, int ABL = 0>
__global__ __launch_bounds__(NB, MINW) void pfb_kernel_os(PfbLaunch p, int n_wg)
{
    constexpr int D = NB / OS;
    constexpr int HALO = OS * (P - 1);
    constexpr int W = F + HALO;
    constexpr int G = 8;
    constexpr int RS = row_stride<NB>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + F * RS;

    const int tid = threadIdx.x;
    int wg;
    if (n_wg < 0) {
        wg = blockIdx.x;                                   // probe: no XCD remap
    } else {
        const int b = blockIdx.x, q = n_wg / 8, r = n_wg % 8, xcd = b % 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const int fb0 = wg * F;
    if (fb0 >= p.n_frames) return;
    const int nf = min(F, p.n_frames - fb0);
    const int64_t n0 = p.n_lo + fb0;

    tw_lds[tid] = p.tw[tid];
    float h[P];
#pragma unroll
    for (int q = 0; q < P; ++q) h[q] = p.ptaps[q * NB + tid];

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cf *>(p.src.base), 0, (int)(p.src_len * (int64_t)sizeof(cf)), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.bins_ring, 0, (int)((int64_t)NB * p.ring_cap * (int64_t)sizeof(cf)), 0x00020000);
    const int64_t m_min = (p.start_sample + tid + D - 1) / D;
    const int64_t m0 = n0 - HALO;
    const int vo_in = (int)((m0 * D - tid - p.src.origin) * (int64_t)sizeof(cf));

    float ur[F], ui[F];
#pragma unroll
    for (int f = 0; f < F; ++f) ur[f] = ui[f] = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < W; j0 += G) {
        v2f x[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            x[g] = (v2f)(0.f);
            if (j0 + g < W && !((ABL & 1) && j0 + g < HALO)) {
                const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo_in, (j0 + g) * D * (int)sizeof(cf),
                                                                     (POL & 1) ? 2 : 0);
                x[g].x = __uint_as_float(r.x);
                x[g].y = __uint_as_float(r.y);
                if (ZH && m0 + j0 + g < m_min) x[g] = (v2f)(0.f);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            constexpr int dummy = 0;
            (void)dummy;
            const int j = j0 + g;
            if (j < W) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const int t = HALO + f - j;          // = OS * q
                    if (t >= 0 && t % OS == 0 && t / OS < ((ABL & 4) ? 1 : P)) {
                        ur[f] = fmaf(h[t / OS], x[g].x, ur[f]);
                        ui[f] = fmaf(h[t / OS], x[g].y, ui[f]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) buf[f * RS + lds_pad(tid)] = make_float2(ur[f], ui[f]);
    __syncthreads();
    if (!(ABL & 2)) {
        using PL = Plan<NB>;
        pfb_pass<NB, PL::r[0], 1, (PL::n < 2 || PL::r[1] != PL::r[0])>(buf, tw_lds, tid);
        if constexpr (PL::n >= 2)
            pfb_pass<NB, PL::r[1], PL::r[0], (PL::n < 3 || PL::r[2] != PL::r[1])>(buf, tw_lds, tid);
        if constexpr (PL::n >= 3) pfb_pass<NB, PL::r[2], PL::r[0] * PL::r[1], true>(buf, tw_lds, tid);
    }
    // Wide epilogue (POL & 4): a lane stores TWO consecutive frames of a bin as one 16-byte access -- half
    // the store instructions for the same bytes.  Needs a full chunk and an even ring index (16-byte
    // alignment, no wrap inside the pair); otherwise the 8-byte form below runs.
    if ((POL & 4) && nf == F && (((n0 - p.n_abs0) & 1) == 0)) {
        constexpr int KQ = NB / (F / 2);                       // bins covered by one store instruction
        const int kq = tid / (F / 2), fp = tid % (F / 2);
        const int64_t n = n0 + 2 * fp;
        const int ridx = (int)((uint64_t)(n - p.n_abs0) & p.ring_mask);
        const int vo = (int)(((int64_t)kq * p.ring_cap + ridx) * (int64_t)sizeof(cf));
        const int so_step = (int)((int64_t)KQ * p.ring_cap * (int64_t)sizeof(cf));
#pragma unroll
        for (int i = 0; i < F / 2; ++i) {
            const int k = kq + i * KQ;
            const int col = (KQ % 16 == 0) ? lds_pad(kq) + i * (KQ + KQ / 16) : lds_pad(k);
            cf a = buf[(2 * fp) * RS + col];
            cf b = buf[(2 * fp + 1) * RS + col];
            if (OS == 2) {
                if (k & 1) b = make_float2(-b.x, -b.y);         // n even, n + 1 odd
            }
            u32x4 o;
            o.x = __float_as_uint(a.x);
            o.y = __float_as_uint(a.y);
            o.z = __float_as_uint(b.x);
            o.w = __float_as_uint(b.y);
            __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, vo, i * so_step, (POL & 2) ? 2 : 0);
        }
    } else {
        const int k0 = tid / F, f_lane = tid % F;
        const int out_so_step = (int)((int64_t)(NB / F) * p.ring_cap * (int64_t)sizeof(cf));
        const int64_t n = n0 + f_lane;
        const int ridx = (int)((uint64_t)(n - p.n_abs0) & p.ring_mask);
        const int vo = (int)(((int64_t)k0 * p.ring_cap + ridx) * (int64_t)sizeof(cf));
        if (f_lane < nf) {
#pragma unroll
            for (int i = 0; i < F; ++i) {
                const int k = k0 + i * (NB / F);
                cf v = ((NB / F) % 16 == 0) ? buf[f_lane * RS + lds_pad(k0) + i * ((NB / F) + (NB / F) / 16)]
                                            : buf[f_lane * RS + lds_pad(k)];
                if (OS == 2) {
                    if ((k & 1) && (n & 1)) v = make_float2(-v.x, -v.y);
                }
                u32x2 o;
                o.x = __float_as_uint(v.x);
                o.y = __float_as_uint(v.y);
                __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, i * out_so_step, (POL & 2) ? 2 : 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Role-split variant: a workgroup is 2*NB threads.  Waves [0, NB/64) are the FIR role (thread rho =
// branch rho: global loads -> sliding-window FIR -> LDS chunk), waves [NB/64, 2NB/64) are the FFT role
// (Stockham passes -> transposed non-temporal stores).  The roles work on the two halves of a
// double-buffered LDS chunk, one chunk apart, and meet at two s_barriers per chunk, so global loads,
// FMAs, butterflies and stores of neighbouring chunks overlap inside ONE workgroup instead of only
// across workgroups; each role's loop has its own (small) register footprint.
//   iteration it:   FIR role : rows 0-7 of chunk it | B1 | rows 8-15, slide, prefetch chunk it+1 | B2
//                   FFT role : passes of chunk it-1 | B1 | epilogue stores of chunk it-1         | B2
__device__ __forceinline__ void wg_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int NB, int OS, int P, int POL, bool ZH>
__global__ __launch_bounds__(2 * NB, 4) void pfb_kernel_rs(PfbLaunch p, int frames_per_wg, int n_wg)
{
    constexpr int D = NB / OS;
    constexpr int HALO = OS * (P - 1);
    constexpr int RS = row_stride<NB>();
    constexpr int FH = F / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf0 = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf0 + 2 * F * RS;

    int wg;
    {
        const int b = blockIdx.x, q = n_wg / 8, r = n_wg % 8, xcd = b % 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const int fb0 = wg * frames_per_wg;
    if (fb0 >= p.n_frames) return;
    const int nfr = min(frames_per_wg, p.n_frames - fb0);
    const int nchunks = (nfr + F - 1) / F;
    const int64_t n0 = p.n_lo + fb0;
    static_assert(Plan<NB>::n == 2 && Plan<NB>::r[0] == Plan<NB>::r[1], "role split needs one frame->wave map");
    const bool fir_role = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < NB;

#ifdef RS_ONLY_FFT
    if (false) {
#else
    if (fir_role) {
#endif
        const int tid = threadIdx.x;
        float h[P];
#pragma unroll
        for (int q = 0; q < P; ++q) h[q] = p.ptaps[q * NB + tid];
        const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<cf *>(p.src.base), 0, (int)(p.src_len * (int64_t)sizeof(cf)), 0x00020000);
        const int64_t m_min = (p.start_sample + tid + D - 1) / D;
        auto frame_off = [&](int64_t m) -> int {
            return (int)((m * D - tid - p.src.origin) * (int64_t)sizeof(cf));
        };
        auto ld = [&](int vo, int so, int64_t m) -> v2f {
            const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo, so, (POL & 1) ? 2 : 0);
            v2f x;
            x.x = __uint_as_float(r.x);
            x.y = __uint_as_float(r.y);
            if (ZH && m < m_min) x = (v2f)(0.f);
            return x;
        };
        v2f w[F + HALO];
        {
            const int64_t m0 = n0 - HALO;
            const int vo = frame_off(m0);
#pragma unroll
            for (int i = 0; i < F + HALO; ++i) w[i] = ld(vo, i * D * (int)sizeof(cf), m0 + i);
        }
        auto fir_rows = [&](cf *buf, int f_lo) {
#pragma unroll
            for (int f0 = 0; f0 < FH; f0 += 4) {
                float ur[4], ui[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ur[j] = ui[j] = 0.f;
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const v2f xv = w[f_lo + f0 + j + HALO - OS * q];
                        ur[j] = fmaf(h[q], xv.x, ur[j]);
                        ui[j] = fmaf(h[q], xv.y, ui[j]);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    buf[(f_lo + f0 + j) * RS + lds_pad(tid)] = make_float2(ur[j], ui[j]);
            }
        };
        for (int it = 0; it < nchunks; ++it) {
            cf *buf = buf0 + (it & 1) * F * RS;
            fir_rows(buf, 0);
            wg_barrier();
            fir_rows(buf, FH);
#pragma unroll
            for (int i = 0; i < HALO; ++i) w[i] = w[i + F];
            {
                const int64_t m0 = n0 + (int64_t)(it + 1) * F;
                const int vo = frame_off(m0);
#pragma unroll
                for (int f = 0; f < F; ++f) w[HALO + f] = ld(vo, f * D * (int)sizeof(cf), m0 + f);
            }
            wg_barrier();
        }
        wg_barrier();      // drain iteration: the FFT role finishes the last chunk
        wg_barrier();
#ifdef RS_ONLY_FIR
    } else if (false) {
#else
    } else {
#endif
        const int tid = threadIdx.x - NB;
        tw_lds[tid] = p.tw[tid];
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.bins_ring, 0, (int)((int64_t)NB * p.ring_cap * (int64_t)sizeof(cf)), 0x00020000);
        const int k0 = tid / F, f_lane = tid % F;
        const int out_so_step = (int)((int64_t)(NB / F) * p.ring_cap * (int64_t)sizeof(cf));
        wg_barrier();      // iteration 0: nothing to transform yet
        wg_barrier();
        for (int it = 1; it <= nchunks; ++it) {
            cf *buf = buf0 + ((it - 1) & 1) * F * RS;
            const int ch = (it - 1) * F;
            const int nf = min(F, nfr - ch);
            {
                using PL = Plan<NB>;
                pfb_pass<NB, PL::r[0], 1, false, true>(buf, tw_lds, tid);
                if constexpr (PL::n >= 2) pfb_pass<NB, PL::r[1], PL::r[0], false, true>(buf, tw_lds, tid);
                if constexpr (PL::n >= 3) pfb_pass<NB, PL::r[2], PL::r[0] * PL::r[1], false, true>(buf, tw_lds, tid);
            }
            wg_barrier();
            {
                const int64_t n = n0 + ch + f_lane;
                const int ridx = (int)((uint64_t)(n - p.n_abs0) & p.ring_mask);
                const int vo = (int)(((int64_t)k0 * p.ring_cap + ridx) * (int64_t)sizeof(cf));
                if (f_lane < nf) {
#pragma unroll
                    for (int i = 0; i < F; ++i) {
                        const int k = k0 + i * (NB / F);
                        cf v = ((NB / F) % 16 == 0) ? buf[f_lane * RS + lds_pad(k0) + i * ((NB / F) + (NB / F) / 16)]
                                                : buf[f_lane * RS + lds_pad(k)];
                        if (OS == 2) {
                            if ((k & 1) && (n & 1)) v = make_float2(-v.x, -v.y);
                        }
                        u32x2 o;
                        o.x = __float_as_uint(v.x);
                        o.y = __float_as_uint(v.y);
                        __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, i * out_so_step, (POL & 2) ? 2 : 0);
                    }
                }
            }
            wg_barrier();
        }
    }
}

template <int NB, int OS, int P, int POL>
void launch_rs(const PfbLaunch &p, hipStream_t s)
{
    static const int fpw_env = env_int("RCF_PFB_FPW", 0);
    int fpw = 64;
    while (fpw > 2 * F && (p.n_frames + fpw - 1) / fpw < 1024) fpw >>= 1;
    if (fpw_env > 0) fpw = fpw_env;
    const int n_wg = (p.n_frames + fpw - 1) / fpw;
    const size_t lds = ((size_t)2 * F * row_stride<NB>() + NB) * sizeof(cf);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pfb_kernel_rs<NB, OS, P, POL, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pfb_kernel_rs<NB, OS, P, POL, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const bool zh = (p.n_lo - (int64_t)OS * (P - 1)) * (NB / OS) - (NB - 1) < p.start_sample;
    if (zh) hipLaunchKernelGGL((pfb_kernel_rs<NB, OS, P, POL, true>), dim3(n_wg), dim3(2 * NB), lds, s, p, fpw, n_wg);
    else    hipLaunchKernelGGL((pfb_kernel_rs<NB, OS, P, POL, false>), dim3(n_wg), dim3(2 * NB), lds, s, p, fpw, n_wg);
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <int NB, int OS, int P, int MINW, int FB, int POL, int ABL = 0>
void launch_one(const PfbLaunch &p, hipStream_t s)
{
    static const int fpw_env = env_int("RCF_PFB_FPW", 0);
    // One 16-frame chunk per workgroup measured fastest on MI355X: more, shorter workgroups hide the
    // serial load -> FIR -> FFT -> store phases of a chunk better than a longer chunk loop saves on
    // halo re-reads (the halo rows are served by L2 / Infinity Cache, not HBM).
    int fpw = F;
    if (fpw_env > 0) fpw = fpw_env;
    const int n_wg = (p.n_frames + fpw - 1) / fpw;
    const size_t lds = ((size_t)F * row_stride<NB>() + NB) * sizeof(cf);
    // zero-history handling is needed only while a launch can still reach samples before start_sample
    const bool zh = (p.n_lo - (int64_t)OS * (P - 1)) * (NB / OS) - (NB - 1) < p.start_sample;
    if (zh) hipLaunchKernelGGL((pfb_kernel<NB, OS, P, MINW, FB, POL, true, ABL>), dim3(n_wg), dim3(NB), lds, s, p, fpw, n_wg);
    else    hipLaunchKernelGGL((pfb_kernel<NB, OS, P, MINW, FB, POL, false, ABL>), dim3(n_wg), dim3(NB), lds, s, p, fpw, n_wg);
}

template <int NB, int OS, int P, int MINW, int POL, int ABL = 0>
void launch_os(const PfbLaunch &p, hipStream_t s)
{
    const int n_wg = (p.n_frames + F - 1) / F;
    static const int no_remap = env_int("RCF_PFB_NOREMAP", 0);
    const int arg = no_remap ? -n_wg : n_wg;
    const size_t lds = ((size_t)F * row_stride<NB>() + NB) * sizeof(cf);
    const bool zh = (p.n_lo - (int64_t)OS * (P - 1)) * (NB / OS) - (NB - 1) < p.start_sample;
    if (zh) hipLaunchKernelGGL((pfb_kernel_os<NB, OS, P, MINW, POL, true, ABL>), dim3(n_wg), dim3(NB), lds, s, p, arg);
    else    hipLaunchKernelGGL((pfb_kernel_os<NB, OS, P, MINW, POL, false, ABL>), dim3(n_wg), dim3(NB), lds, s, p, arg);
}

int round_p(int P)
{
    if (P <= 4) return 4;
    if (P <= 16) return 16;
    return 0;
}

template <int NB>
bool dispatch_nb(const PfbLaunch &p, int OS, int P, bool probe, hipStream_t s)
{
    const int PR = round_p(P);
    if (PR == 0 || (OS != 1 && OS != 2)) return false;
    if (probe) return true;
    // waves per SIMD the register allocator must allow: 4 workgroups per CU is the LDS limit
    constexpr int MW = NB >= 1024 ? 4 : (NB >= 512 ? 4 : 4 * NB / 256 > 0 ? (4 * NB / 256 > 8 ? 8 : (4 * NB / 256 < 1 ? 1 : 4 * NB / 256)) : 1);
    if (OS == 1) { if (PR == 4) launch_os<NB, 1, 4, MW, 2>(p, s); else launch_os<NB, 1, 16, MW, 2>(p, s); }
    else         { if (PR == 4) launch_os<NB, 2, 4, MW, 2>(p, s); else launch_os<NB, 2, 16, MW, 2>(p, s); }
    return true;
}

bool dispatch(const PfbLaunch &p, bool probe, hipStream_t s)
{
    if (p.D <= 0 || p.NB % p.D) return false;
    const int OS = p.NB / p.D;
    if (p.NB == 256 && OS == 1 && p.P > 8 && p.P <= 14) {          // BASELINE config 2 shape
        if (!probe) {
            static const int variant = env_int("RCF_PFB_VARIANT", 0);
            switch (variant) {
                case 1:  launch_one<256, 1, 14, 3, 16, 2>(p, s); break;
                case 2:  launch_one<256, 1, 14, 4, 8, 2>(p, s); break;
                case 3:  launch_one<256, 1, 14, 3, 8, 2>(p, s); break;
                case 4:  launch_rs<256, 1, 14, 2>(p, s); break;
                case 5:  launch_one<256, 1, 14, 2, 16, 2, 1>(p, s); break;   // ablation: 1-tap FIR
                case 6:  launch_one<256, 1, 14, 2, 16, 2, 2>(p, s); break;   // ablation: no FFT passes
                case 7:  launch_one<256, 1, 14, 2, 16, 2, 3>(p, s); break;   // ablation: both
                case 8:  launch_os<256, 1, 14, 4, 2>(p, s); break;
                case 9:  launch_os<256, 1, 14, 3, 2>(p, s); break;
                case 10: launch_os<256, 1, 14, 2, 2>(p, s); break;
                case 11: launch_one<256, 1, 14, 2, 16, 2>(p, s); break;   // sliding-window kernel
                case 12: launch_os<256, 1, 14, 4, 6>(p, s); break;       // 16-byte stores
                case 14: launch_os<256, 1, 14, 4, 2, 1>(p, s); break;    // ablation: no halo loads
                case 15: launch_os<256, 1, 14, 4, 2, 6>(p, s); break;    // ablation: no FIR math, no FFT
                case 16: launch_os<256, 1, 14, 4, 2, 7>(p, s); break;    // ablation: all three
                case 13: launch_os<256, 1, 14, 4, 4>(p, s); break;       // 16-byte stores, default cache policy
                default: launch_os<256, 1, 14, 4, 2>(p, s); break;
            }
        }
        return true;
    }
    switch (p.NB) {
        case 64:   return dispatch_nb<64>(p, OS, p.P, probe, s);
        case 128:  return dispatch_nb<128>(p, OS, p.P, probe, s);
        case 256:  return dispatch_nb<256>(p, OS, p.P, probe, s);
        case 512:  return dispatch_nb<512>(p, OS, p.P, probe, s);
        case 1024: return dispatch_nb<1024>(p, OS, p.P, probe, s);
        default:   return false;
    }
}

}  // namespace

// taps buffer must hold round-up(P) rows: see pfb_padded_p()
bool pfb_supported(int NB, int D, int P)
{
    PfbLaunch p{};
    p.NB = NB; p.D = D; p.P = P;
    return dispatch(p, true, nullptr);
}

int pfb_padded_p(int NB, int D, int P)
{
    if (NB == 256 && D == 256 && P > 8 && P <= 14) return 14;
    return round_p(P);
}

void launch_pfb(const PfbLaunch &p, hipStream_t s)
{
    if (p.n_frames <= 0) return;
    dispatch(p, false, s);
}

}  // namespace rcfx
