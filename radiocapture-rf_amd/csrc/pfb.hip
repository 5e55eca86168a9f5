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
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
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
template <int NB, int R, int NS, bool END_WG>
__device__ __forceinline__ void pfb_pass(cf *buf, const cf *tw_lds, int tid)
{
    constexpr bool WAVE_LOCAL = (64 % (NB / R)) == 0;
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
        }
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i) Dft<R, +1>::run(v[i]);
    if (WAVE_LOCAL) wave_sync(); else __syncthreads();   // every butterfly has read before any writes
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int frame = (tid + i * NB) / BPF;
        Pass::store(buf + frame * RS, NS, j, v[i]);
    }
    if (WAVE_LOCAL && !END_WG) wave_sync(); else __syncthreads();
}

template <int NB, int OS, int P, int MINW, int FB, int POL, bool ZH>
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
                for (int q = 0; q < P; ++q)
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

        {
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
                    cf v = buf[f_lane * RS + lds_pad(k)];
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

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <int NB, int OS, int P, int MINW, int FB, int POL>
void launch_one(const PfbLaunch &p, hipStream_t s)
{
    static const int fpw_env = env_int("RCF_PFB_FPW", 0);
    int fpw = 32;
    while (fpw > F && (p.n_frames + fpw - 1) / fpw < 2048) fpw >>= 1;
    if (fpw_env > 0) fpw = fpw_env;
    const int n_wg = (p.n_frames + fpw - 1) / fpw;
    const size_t lds = ((size_t)F * row_stride<NB>() + NB) * sizeof(cf);
    // zero-history handling is needed only while a launch can still reach samples before start_sample
    const bool zh = (p.n_lo - (int64_t)OS * (P - 1)) * (NB / OS) - (NB - 1) < p.start_sample;
    if (zh) hipLaunchKernelGGL((pfb_kernel<NB, OS, P, MINW, FB, POL, true>), dim3(n_wg), dim3(NB), lds, s, p, fpw, n_wg);
    else    hipLaunchKernelGGL((pfb_kernel<NB, OS, P, MINW, FB, POL, false>), dim3(n_wg), dim3(NB), lds, s, p, fpw, n_wg);
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
    constexpr int MW = NB >= 1024 ? 1 : 2;      // waves per SIMD the register allocator must allow
    if (OS == 1) { if (PR == 4) launch_one<NB, 1, 4, MW, 16, 2>(p, s); else launch_one<NB, 1, 16, MW, 16, 2>(p, s); }
    else         { if (PR == 4) launch_one<NB, 2, 4, MW, 16, 2>(p, s); else launch_one<NB, 2, 16, MW, 16, 2>(p, s); }
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
                case 4:  launch_one<256, 1, 14, 4, 4, 2>(p, s); break;
                default: launch_one<256, 1, 14, 2, 16, 2>(p, s); break;
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
