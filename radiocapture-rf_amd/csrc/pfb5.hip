// pfb5.hip -- polyphase filterbank for bin counts with a factor 25: NB = 20 * 20 * R3, R3 in {1, 2, 4, 8}
// (400, 800, 1600, 3200 bins).
//
// Why these sizes: the reference's channel (rc_frontend/channel.py:31-35) is freq_xlating_fir_filter_ccc(D, h, f, fs)
// with D = int(fs / cr) / 2 and T = |h| ~ 3.64 D taps; SURVEY.md 7.2 shows that ALL such channels on the
// fs / NB grid are one NB-bin filterbank with the same D and h.  At fs = 20 Msps, cr = 12.5 kHz: D = 800,
// T = 2909, and the grid that carries the reference's channel plans is 12.5 kHz (NB = 1600, OS = NB / D = 2,
// 2 taps per branch) or 6.25 kHz (NB = 3200, OS = 4, 1 tap per branch) -- 2^6 5^2 and 2^7 5^2, not powers of two
// (10 Msps: 800 bins, 5 Msps: 400).  Bin k of this kernel IS the reference's channel at offset k fs / NB: same
// prototype, same decimation, same 25 kS/s output rate, phases exact (SURVEY 7.3 discusses the float32-phase
// difference to GNU Radio's own; tests/test_gpu_round2.py reports it).
//
//   out_k[n] = e^{-j 2 pi k n D / NB} * sum_{rho<NB} e^{+j 2 pi k rho / NB} * u_rho[n]
//   u_rho[n] = sum_{q<P} h[NB q + rho] * x[n D - rho - NB q]
//
// Mapping: workgroup = 320 threads, one chunk of F = 16 / R3 output frames (4 for 1600 bins, 2 for 3200); 53 KB of
// LDS, three workgroups per CU; two workgroup barriers per chunk.
//   NB-point inverse-sign DFT = 20 x 20 x R3 Stockham (Ns = 1, 20, 400), 20 = 5 x 4 as a Good-Thomas prime-factor
//   butterfly held in registers (fft_core.hpp: no twiddles inside a butterfly):
//   * phase A, thread = (frame, j < 20 R3): the branch FIR for the 20 inputs rho = j + 20 R3 t of first-pass
//     butterfly j straight from global memory (x[m D - rho] for consecutive j is a reversed unit-stride run of the
//     interleaved stream; P <= 4 real taps, rows m = n - OS q), the radix-20 butterfly in registers, results to LDS
//     at 20 j + f (rows of 20 padded to 21: the stride-20 writes fall in distinct banks).  No LDS read, so no
//     read-before-write hazard: the first pass costs one LDS write and nothing else.
//   * barrier.
//   * phase B, thread = (k < 20, g < R3, frame) with the frame fastest: second-pass butterfly j = 20 g + k reads
//     positions j + 20 R3 t and writes 400 g + k + 20 f -- every position it touches is = k mod 20, and so is every
//     position of the radix-R3 finish for bins k + 20 i.  The 16 lanes (g, frame) of one k are therefore a closed
//     group inside one wavefront: the in-place hand-offs need wave-local ordering only (DS operations of a wave
//     execute in order), no s_barrier.  Twiddles W_400^{k t} / W_NB^{jj t}: one exact table entry per butterfly,
//     the other powers by binary powering in registers.
//   * the finish writes its bins back in place (bin b at position b, bin phase factor e^{-j 2 pi k n D / NB} = a
//     power of -j applied), one more barrier, then the chunk leaves as WHOLE FRAMES: the output is a frame-major ring
//     bins_ring[(n & mask) NB + k] and every wavefront store is 512 contiguous bytes.  (Per-bin lines would get 32- or
//     16-byte pieces from a 4- or 2-frame chunk: per-bin rings measured 2x slower than the whole rest of the kernel, nt
//     or not; pfb.hip's 16-frame tiles, where those pieces are dense, 4.6x slower than this -- a store that does not
//     fill its 128-byte line is a read-modify-write further down.)  Consumers read one bin with stride NB
//     (StreamView.stride); bins open as channels leave through the tap matrix below.
//   * shapes whose input window is larger than the F frame rows (3200 bins at D = 1600, 800 bins at OS = 1) take the
//     window's size of LDS instead and run two workgroups per CU (pfb5_buf): staging the window is worth more than
//     the third workgroup (0.445 -> 0.48, 0.37 -> 0.49 of the HBM peak).
// Bound: HBM.  Algorithmic bytes per input sample = 8 (read) + 8 NB / D (written) = 24 at OS = 2, 40 at OS = 4.
// Build notes: no SLP vectoriser (packed v_pk_*_f32 arithmetic issues slower than the scalar pair on gfx950: 0.137 ->
// 0.111 ms), no implicit contraction (instantiations must round alike) with the FMAs of the complex products spelled
// out (RCF_EXPLICIT_FMA).  Tried and dropped (git history): one frame per wavefront, 1600 = 25 x 64 with the 64-point
// part done across the lanes by shuffles -- no barriers at all, but 1.7x the arithmetic: 0.118 ms against 0.111.
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#define RCF_EXPLICIT_FMA 1
#include "fft_core.hpp"
#include <hip/hip_ext.h>
#include "rcf_internal.h"
#include "fast_atan2f_gr.hpp"
#include "rotator.hpp"

namespace rcfx {

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef RCF_P5_STORE_AUX
#define RCF_P5_STORE_AUX 2          // non-temporal
#endif
constexpr int kThreads5 = 320;

// padded extent of one frame in LDS: pad5(NB - 1) + 1
constexpr int pfb5_row_stride(int NB, int R) { return NB + NB / R - 1; }

// LDS of one workgroup, in complex samples: the F padded frame rows of the FFT -- or, where the chunk's input window is
// larger than that but still fits TWO workgroups per CU (64 granules of 1280 bytes each), the window: staging it is
// worth more than the third workgroup (3200 bins, D = 1600: window 8000 samples = 64 KB against 53.7 KB of rows)
constexpr int pfb5_win(int NB, int F, int OS, int P) { return (F - 1 + OS * (P - 1)) * (NB / OS) + NB; }
constexpr int pfb5_win_rounds(int win) { return (win + 1 + 2 * kThreads5 - 1) / (2 * kThreads5) * (2 * kThreads5); }   // DMA rounds of 640 samples
constexpr int pfb5_buf(int NB, int R, int F, int OS, int P)
{
    const int rows = F * pfb5_row_stride(NB, R), win = pfb5_win_rounds(pfb5_win(NB, F, OS, P));
    if (!(win > rows && (size_t)win * 8 <= (size_t)64 * 1280)) return rows;
    // ... and behind the window as many whole rows of the prototype as the two-workgroup budget still holds
    int extra = (int)(((size_t)64 * 1280 - (size_t)win * 8) / ((size_t)NB * 4));
    if (extra > P) extra = P;
    return win + extra * (NB / 2);
}

// padded index: one spare complex after every R
template <int R> __device__ __forceinline__ constexpr int pad5(int i) { return i + i / R; }

__device__ __forceinline__ void wave_sync5()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The discriminator fused into the bank (PfbLaunch::fm_ring): FM = 0 the bank as it always was;
// FM_BOTH the frame-major discriminator ring fm_ring[(n & mask) NB + k] beside the bins ring; FM_ONLY the discriminator
// ring INSTEAD of the bins ring (8 + 8 bytes per input sample at OS = 2 instead of 8 + 16, and no tap_finalize pass behind
// it: 48 -> 16 bytes per sample for "every channel.py channel demodulated"); FM_HALO = the pass a workgroup runs on the
// chunk BEFORE its span: nothing is stored, the chunk's last frame stays in the threads' registers (zprev) as the
// predecessor of the span's first discriminator sample.
enum { FM_OFF = 0, FM_BOTH = 1, FM_ONLY = 2, FM_HALO = 3 };
constexpr int pfb5_bins_per_thread(int NB) { return (NB + kThreads5 - 1) / kThreads5; }

// one chunk of one front-end's bank (shared by the single-front-end kernel and the grouped one: same instructions, same bits)
// -DRCF_FM_RCP=1: the fused discriminator's quotient as num * v_rcp_f32(den) (fast_atan2f_gr.hpp) instead of the correctly
// rounded division.  Measured at 1600 bins, discriminator ring only: 0.232 against 0.236 ms per 2^25 samples -- not worth
// giving up the bits of tap_finalize's discriminator (and of gr::fast_atan2f) for.
#ifndef RCF_FM_RCP
#define RCF_FM_RCP 0
#endif

// The words one workgroup hands to the next (look-back form).  Two ways, chosen per launch (PfbLaunch::fm_local):
//   agent scope   atomic stores that write through and atomic loads that miss the vector cache: right wherever the two
//                 workgroups run;
//   one XCD's L2  plain stores (the vector cache writes through to L2, the write-back L2 keeps the line) and the same loads
//                 (they miss the vector cache and are served by L2): right when producer and consumer share an L2 -- which the
//                 chunk map arranges (neighbouring chunks are eight blocks apart, block b runs on XCD b mod 8) and which the
//                 host VERIFIES on the device before it asks for this mode (pfb5_xcd_map_ok: a probe launch reads every
//                 block's XCC_ID).  Measured at 1600 bins: 0.245 -> 0.229 ms per 2^25 samples.  Either way the rows are HBM
//                 traffic in the counters (+ 4 + 4 B per input sample: the ring of rows, 52 MB, does not stay in 4 MB of L2 per
//                 XCD next to the streamed input), and a hand-over that does not arrive is counted (fm_err).
__device__ __forceinline__ void ho_store(const bool local, unsigned long long *row, const int i, const unsigned long long v)
{
    if (local) row[i] = v;
    else       __hip_atomic_store(row + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ho_load(const bool local, const __amdgpu_buffer_rsrc_t rsrc, const unsigned long long *base,
                                                      const size_t word)
{
    // (the load is the agent-scope one either way -- it has to miss the vector cache, and sc0 alone does not: measured, a
    // poll with sc0 never sees the flag; what the local mode changes is the STORE: plain, so that the line stays in L2)
    (void)local; (void)rsrc;
    return __hip_atomic_load(base + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// where entry e of gr::fast_atan2f's table sits in the chunk's LDS, as the PAIR (tab[e], tab[e + 1]) one lookup needs: the
// frame rows have a spare complex after every R -- 256 / F of them are used per row
template <int R, int R3>
__device__ __forceinline__ int pfb5_tab_slot(int e)
{
    constexpr int NB = R * R * R3, F = 16 / R3, PER = 256 / F, RS = pfb5_row_stride(NB, R);
    static_assert(PER <= NB / R - 1 && (PER & (PER - 1)) == 0, "the rows' spare slots hold the 256 table pairs");
    return (e / PER) * RS + (e % PER) * (R + 1) + R;
}

template <int R, int R3, int OS, int P, bool ZH, int FM = FM_OFF, bool LB = false>
__device__ __forceinline__ void pfb5_chunk(const PfbLaunch &p, const int wg, const int tid, cf *buf, cf *zprev = nullptr,
                                           const cf tabpair = cf{0.f, 0.f}, unsigned long long *halo_row = nullptr,
                                           const bool own_halo = false)
{
    constexpr int NB = R * R * R3;
    constexpr int N2 = R * R;                  // W_{N2}^n = e^{+2 pi i n / (R R)} = tw[n R3]
    constexpr int F = 16 / R3;                 // frames per chunk
    constexpr int D = NB / OS;
    constexpr int BPF = NB / R;                // butterflies per frame and pass (= R R3)
    // LDS row stride of a frame (complex) = the padded extent of a frame, which is odd (= -1 mod 16 for 1600 bins):
    // in the second pass's stores the 16 lanes (g, frame) of a k write dwords 2 RS frame + 840 g (+ const) =
    // -2 frame + 8 g mod 32 -- sixteen distinct bank pairs.  (With a stride of 8 mod 16 those stores were 4-way
    // conflicts and LDS conflict cycles were 5x the LDS instructions.)  NOT one more: gfx950 hands out LDS in granules
    // of 1280 bytes (tools/occ_probe.hip: 53760 bytes -> three workgroups per CU, 53776 -> two), and F (extent + 2)
    // complex -- the stride this kernel had first -- is 53792 bytes: 32 bytes too many for the third workgroup.
    constexpr int RS = pfb5_row_stride(NB, R);
    constexpr int BUF = pfb5_buf(NB, R, F, OS, P);       // complex samples of LDS (>= F RS)
    static_assert((size_t)F * RS * sizeof(cf) <= 42 * 1280, "three workgroups per CU need <= 42 LDS granules each");
    static_assert((size_t)BUF * sizeof(cf) <= 64 * 1280, "at least two workgroups per CU");
    static_assert(R == 20 && F * BPF == kThreads5, "one butterfly per thread and pass");
    static_assert(OS == 1 || OS == 2 || OS == 4, "bin phase factor must be a power of -j");
    // -DRCF_PFB5_TRACE: two workgroups print the cycle counts of their phases (how the time of this kernel was found:
    // phase A ~45 % before the window was staged through LDS, phase B ~30 %; hipcc ... -DRCF_PFB5_TRACE -c pfb5.hip, link as another librcf, RCF_LIBRCF=...)
#ifdef RCF_PFB5_TRACE
    long long ts[8];
    ts[0] = clock64();
#define TS(i) ts[i] = clock64()
#else
#define TS(i)
#endif
    const int fb0 = wg * F;
    if (fb0 >= p.n_frames) return;
    const int nf = min(F, p.n_frames - fb0);
    const int64_t n0 = p.n_lo + fb0;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cf *>(p.src.base), 0, (int)(p.src_len * (int64_t)sizeof(cf)), 0x00020000);
    // Phase B's twiddle seeds (one table entry per second-pass butterfly, one per finish sweep: dependent-address loads
    // from L2) requested NOW, at kernel entry, a whole phase A ahead of their use -- for the 3200-bin banks (0.487 ->
    // 0.54 of the HBM peak at D = 800, 0.478 -> 0.51 at D = 1600) and the 400-bin one (one seed: 0.650 -> 0.660).
    // NOT for 800 / 1600 bins: measured 0.606 -> 0.47 and 0.598 -> 0.49 there (phase A then starts behind six more
    // loads per thread).
    constexpr bool SE = R3 == 8 || R3 == 1;
    cf seed2 = make_float2(0.f, 0.f), seedf[(R + R3 - 1) / R3];
    if constexpr (SE) {
        const int g_ = (tid / F) % R3, k_ = tid / (F * R3);
        seed2 = p.tw[k_ * R3];
#pragma unroll
        for (int s_ = 0; s_ < (R + R3 - 1) / R3; ++s_) {
            const int i_ = g_ + R3 * s_;
            seedf[s_] = p.tw[(k_ + R * (i_ < R ? i_ : 0)) % NB];
        }
    }

    // ---- phase A: branch FIR + first radix-20 pass.  u[t] = sum_q h[NB q + rho_t] x[(n - OS q) D - rho_t],
    // rho_t = j + BPF t; X[f] -> buf[20 j + f]
    // Every input sample of the chunk's window is used by up to OS P (frame, branch) pairs -- by different threads.
    // Where the window (WIN samples) fits the chunk's LDS buffer, which is idle until the first pass writes it, the
    // workgroup first stages the window there with coalesced loads -- ONE round trip, 18 loads per thread instead
    // of 40 in two dependent rounds, each sample fetched once per workgroup instead of 2.3 times -- and the branch
    // FIR reads LDS (consecutive lanes = consecutive samples, conflict free).  Shapes whose window does not fit
    // (OS = 1 with 4 taps per branch) keep the direct form.
    constexpr int WIN = (F - 1 + OS * (P - 1)) * D + NB;
    constexpr bool STAGE = WIN <= BUF;
    {
        const int frame = tid / BPF, j = tid % BPF;
        const int64_t n = n0 + frame;
        cf vv[R];
#pragma unroll
        for (int t = 0; t < R; ++t) vv[t] = make_float2(0.f, 0.f);
        if constexpr (STAGE) {
            const int64_t m_lo = (n0 - OS * (P - 1)) * D - (NB - 1);          // first sample of the window
            float h[P][R];
            int odd = 0;
            int hq_staged = 0;                                               // (compile-time after inlining)
            const float *h_lds = nullptr;
            if constexpr (!ZH) {
                // steady state: the window goes global -> LDS directly (buffer_load_dwordx4 ... lds, 16 bytes per lane,
                // a wavefront = 1 KB contiguous on both sides): no staging registers, 9 requests per thread instead of
                // 18 loads + 18 LDS writes.  The window starts at an even sample so that every request is 16-byte
                // aligned; rounds past its end read zeros (descriptor range) into LDS nobody looks at.
                constexpr int PER = kThreads5 * 2;                            // samples per round of the workgroup
                constexpr int NLD = (WIN + 1 + PER - 1) / PER;
                static_assert((size_t)NLD * PER * sizeof(cf) <= (size_t)BUF * sizeof(cf), "window rounds fit the buffer");
                {
                    // the prototype rows that do not fit LDS come from L2 with dependent addresses: requested BEFORE the
                    // window's DMA rounds rather than behind them (+1 % at 1600 bins: 0.604 -> 0.610 of the HBM peak)
                    constexpr int WB_ = NLD * PER * (int)sizeof(cf);
                    constexpr int HQA_ = ((int)(BUF * sizeof(cf)) - WB_) / (NB * (int)sizeof(float));
                    constexpr int HQA = HQA_ > P ? P : HQA_;
#pragma unroll
                    for (int q = HQA; q < P; ++q)
#pragma unroll
                        for (int t = 0; t < R; ++t) h[q][t] = p.ptaps[q * NB + j + BPF * t];
                    __builtin_amdgcn_sched_barrier(0);
                }
                odd = (int)((m_lo - p.src.origin) & 1);
                const int vo0 = (int)((m_lo - odd - p.src.origin) * (int64_t)sizeof(cf)) + tid * 16;
                unsigned char *lds_wave = reinterpret_cast<unsigned char *>(buf) + (tid >> 6) * (64 * 16);
#pragma unroll
                for (int r = 0; r < NLD; ++r)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void *)(lds_wave + r * PER * (int)sizeof(cf)),
                                                             16, vo0, r * PER * (int)sizeof(cf), 0, 0);
                // The prototype taps a thread needs (P x 20 floats, the same for every chunk) are 40 dependent-address
                // loads from L2 per thread -- global loads are what this kernel pays for most (19 table loads in the
                // second pass cost 13 %).  As many whole rows h[q][.] as fit behind the window in the buffer go the
                // same direct way, one or two requests per thread, and are read back from LDS (consecutive lanes =
                // consecutive floats); the rest still comes from L2.
                constexpr int WIN_BYTES = NLD * PER * (int)sizeof(cf);
                constexpr int HROW = NB * (int)sizeof(float);
                constexpr int HQ_ = ((int)(BUF * sizeof(cf)) - WIN_BYTES) / HROW;
                constexpr int HQ = HQ_ > P ? P : HQ_;                         // rows of h staged in LDS
                if constexpr (HQ > 0) {
                    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(p.ptaps), 0, P * HROW, 0x00020000);
                    constexpr int HR = (HQ * HROW + kThreads5 * 16 - 1) / (kThreads5 * 16);
#pragma unroll
                    for (int r = 0; r < HR; ++r)
                        if ((r * kThreads5 + tid) * 16 < HQ * HROW)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                h_rsrc, (__attribute__((address_space(3))) void *)(lds_wave + WIN_BYTES + r * kThreads5 * 16), 16,
                                tid * 16, r * kThreads5 * 16, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0);                               // the DMA writes count on vmcnt
                hq_staged = HQ;
                h_lds = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(buf) + WIN_BYTES);
            } else {
                constexpr int NLD = (WIN + kThreads5 - 1) / kThreads5;
                const int vo0 = (int)((m_lo + tid - p.src.origin) * (int64_t)sizeof(cf));
                cf xs[NLD];
#pragma unroll
                for (int r = 0; r < NLD; ++r) {
                    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo0, r * kThreads5 * (int)sizeof(cf), 0);
                    xs[r] = make_float2(__uint_as_float(w.x), __uint_as_float(w.y));
                }
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int t = 0; t < R; ++t) h[q][t] = p.ptaps[q * NB + j + BPF * t];
#pragma unroll
                for (int r = 0; r < NLD; ++r) {
                    const int idx = tid + r * kThreads5;
                    if (m_lo + idx < p.start_sample) xs[r] = make_float2(0.f, 0.f);
                    if (NLD * kThreads5 == WIN || idx < WIN) buf[idx] = xs[r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < P; ++q)
                if (q < hq_staged) {
#pragma unroll
                    for (int t = 0; t < R; ++t) h[q][t] = h_lds[q * NB + j + BPF * t];
                }
            // x[(n - OS q) D - j - BPF t] = window[(frame + OS (P - 1 - q)) D + NB - 1 - j - BPF t]
            const cf *sb = buf + frame * D + BPF - 1 - j + odd;             // q = P - 1, t = R - 1
#pragma unroll
            for (int q = 0; q < P; ++q) {
                cf x[R];
#pragma unroll
                for (int t = 0; t < R; ++t) x[t] = sb[OS * (P - 1 - q) * D + BPF * (R - 1 - t)];
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    vv[t].x = fmaf(h[q][t], x[t].x, vv[t].x);
                    vv[t].y = fmaf(h[q][t], x[t].y, vv[t].y);
                }
            }
            __syncthreads();                                 // every thread has read the window before it is overwritten
        } else {
            // lowest address this thread reads: row n - OS (P - 1), branch j + BPF (R - 1); everything else is a
            // compile-time constant above it
            const int vo = (int)(((n - OS * (P - 1)) * D - j - BPF * (R - 1) - p.src.origin) * (int64_t)sizeof(cf));
#pragma unroll
            for (int q = 0; q < P; ++q) {
                cf x[R];
                float h[R];
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    // x[(n - OS q) D - j - BPF t]
                    const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(
                        in_rsrc, vo + (OS * (P - 1 - q) * D + BPF * (R - 1 - t)) * (int)sizeof(cf), 0, 0);
                    x[t] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
                    h[t] = p.ptaps[q * NB + j + BPF * t];
                }
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    if (ZH && (n - OS * q) * D - (j + BPF * t) < p.start_sample) x[t] = make_float2(0.f, 0.f);
                    vv[t].x = fmaf(h[t], x[t].x, vv[t].x);
                    vv[t].y = fmaf(h[t], x[t].y, vv[t].y);
                }
            }
        }
        Dft<R, +1>::run(vv);
        cf *o = buf + frame * RS + j * (R + 1);              // pad5(j R + f) = j (R + 1) + f for f < R
#pragma unroll
        for (int f = 0; f < R; ++f) o[f] = vv[Dft<R, +1>::reg_of(f)];
        if constexpr (FM == FM_BOTH || FM == FM_ONLY) {
            // gr::fast_atan2f's table for the copy-out's discriminator.  The LDS budget has no 1 KB left (53 728 of the 53 760
            // bytes three workgroups per CU allow), but the frame rows have a spare complex after every R: entry e of the table,
            // as the PAIR (tab[e], tab[e + 1]) one lookup needs, sits in spare slot e -- PADS = NB / R - 1 slots per row.  The
            // window DMA of every chunk runs over them, so thread e (which keeps its pair in two registers for the whole
            // span) puts it back here, once the window has been read and together with the first pass's own writes.
            if constexpr (!LB)
                if (tid < 256) buf[pfb5_tab_slot<R, R3>(tid)] = tabpair;
        }
    }
    TS(1);
    __syncthreads();
    TS(2);
    // (look-back form: the table pair is requested HERE, a whole second pass ahead of the barrier it is written behind -- at
    // kernel entry two more dependent-address loads per thread set the first pass back, the reason the twiddle seeds are not
    // requested there either)
    cf tabpair_lb = make_float2(0.f, 0.f);
    if constexpr (LB && (FM == FM_BOTH || FM == FM_ONLY))
        if (tid < 256) tabpair_lb = make_float2(p.atan_tab[tid], p.atan_tab[tid + 1]);

    // ---- phase B: second pass + radix-R3 finish + stores, closed inside the 16 lanes of one k
    {
        const int frame = tid % F, g = (tid / F) % R3, k = tid / (F * R3);      // F R3 = 16
        cf *fbuf = buf + frame * RS;
        // pass 2 (Ns = R): j = R g + k; v[t] = buf[j + t BPF] W_{R R}^{k t}; X[f] -> buf[N2 g + k + f R]
        {
            cf vv[R];
            const cf *rd = fbuf + pad5<R>(R * g + k);        // pad5(j + t BPF) = pad5(j) + t (BPF + R3): BPF = R R3
#pragma unroll
            for (int t = 0; t < R; ++t) vv[t] = rd[t * (BPF + R3)];
            {
                // W_{R R}^{k t}, t < R, from ONE table entry by binary powering (depth <= 5 multiplications, a few
                // 1e-7 of error): nineteen table loads per thread were a third of the kernel's L2 traffic
                cf w[R];
                twiddle_powers<R>(SE ? seed2 : p.tw[k * R3], w);
#pragma unroll
                for (int t = 1; t < R; ++t) vv[t] = cmul(vv[t], w[t]);
            }
            Dft<R, +1>::run(vv);
            wave_sync5();                                    // the group's butterflies have read before any writes
            cf *o = fbuf + pad5<R>(N2 * g + k);              // pad5(N2 g + k + f R) = pad5(N2 g + k) + f (R + 1)
#pragma unroll
            for (int f = 0; f < R; ++f) o[f * (R + 1)] = vv[Dft<R, +1>::reg_of(f)];
            wave_sync5();
        }
        // finish (Ns = N2, radix R3): butterfly jj = k + R i reads positions jj + t N2, twiddles W_NB^{jj t},
        // yields bins jj + f N2 -- written back in place (bin b at position b), with the bin phase factor;
        // lane g takes i = g, g + R3, ...
        const int64_t n = n0 + frame;
        const int nph = (int)(n & 3);
        auto phase = [&](cf z, int kk) {                     // e^{-j 2 pi kk n / OS} = (-j)^{(4 / OS) kk n}
            if (OS == 1) return z;
            const int e = ((4 / OS) * (kk & 3) * nph) & 3;
            if (e == 1) return make_float2(z.y, -z.x);       // * -j
            if (e == 2) return make_float2(-z.x, -z.y);
            if (e == 3) return make_float2(-z.y, z.x);       // * +j
            return z;
        };
        constexpr int SWEEPS = (R + R3 - 1) / R3;
#pragma unroll
        for (int s = 0; s < SWEEPS; ++s) {
            const int i = g + R3 * s;
            if (R % R3 != 0 && i >= R) break;
            const int jj = k + R * i;
            cf *pos = fbuf + pad5<R>(jj);
            cf w[R3];
#pragma unroll
            for (int t = 0; t < R3; ++t) w[t] = pos[t * (N2 + N2 / R)];
            if (R3 > 1) {
                {
                    cf wp[R3 > 1 ? R3 : 2];
                    twiddle_powers<(R3 > 1 ? R3 : 2)>(SE ? seedf[s] : p.tw[jj], wp);       // W_NB^{jj t} from one entry
#pragma unroll
                    for (int t = 1; t < R3; ++t) w[t] = cmul(w[t], wp[t]);
                }
                Dft<R3, +1>::run(w);
            }
#pragma unroll
            for (int f = 0; f < R3; ++f)
                pos[f * (N2 + N2 / R)] = phase(R3 > 1 ? w[Dft<R3, +1>::reg_of(f)] : w[0], jj + f * N2);
        }
    }
    TS(3);
    if constexpr (FM == FM_HALO) {
        // the chunk before the span: its last frame, bin tid + 320 bb in zprev[bb], is all that is wanted of it
        __syncthreads();
        constexpr int NBT = pfb5_bins_per_thread(NB);
        const cf *row = buf + (F - 1) * RS;
#pragma unroll
        for (int bb = 0; bb < NBT; ++bb) {
            const int bin = tid + bb * kThreads5;
            const cf z = (NB % kThreads5 == 0 || bin < NB) ? row[pad5<R>(bin)] : make_float2(0.f, 0.f);
            if constexpr (LB) {
                // (a global row instead of registers: the thread reads its own words back after the chunk proper)
                if (NB % kThreads5 == 0 || bin < NB)
                    ho_store(p.fm_local != 0, halo_row, bin, ((unsigned long long)__float_as_uint(z.y) << 32) | __float_as_uint(z.x));
            } else {
                zprev[bb] = z;
            }
        }
        return;
    }
    // ---- taps first, copy-out last: bins that are open as channels are copied into the launch's compact tap matrix,
    // tap_mat[(frame - n_lo) tap_pitch + slot - tap_first] -- lanes = consecutive slots, so every wavefront store is 512 contiguous
    // bytes of a row.  Rotators and the discriminator are tap_finalize_kernel's business (tapfin.hip).  The bin numbers
    // are requested before the barrier.
    constexpr int TAP_IT = 5;                                // 5 x 320 slots cover all 1600 bins; 3200 bins loop
    // slots below tap_first are whole aligned runs of 16 bins: tap_finalize reads those from the ring this kernel
    // writes anyway, and nothing is copied for them here
    const int n_mat = p.n_taps - p.tap_first;
    int tap_bin[TAP_IT];
#pragma unroll
    for (int it = 0; it < TAP_IT; ++it) {
        const int sl = tid + it * kThreads5;
        tap_bin[it] = sl < n_mat ? p.tap_bins[p.tap_first + sl] : -1;
    }
    // (fused discriminator: the bins' rotator increments, requested before the barrier like the tap numbers -- carried across
    // the chunks of a span they would be 2 NBT registers held through the FIR and both FFT passes)
    cf finc[FM == FM_BOTH || FM == FM_ONLY ? pfb5_bins_per_thread(NB) : 1];
    if constexpr (FM == FM_BOTH || FM == FM_ONLY) {
#pragma unroll
        for (int bb = 0; bb < pfb5_bins_per_thread(NB); ++bb) {
            const int bin = tid + bb * kThreads5;
            finc[bb] = (NB % kThreads5 == 0 || bin < NB) ? p.fm_inc[bin] : make_float2(1.f, 0.f);
        }
        if constexpr (LB)
            if (tid < 256) buf[pfb5_tab_slot<R, R3>(tid)] = tabpair_lb;      // (spare slots: no pass of the FFT touches them)
    }
    __syncthreads();
    TS(4);
    if (n_mat > 0) {
        float2 *trow = p.tap_mat + (size_t)fb0 * p.tap_pitch;        // column = slot - tap_first
#pragma unroll
        for (int it = 0; it < TAP_IT; ++it) {
            const int sl = tid + it * kThreads5;
            if (tap_bin[it] < 0) break;
            const cf *col = buf + pad5<R>(tap_bin[it]);
#pragma unroll
            for (int f = 0; f < F; ++f)
                if (f < nf) trow[(size_t)f * p.tap_pitch + sl] = col[f * RS];
        }
        for (int sl = tid + TAP_IT * kThreads5; sl < n_mat; sl += kThreads5) {          // more than 1600 taps (3200 bins)
            const cf *col = buf + pad5<R>(p.tap_bins[p.tap_first + sl]);
            for (int f = 0; f < nf; ++f) trow[(size_t)f * p.tap_pitch + sl] = col[f * RS];
        }
    }

    TS(5);
    if constexpr (FM == FM_BOTH || FM == FM_ONLY) {
        // ---- copy-out with the discriminator fused in: frame f of bin k leaves as fm_ring[((n0 + f) & mask) NB + k] =
        // fast_atan2f(bin[n] conj(bin[n - 1]) x inc_k) -- tap_finalize's discriminator-only arithmetic (tapfin.hip: the
        // discriminator of the ROTATED stream is the bare product turned by the rotator's one increment).
        // Lanes = consecutive bins: every wavefront store is 256 contiguous bytes of a frame row.  A thread walks ITS bins
        // down the frames, so the predecessor is a register; across chunks it is zprev (span form) or the row the
        // workgroup before this one published (look-back form).
        constexpr int NBT = pfb5_bins_per_thread(NB);
        auto lookup = [&](int e, float &t0, float &t1) {
            const cf pr = buf[pfb5_tab_slot<R, R3>(e)];
            t0 = pr.x; t1 = pr.y;
        };
        auto emit = [&](const int f, const int bin, const cf z, const cf prev, const cf inc) {
            // volk_32fc_x2_multiply_conjugate_32fc (unfused), then the turn by the rotator's increment
            const float tr = __fadd_rn(__fmul_rn(z.x, prev.x), __fmul_rn(z.y, prev.y));
            const float ti = __fsub_rn(__fmul_rn(z.y, prev.x), __fmul_rn(z.x, prev.y));
            const float ur = __fsub_rn(__fmul_rn(tr, inc.x), __fmul_rn(ti, inc.y));
            const float ui = __fadd_rn(__fmul_rn(tr, inc.y), __fmul_rn(ti, inc.x));
#ifdef RCF_X_NOATAN
            const float fm = ui + ur;
#elif defined(RCF_X_NODISC)
            const float fm = z.x;
#else
            const float fm = fast_atan2f_gr_lut<decltype(lookup), (RCF_FM_RCP != 0)>(ui, ur, lookup);
#endif
            // one descriptor per frame row (scalar arithmetic, redone per bin column: a table of F of them is 64
            // SGPRs at 400 bins and went to scratch)
            const int64_t slot = (int64_t)((uint64_t)(n0 + f - p.n_abs0) & p.ring_mask);
            const __amdgpu_buffer_rsrc_t fm_rsrc =
                __builtin_amdgcn_make_buffer_rsrc(p.fm_ring + slot * NB, 0, NB * (int)sizeof(float), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fm), fm_rsrc, bin * (int)sizeof(float), 0, RCF_P5_STORE_AUX);
            if constexpr (FM == FM_BOTH) {
                const __amdgpu_buffer_rsrc_t iq_rsrc =
                    __builtin_amdgcn_make_buffer_rsrc(p.bins_ring + slot * NB, 0, NB * (int)sizeof(cf), 0x00020000);
                u32x2 o;
                o.x = __float_as_uint(z.x);
                o.y = __float_as_uint(z.y);
                __builtin_amdgcn_raw_buffer_store_b64(o, iq_rsrc, bin * (int)sizeof(cf), 0, RCF_P5_STORE_AUX);
            }
        };
        if constexpr (LB) {
            // ---- look-back form: one chunk per workgroup, no arithmetic done twice, no workgroup barrier added.
            // A thread handles the same bins (tid + 320 bb) in every chunk, so the hand-over is WAVE to WAVE: wave w of this
            // workgroup publishes its 64 x NBT words of the chunk's last frame and its own flag, wave w of the next chunk's
            // workgroup waits for that flag only.
            // Every word that is handed over is itself an agent-scope atomic access (coherent in L2 whichever CU asks: the
            // stores write through, the loads bypass the vector cache), so no cache maintenance is wanted -- an agent-scope
            // release / acquire FENCE pair writes back and invalidates the whole L2 on this chip and made the kernel 7x
            // slower.  What the order needs is that a wave's row stores have been acknowledged before its flag store is
            // issued: s_waitcnt vmcnt(0) of that wave alone.
            const int wave = tid >> 6;
            const int my_slot = wg % p.fm_slots;
            const bool local = p.fm_local != 0;
            const __amdgpu_buffer_rsrc_t edge_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                p.fm_edge, 0, (int)((size_t)(p.fm_slots + 9) * NB * sizeof(unsigned long long)), 0x00020000);
            const __amdgpu_buffer_rsrc_t flag_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                p.fm_flag, 0, (int)((size_t)p.fm_slots * 8 * sizeof(unsigned long long)), 0x00020000);
            // (1) the chunk's LAST frame -> this chunk's edge row
            {
                unsigned long long *edge = p.fm_edge + (size_t)my_slot * NB;
                const cf *last = buf + (nf - 1) * RS;
#pragma unroll
                for (int bb = 0; bb < NBT; ++bb) {
                    const int bin = tid + bb * kThreads5;
                    if (NB % kThreads5 != 0 && bin >= NB) break;
                    const cf z = last[pad5<R>(bin)];
                    ho_store(local, edge, bin, ((unsigned long long)__float_as_uint(z.y) << 32) | __float_as_uint(z.x));
                }
                // ... and the wave's flag, once its words have landed (measured: with the flag behind the frames 1 .. nf - 1
                // work the wait is for THEIR streaming stores too, 0.248 -> 0.278 ms)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((tid & 63) == 0)
                    ho_store(local, p.fm_flag, my_slot * 8 + wave, p.fm_tag + (unsigned long long)(unsigned)wg);
            }
            // (2) the predecessor of frame 0 is the last frame of the chunk before: the row that chunk's workgroup published
            //     (same XCD, dispatched eight blocks earlier: usually there by now) or, for the first workgroup of an XCD's
            //     range, the row this workgroup computed for itself before its own chunk.  Its words are REQUESTED here, if the
            //     flag is already up, and used last: the loads (they bypass the vector cache) fly during the frames 1 .. nf - 1 work.
            size_t src_w = (size_t)(halo_row - p.fm_edge);             // the predecessor row's first word in the edge buffer
            size_t flag_w = 0;
            unsigned long long want = 0;
            bool have = own_halo;
            if (!own_halo) {
                const int ps = (wg - 1) % p.fm_slots;
                src_w = (size_t)ps * NB;
                flag_w = (size_t)ps * 8 + wave;
                want = p.fm_tag + (unsigned long long)(unsigned)(wg - 1);
                have = ho_load(local, flag_rsrc, p.fm_flag, flag_w) == want;
                asm volatile("" ::: "memory");                         // (the row's loads are issued after the flag was seen)
            }
            unsigned long long pw[NBT];
#pragma unroll
            for (int bb = 0; bb < NBT; ++bb) pw[bb] = 0;
            if (have) {
#pragma unroll
                for (int bb = 0; bb < NBT; ++bb) {
                    const int bin = tid + bb * kThreads5;
                    if (NB % kThreads5 == 0 || bin < NB) pw[bb] = ho_load(local, edge_rsrc, p.fm_edge, src_w + bin);
                }
            }
            // (3) frames 1 .. nf - 1: the predecessor is in LDS
#pragma unroll
            for (int bb = 0; bb < NBT; ++bb) {
                const int bin = tid + bb * kThreads5;
                if (NB % kThreads5 != 0 && bin >= NB) break;
                cf prev = buf[pad5<R>(bin)];
                const cf inc = finc[bb];
#pragma unroll
                for (int f = 1; f < F; ++f) {
                    if (f >= nf) break;
                    const cf z = buf[f * RS + pad5<R>(bin)];
                    emit(f, bin, z, prev, inc);
                    prev = z;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (4) frame 0.  A predecessor that was not there before is waited for now -- a bounded wait: one that never
            //     arrives is counted (fm_err) and the frame's samples are computed against whatever the row holds.
            if (!have) {
                int tries = 0;
                while (ho_load(local, flag_rsrc, p.fm_flag, flag_w) != want) {
                    __builtin_amdgcn_s_sleep(4);
                    asm volatile("" ::: "memory");                     // (every poll is a load)
                    if (++tries > (1 << 22)) { if ((tid & 63) == 0) atomicAdd(p.fm_err, 1); break; }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int bb = 0; bb < NBT; ++bb) {
                    const int bin = tid + bb * kThreads5;
                    if (NB % kThreads5 == 0 || bin < NB) pw[bb] = ho_load(local, edge_rsrc, p.fm_edge, src_w + bin);
                }
            }
#pragma unroll
            for (int bb = 0; bb < NBT; ++bb) {
                const int bin = tid + bb * kThreads5;
                if (NB % kThreads5 != 0 && bin >= NB) break;
                const cf prev = make_float2(__uint_as_float((unsigned)pw[bb]), __uint_as_float((unsigned)(pw[bb] >> 32)));
                emit(0, bin, buf[pad5<R>(bin)], prev, finc[bb]);
            }
            return;
        }
#pragma unroll
        for (int bb = 0; bb < NBT; ++bb) {
            const int bin = tid + bb * kThreads5;
            if (NB % kThreads5 != 0 && bin >= NB) break;
            cf prev = zprev[bb];
            const cf inc = finc[bb];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (f >= nf) break;
                const cf z = buf[f * RS + pad5<R>(bin)];
                emit(f, bin, z, prev, inc);
                prev = z;
            }
            zprev[bb] = prev;
            // (one bin's frames at a time: left to itself the scheduler interleaves all NBT x F discriminators -- every
            // LDS load hoisted to the top, 168 VGPRs and 40 of them spilled)
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // ---- copy-out: whole frames, bins consecutive across lanes -- every wavefront store is 512 contiguous bytes of
    // the frame-major ring bins_ring[(n & mask) NB + k]
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (f >= nf) break;
        // one descriptor per frame row, based at the row itself: the ring (3200 bins x 2^17 frames = 3.4 GB) has no
        // 2 GiB limit, offsets inside a row stay 32-bit
        const int64_t slot = (int64_t)((uint64_t)(n0 + f - p.n_abs0) & p.ring_mask);
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.bins_ring + slot * NB, 0, NB * (int)sizeof(cf), 0x00020000);
        constexpr int so = 0;
        const cf *row = buf + f * RS;
#pragma unroll
        for (int bb = 0; bb < NB / kThreads5; ++bb) {
            const int bin = tid + bb * kThreads5;
            const cf z = row[pad5<R>(bin)];
            u32x2 o;
            o.x = __float_as_uint(z.x);
            o.y = __float_as_uint(z.y);
            __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, bin * (int)sizeof(cf), so, RCF_P5_STORE_AUX);
        }
        if (NB % kThreads5 != 0) {
            const int bin = tid + (NB / kThreads5) * kThreads5;
            if (bin < NB) {
                const cf z = row[pad5<R>(bin)];
                u32x2 o;
                o.x = __float_as_uint(z.x);
                o.y = __float_as_uint(z.y);
                __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, bin * (int)sizeof(cf), so, RCF_P5_STORE_AUX);
            }
        }
    }
#ifdef RCF_PFB5_TRACE
    TS(6);
    __builtin_amdgcn_s_waitcnt(0);
    TS(7);
    if ((wg == 1000 || wg == 3000) && (tid == 0 || tid == 256))
        printf("wg %d tid %d: A %lld bar1 %lld B %lld bar2 %lld taps %lld copy %lld acks %lld\n", wg, tid, ts[1] - ts[0], ts[2] - ts[1],
               ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6]);
#endif
}

template <int R, int R3, int OS, int P, bool ZH>
__global__ __launch_bounds__(kThreads5, 3) void pfb5_kernel(PfbLaunch p, int n_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    // neighbouring chunks (they share input rows and complete each other's 128-byte output lines) on one XCD
    const int b = blockIdx.x, q8 = n_wg / 8, r8 = n_wg % 8, xcd = b % 8;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    pfb5_chunk<R, R3, OS, P, ZH>(p, wg, threadIdx.x, buf);
}

// The bank with the discriminator fused in: a workgroup walks `span` consecutive chunks.  It first runs the chunk BEFORE its
// span without storing anything (the halo: one chunk of arithmetic per span, 1 / span of overhead) -- the span's first
// discriminator sample needs the frame before it, and no other workgroup's output can be waited for inside a launch.
template <int R, int R3, int OS, int P, bool ZH, int FM>
__device__ __forceinline__ void pfb5_fm_span(const PfbLaunch &p, const int wg, const int tid, cf *buf)
{
    constexpr int NB = R * R * R3, F = 16 / R3, NBT = pfb5_bins_per_thread(NB);
    const int n_chunks = (p.n_frames + F - 1) / F;
    const int c0 = wg * p.fm_span, c1 = min(c0 + p.fm_span, n_chunks);
    if (c0 >= n_chunks) return;
    cf zprev[NBT];
    const cf tabpair = tid < 256 ? make_float2(p.atan_tab[tid], p.atan_tab[tid + 1]) : make_float2(0.f, 0.f);
#ifndef RCF_X_NOHALO
    pfb5_chunk<R, R3, OS, P, ZH, FM_HALO>(p, c0 - 1, tid, buf, zprev);
#else
    for (int bb = 0; bb < NBT; ++bb) zprev[bb] = make_float2(0.f, 0.f);
#endif
    for (int c = c0; c < c1; ++c) {
        // Nothing may be carried from chunk to chunk but zprev / tabpair.  Everything a chunk derives from the thread index
        // (twenty 64-bit prototype-row addresses, the LDS addresses of three phases) is loop invariant, and hoisted out of
        // this loop it cost 168 VGPRs + 34 spilled: the thread index is made opaque per chunk, the chunk recomputes them
        // as the one-chunk kernel does (121 VGPRs, none spilled, three workgroups per CU).
        int tid_c = tid;
        asm volatile("" : "+v"(tid_c) :: "memory");
        __syncthreads();                                   // the chunk before has been read out of LDS
        pfb5_chunk<R, R3, OS, P, ZH, FM>(p, c, tid_c, buf, zprev, tabpair);
    }
}

// (HIP's second launch bound is waves per SIMD, not workgroups per CU: three workgroups of five waves need FOUR per SIMD,
// i.e. <= 128 VGPRs; the shapes whose LDS allows two workgroups only get the 168 of three)
constexpr int pfb5_fm_waves(int R, int R3, int OS, int P)
{
    return (size_t)pfb5_buf(R * R * R3, R, 16 / R3, OS, P) * sizeof(cf) <= (size_t)42 * 1280 ? 4 : 3;
}

template <int R, int R3, int OS, int P, bool ZH, int FM>
__global__ __launch_bounds__(kThreads5, pfb5_fm_waves(R, R3, OS, P)) void pfb5_fm_kernel(PfbLaunch p, int n_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    const int b = blockIdx.x, q8 = n_wg / 8, r8 = n_wg % 8, xcd = b % 8;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    pfb5_fm_span<R, R3, OS, P, ZH, FM>(p, wg, threadIdx.x, buf);
}

// The look-back form of the same: ONE chunk per workgroup, as pfb5_kernel has it, and the frame before a chunk's first one
// handed over by the workgroup of the chunk before (pfb5_chunk: LB).  Neighbouring chunks run on one XCD, eight blocks
// apart in dispatch order: the wait at the end of the copy-out practically never waits.  Only the first workgroup of each
// XCD's range (its predecessor chunk belongs to the END of another XCD's range, dispatched much later) computes the frame
// before its chunk itself: eight halo chunks per launch instead of one per span.
template <int R, int R3, int OS, int P, bool ZH, int FM>
__global__ __launch_bounds__(kThreads5, pfb5_fm_waves(R, R3, OS, P)) void pfb5_fmlb_kernel(PfbLaunch p, int n_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    constexpr int NB = R * R * R3;
    const int tid = threadIdx.x;
    const int b = blockIdx.x, q8 = n_wg / 8, r8 = n_wg % 8, xcd = b % 8;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
    const bool own_halo = b / 8 == 0;
    const cf tabpair = make_float2(0.f, 0.f);              // (requested inside the chunk: pfb5_chunk, LB)
    unsigned long long *halo_row = p.fm_edge + (size_t)(p.fm_slots + xcd) * NB;
    if (own_halo) {
        int tid_h = tid;
        asm volatile("" : "+v"(tid_h) :: "memory");        // (nothing of the halo chunk's addressing survives into the chunk proper)
        pfb5_chunk<R, R3, OS, P, ZH, FM_HALO, true>(p, wg - 1, tid_h, buf, nullptr, tabpair, halo_row);
        __syncthreads();
    }
    pfb5_chunk<R, R3, OS, P, ZH, FM, true>(p, wg, tid, buf, nullptr, tabpair, halo_row, own_halo);
}

// ... and the banks of G front-ends with the discriminator fused in, in ONE launch (rcf_group.cpp): the same hand-over inside
// every front-end's run of chunks; the first chunk of a front-end (its predecessor frame is the previous BLOCK's last one)
// computes the chunk before it itself, like the first workgroup of an XCD's range.  Steady state only.
template <int R, int R3, int OS, int P, int FM>
__global__ __launch_bounds__(kThreads5, pfb5_fm_waves(R, R3, OS, P)) void pfb5_fmlb_group_kernel(const PfbLaunch *__restrict__ pls, GroupMap gm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    constexpr int NB = R * R * R3;
    const int tid = threadIdx.x;
    int fe, wg;
    group_resolve(gm, blockIdx.x, fe, wg);
    const PfbLaunch p = pls[fe];
    const bool own_halo = blockIdx.x / 8 == 0 || wg == 0;
    const cf tabpair = make_float2(0.f, 0.f);
    // (rows fm_slots .. fm_slots + 7: the XCD ranges' first workgroups; row fm_slots + 8: the front-end's first chunk)
    unsigned long long *halo_row = p.fm_edge + (size_t)(p.fm_slots + (wg == 0 ? 8 : (int)(blockIdx.x % 8))) * NB;
    if (own_halo) {
        int tid_h = tid;
        asm volatile("" : "+v"(tid_h) :: "memory");
        pfb5_chunk<R, R3, OS, P, false, FM_HALO, true>(p, wg - 1, tid_h, buf, nullptr, tabpair, halo_row);
        __syncthreads();
    }
    pfb5_chunk<R, R3, OS, P, false, FM, true>(p, wg, tid, buf, nullptr, tabpair, halo_row, own_halo);
}

template <int R, int R3, int OS, int P>
void launch5_fmlb_group(const PfbLaunch &shape, const PfbLaunch *d_pls, const GroupMap &gm, hipStream_t s)
{
    constexpr int NB = R * R * R3, F = 16 / R3;
    const size_t lds = (size_t)pfb5_buf(NB, R, F, OS, P) * sizeof(cf);
    if (shape.fm_mode == FM_ONLY) {
        static DynLdsAttr attr;
        attr.ensure(reinterpret_cast<const void *>(pfb5_fmlb_group_kernel<R, R3, OS, P, FM_ONLY>), lds);
        hipLaunchKernelGGL((pfb5_fmlb_group_kernel<R, R3, OS, P, FM_ONLY>), dim3(gm.total_wg), dim3(kThreads5), lds, s, d_pls, gm);
    } else {
        static DynLdsAttr attr;
        attr.ensure(reinterpret_cast<const void *>(pfb5_fmlb_group_kernel<R, R3, OS, P, FM_BOTH>), lds);
        hipLaunchKernelGGL((pfb5_fmlb_group_kernel<R, R3, OS, P, FM_BOTH>), dim3(gm.total_wg), dim3(kThreads5), lds, s, d_pls, gm);
    }
}

// The banks of G front-ends in ONE launch (rcf_group.cpp; see pfb_group_kernel_os in pfb.hip): steady state only
template <int R, int R3, int OS, int P>
__global__ __launch_bounds__(kThreads5, 3) void pfb5_group_kernel(const PfbLaunch *__restrict__ pls, GroupMap gm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    int fe, wg;
    group_resolve(gm, blockIdx.x, fe, wg);
    const PfbLaunch p = pls[fe];
    pfb5_chunk<R, R3, OS, P, false>(p, wg, threadIdx.x, buf);
}

template <int R, int R3, int OS, int P>
void launch5_group(const PfbLaunch *d_pls, const GroupMap &gm, hipStream_t s)
{
    constexpr int NB = R * R * R3, F = 16 / R3;
    const size_t lds = (size_t)pfb5_buf(NB, R, F, OS, P) * sizeof(cf);
    static DynLdsAttr attr;
    attr.ensure(reinterpret_cast<const void *>(pfb5_group_kernel<R, R3, OS, P>), lds);
    hipLaunchKernelGGL((pfb5_group_kernel<R, R3, OS, P>), dim3(gm.total_wg), dim3(kThreads5), lds, s, d_pls, gm);
}

// chunks per workgroup of the fused-discriminator kernel: the span that wastes least -- 1 / (span + 1) of a workgroup's
// arithmetic is the halo chunk, and the last round of workgroups over the chip's slots (256 CUs x 3) should be full
int pfb5_fm_span_for(int n_chunks)
{
    static const int forced = [] { const char *e = getenv("RCF_PFB5_FM_SPAN"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    constexpr int kSlots = 256 * 3;
    int best = 1;
    double best_score = 0.0;
    for (int span = 1; span <= 16; ++span) {
        const int n_wg = (n_chunks + span - 1) / span;
        const int rounds = (n_wg + kSlots - 1) / kSlots;
        const double fill = n_wg >= kSlots ? (double)n_wg / ((double)rounds * kSlots) : 1.0;   // (a launch below one round: latency)
        const double score = fill * span / (span + 1.0) * (n_wg >= kSlots || span == 1 ? 1.0 : (double)n_wg * span / n_chunks);
        if (score > best_score + 1e-9) { best_score = score; best = span; }
    }
    return best;
}

template <int R, int R3, int OS, int P>
void launch5_fm(const PfbLaunch &p0, hipStream_t s)
{
    constexpr int NB = R * R * R3, F = 16 / R3;
    PfbLaunch p = p0;
    const int n_chunks = (p.n_frames + F - 1) / F;
    if (p.fm_span <= 0) p.fm_span = pfb5_fm_span_for(n_chunks);
    const int n_wg = (n_chunks + p.fm_span - 1) / p.fm_span;
    const size_t lds = (size_t)pfb5_buf(NB, R, F, OS, P) * sizeof(cf);
    // zero history also when the HALO chunk before the launch's first frame reaches before the stream's start
    const bool zh = (p.n_lo - F - (int64_t)OS * (P - 1)) * (NB / OS) - (NB - 1) < p.start_sample;
#define RCF_FM_GO(ZH_, FM_)                                                                                       \
    do {                                                                                                           \
        static DynLdsAttr attr;                                                                                    \
        attr.ensure(reinterpret_cast<const void *>(pfb5_fm_kernel<R, R3, OS, P, ZH_, FM_>), lds);                  \
        RCF_PFB_LAUNCH(p, (pfb5_fm_kernel<R, R3, OS, P, ZH_, FM_>), dim3(n_wg), dim3(kThreads5), lds, s, p, n_wg); \
    } while (0)
#define RCF_FMLB_GO(ZH_, FM_)                                                                                         \
    do {                                                                                                             \
        static DynLdsAttr attr;                                                                                      \
        attr.ensure(reinterpret_cast<const void *>(pfb5_fmlb_kernel<R, R3, OS, P, ZH_, FM_>), lds);                  \
        RCF_PFB_LAUNCH(p, (pfb5_fmlb_kernel<R, R3, OS, P, ZH_, FM_>), dim3(n_chunks), dim3(kThreads5), lds, s, p, n_chunks); \
    } while (0)
    if (p.fm_edge) {                                     // look-back form: one chunk per workgroup
        if (p.fm_mode == FM_ONLY) { if (zh) RCF_FMLB_GO(true, FM_ONLY); else RCF_FMLB_GO(false, FM_ONLY); }
        else                      { if (zh) RCF_FMLB_GO(true, FM_BOTH); else RCF_FMLB_GO(false, FM_BOTH); }
        return;
    }
    if (p.fm_mode == FM_ONLY) { if (zh) RCF_FM_GO(true, FM_ONLY); else RCF_FM_GO(false, FM_ONLY); }
    else                      { if (zh) RCF_FM_GO(true, FM_BOTH); else RCF_FM_GO(false, FM_BOTH); }
#undef RCF_FM_GO
#undef RCF_FMLB_GO
}

template <int R, int R3, int OS, int P>
void launch5(const PfbLaunch &p, hipStream_t s)
{
    constexpr int NB = R * R * R3, F = 16 / R3;
    const int n_wg = (p.n_frames + F - 1) / F;
    const size_t lds = (size_t)pfb5_buf(NB, R, F, OS, P) * sizeof(cf);
    static DynLdsAttr attr_f, attr_t;
    attr_f.ensure(reinterpret_cast<const void *>(pfb5_kernel<R, R3, OS, P, false>), lds);
    attr_t.ensure(reinterpret_cast<const void *>(pfb5_kernel<R, R3, OS, P, true>), lds);
    const bool zh = (p.n_lo - (int64_t)OS * (P - 1)) * (NB / OS) - (NB - 1) < p.start_sample;
    if (zh) RCF_PFB_LAUNCH(p, (pfb5_kernel<R, R3, OS, P, true>), dim3(n_wg), dim3(kThreads5), lds, s, p, n_wg);
    else    RCF_PFB_LAUNCH(p, (pfb5_kernel<R, R3, OS, P, false>), dim3(n_wg), dim3(kThreads5), lds, s, p, n_wg);
}

}  // namespace

// shapes: (NB, OS) with taps per branch P <= 2 (OS 2, 4) or <= 4 (OS 1); the reference's own prototype gives
// P = ceil(3.64 D / NB) = 2 at OS = 2 and 1 at OS = 4
bool pfb5_dispatch(const PfbLaunch &p, bool probe, hipStream_t s)
{
    if (p.D <= 0 || p.NB % p.D) return false;
    const int OS = p.NB / p.D;
    const int PR = pfb5_padded_p(p.NB, p.D, p.P);
    if (PR == 0) return false;
#define RCF_PFB5(R_, R3_, OS_, P_)                                  \
    if (p.NB == R_ * R_ * R3_ && OS == OS_ && PR == P_) {            \
        if (!probe) launch5<R_, R3_, OS_, P_>(p, s);                 \
        return true;                                                 \
    }
    if (p.fm_ring) {
        // the discriminator fused in: the shapes the reference's channel rule produces (OS = 2: 12.5 kHz raster, OS = 4: 6.25 kHz)
#define RCF_PFB5F(R_, R3_, OS_, P_)                                 \
        if (p.NB == R_ * R_ * R3_ && OS == OS_ && PR == P_) {        \
            if (!probe) launch5_fm<R_, R3_, OS_, P_>(p, s);          \
            return true;                                             \
        }
        RCF_PFB5F(20, 4, 2, 2) RCF_PFB5F(20, 8, 4, 1) RCF_PFB5F(20, 2, 2, 2) RCF_PFB5F(20, 1, 2, 2)
#undef RCF_PFB5F
        return false;
    }
    RCF_PFB5(20, 4, 2, 2) RCF_PFB5(20, 4, 2, 1) RCF_PFB5(20, 4, 1, 4) RCF_PFB5(20, 4, 4, 1)      // 1600 bins
    RCF_PFB5(20, 8, 4, 1) RCF_PFB5(20, 8, 2, 2) RCF_PFB5(20, 8, 2, 1) RCF_PFB5(20, 8, 1, 4)      // 3200 bins
    RCF_PFB5(20, 2, 2, 2) RCF_PFB5(20, 2, 2, 1) RCF_PFB5(20, 2, 1, 4) RCF_PFB5(20, 2, 4, 1)      // 800 bins
    RCF_PFB5(20, 1, 2, 2) RCF_PFB5(20, 1, 2, 1) RCF_PFB5(20, 1, 1, 4) RCF_PFB5(20, 1, 4, 1)      // 400 bins
#undef RCF_PFB5
    return false;
}

bool pfb5_dispatch_group(const PfbLaunch &p, const PfbLaunch *d_pls, const GroupMap &gm, hipStream_t s)
{
    if (p.D <= 0 || p.NB % p.D) return false;
    const int OS = p.NB / p.D;
    const int PR = pfb5_padded_p(p.NB, p.D, p.P);
    if (PR == 0) return false;
    if (p.fm_ring) {
        if (!p.fm_edge) return false;                    // (the span form has no grouped kernel: one by one)
#define RCF_PFB5FG(R_, R3_, OS_, P_)                                \
        if (p.NB == R_ * R_ * R3_ && OS == OS_ && PR == P_) {        \
            launch5_fmlb_group<R_, R3_, OS_, P_>(p, d_pls, gm, s);   \
            return true;                                             \
        }
        RCF_PFB5FG(20, 4, 2, 2) RCF_PFB5FG(20, 8, 4, 1) RCF_PFB5FG(20, 2, 2, 2) RCF_PFB5FG(20, 1, 2, 2)
#undef RCF_PFB5FG
        return false;
    }
#define RCF_PFB5G(R_, R3_, OS_, P_)                                 \
    if (p.NB == R_ * R_ * R3_ && OS == OS_ && PR == P_) {            \
        launch5_group<R_, R3_, OS_, P_>(d_pls, gm, s);               \
        return true;                                                 \
    }
    // the shapes the reference's channel rule produces (OS = 2: 12.5 kHz raster, OS = 4: 6.25 kHz) -- the others run one by one
    RCF_PFB5G(20, 4, 2, 2) RCF_PFB5G(20, 8, 4, 1) RCF_PFB5G(20, 8, 2, 2) RCF_PFB5G(20, 2, 2, 2) RCF_PFB5G(20, 1, 2, 2)
#undef RCF_PFB5G
    return false;
}

// whether the launch's halo chunks (the chunk before its first frame, recomputed by the workgroups that have no predecessor
// inside the launch) reach before the stream's start: such a launch takes the zero-history kernel, alone
bool pfb5_fm_sees_zero_history(const PfbLaunch &p)
{
    if (p.D <= 0 || p.NB % p.D || p.NB % 400) return true;
    const int R3 = p.NB / 400, F = 16 / R3, OS = p.NB / p.D;
    const int PR = pfb5_padded_p(p.NB, p.D, p.P);
    return (p.n_lo - F - (int64_t)OS * (PR - 1)) * (int64_t)p.D - (p.NB - 1) < p.start_sample;
}

namespace {
__global__ void pfb5_xcc_probe_kernel(int *out)
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
}
}  // namespace

// whether block b of a one-dimensional grid runs on XCD b mod 8 on this device -- what the chunk maps of the filterbank
// kernels assume for LOCALITY, and what the look-back hand-over through one XCD's L2 needs for CORRECTNESS: asked of the
// device itself, once per device and process (a 4096-block probe launch with the banks' LDS footprint; every block
// reports the XCC_ID register).  Anything unexpected (fewer XCDs, another dispatch order, a failed launch): false.
bool pfb5_xcd_map_ok(int device, hipStream_t s)
{
    static std::mutex mu;
    static std::map<int, bool> known;
    std::lock_guard<std::mutex> g(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    bool ok = false;
    constexpr int n = 4096;
    int *d = nullptr;
    std::vector<int> h((size_t)n, -1);
    if (hipMalloc(&d, n * sizeof(int)) == hipSuccess) {
        static DynLdsAttr attr;
        attr.ensure(reinterpret_cast<const void *>(pfb5_xcc_probe_kernel), 53 * 1024);
        hipLaunchKernelGGL(pfb5_xcc_probe_kernel, dim3(n), dim3(kThreads5), 53 * 1024, s, d);
        if (hipMemcpyAsync(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess &&
            hipStreamSynchronize(s) == hipSuccess) {
            ok = true;
            for (int b = 0; b < 8 && ok; ++b)
                for (int c = 0; c < b; ++c)
                    if (h[(size_t)b] == h[(size_t)c]) ok = false;          // eight different XCDs
            for (int b = 0; b < n && ok; ++b)
                if (h[(size_t)b] != h[(size_t)(b % 8)]) ok = false;
        }
        (void)hipFree(d);
    }
    (void)hipGetLastError();
    known[device] = ok;
    return ok;
}

bool pfb5_fm_supported(int NB, int D, int P)
{
    PfbLaunch p{};
    p.NB = NB; p.D = D; p.P = P;
    p.fm_ring = reinterpret_cast<float *>(0x1);          // (never dereferenced: probe)
    return pfb5_dispatch(p, true, nullptr);
}

// samples of input history the halo chunk of a launch's first workgroup reaches back over: F + OS (P - 1) + 1 frames and
// the prototype's span
size_t pfb5_fm_history(int NB, int D, int P)
{
    const int R3 = NB / 400, F = 16 / (R3 > 0 ? R3 : 1), OS = NB / D;
    return (size_t)(F + OS * (pfb5_padded_p(NB, D, P) - 1) + 2) * (size_t)D + (size_t)NB;
}

namespace {
__global__ void pfb5_fm_inc_kernel(const double *__restrict__ dangle, float2 *__restrict__ inc, int NB)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= NB) return;
    float2 v = make_float2(1.f, 0.f);
    if (dangle[k] != 0.0) {                                // (tapfin.hip: TapInfo::inc_r / inc_i, the same expression)
        double sn, cs;
        sincos_fast(dangle[k], sn, cs);
        v = make_float2((float)cs, (float)sn);
    }
    inc[k] = v;
}
}  // namespace

void launch_pfb5_fm_inc(const double *d_dangle, float2 *d_inc, int NB, hipStream_t s)
{
    hipLaunchKernelGGL(pfb5_fm_inc_kernel, dim3((NB + 255) / 256), dim3(256), 0, s, d_dangle, d_inc, NB);
}

// rows of the polyphase table the instantiation for (NB / D, P) reads (zero padded by rcf_pfb_open); 0: no kernel.
// Instantiated: OS = 1 with 4 taps per branch, OS = 2 with 1 or 2, OS = 4 with 1 -- fewer taps run the next larger one.
int pfb5_padded_p(int NB, int D, int P)
{
    if (D <= 0 || NB % D || P < 1) return 0;
    const int OS = NB / D;
    if (OS == 1) return P <= 4 ? 4 : 0;
    if (OS == 2) return P <= 1 ? 1 : (P <= 2 ? 2 : 0);
    if (OS == 4) return P <= 1 ? 1 : 0;
    return 0;
}

}  // namespace rcfx
