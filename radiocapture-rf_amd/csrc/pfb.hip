// pfb.hip -- polyphase filterbank channelizer for gfx950: polyphase FIR in registers fused with an
// in-LDS NB-point inverse-sign FFT and an epilogue that writes 16-frame tiles.
//
// Math: with prototype h (T taps), NB bins, decimation D (OS = NB / D),
//   out_k[n] = e^{-j 2 pi k n D / NB} * sum_{rho<NB} e^{+j 2 pi k rho / NB} * u_rho[n]
//   u_rho[n] = sum_{p<P} h[NB p + rho] * x[n D - rho - NB p]
// which is exactly freq_xlating_fir_filter_ccc(D, h, k fs / NB, fs) -- the block the reference
// instantiates once PER CHANNEL at /root/reference/rc_frontend/channel.py:35 -- for all NB on-grid
// frequencies at once, with mathematically exact phases.
//
// Mapping (workgroup = NB threads, thread rho = branch rho, ONE chunk of F = 16 output frames per workgroup):
//   * x[m D - rho] for consecutive rho is a reversed unit-stride run of the interleaved cf32 stream:
//     every wavefront load is one contiguous 512-byte segment (buffer-descriptor addressing: one VGPR offset,
//     scalar row offsets).  The chunk needs F + OS (P-1) input rows; the OS (P-1) halo rows are shared with the
//     neighbouring chunk, which the block -> chunk map keeps on the same XCD, so they come out of L2.
//   * output-stationary branch FIR: the thread holds the 16 output accumulators and streams the input rows
//     through them in groups of 8 (row j feeds output f with tap (HALO + f - j) / OS; fully unrolled, every
//     index a compile-time constant), P real taps in registers, 2 FMA per tap per output, no LDS traffic.
//     ~95 VGPRs => 4 workgroups per CU (the LDS limit).  This kernel is latency / occupancy bound, not
//     instruction bound: a sliding-window form (window in registers, 160 VGPRs, 3 workgroups per CU) was 8 %
//     slower, a role-split form (FIR waves + FFT waves, double-buffered LDS) and packed-f32 FMAs gained nothing,
//     spilling to reach more workgroups lost 30-55 % (history: git log).
//   * the 16 frames of u are parked in LDS ([frame][branch], rows padded 1-in-16 + 2), transformed by
//     radix-16/8/4/2 Stockham passes (fft_core.hpp) with exact twiddles read from an LDS table, then
//     read back transposed so that each bin's F consecutive outputs leave as one 128-byte line, and the lines of
//     consecutive bins are consecutive in memory: the output ring is TILED, [tile of 16 frames][bin][16], so the
//     whole chunk is one contiguous run of 16 NB samples for this kernel while a bin's 16 frames stay one line for
//     the stage-2 FIR and the egress pump (PfbLaunch in rcf_internal.h; non-temporal stores).  The first layout,
//     one ring per bin, scattered a chunk over NB separate lines: 0.1099 vs 0.1055 ms at 256 bins.
// Bound: HBM.  Algorithmic bytes per input sample = 8 (read) + 8 NB / D (write) = 16 at OS = 1.
#include <cstdlib>

#include "fft_core.hpp"
#include <hip/hip_ext.h>
#ifndef RCF_PFB_LOAD_AUX
#define RCF_PFB_LOAD_AUX 0
#endif
#ifndef RCF_PFB_STORE_AUX
#define RCF_PFB_STORE_AUX 18       // nt | sc1 (bits 1 and 4).  256 / 512 / 1024 bins, two alternating runs each, of the HBM peak: plain 0.658-0.663 / 0.626-0.629 / -, sc0 0.660 / 0.627-0.630 / -, sc1 0.664 / 0.641-0.643 / -, nt 0.669-0.673 / 0.661-0.669 / 0.600, nt sc0 0.670 / 0.660-0.664 / 0.598, sc0 sc1 0.659-0.661 / 0.637 / 0.562, nt sc1 0.675-0.679 / 0.676-0.681 / 0.602-0.607, all three 0.675-0.676 / 0.672-0.679 / 0.606
#endif
#include "rcf_internal.h"
#include "fir_small.hpp"

namespace rcfx {

namespace {

constexpr int F = 16;   // frames per LDS chunk
int env_int(const char *name, int dflt);
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

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
template <int NB, int R, int NS, bool END_WG, int TWS = 1>
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
            const cf w = tw_lds[(k * t) * (NB / (NS * R)) * TWS];   // exact table entry e^{+2 pi i k t / (NS R)}
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
        Pass::template store_t<NS>(buf + frame * RS, j, v[i]);
    }
    if (WAVE_LOCAL && !END_WG) wave_sync(); else __syncthreads();
}

// the FFT passes over the chunk in LDS (after the barrier that follows the branch sums' LDS writes)
// TWS: the table holds e^{+2 pi i n / (TWS NB)} (the two-branch kernel keeps ONE table, of the whole bank's size)
template <int NB, int TWS = 1>
__device__ __forceinline__ void pfb_fft(cf *buf, const cf *tw_lds, int tid)
{
    using PL = Plan<NB>;
    pfb_pass<NB, PL::r[0], 1, (PL::n < 2 || PL::r[1] != PL::r[0]), TWS>(buf, tw_lds, tid);
    if constexpr (PL::n >= 2)
        pfb_pass<NB, PL::r[1], PL::r[0], (PL::n < 3 || PL::r[2] != PL::r[1]), TWS>(buf, tw_lds, tid);
    // a third pass (radix 2 for 512 bins, 4 for 1024) is NOT run over LDS: its butterfly j reads and writes
    // the same R3 positions j + t NB/R3, and those are exactly the bins one epilogue lane handles (bins
    // k0 + i NB/F), so it is done in registers on the way out -- one LDS round trip and two barriers less
}

// epilogue: LDS read transposed (lane = (bin k0, frame f_lane)), the radix-2 / 4 finish of 512 / 1024-bin banks in
// registers, the OS = 2 bin phase factor, and the stores into the tiled ring
template <int NB, int OS>
__device__ __forceinline__ void pfb_epilogue(const cf *buf, const cf *tw_lds, const PfbLaunch &p,
                                             const __amdgpu_buffer_rsrc_t out_rsrc, int64_t n0, int nf, int tid)
{
    using PL = Plan<NB>;
    constexpr int RS = row_stride<NB>();
    const int k0 = tid / F, f_lane = tid % F;
    // tiled ring: (i >> 4) tile_pitch + 16 k + (i & 15); the lane's bins k0 + i NB / F are NB / F lines apart
    constexpr int out_so_step = (NB / F) * F * (int)sizeof(cf);
    const int64_t n = n0 + f_lane;
    const int64_t ridx = (int64_t)((uint64_t)(n - p.n_abs0) & p.ring_mask);
    const int vo = (int)(((ridx >> 4) * p.tile_pitch + k0 * F + (ridx & 15)) * (int64_t)sizeof(cf));
    if (f_lane >= nf) return;
    cf vv[F];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        const int k = k0 + i * (NB / F);
        vv[i] = ((NB / F) % 16 == 0) ? buf[f_lane * RS + lds_pad(k0) + i * ((NB / F) + (NB / F) / 16)]
                                     : buf[f_lane * RS + lds_pad(k)];
    }
    if constexpr (PL::n >= 3) {
        constexpr int R3 = PL::r[2];                 // bins j + t NB/R3 = registers i + t F/R3
        static_assert(PL::r[0] * PL::r[1] * R3 == NB && F % R3 == 0, "final radix must divide the chunk");
#pragma unroll
        for (int i = 0; i < F / R3; ++i) {
            const int j = k0 + i * (NB / F);         // butterfly index, < NB / R3
            cf w[R3];
#pragma unroll
            for (int t = 0; t < R3; ++t) w[t] = vv[i + t * (F / R3)];
#pragma unroll
            for (int t = 1; t < R3; ++t) w[t] = cmul(w[t], tw_lds[(j * t) & (NB - 1)]);   // W_NB^{j t}, exact entry
            Dft<R3, +1>::run(w);
#pragma unroll
            for (int f = 0; f < R3; ++f) vv[i + f * (F / R3)] = w[Dft<R3, +1>::reg_of(f)];
        }
    }
#pragma unroll
    for (int i = 0; i < F; ++i) {
        const int k = k0 + i * (NB / F);
        cf v = vv[i];
        if (OS == 2) {
            if ((k & 1) && (n & 1)) v = make_float2(-v.x, -v.y);
        }
        u32x2 o;
        o.x = __float_as_uint(v.x);
        o.y = __float_as_uint(v.y);
        __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, i * out_so_step, RCF_PFB_STORE_AUX);   // nt: +4 %
    }
}

// one chunk (16 frames) of one front-end's bank: everything the workgroup does once it knows WHICH chunk is its own.
// Shared by the single-front-end kernel (launch record in the kernel arguments) and the grouped one (records of G
// front-ends in the arena): the same instructions in the same order, so the bins are the same bits either way.
template <int NB, int OS, int P, bool ZH>
__device__ __forceinline__ void pfb_os_chunk(const PfbLaunch &p, const int wg, const int tid, cf *buf, cf *tw_lds)
{
    constexpr int D = NB / OS;
    constexpr int HALO = OS * (P - 1);
    constexpr int W = F + HALO;
    constexpr int RS = row_stride<NB>();
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
        p.bins_ring, 0, (int)((int64_t)((p.ring_mask + 1) >> kPfbTileLog2) * p.tile_pitch * (int64_t)sizeof(cf)), 0x00020000);
    const int64_t m_min = (p.start_sample + tid + D - 1) / D;
    const int64_t m0 = n0 - HALO;
    const int vo_in = (int)((m0 * D - tid - p.src.origin) * (int64_t)sizeof(cf));

    float ur[F], ui[F];
#pragma unroll
    for (int f = 0; f < F; ++f) ur[f] = ui[f] = 0.f;
    {
        // Rows in two pinned groups of <= 16, the second requested before the first is consumed (sched_barrier keeps
        // the scheduler from re-interleaving the fully unrolled schedule).  Left to itself the compiler kept ~5 loads in
        // flight at the tail of the row loop; measured at 256 bins, block 2^25 (same session): compiler's schedule
        // 0.1054-0.1061 ms, 8 rows single-buffered 0.1054, 4 + 4 double-buffered 0.1054, 8 + 8 0.1024, 16 single 0.1014-
        // 0.1029, 15 + 14 0.1019-0.1020, all 29 up front 0.1019-0.1023: 0.635 -> 0.66 of the HBM peak (73-78 VGPRs,
        // still four workgroups per CU: the LDS limit).  The zero-history instantiation keeps groups of 8.
        constexpr int GX = ZH ? 8 : ((W + 1) / 2 < 16 ? (W + 1) / 2 : 16);
        constexpr bool DBX = !ZH;
        constexpr int NG = (W + GX - 1) / GX;
        v2f xb[DBX ? 2 : 1][GX];
        auto load_group = [&](v2f (&x)[GX], int j0) {
#pragma unroll
            for (int g = 0; g < GX; ++g) {
                x[g] = (v2f)(0.f);
                if (j0 + g < W) {
                    const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo_in, (j0 + g) * D * (int)sizeof(cf), RCF_PFB_LOAD_AUX);
                    x[g].x = __uint_as_float(r.x);
                    x[g].y = __uint_as_float(r.y);
                    if (ZH && m0 + j0 + g < m_min) x[g] = (v2f)(0.f);
                }
            }
        };
        auto use_group = [&](const v2f (&x)[GX], int j0) {
#pragma unroll
            for (int g = 0; g < GX; ++g) {
                const int j = j0 + g;
                if (j < W) {
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        const int t = HALO + f - j;
                        if (t >= 0 && t % OS == 0 && t / OS < P) {
                            ur[f] = fmaf(h[t / OS], x[g].x, ur[f]);
                            ui[f] = fmaf(h[t / OS], x[g].y, ui[f]);
                        }
                    }
                }
            }
        };
        if constexpr (DBX) {
            load_group(xb[0], 0);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (gi + 1 < NG) load_group(xb[(gi + 1) & 1], (gi + 1) * GX);
                __builtin_amdgcn_sched_barrier(0);
                use_group(xb[gi & 1], gi * GX);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                load_group(xb[0], gi * GX);
                __builtin_amdgcn_sched_barrier(0);
                use_group(xb[0], gi * GX);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) buf[f * RS + lds_pad(tid)] = make_float2(ur[f], ui[f]);
    __syncthreads();
    pfb_fft<NB>(buf, tw_lds, tid);
    pfb_epilogue<NB, OS>(buf, tw_lds, p, out_rsrc, n0, nf, tid);
}

// n_wg < 0: probe without the XCD-aware block -> chunk map (RCF_PFB_NOREMAP=1)
// sr: the stage-2 rider.  The stage-2 launch of the PREVIOUS block (fir_small_tile: short xlating FIR + discriminator per
// active bin) is a latency-bound tail of ~19 us behind a bandwidth-bound 100 us kernel -- 14 % of the step.  Its work
// items are independent of this block's chunks (the frames they read were finished by the previous launch), so they are
// the FIRST sr.n_wgs workgroups of this launch and run beside the first rounds of filterbank chunks: the tail costs its
// 50 MB of traffic and nothing else.
template <int NB, int OS, int P, int MINW, bool ZH>
__global__ __launch_bounds__(NB, MINW) void pfb_kernel_os(PfbLaunch p, int n_wg, S2Rider sr)
{
    constexpr int RS = row_stride<NB>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int bid = blockIdx.x;
    if constexpr (NB == kSmallThreads && !ZH) {
        if (sr.n_wgs > 0) {
            // the riders come in sr.n_batches batches of sr.batch_wgs workgroups, one batch at the head of every sr.period
            // blocks of the grid (all multiples of 8: the chunks keep their XCDs): spread over the launch they run beside
            // the bandwidth-bound chunks instead of ahead of or behind them
            const int q = bid / sr.period, r = bid - q * sr.period;
            if (q < sr.n_batches && r < sr.batch_wgs) {
                const int rid = q * sr.batch_wgs + r;
                if (rid < sr.n_chans * sr.n_tiles)
                    fir_small_tile<kSmallRiderPerThread>(sr.chans, rid % sr.n_chans, rid / sr.n_chans, sr.D, sr.T, sr.KB, sr.ring_mask, sr.atan_tab, smem_raw);
                return;
            }
            bid -= (q < sr.n_batches ? q + 1 : sr.n_batches) * sr.batch_wgs;
        }
    }
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + F * RS;

    const int tid = threadIdx.x;
    if (p.rider_n8[0] + p.rider_n8[1] && bid < kPfbRiderWgs)
        pfb_copy_rider(p, bid, min(kPfbRiderWgs, (int)gridDim.x - (int)sr.n_wgs), tid, NB);
    int wg;
    if (n_wg < 0) {
        wg = bid;                                          // probe: no XCD remap
    } else {
        const int b = bid, q = n_wg / 8, r = n_wg % 8, xcd = b % 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    pfb_os_chunk<NB, OS, P, ZH>(p, wg, tid, buf, tw_lds);
}

// The banks of G front-ends in ONE launch (rcf_group.cpp): grid = the chunks of all of them, the XCD-aware map runs over
// the whole grid (neighbouring chunks of one front-end stay on one XCD), a prefix table says whose chunk a workgroup
// has.  The launch record comes out of the arena by scalar loads -- the address is uniform -- instead of the kernel
// arguments; steady state only (a front-end that still sees zero history is launched on its own).
template <int NB, int OS, int P, int MINW>
__global__ __launch_bounds__(NB, MINW) void pfb_group_kernel_os(const PfbLaunch *__restrict__ pls, GroupMap gm)
{
    constexpr int RS = row_stride<NB>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + F * RS;
    int fe, wg;
    group_resolve(gm, blockIdx.x, fe, wg);
    const PfbLaunch p = pls[fe];
    pfb_os_chunk<NB, OS, P, false>(p, wg, threadIdx.x, buf, tw_lds);
}

// Persistent form of the kernel above: the grid is one resident round of workgroups (8 XCDs x wg_per_xcd), each
// walks the chunks  c = first(xcd) + li, + wg_per_xcd, ...  of its XCD's contiguous chunk range, and the first PF
// input rows of the NEXT chunk are requested right after the branch sums went to LDS, so that they travel while the
// FFT passes and the epilogue of the current chunk run (registers: the 2 F accumulators are dead by then).
template <int NB, int OS, int P, int MINW, bool ZH, int PF>
__global__ __launch_bounds__(NB, MINW) void pfb_kernel_pp(PfbLaunch p, int n_chunks)
{
    constexpr int D = NB / OS;
    constexpr int HALO = OS * (P - 1);
    constexpr int W = F + HALO;
    constexpr int G = 8;
    constexpr int RS = row_stride<NB>();
    static_assert(PF <= W && PF % G == 0, "prefetch depth");
    static_assert(F == (1 << kPfbTileLog2), "ring tiles are one chunk long");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + F * RS;

    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % 8, li = blockIdx.x / 8, step = gridDim.x / 8;
    const int q = n_chunks / 8, r = n_chunks % 8;
    const int c_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int c_hi = c_lo + (xcd < r ? q + 1 : q);
    int c = c_lo + li;
    if (c >= c_hi) return;

    tw_lds[tid] = p.tw[tid];
    float h[P];
#pragma unroll
    for (int qq = 0; qq < P; ++qq) h[qq] = p.ptaps[qq * NB + tid];

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cf *>(p.src.base), 0, (int)(p.src_len * (int64_t)sizeof(cf)), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.bins_ring, 0, (int)((int64_t)((p.ring_mask + 1) >> kPfbTileLog2) * p.tile_pitch * (int64_t)sizeof(cf)), 0x00020000);
    const int64_t m_min = (p.start_sample + tid + D - 1) / D;
    // byte offset of row 0 of chunk 0 for this lane; a chunk advances it by F * D samples
    const int vo_in0 = (int)(((p.n_lo - HALO) * D - tid - p.src.origin) * (int64_t)sizeof(cf));
    constexpr int CHUNK_BYTES = F * D * (int)sizeof(cf);

    v2f xp[PF > 0 ? PF : 1];
    {
        const int vo = vo_in0 + c * CHUNK_BYTES;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const u32x2 rr = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo, j * D * (int)sizeof(cf), 0);
            xp[j].x = __uint_as_float(rr.x);
            xp[j].y = __uint_as_float(rr.y);
        }
    }
    for (;;) {
        const int fb0 = c * F;
        const int nf = min(F, p.n_frames - fb0);
        const int64_t n0 = p.n_lo + fb0;
        const int64_t m0 = n0 - HALO;
        const int vo_in = vo_in0 + c * CHUNK_BYTES;

        float ur[F], ui[F];
#pragma unroll
        for (int f = 0; f < F; ++f) ur[f] = ui[f] = 0.f;
#pragma unroll
        for (int j0 = 0; j0 < W; j0 += G) {
            v2f x[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                x[g] = (v2f)(0.f);
                if (j0 + g < W) {
                    if (j0 + g < PF) {
                        x[g] = xp[j0 + g];
                    } else {
                        const u32x2 rr = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo_in, (j0 + g) * D * (int)sizeof(cf), 0);
                        x[g].x = __uint_as_float(rr.x);
                        x[g].y = __uint_as_float(rr.y);
                    }
                    if (ZH && m0 + j0 + g < m_min) x[g] = (v2f)(0.f);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = j0 + g;
                if (j < W) {
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        const int t = HALO + f - j;          // = OS * q
                        if (t >= 0 && t % OS == 0 && t / OS < P) {
                            ur[f] = fmaf(h[t / OS], x[g].x, ur[f]);
                            ui[f] = fmaf(h[t / OS], x[g].y, ui[f]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int f = 0; f < F; ++f) buf[f * RS + lds_pad(tid)] = make_float2(ur[f], ui[f]);
        const int c_next = c + step;
        const bool more = c_next < c_hi;
        if (PF > 0) {
            // next chunk's first rows; past the end the offset stays inside the descriptor or reads as zero
            const int vo = vo_in0 + (more ? c_next : c) * CHUNK_BYTES;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const u32x2 rr = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, vo, j * D * (int)sizeof(cf), 0);
                xp[j].x = __uint_as_float(rr.x);
                xp[j].y = __uint_as_float(rr.y);
            }
        }
        __syncthreads();
        pfb_fft<NB>(buf, tw_lds, tid);
        pfb_epilogue<NB, OS>(buf, tw_lds, p, out_rsrc, n0, nf, tid);
        if (!more) break;
        c = c_next;
        __syncthreads();                       // the epilogue's LDS reads before the next chunk's branch sums
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Two-branch form for critically sampled banks of NB = 2 NH >= 512 bins: the workgroup is NH threads -- the size, the LDS
// footprint and therefore the residency (four independent workgroups per CU at 512 bins, two at 1024) of the kernel for
// HALF as many bins -- and thread r owns the two ADJACENT branches 2r and 2r + 1:
//   * x[mD - 2r - 1], x[mD - 2r] are one 16-byte load: half the load instructions, a wavefront reads 1 KB contiguous;
//   * decimation in time over the branch index: out[k'] = E[k'] + W_NB^{k'} O[k'], out[k' + NH] = E[k'] - W_NB^{k'} O[k']
//     with E / O the NH-point transforms of the even / odd branch sums.  The two transforms go through the SAME
//     16-frame LDS buffer one after the other (the odd sums wait in their 32 registers while E is transformed, E's
//     transposed read-back waits in 32 while O is), and the combine is done in registers on the way out -- the
//     radix-2 pass the persistent form ran as its epilogue, now between two half-size transforms;
//   * the ring's tile is unchanged: bins k' and k' + NH of the lane's 16 frames leave as whole 128-byte lines, four
//     consecutive bins of a wavefront store = 512 contiguous bytes, in both halves of the tile.
// What the 512-thread persistent form could not get: its two barrier-coupled workgroups per CU leave the memory pipe
// idle whenever both are inside their transforms; four independent half-size workgroups overlap their phases the way
// the 256-bin kernel's do.
template <int NH>
__device__ __forceinline__ void pfb2_read_bins(const cf *buf, const cf *tw_lds, int k0, int f_lane, cf (&vv)[F])
{
    using PL = Plan<NH>;
    constexpr int RS = row_stride<NH>();
#pragma unroll
    for (int i = 0; i < F; ++i) {
        const int k = k0 + i * (NH / F);
        vv[i] = ((NH / F) % 16 == 0) ? buf[f_lane * RS + lds_pad(k0) + i * ((NH / F) + (NH / F) / 16)]
                                     : buf[f_lane * RS + lds_pad(k)];
    }
    if constexpr (PL::n >= 3) {                      // the NH-point transform's own last pass, in registers (as pfb_epilogue)
        constexpr int R3 = PL::r[2];
#pragma unroll
        for (int i = 0; i < F / R3; ++i) {
            const int j = k0 + i * (NH / F);
            cf w[R3];
#pragma unroll
            for (int t = 0; t < R3; ++t) w[t] = vv[i + t * (F / R3)];
#pragma unroll
            for (int t = 1; t < R3; ++t) w[t] = cmul(w[t], tw_lds[2 * ((j * t) & (NH - 1))]);
            Dft<R3, +1>::run(w);
#pragma unroll
            for (int f = 0; f < R3; ++f) vv[i + f * (F / R3)] = w[Dft<R3, +1>::reg_of(f)];
        }
    }
}

template <int NH, int P, bool ZH>
__device__ __forceinline__ void pfb_2b_chunk(const PfbLaunch &p, const int wg, const int tid, cf *buf, cf *tw_lds)
{
    constexpr int NB = 2 * NH, D = NB;
    constexpr int HALO = P - 1;
    constexpr int W = F + HALO;
    // rows in flight per thread: measured on the chip (block 2^25, P = 14), 512 bins: G = 4 single-buffered 0.575 of the
    // HBM peak (= the persistent form it replaces), 6: 0.61, 8: 0.635, 4 + 4 double-buffered 0.625, and at THREE
    // workgroups per CU (168 VGPRs) 8 + 8 double-buffered 0.65 (6 + 6, 10 + 10, 16 single: the same within 1 %);
    // 1024 bins (512-thread workgroups, 128 VGPRs): 4 + 4 double-buffered 0.59, 8 single 0.585, 4 single 0.525
    constexpr int G = ZH ? 4 : (NH == 256 ? 8 : 4);
    constexpr bool DB = !ZH;
    constexpr int RS = row_stride<NH>();
    const int fb0 = wg * F;
    if (fb0 >= p.n_frames) return;
    const int nf = min(F, p.n_frames - fb0);
    const int64_t n0 = p.n_lo + fb0;

    tw_lds[tid] = p.tw[tid];
    tw_lds[tid + NH] = p.tw[tid + NH];
    float he[P], ho[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const float2 hh = *reinterpret_cast<const float2 *>(p.ptaps + q * NB + 2 * tid);
        he[q] = hh.x;
        ho[q] = hh.y;
    }

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cf *>(p.src.base), 0, (int)(p.src_len * (int64_t)sizeof(cf)), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.bins_ring, 0, (int)((int64_t)((p.ring_mask + 1) >> kPfbTileLog2) * p.tile_pitch * (int64_t)sizeof(cf)), 0x00020000);
    const int64_t m_min_e = (p.start_sample + 2 * tid + D - 1) / D;
    const int64_t m_min_o = (p.start_sample + 2 * tid + 1 + D - 1) / D;
    const int64_t m0 = n0 - HALO;
    // rows of this chunk's window that still lie before the stream's start, per branch (zero-history launches only)
    const int jz_e = (int)max((int64_t)0, min((int64_t)W, m_min_e - m0));
    const int jz_o = (int)max((int64_t)0, min((int64_t)W, m_min_o - m0));
    // the 16 bytes at sample m D - 2 tid - 1: .xy = the odd branch's sample, .zw = the even branch's
    const int vo_in = (int)((m0 * D - 2 * tid - 1 - p.src.origin) * (int64_t)sizeof(cf));

    float er[F], ei[F], orr[F], oi[F];
#pragma unroll
    for (int f = 0; f < F; ++f) er[f] = ei[f] = orr[f] = oi[f] = 0.f;
    // Rows in groups of G; with DB the NEXT group's loads are issued before the current group's FMAs (two groups of
    // registers), without it a group is loaded, waited for and consumed.  The schedule is fully unrolled, and left
    // alone the scheduler hoists seventeen 16-byte loads above the first FMA (68 VGPRs of rows) and spills the
    // accumulators: sched_barrier pins the group structure.
    constexpr int NG = (W + G - 1) / G;
    v4f xb[DB ? 2 : 1][G];
    auto load_group = [&](v4f (&x)[G], int j0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            x[g] = (v4f)(0.f);
            if (j0 + g < W) {
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, vo_in, (j0 + g) * D * (int)sizeof(cf), RCF_PFB_LOAD_AUX);
                x[g].x = __uint_as_float(r.x);
                x[g].y = __uint_as_float(r.y);
                x[g].z = __uint_as_float(r.z);
                x[g].w = __uint_as_float(r.w);
                if (ZH && j0 + g < jz_o) x[g].x = x[g].y = 0.f;
                if (ZH && j0 + g < jz_e) x[g].z = x[g].w = 0.f;
            }
        }
    };
    auto use_group = [&](const v4f (&x)[G], int j0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int j = j0 + g;
            if (j < W) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const int t = HALO + f - j;
                    if (t >= 0 && t < P) {
                        er[f] = fmaf(he[t], x[g].z, er[f]);
                        ei[f] = fmaf(he[t], x[g].w, ei[f]);
                        orr[f] = fmaf(ho[t], x[g].x, orr[f]);
                        oi[f] = fmaf(ho[t], x[g].y, oi[f]);
                    }
                }
            }
        }
    };
    if constexpr (DB) {
        load_group(xb[0], 0);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + 1 < NG) load_group(xb[(gi + 1) & 1], (gi + 1) * G);
            __builtin_amdgcn_sched_barrier(0);
            use_group(xb[gi & 1], gi * G);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            load_group(xb[0], gi * G);
            __builtin_amdgcn_sched_barrier(0);
            use_group(xb[0], gi * G);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int k0 = tid / F, f_lane = tid % F;
    cf vE[F], vO[F];
#pragma unroll
    for (int f = 0; f < F; ++f) buf[f * RS + lds_pad(tid)] = make_float2(er[f], ei[f]);
    __syncthreads();
    pfb_fft<NH, 2>(buf, tw_lds, tid);
    pfb2_read_bins<NH>(buf, tw_lds, k0, f_lane, vE);
    __syncthreads();                               // every lane has read E before the odd sums overwrite the buffer
#pragma unroll
    for (int f = 0; f < F; ++f) buf[f * RS + lds_pad(tid)] = make_float2(orr[f], oi[f]);
    __syncthreads();
    pfb_fft<NH, 2>(buf, tw_lds, tid);
    pfb2_read_bins<NH>(buf, tw_lds, k0, f_lane, vO);

    constexpr int out_so_step = (NH / F) * F * (int)sizeof(cf);
    constexpr int out_so_half = NH * F * (int)sizeof(cf);
    const int64_t n = n0 + f_lane;
    const int64_t ridx = (int64_t)((uint64_t)(n - p.n_abs0) & p.ring_mask);
    const int vo = (int)(((ridx >> 4) * p.tile_pitch + k0 * F + (ridx & 15)) * (int64_t)sizeof(cf));
    if (f_lane >= nf) return;
#pragma unroll
    for (int i = 0; i < F; ++i) {
        const int k = k0 + i * (NH / F);
        const cf t = cmul(vO[i], tw_lds[k]);
        const cf lo = cadd(vE[i], t), hi = csub(vE[i], t);
        u32x2 o;
        o.x = __float_as_uint(lo.x);
        o.y = __float_as_uint(lo.y);
        __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, i * out_so_step, RCF_PFB_STORE_AUX);
        o.x = __float_as_uint(hi.x);
        o.y = __float_as_uint(hi.y);
        __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, out_so_half + i * out_so_step, RCF_PFB_STORE_AUX);
    }
}

template <int NH, int P, int MINW, bool ZH>
__global__ __launch_bounds__(NH, MINW) void pfb_kernel_2b(PfbLaunch p, int n_wg)
{
    constexpr int RS = row_stride<NH>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + F * RS;                       // e^{+2 pi i n / NB}, n < NB

    const int tid = threadIdx.x;
    if (p.rider_n8[0] + p.rider_n8[1] && (int)blockIdx.x < kPfbRiderWgs)
        pfb_copy_rider(p, blockIdx.x, min(kPfbRiderWgs, (int)gridDim.x), tid, NH);
    int wg;
    if (n_wg < 0) {
        wg = blockIdx.x;
    } else {
        const int b = blockIdx.x, q = n_wg / 8, r = n_wg % 8, xcd = b % 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    pfb_2b_chunk<NH, P, ZH>(p, wg, tid, buf, tw_lds);
}

// grouped form (see pfb_group_kernel_os)
template <int NH, int P, int MINW>
__global__ __launch_bounds__(NH, MINW) void pfb_group_kernel_2b(const PfbLaunch *__restrict__ pls, GroupMap gm)
{
    constexpr int RS = row_stride<NH>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + F * RS;
    int fe, wg;
    group_resolve(gm, blockIdx.x, fe, wg);
    const PfbLaunch p = pls[fe];
    pfb_2b_chunk<NH, P, false>(p, wg, threadIdx.x, buf, tw_lds);
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

bool pfb_persistent(int NB)
{
    static const int pp_env = env_int("RCF_PFB_PP", -1);
    return pp_env < 0 ? NB >= 512 : pp_env != 0;
}

// critically sampled banks of >= 512 bins run the two-branch form (RCF_PFB_2B=0: the persistent form instead)
bool pfb_two_branch(int NB, int OS)
{
    // (256 bins as 128-thread two-branch workgroups, eight per CU: 0.56-0.60 against the plain kernel's 0.63 -- not kept)
    static const int env = env_int("RCF_PFB_2B", 1);
    return env != 0 && OS == 1 && NB >= 512;
}

template <int NB, int OS, int P, int MINW>
void launch_os(const PfbLaunch &p, hipStream_t s, const S2Rider *sr_in)
{
    const int n_wg = (p.n_frames + F - 1) / F;
    S2Rider sr{};
    if (sr_in) {
        sr = *sr_in;
        // batches (RCF_S2_RIDER_BATCHES; 1 = all riders first): batch size and period are multiples of 8.  Fused launch on one box, filterbank alone 101.3 us: 1 batch 114.8, 4: 114.4, 16: 114.3, 64: 112.7 us (all riders LAST: the same as 64) -- the rider costs its ~50 MB of traffic wherever it sits, the trailing launch cost 18 us
        static const int nb_env = env_int("RCF_S2_RIDER_BATCHES", 64);
        const int work = sr.n_chans * sr.n_tiles;
        int nbat = std::max(1, std::min(nb_env, (work + 7) / 8));
        sr.batch_wgs = ((work + nbat - 1) / nbat + 7) & ~7;
        sr.n_batches = (work + sr.batch_wgs - 1) / sr.batch_wgs;
        sr.n_wgs = sr.n_batches * sr.batch_wgs;
        const int total = n_wg + sr.n_wgs;
        sr.period = std::max(sr.batch_wgs + 8, (total / sr.n_batches) & ~7);
        if ((long long)sr.period * (sr.n_batches - 1) + sr.batch_wgs > total) {     // (a tiny launch: everything first)
            sr.n_batches = 1; sr.batch_wgs = (work + 7) & ~7; sr.n_wgs = sr.batch_wgs; sr.period = n_wg + sr.n_wgs + 8;
        }
    }
    static const int no_remap = env_int("RCF_PFB_NOREMAP", 0);
    const int arg = no_remap ? -n_wg : n_wg;
    const size_t lds = ((size_t)F * row_stride<NB>() + NB) * sizeof(cf);
    const bool zh = (p.n_lo - (int64_t)OS * (P - 1)) * (NB / OS) - (NB - 1) < p.start_sample;
    if constexpr (OS == 1 && NB >= 512) {
        if (pfb_two_branch(NB, OS)) {
            constexpr int NH = NB / 2;
            // waves per SIMD the register budget is for: 3 / 2 workgroups per CU.  The zero-history instantiation (the
            // first launch after rcf_pfb_open only) masks rows and gets 256 VGPRs instead of spilling
            constexpr int MW2 = NH == 256 ? 3 : 4;
            const size_t lds2 = ((size_t)F * row_stride<NH>() + NB) * sizeof(cf);
            static DynLdsAttr attr_zh, attr;
            if (zh) {
                attr_zh.ensure((const void *)pfb_kernel_2b<NH, P, 2, true>, lds2);
                RCF_PFB_LAUNCH(p, (pfb_kernel_2b<NH, P, 2, true>), dim3(n_wg), dim3(NH), lds2, s, p, arg);
            } else {
                attr.ensure((const void *)pfb_kernel_2b<NH, P, MW2, false>, lds2);
                RCF_PFB_LAUNCH(p, (pfb_kernel_2b<NH, P, MW2, false>), dim3(n_wg), dim3(NH), lds2, s, p, arg);
            }
            return;
        }
    }
    // 512 / 1024 bins run the persistent form (2 / 1 workgroups per CU: +2 % / +10 %); at 256 bins and below four
    // independent workgroups per CU already overlap their phases and the persistent form's extra barrier per chunk
    // costs 8 % (measured, block 2^25).  RCF_PFB_PP=0 / 1 forces it off / on.
    if (pfb_persistent(NB) && !zh) {
        constexpr int PF = NB >= 1024 ? 8 : 16;
        const int wg_per_cu = NB <= 256 ? 4 : (NB == 512 ? 2 : 1);
        static const int cus = [] {
            int d = 0, n = 256;
            (void)hipGetDevice(&d);
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d);
            return n > 8 ? n : 256;
        }();
        int grid = 8 * (cus / 8) * wg_per_cu;                 // one resident round, the same count on every XCD
        if (grid > ((n_wg + 7) / 8) * 8) grid = ((n_wg + 7) / 8) * 8;
        RCF_PFB_LAUNCH(p, (pfb_kernel_pp<NB, OS, P, MINW, false, PF>), dim3(grid), dim3(NB), lds, s, p, n_wg);
        return;
    }
    if (zh) { S2Rider none{}; RCF_PFB_LAUNCH(p, (pfb_kernel_os<NB, OS, P, MINW, true>), dim3(n_wg), dim3(NB), lds, s, p, arg, none); }
    else {
        size_t lds_s2 = 0;
        if (sr.n_wgs > 0) lds_s2 = ((size_t)sr.KB * sr.D + 2 * sr.T + sr.KB + 1) * sizeof(float2) + 264 * sizeof(float);
        RCF_PFB_LAUNCH(p, (pfb_kernel_os<NB, OS, P, MINW, false>), dim3(n_wg + sr.n_wgs), dim3(NB), std::max(lds, lds_s2), s, p, arg, sr);
    }
}

// grouped launch of one shape; false: this shape has no grouped form (the oversampled persistent kernels), the caller
// launches its members one by one
template <int NB, int OS, int P, int MINW>
bool launch_os_group(const PfbLaunch *d_pls, const GroupMap &gm, hipStream_t s)
{
    if constexpr (OS == 1 && NB >= 512) {
        if (!pfb_two_branch(NB, OS)) return false;
        constexpr int NH = NB / 2;
        constexpr int MW2 = NH == 256 ? 3 : 4;
        const size_t lds2 = ((size_t)F * row_stride<NH>() + NB) * sizeof(cf);
        static DynLdsAttr attr;
        attr.ensure((const void *)pfb_group_kernel_2b<NH, P, MW2>, lds2);
        hipLaunchKernelGGL((pfb_group_kernel_2b<NH, P, MW2>), dim3(gm.total_wg), dim3(NH), lds2, s, d_pls, gm);
        return true;
    } else {
        if (pfb_persistent(NB)) return false;
        const size_t lds = ((size_t)F * row_stride<NB>() + NB) * sizeof(cf);
        hipLaunchKernelGGL((pfb_group_kernel_os<NB, OS, P, MINW>), dim3(gm.total_wg), dim3(NB), lds, s, d_pls, gm);
        return true;
    }
}

// taps per branch the kernels are instantiated for.  14 is what the reference's own low_pass_2 rule with a
// Blackman-Harris window gives a critically sampled bank of ANY size (transition 0.2 bin, 60 dB -> 13.6 taps
// per branch), so that case gets its exact row count instead of 16.
int round_p(int P, int OS)
{
    if (P <= 4) return 4;
    if (OS == 1 && P > 8 && P <= 14) return 14;
    if (P <= 16) return 16;
    return 0;
}

// d_pls != nullptr: the grouped launch of this shape (p is the members' common shape; gm the chunk map)
template <int NB>
bool dispatch_nb(const PfbLaunch &p, int OS, int P, bool probe, hipStream_t s, const PfbLaunch *d_pls = nullptr,
                 const GroupMap *gm = nullptr, const S2Rider *sr = nullptr)
{
    const int PR = round_p(P, OS);
    if (PR == 0 || (OS != 1 && OS != 2)) return false;
    if (probe) return true;
    // waves per SIMD the register allocator must allow: 4 workgroups per CU is the LDS limit
    constexpr int MW = NB >= 1024 ? 4 : (NB >= 512 ? 4 : 4 * NB / 256 > 0 ? (4 * NB / 256 > 8 ? 8 : (4 * NB / 256 < 1 ? 1 : 4 * NB / 256)) : 1);
    if (d_pls) {
        if (OS == 1) {
            if (PR == 4) return launch_os_group<NB, 1, 4, MW>(d_pls, *gm, s);
            if (PR == 14) return launch_os_group<NB, 1, 14, MW>(d_pls, *gm, s);
            return launch_os_group<NB, 1, 16, MW>(d_pls, *gm, s);
        }
        if (PR == 4) return launch_os_group<NB, 2, 4, MW>(d_pls, *gm, s);
        return launch_os_group<NB, 2, 16, MW>(d_pls, *gm, s);
    }
    if (OS == 1) {
        if (PR == 4) launch_os<NB, 1, 4, MW>(p, s, sr);
        else if (PR == 14) launch_os<NB, 1, 14, MW>(p, s, sr);
        else launch_os<NB, 1, 16, MW>(p, s, sr);
    }
    else         { if (PR == 4) launch_os<NB, 2, 4, MW>(p, s, sr); else launch_os<NB, 2, 16, MW>(p, s, sr); }
    return true;
}

bool dispatch(const PfbLaunch &p, bool probe, hipStream_t s, const PfbLaunch *d_pls = nullptr, const GroupMap *gm = nullptr,
              const S2Rider *sr = nullptr)
{
    if (p.D <= 0 || p.NB % p.D) return false;
    const int OS = p.NB / p.D;
    switch (p.NB) {
        case 64:   return dispatch_nb<64>(p, OS, p.P, probe, s, d_pls, gm, sr);
        case 128:  return dispatch_nb<128>(p, OS, p.P, probe, s, d_pls, gm, sr);
        case 256:  return dispatch_nb<256>(p, OS, p.P, probe, s, d_pls, gm, sr);
        case 512:  return dispatch_nb<512>(p, OS, p.P, probe, s, d_pls, gm, sr);
        case 1024: return dispatch_nb<1024>(p, OS, p.P, probe, s, d_pls, gm, sr);
        default:   return d_pls ? pfb5_dispatch_group(p, d_pls, *gm, s) : pfb5_dispatch(p, probe, s);      // 400 / 800 / 1600 / 3200 bins (pfb5.hip)
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
    if (NB % 25 == 0) return pfb5_padded_p(NB, D, P);
    return round_p(P, D > 0 ? NB / D : 1);
}

bool pfb_takes_rider(const PfbLaunch &p)
{
    if (pfb_frame_major(p.NB)) return false;            // pfb5_kernel: measured, +5.7 us on the 1600-bin launch for 4.7 saved
    // (a launch that still sees zero history runs the plain form whatever the bin count: being wrong about that one
    // launch costs a late start, nothing else)
    return pfb_two_branch(p.NB, p.D > 0 ? p.NB / p.D : 1) || !pfb_persistent(p.NB);
}

void launch_pfb(const PfbLaunch &p, hipStream_t s, const S2Rider *sr)
{
    if (p.n_frames <= 0) return;
    dispatch(p, false, s, nullptr, nullptr, sr);
}

// whether THIS launch runs the kernel that can carry a stage-2 rider: the 256-bin steady-state kernel (its workgroups have
// the small-T tile's 256 threads)
bool pfb_can_carry_s2(const PfbLaunch &p)
{
    if (p.NB != kSmallThreads || pfb_frame_major(p.NB) || p.D <= 0 || p.n_frames <= 0) return false;
    const int OS = p.NB / p.D;
    if ((OS != 1 && OS != 2) || round_p(p.P, OS) == 0 || pfb_persistent(p.NB) || pfb_two_branch(p.NB, OS)) return false;
    return !pfb_sees_zero_history(p);
}

// whether this launch still reaches samples before the bank's start (it then runs the masking instantiation, alone)
bool pfb_sees_zero_history(const PfbLaunch &p)
{
    const int OS = p.D > 0 ? p.NB / p.D : 1;
    const int P = pfb_padded_p(p.NB, p.D, p.P);
    return (p.n_lo - (int64_t)OS * (P - 1)) * (int64_t)p.D - (p.NB - 1) < p.start_sample;
}

// frames per chunk (= per workgroup) of this shape's kernel
int pfb_chunk_frames(int NB) { return pfb_frame_major(NB) ? 16 / (NB / 400) : F; }

bool launch_pfb_group(const PfbLaunch &shape, const PfbLaunch *d_pls, const GroupMap &gm, hipStream_t s)
{
    if (gm.total_wg <= 0) return true;
    return dispatch(shape, false, s, d_pls, &gm);
}

}  // namespace rcfx
