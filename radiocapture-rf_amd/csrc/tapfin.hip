// tapfin.hip -- filterbank taps: the frame-major banks' tap matrix / ring -> per-channel IQ and discriminator rings
// (tap_finalize_kernel, tap_finalize_group_kernel; gfx950).
//
// Replaces, for every tapped bin of a reference-grid bank at once, the reference's per-channel rotation inside
// freq_xlating_fir_filter_ccc (/root/reference/rc_frontend/channel.py:29-38) and the consumers' analog.quadrature_demod_cf
// (/root/reference/p25_control_demod.py:120-121).  A translation unit of its own (it was part of fir.hip) because it is
// built WITHOUT the SLP vectoriser: the vectoriser pairs the two arctangents of an output pair into v_pk_* float32
// instructions, which issue no faster than the scalar pair on gfx950 and cost the moves that pack them -- 1600 taps 0.246
// -> 0.237 ms, discriminator-only 0.181 -> 0.167 ms per 2^25-sample launch (same box, alternating) -- while fir.hip's
// vector FIR kernels are faster WITH it (Makefile).  Built without implicit FMA contraction like fir.hip: GNU Radio's
// rotator and discriminator are unfused float32 arithmetic.
#include <type_traits>
#include "rcf_internal.h"
#include "rotator.hpp"
#include "fast_atan2f_gr.hpp"

namespace rcfx {

namespace {

constexpr int kThreads = 256;

// The frame-major banks (pfb5.hip) leave the tapped bins of a launch as a compact frame-major matrix, row r = the
// bank's frame k_first + r, one column per tap.  A workgroup takes 16 taps x 128 outputs: it reads the rows the way
// they lie (16 taps = one 128-byte piece of a row), applies each tap's rotator (GNU Radio's increment and / or the
// source shift: rotate_value(), the FIR bank's epilogue) into LDS, then turns the tile round -- 16 lanes x 2
// consecutive outputs of ONE tap = two 128-byte lines of that channel's IQ ring and one of its discriminator ring per
// store -- and writes IQ and the discriminator (which needs the output before: the row above in LDS; rows before the
// launch come from the ring).  The 128 outputs of a tap are ALIGNED to 32 in the tap's own ring index (every tap has
// its own k_abs0), so that every store is whole lines: a partial line costs a read-modify-write in the memory system
// (32-byte pieces measured 4-5x slower than lines).  Every tap column of the LDS tile has its OWN row origin
// (LDS row 0 = the row before the tap's first output of the tile): a lane stages exactly the 129 rows its tap reads --
// nine rounds of sixteen -- where a common origin for the sixteen taps meant 161 rows and eleven predicated rounds.
#ifndef RCF_TAPFIN_WGS
#define RCF_TAPFIN_WGS 5      // workgroups per CU the finalize kernels are compiled for (96 VGPRs; tools/variant_lib.sh for A/B)
#endif
constexpr int kTapOut = 128, kTapCols = 16, kTapAlign = 32, kTapLdsRows = kTapOut + 1,
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
    // 128-byte pieces of the SAME matrix rows (whole rows between them), not one piece from each of 129 rows apart
    const int s0 = bx * kTapCols, r0 = by * kTapOut;
    for (int i = tid; i < 257; i += kThreads) tab[i] = atan_tab[i];
    {
        const int sl = tid & (kTapCols - 1), rr = tid >> 4;
        const int slot = s0 + sl;
        if (slot < n_taps) {
            const TapLaunch L = taps[slot];
            const bool idle = L.dangle == 0.0 && L.dlogmag == 0.0 && L.angle0 == 0.0 && L.logmag0 == 0.0;
            const int b0 = group_bin0[bx];                      // >= 0: this group's taps are 16 consecutive bins of the ring
            // all of a lane's rows are requested before the first is used (the loop below would otherwise pay one
            // memory round trip per row: nine in a row)
            constexpr int NIT = (kTapLdsRows + 15) / 16;
            float2 z[NIT];
            // the rows THIS tap's 128 outputs (and the output before them) come from: its tile starts a_own rows before r0
            // (its ring index aligned to 32); the tap's LDS column starts there too, so exactly these 129 rows are staged
            const int a_own = (int)((uint64_t)(k_first - L.k_abs0 + r0) & (kTapAlign - 1));
            const int r_lds0 = r0 - a_own - 1;                  // matrix row of this tap's LDS row 0 (the output before the tile's first)
            // The per-row index arithmetic in 32 bits (the kernel is vector-issue bound and int64 compares / masks / multiplies
            // were a fifth of its instructions): a row r is FRESH (a frame of this launch inside the tap's range) for r in
            // [rf_lo, rf_hi], OLD (an output of an earlier launch, read back from the tap's ring) for r in [ro_lo, -1]; ring
            // positions wrap in uint32 (the ring is a power of two of at most 2^31 samples)
            const TapRows tr_ = tap_rows(L, k_first, n_rows);
            const int rf_lo = tr_.lo, rf_hi = tr_.hi, ro_lo = tr_.old_lo;
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
                // (rows before the launch came out of the ring rotated already)
                if (!idle && r >= 0) v = walk.rotate(L, v.x, v.y);
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
                const int lr = 2 * (q + 16 * j) + 1;                    // LDS row of ra: the tap's row 0 is r0 - a - 1
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

}  // namespace

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

}  // namespace rcfx
