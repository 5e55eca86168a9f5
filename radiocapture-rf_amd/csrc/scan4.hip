// scan4.hip -- four-step (N = N1 x N2) variant of the scan FFT for transforms that do not fit one
// workgroup's LDS: 2^15 .. 2^20 points (BASELINE configs[2]: 1M-point scan at 100 Msps).
//
// With n = N2 n1 + n2 and k = k1 + N1 k2:
//   X[k1 + N1 k2] = sum_{n2} W_N2^{n2 k2} * ( W_N^{n2 k1} * sum_{n1} W_N1^{n1 k1} x[N2 n1 + n2] )
// N1 is kept SMALL (256; 128 for N = 2^15) so that the column kernel's LDS tile is 35 KB and four
// workgroups share a CU (load / FFT / store phases of different tiles overlap); the long dimension N2
// (up to 4096) runs along contiguous rows in scan.hip's single-pass FFT machinery.
// Kernel A (here, columns): a workgroup takes 16 consecutive columns n2 (16 x 8 B = one 128-byte line per
//   row), applies the Blackman-Harris window on load, runs 16 length-N1 FFTs in LDS, multiplies by
//   W_N^{n2 k1} (two tables: W_N^{1024 j} * W_N^{i}) and writes scratch[k1][n2] in 128-byte runs.
// Kernel B (scan.hip, rows): length-N2 FFTs along contiguous rows, log-magnitude stored in row order
//   (fully coalesced); the un-permute k = k1 + N1 k2 and the fftshift happen once, on the emitted vector.
// HBM traffic per sample: 8 (read) + 8 (scratch write) + 8 (scratch read) + 4 (write) = 28 B against
// 12 B algorithmic -- the four-step ceiling stated in SURVEY.md 8(d) (~43 % of the algorithmic roofline).
#include <cstdlib>

#include "fft_core.hpp"
#include "rcf_internal.h"

namespace rcfx {

void launch_scan4_rows(const ScanLaunch &p, const cf *tw2, int N1, int N2, hipStream_t s);   // scan.hip

namespace {


template <int N> struct P4;
template <> struct P4<128> { static constexpr int r[2] = {16, 8}; };
template <> struct P4<256> { static constexpr int r[2] = {16, 16}; };

template <int N> constexpr int rs4() { return lds_padded_len(N) + 1; }   // odd: transposed accesses spread

// one radix-R pass over CW length-N transforms in LDS, N threads, forward sign, twiddles e^{-2 pi i n/N}
template <int N, int R, int NS, int CW>
__device__ __forceinline__ void pass4(cf *buf, const cf *tw_lds, int tid)
{
    constexpr int BPF = N / R;
    constexpr int CNT = CW / R;
    constexpr int RS = rs4<N>();
    using Pass = StockhamPass<N, R, -1>;
    const int j = tid % BPF;
    cf v[CNT][R];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int row = (tid + i * N) / BPF;
        Pass::load(buf + row * RS, j, v[i]);
    }
    if (NS > 1) {
        const int k = j & (NS - 1);
#pragma unroll
        for (int t = 1; t < R; ++t) {
            const cf w = tw_lds[(k * t) * (N / (NS * R))];
#pragma unroll
            for (int i = 0; i < CNT; ++i) v[i][t] = cmul(v[i][t], w);
        }
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i) Dft<R, -1>::run(v[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int row = (tid + i * N) / BPF;
        Pass::template store_t<NS>(buf + row * RS, j, v[i]);
    }
    __syncthreads();
}

struct Scan4Args {
    ScanLaunch p;
    int N1, N2;
    const cf *tw1;      // e^{-2 pi i n / N1}
    const cf *tlo;      // W_N^{i},        i < 1024
    const cf *thi;      // W_N^{1024 j},   j < N / 1024
};

// grid: (N2 / CW, frames)
template <int N1, int CW>
__global__ __launch_bounds__(N1) void scan4_cols(Scan4Args a)
{
    constexpr int RS = rs4<N1>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    cf *tw_lds = buf + CW * RS;
    const int tid = threadIdx.x;
    const int N2 = a.N2, N = a.p.N;
    const int c0 = blockIdx.x * CW;
    const int fl = blockIdx.y;
    tw_lds[tid] = a.tw1[tid];
    const int64_t s0 = a.p.s0 + (int64_t)fl * N;
    // load: lane -> (column c = e % 16, row n1 = e / 16): 16 lanes read one 128-byte line
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int e = tid + i * N1;
        const int c = e % CW, n1 = e / CW;
        const int n = N2 * n1 + c0 + c;
        // (every input sample is read once: non-temporal, FFT pair 6.24 -> 6.07 ms per 1000 frames; the window stays cached)
        typedef float v2f_ __attribute__((ext_vector_type(2)));
        const v2f_ t_ = __builtin_nontemporal_load(reinterpret_cast<const v2f_ *>(a.p.src.base + ((uint64_t)(s0 + n - a.p.src.origin) & a.p.src.mask)));
        cf x = make_float2(t_.x, t_.y);
        const float w = a.p.window[n];
        buf[c * RS + lds_pad(n1)] = make_float2(__fmul_rn(x.x, w), __fmul_rn(x.y, w));
    }
    __syncthreads();
    pass4<N1, P4<N1>::r[0], 1, CW>(buf, tw_lds, tid);
    pass4<N1, P4<N1>::r[1], P4<N1>::r[0], CW>(buf, tw_lds, tid);
    cf *scr = a.p.scratch + (size_t)fl * N;
    // W_N^{n2 k1}: this thread's column n2 is fixed and its k1 advances by N1 / CW per element, so after the first
    // element (two table entries: W_N^{1024 j} W_N^{i}) the factor advances by one constant, W_N^{n2 N1 / CW}
    // (again two table entries).  Sixteen steps of a float32 recurrence: a few 1e-7 of error, against 32 dependent
    // table loads per thread before.
    {
        const int c = tid % CW, k10 = tid / CW;
        const unsigned n2c = (unsigned)(c0 + c);
        const unsigned m0 = n2c * (unsigned)k10;                              // < N
        const unsigned ms = n2c * (unsigned)(N1 / CW);                        // < N
        cf w = cmul(a.thi[m0 >> 10], a.tlo[m0 & 1023]);
        const cf step = cmul(a.thi[ms >> 10], a.tlo[ms & 1023]);
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const int k1 = k10 + i * (N1 / CW);
            {
                // streamed out: the row pass reads it back a whole launch (268 MB) later (non-temporal: FFT pair 6.39 ->
                // 6.29 ms per 1000 frames; the row pass's own stores measured 2 % SLOWER that way)
                const cf z_ = cmul(buf[c * RS + lds_pad(k1)], w);
                typedef float v2f_ __attribute__((ext_vector_type(2)));
                v2f_ o_; o_.x = z_.x; o_.y = z_.y;
                __builtin_nontemporal_store(o_, reinterpret_cast<v2f_ *>(scr + (size_t)k1 * N2 + c0 + c));
            }
            w = cmul(w, step);
        }
    }
}

template <int N1, int CW>
void launch_cols(const Scan4Args &a, hipStream_t s)
{
    const size_t lds = ((size_t)CW * rs4<N1>() + N1) * sizeof(cf);
    static DynLdsAttr attr;
    attr.ensure(reinterpret_cast<const void *>(scan4_cols<N1, CW>), lds);
    hipLaunchKernelGGL((scan4_cols<N1, CW>), dim3(a.N2 / CW, a.p.n_frames), dim3(N1), lds, s, a);
}

}  // namespace

bool scan4_split(int N, int *N1, int *N2)
{
    if (N < (1 << 15) || N > (1 << 20) || (N & (N - 1))) return false;
    *N1 = N >= (1 << 16) ? 256 : 128;
    *N2 = N / *N1;
    return true;
}

bool scan4_supported(int N)
{
    int a, b;
    return scan4_split(N, &a, &b);
}

// tw layout prepared by the host (rcf_scan_start): [tw1 (N1) | tw2 (N2) | tlo (1024) | thi (N/1024)]
void launch_scan4_fft(const ScanLaunch &p, hipStream_t s)
{
    Scan4Args a{};
    a.p = p;
    if (!scan4_split(p.N, &a.N1, &a.N2)) return;
    a.tw1 = p.tw;
    const cf *tw2 = p.tw + a.N1;
    a.tlo = tw2 + a.N2;
    a.thi = a.tlo + 1024;
    // N1 = 256: 32 columns per workgroup = 256-byte runs per row on both sides, 72 KB of LDS, two workgroups per CU --
    // 2.7 % faster than 16 columns (128-byte runs, 37 KB, four per CU): 6.61 -> 6.43 ms per 1000 frames at N = 2^20
    // (RCF_SCAN_CW32=0 keeps the 16-column form)
    static const bool cw32 = [] { const char *e = getenv("RCF_SCAN_CW32"); return !e || atoi(e) != 0; }();
    if (a.N1 == 128) launch_cols<128, 16>(a, s);
    else if (cw32)   launch_cols<256, 32>(a, s);
    else             launch_cols<256, 16>(a, s);
    launch_scan4_rows(p, tw2, a.N1, a.N2, s);
}

}  // namespace rcfx
