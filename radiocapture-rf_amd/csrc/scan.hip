// scan.hip -- spectrum scan chain of /root/reference/fft_vector.py:37-60 on gfx950:
//   stream_to_vector(N) -> fft_vcc(N, forward, blackmanharris(N), shift) -> complex_to_mag_squared
//   -> nlog10_ff(1, N, 1) -> moving_average_ff(L, 1, ., N) -> head / skiphead (keep frame n_frames-1)
//
// Kernel 1 (scan_fft_kernel): one workgroup transforms 16*NT/N frames entirely in LDS (NT threads,
// 16 points per thread per pass, radix-16/../2 Stockham passes from fft_core.hpp); the window is
// applied on the coalesced load, and |X|^2 -> log10 -> +1 -> fftshift are fused into the store, so a
// frame costs 8 B/sample of HBM reads and 4 B/sample of writes.
// Kernel 2 (movsum_kernel): GNU Radio's float32 running sum, bit-faithful in operation order
// (sum += newest; emit; sum -= oldest), one thread per bin walking the chunk's frames in order.
// Larger transforms (N > 16384) use the four-step variant in scan4.hip.
#include "fft_core.hpp"
#include "rcf_internal.h"

namespace rcfx {

bool scan4_supported(int N);
void launch_scan4_fft(const ScanLaunch &p, hipStream_t s);

namespace {

template <int N> struct SPlan;
template <> struct SPlan<256>   { static constexpr int n = 2; static constexpr int r[4] = {16, 16, 1, 1}; };
template <> struct SPlan<512>   { static constexpr int n = 3; static constexpr int r[4] = {16, 16, 2, 1}; };
template <> struct SPlan<1024>  { static constexpr int n = 3; static constexpr int r[4] = {16, 16, 4, 1}; };
template <> struct SPlan<2048>  { static constexpr int n = 3; static constexpr int r[4] = {16, 16, 8, 1}; };
template <> struct SPlan<4096>  { static constexpr int n = 3; static constexpr int r[4] = {16, 16, 16, 1}; };
template <> struct SPlan<8192>  { static constexpr int n = 4; static constexpr int r[4] = {16, 16, 16, 2}; };
template <> struct SPlan<16384> { static constexpr int n = 4; static constexpr int r[4] = {16, 16, 16, 4}; };

template <int N> constexpr int scan_threads() { return N / 16 > 256 ? N / 16 : 256; }
template <int N> constexpr int scan_fpw() { return scan_threads<N>() * 16 / N; }          // frames per workgroup
template <int N> constexpr int scan_rs() { return lds_padded_len(N) + 1; }

template <int N, int NT, int R, int NS>
__device__ __forceinline__ void scan_pass(cf *buf, const cf *__restrict__ tw, int tid)
{
    constexpr int BPF = N / R;
    constexpr int CNT = 16 / R;               // butterflies per thread per pass
    constexpr int RS = scan_rs<N>();
    using Pass = StockhamPass<N, R, -1>;
    cf v[CNT][R];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int b = tid + i * NT;
        const int frame = b / BPF, j = b % BPF;
        Pass::load(buf + frame * RS, j, v[i]);
        Pass::twiddle(tw, NS, j, v[i]);
        Dft<R, -1>::run(v[i]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int b = tid + i * NT;
        const int frame = b / BPF, j = b % BPF;
        Pass::template store_t<NS>(buf + frame * RS, j, v[i]);
    }
    __syncthreads();
}

__device__ __forceinline__ float logmag_gr(cf X)
{
    // complex_to_mag_squared (unfused) -> volk log2 (log2f, -inf -> -127) * (1/log2(10)) -> + 1
    const float p = __fadd_rn(__fmul_rn(X.x, X.x), __fmul_rn(X.y, X.y));
    float l2 = log2f(p);
    if (isinf(l2)) l2 = copysignf(127.0f, l2);
    const float scale = 0.30102999566398120f;   // 1 / log2(10)
    return __fadd_rn(__fmul_rn(l2, scale), 1.0f);
}

template <int N>
__global__ __launch_bounds__(scan_threads<N>()) void scan_fft_kernel(ScanLaunch p)
{
    constexpr int NT = scan_threads<N>();
    constexpr int FPW = scan_fpw<N>();
    constexpr int RS = scan_rs<N>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    const int tid = threadIdx.x;
    const int fl0 = blockIdx.x * FPW;          // first local frame of this workgroup

    // load + window
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * NT;
        const int fl = e / N, idx = e % N;
        cf x = make_float2(0.f, 0.f);
        if (fl0 + fl < p.n_frames) {
            const int64_t s = p.s0 + (int64_t)(fl0 + fl) * N + idx;
            x = p.src.base[(uint64_t)(s - p.src.origin) & p.src.mask];
            const float w = p.window[idx];
            x = make_float2(__fmul_rn(x.x, w), __fmul_rn(x.y, w));
        }
        buf[fl * RS + lds_pad(idx)] = x;
    }
    __syncthreads();

    scan_pass<N, NT, SPlan<N>::r[0], 1>(buf, p.tw, tid);
    if constexpr (SPlan<N>::n >= 2) scan_pass<N, NT, SPlan<N>::r[1], SPlan<N>::r[0]>(buf, p.tw, tid);
    if constexpr (SPlan<N>::n >= 3)
        scan_pass<N, NT, SPlan<N>::r[2], SPlan<N>::r[0] * SPlan<N>::r[1]>(buf, p.tw, tid);
    if constexpr (SPlan<N>::n >= 4)
        scan_pass<N, NT, SPlan<N>::r[3], SPlan<N>::r[0] * SPlan<N>::r[1] * SPlan<N>::r[2]>(buf, p.tw, tid);

    // |X|^2 -> log10 + 1 -> fftshift -> ring slot of the frame
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * NT;
        const int fl = e / N, k = e % N;
        if (fl0 + fl < p.n_frames) {
            const int f = p.f0 + fl0 + fl;
            const float v = logmag_gr(buf[fl * RS + lds_pad(k)]);
            p.vring[(size_t)(f % p.R) * N + ((k + N / 2) & (N - 1))] = v;
        }
    }
}

// One thread per bin; the add/subtract chain is inherently sequential in float32 (that IS the
// specified result), but the loads are not: U frames' newest/oldest values are fetched up front so the
// chain runs on registers.  64-thread workgroups so that a 16384-bin spectrum still spreads over all CUs.
// Four-step frames (n1 > 0) sit in the ring in row order p = k1 * n2 + k2 (frequency k = k1 + n1 * k2, not
// yet fft-shifted): the running sum does not care about the order of bins, so the un-permute and the
// fftshift are applied only to the ONE emitted vector.
constexpr int kMovThreads = 64;
// U frames per batch: the 2 U loads of a batch are independent, the 2 U adds behind them are the sequential part.
// A scan of N bins is N independent chains and nothing else: at N = 16384 (the reference's size) that is one
// wavefront per CU, so the only latency hiding there is per thread -- U = 32 (208 -> ~60 us per 512-frame launch);
// at N = 2^20 there are 16 K wavefronts and U = 8 keeps the registers down.
template <int U>
__global__ __launch_bounds__(kMovThreads) void movsum_kernel(const float *__restrict__ vring, int N, int R, int L, int f0,
                                                            int n_frames, int emit_frame, float *__restrict__ sum,
                                                            float *__restrict__ out_base, int n1, int n2)
{
    const int k = blockIdx.x * kMovThreads + threadIdx.x;
    if (k >= N) return;
    float *out = out_base;
    if (n1 > 0) {
        const int k1 = k / n2, k2 = k - k1 * n2;
        out = out_base + (((k1 + n1 * k2) + N / 2) & (N - 1)) - k;      // so that out[k] lands on the shifted bin
    }
    float s = sum[k];
    int f = f0;
    const int f_end = f0 + n_frames;
    for (; f + U <= f_end; f += U) {
        float vn[U], vo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (both rows are read exactly once by this kernel: non-temporal loads, running sum 1.80 -> 1.70 ms per 1000 frames)
            vn[u] = __builtin_nontemporal_load(vring + (size_t)((f + u) % R) * N + k);
            const int fo = f + u - (L - 1);
            vo[u] = fo >= 0 ? __builtin_nontemporal_load(vring + (size_t)(fo % R) * N + k) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s = __fadd_rn(s, vn[u]);
            if (f + u == emit_frame) out[k] = s;
            if (f + u - (L - 1) >= 0) s = __fsub_rn(s, vo[u]);
        }
    }
    for (; f < f_end; ++f) {
        s = __fadd_rn(s, vring[(size_t)(f % R) * N + k]);
        if (f == emit_frame) out[k] = s;
        const int fo = f - (L - 1);
        if (fo >= 0) s = __fsub_rn(s, vring[(size_t)(fo % R) * N + k]);
    }
    sum[k] = s;
}

// Small spectra (N <= 2^17): with one thread per bin there are too few wavefronts to keep HBM busy.  Here a
// workgroup of kCoopWaves wavefronts owns 64 bins; the frames go by in tiles of TT: every wavefront fetches its share
// of the tile's newest and oldest rows (64 bins = one 256-byte run per row) into registers, the tile is handed over
// through LDS, and wavefront 0 alone runs the sequential add/subtract chain from LDS while everyone's loads for the
// NEXT tile are already in flight: 2 TT loads per bin outstanding instead of 64.
constexpr int kCoopWaves = 4;
template <int TT>
__global__ __launch_bounds__(64 * kCoopWaves) void movsum_coop_kernel(const float *__restrict__ vring, int N, int R, int L,
                                                                      int f0, int n_frames, int emit_frame,
                                                                      float *__restrict__ sum, float *__restrict__ out_base,
                                                                      int n1, int n2)
{
    constexpr int RPW = TT / kCoopWaves;                   // rows per wavefront per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *lds = reinterpret_cast<float *>(smem_raw);      // [2 buffers][new | old][TT][64]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.x * 64 + lane;
    const bool live = k < N;
    const int kk = live ? k : 0;
    float *out = out_base;
    if (n1 > 0) {
        const int k1 = kk / n2, k2 = kk - k1 * n2;
        out = out_base + (((k1 + n1 * k2) + N / 2) & (N - 1)) - kk;
    }
    float s = (w == 0 && live) ? sum[kk] : 0.f;
    const int f_end = f0 + n_frames;
    float vn[RPW], vo[RPW];
    // ring slots of a tile: one modulo per tile, then +1 per frame with a conditional wrap (TT <= R, checked by the
    // launcher); rows past the end / before frame 0 are fetched from a valid slot and never enter the chain
    auto fetch = [&](int ft) {
        const int sn0 = ft % R;
        const int so0 = (int)(((int64_t)ft - (L - 1) + (int64_t)R * (1 + (L - 1) / R)) % R);
        const float *col = vring + kk;
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int d = w + u * kCoopWaves;
            int sn = sn0 + d, so = so0 + d;
            if (sn >= R) sn -= R;
            if (so >= R) so -= R;
            vn[u] = col[(size_t)sn * N];
            vo[u] = col[(size_t)so * N];
        }
    };
    fetch(f0);
    int t = 0;
    for (int ft = f0; ft < f_end; ft += TT, t ^= 1) {
        float *bn = lds + (size_t)t * (2 * TT * 64), *bo = bn + TT * 64;
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            bn[(w + u * kCoopWaves) * 64 + lane] = vn[u];
            bo[(w + u * kCoopWaves) * 64 + lane] = vo[u];
        }
        __syncthreads();
        if (ft + TT < f_end) fetch(ft + TT);
        if (w == 0) {
            const int nr = min(TT, f_end - ft);
            for (int r0 = 0; r0 < nr; r0 += 16) {
                float a[16], b[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { a[u] = bn[(r0 + u) * 64 + lane]; b[u] = bo[(r0 + u) * 64 + lane]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int f = ft + r0 + u;
                    if (f < f_end) {
                        s = __fadd_rn(s, a[u]);
                        if (f == emit_frame && live) out[kk] = s;
                        if (f - (L - 1) >= 0) s = __fsub_rn(s, b[u]);
                    }
                }
            }
        }
    }
    if (w == 0 && live) sum[kk] = s;
}

// Four-step, second half: length-N2 FFTs along the CONTIGUOUS rows of the scratch matrix [N1][N2] the
// column kernel (scan4.hip) produced, then |X|^2 -> log10 -> +1.  Output stays in row order
// (p = k1 * N2 + k2): fully coalesced float stores, no transpose -- movsum_kernel un-permutes the one
// emitted vector.  grid = (N1 / FPW, frames).
template <int N2>
__global__ __launch_bounds__(scan_threads<N2>()) void scan4_rows_kernel(ScanLaunch p, const cf *__restrict__ tw2, int N1)
{
    constexpr int NT = scan_threads<N2>();
    constexpr int FPW = scan_fpw<N2>();                  // rows per workgroup
    constexpr int RS = scan_rs<N2>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *buf = reinterpret_cast<cf *>(smem_raw);
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * FPW;
    const int fl = blockIdx.y;
    const cf *scr = p.scratch + (size_t)fl * p.N + (size_t)r0 * N2;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * NT;
        const int r = e / N2, idx = e % N2;
        buf[r * RS + lds_pad(idx)] = scr[(size_t)r * N2 + idx];
    }
    __syncthreads();
    scan_pass<N2, NT, SPlan<N2>::r[0], 1>(buf, tw2, tid);
    if constexpr (SPlan<N2>::n >= 2) scan_pass<N2, NT, SPlan<N2>::r[1], SPlan<N2>::r[0]>(buf, tw2, tid);
    if constexpr (SPlan<N2>::n >= 3)
        scan_pass<N2, NT, SPlan<N2>::r[2], SPlan<N2>::r[0] * SPlan<N2>::r[1]>(buf, tw2, tid);
    if constexpr (SPlan<N2>::n >= 4)
        scan_pass<N2, NT, SPlan<N2>::r[3], SPlan<N2>::r[0] * SPlan<N2>::r[1] * SPlan<N2>::r[2]>(buf, tw2, tid);
    const int f = p.f0 + fl;
    float *dst = p.vring + (size_t)(f % p.R) * p.N + (size_t)r0 * N2;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * NT;
        const int r = e / N2, k2 = e % N2;
        dst[(size_t)r * N2 + k2] = logmag_gr(buf[r * RS + lds_pad(k2)]);
    }
    (void)N1;
}

template <int N2>
void launch_rows(const ScanLaunch &p, const cf *tw2, int N1, hipStream_t s)
{
    constexpr int FPW = scan_fpw<N2>();
    const size_t lds = (size_t)FPW * scan_rs<N2>() * sizeof(cf);
    static DynLdsAttr attr;
    attr.ensure(reinterpret_cast<const void *>(&scan4_rows_kernel<N2>), lds);
    hipLaunchKernelGGL((scan4_rows_kernel<N2>), dim3(N1 / FPW, p.n_frames), dim3(scan_threads<N2>()), lds, s, p, tw2, N1);
}

template <int N>
void launch_n(const ScanLaunch &p, hipStream_t s)
{
    constexpr int FPW = scan_fpw<N>();
    const int n_wg = (p.n_frames + FPW - 1) / FPW;
    const size_t lds = (size_t)FPW * scan_rs<N>() * sizeof(cf);
    static DynLdsAttr attr;
    attr.ensure(reinterpret_cast<const void *>(&scan_fft_kernel<N>), lds);
    hipLaunchKernelGGL((scan_fft_kernel<N>), dim3(n_wg), dim3(scan_threads<N>()), lds, s, p);
}

}  // namespace

bool scan_supported(int N)
{
    if (N >= 256 && N <= 16384 && (N & (N - 1)) == 0) return true;
    return scan4_supported(N);
}

void launch_scan_fft(const ScanLaunch &p, hipStream_t s)
{
    if (p.n_frames <= 0) return;
    switch (p.N) {
        case 256:   launch_n<256>(p, s); break;
        case 512:   launch_n<512>(p, s); break;
        case 1024:  launch_n<1024>(p, s); break;
        case 2048:  launch_n<2048>(p, s); break;
        case 4096:  launch_n<4096>(p, s); break;
        case 8192:  launch_n<8192>(p, s); break;
        case 16384: launch_n<16384>(p, s); break;
        default:    launch_scan4_fft(p, s); break;
    }
}

void launch_scan_movsum(float *vring, int N, int R, int L, int f0, int n_frames, int emit_frame, float *sum,
                        float *out, hipStream_t s)
{
    if (n_frames <= 0) return;
    int n1 = 0, n2 = 0;
    if (N > 16384 && !scan4_split(N, &n1, &n2)) return;
    static const int coop = [] { const char *e = getenv("RCF_SCAN_MOVSUM_COOP"); return e ? atoi(e) : 1; }();
    if (N <= (1 << 15) && coop && R >= 128) {   // measured: 16384 bins 53 -> 34 us per 512 frames, 131072 bins 2x slower
        constexpr int TT = 128;
        const size_t lds = sizeof(float) * 2 * 2 * TT * 64;
        static DynLdsAttr attr;
        attr.ensure(reinterpret_cast<const void *>(movsum_coop_kernel<TT>), lds);
        hipLaunchKernelGGL(movsum_coop_kernel<TT>, dim3((N + 63) / 64), dim3(64 * kCoopWaves), lds, s, vring, N, R, L, f0,
                           n_frames, emit_frame, sum, out, n1, n2);
    } else if (N <= (1 << 17))
        hipLaunchKernelGGL(movsum_kernel<32>, dim3((N + kMovThreads - 1) / kMovThreads), dim3(kMovThreads), 0, s, vring, N, R, L,
                           f0, n_frames, emit_frame, sum, out, n1, n2);
    else
        hipLaunchKernelGGL(movsum_kernel<8>, dim3((N + kMovThreads - 1) / kMovThreads), dim3(kMovThreads), 0, s, vring, N, R, L,
                           f0, n_frames, emit_frame, sum, out, n1, n2);
}

// rows of the four-step transform (called from scan4.hip after the column kernel)
void launch_scan4_rows(const ScanLaunch &p, const cf *tw2, int N1, int N2, hipStream_t s)
{
    switch (N2) {
        case 256:  launch_rows<256>(p, tw2, N1, s); break;
        case 512:  launch_rows<512>(p, tw2, N1, s); break;
        case 1024: launch_rows<1024>(p, tw2, N1, s); break;
        case 2048: launch_rows<2048>(p, tw2, N1, s); break;
        case 4096: launch_rows<4096>(p, tw2, N1, s); break;
        default: break;
    }
}

}  // namespace rcfx
