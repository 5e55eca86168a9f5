// peaks.hip -- device peak picker: /root/reference/fft_peak_detection.py:54-72 on the spectrum that
// already sits in HBM after a scan (a 2^20-bin spectrum costs ~50 ms on the host -- D2H plus
// scipy-style sequential walks -- which would dwarf the ~5 ms the FFTs take).
//
//   data[i] += abs(min(data))   (float32)         -> k_min, k_prep
//   mean = sum(data) / n        (float64)         -> k_prep (block partials) + k_tables (ordered final sum)
//   scipy.signal.find_peaks(data, width=[min_w, max_w], prominence=p), keep data[line] > 2 mean
//                                                -> k_pick (one thread per sample), k_sort
// Bit-exactness with SciPy: every comparison and the half-prominence interpolation run in float64 on
// the float32->float64 widened samples, exactly the operations SciPy's Cython kernels perform; the file
// is built with -ffp-contract=off.  Prominence needs min(x) over the stretch between the peak and the
// first strictly higher sample on each side: the walk skips 32-sample and 1024-sample blocks whose
// maximum does not exceed the peak (block max / min tables), so a thread does O(32 + 32 + N/1024)
// steps instead of O(N).  The bases' positions never bound the width walk when prominence > 0 (the
// half-prominence level lies strictly above both base values), so only the minima are needed.
// The float64 mean is reduced as a fixed tree (block partials summed in index order); it equals the
// reference's left-to-right sum bit for bit whenever the additions are exact, which holds for these
// spectra (float32 values sharing a 2^-18 grid, total < 2^33); otherwise it differs in the last bits
// and only a peak within 1e-13 (relative) of the 2*mean gate could be affected.
#include "rcf_internal.h"

namespace rcfx {

namespace {

constexpr int B1 = 32;      // level-1 block
constexpr int B2 = 1024;    // level-2 block (32 level-1 blocks)

constexpr int kMinBlocks = 256;

__device__ __forceinline__ float block_min256(float m, float *red)
{
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    return red[0];
}

// partial minima (min is exact and order-free): kMinBlocks blocks x 256 threads, grid-stride
__global__ __launch_bounds__(256) void k_min(const float *__restrict__ spec, int n, float *part_min)
{
    __shared__ float red[256];
    float m = spec[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += kMinBlocks * 256) m = fminf(m, spec[i]);
    m = block_min256(m, red);
    if (threadIdx.x == 0) part_min[blockIdx.x] = m;
}

// x[i] = spec[i] + shift (float32); level-1 max / min; per-1024 partial sums (double, index order inside)
__global__ __launch_bounds__(256) void k_prep(const float *__restrict__ spec, int n, const float *shift_p,
                                              float *__restrict__ x, float *__restrict__ max1,
                                              float *__restrict__ min1, double *__restrict__ part)
{
    __shared__ double ps[256];
    __shared__ float red[256];
    // shift = |min(data)|: every workgroup reduces the kMinBlocks partial minima itself (256 floats)
    const float shift = fabsf(block_min256(shift_p[threadIdx.x], red));
    const int base = blockIdx.x * B2;
    const int t = threadIdx.x;
    double s = 0.0;
    // thread t owns samples base + 4t .. base + 4t + 3 (index order inside the thread)
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = base + 4 * t + j;
        v[j] = i < n ? __fadd_rn(spec[i], shift) : 0.f;
        if (i < n) { x[i] = v[j]; s += (double)v[j]; }
    }
    ps[t] = s;
    float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    float mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
    // 8 threads = one level-1 block of 32 samples
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        mn = fminf(mn, __shfl_xor(mn, off, 64));
    }
    if ((t & 7) == 0) {
        const int b = (base + 4 * t) / B1;
        if (base + 4 * t < n) { max1[b] = mx; min1[b] = mn; }
    }
    __syncthreads();
    if (t == 0) {
        double a = 0.0;
        for (int i = 0; i < 256; ++i) a += ps[i];
        part[blockIdx.x] = a;
    }
}

// level-2 tables from level-1, ordered final sum -> mean
__global__ __launch_bounds__(1024) void k_tables(const float *__restrict__ max1, const float *__restrict__ min1,
                                                 int n1, float *__restrict__ max2, float *__restrict__ min2, int n2,
                                                 const double *__restrict__ part, int n, double *mean_out)
{
    for (int b = threadIdx.x; b < n2; b += 1024) {
        float mx = -INFINITY, mn = INFINITY;
        for (int j = 0; j < B2 / B1; ++j) {
            const int i = b * (B2 / B1) + j;
            if (i < n1) { mx = fmaxf(mx, max1[i]); mn = fminf(mn, min1[i]); }
        }
        max2[b] = mx;
        min2[b] = mn;
    }
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int i = 0; i < n2; ++i) a += part[i];
        *mean_out = a / (double)n;
    }
}

struct PickArgs {
    const float *x;
    const float *max1, *min1, *max2, *min2;
    int n, n1, n2;
    double min_w, max_w, prominence;
    const double *mean;
    int64_t *out;          // unordered survivors
    int *count;
    int cap;
};

// min over the stretch from p outward (dir = -1 / +1) up to, not including, the first sample > h
__device__ double walk_min(const PickArgs &a, int p, double h, int dir)
{
    const float *x = a.x;
    const float hf_note = 0.f;
    (void)hf_note;
    double m = h;
    int i = p;
    const int last = a.n - 1;
    // 1) samples up to the level-1 boundary
    while (true) {
        i += dir;
        if (i < 0 || i > last) return m;
        const double v = (double)x[i];
        if (v > h) return m;
        if (v < m) m = v;
        if (dir < 0 ? (i % B1 == 0) : (i % B1 == B1 - 1)) break;
    }
    // 2) level-1 blocks up to the level-2 boundary, 3) level-2 blocks, descending when a block holds a higher sample
    int b1 = i / B1 + dir;
    while (b1 >= 0 && b1 < a.n1) {
        if (dir < 0 ? (b1 % (B2 / B1) == (B2 / B1) - 1) : (b1 % (B2 / B1) == 0)) {
            // at a level-2 boundary: skip whole level-2 blocks
            int b2 = b1 / (B2 / B1);
            while (b2 >= 0 && b2 < a.n2 && !((double)a.max2[b2] > h)) {
                const double v = (double)a.min2[b2];
                if (v < m) m = v;
                b2 += dir;
            }
            if (b2 < 0 || b2 >= a.n2) return m;
            b1 = dir < 0 ? b2 * (B2 / B1) + (B2 / B1) - 1 : b2 * (B2 / B1);
        }
        if ((double)a.max1[b1] > h) {
            // the stop is inside this block
            int j = dir < 0 ? b1 * B1 + B1 - 1 : b1 * B1;
            if (j > last) j = last;
            for (;; j += dir) {
                const double v = (double)x[j];
                if (v > h) return m;
                if (v < m) m = v;
            }
        }
        const double v = (double)a.min1[b1];
        if (v < m) m = v;
        b1 += dir;
    }
    return m;
}

__global__ __launch_bounds__(256) void k_pick(PickArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int last = a.n - 1;
    if (i < 1 || i >= last) return;
    const float *x = a.x;
    const double xi = (double)x[i];
    if (!((double)x[i - 1] < xi)) return;
    int e = i + 1;
    while (e < last && (double)x[e] == xi) ++e;          // plateau
    if (!((double)x[e] < xi)) return;
    const int pk = (i + e - 1) / 2;
    const double top = (double)x[pk];
    const double lmin = walk_min(a, pk, top, -1);
    const double rmin = walk_min(a, pk, top, +1);
    const double prom = top - (lmin > rmin ? lmin : rmin);
    if (!(prom >= a.prominence)) return;
    const double level = top - prom * 0.5;
    // width at half prominence; walks are bounded: a side longer than max_w already fails the window
    const int bound = (int)fmin(a.max_w + 2.0, (double)a.n);
    int j = pk;
    while (j > 0 && level < (double)x[j]) {
        --j;
        if (pk - j > bound) return;
    }
    double left = (double)j;
    if ((double)x[j] < level) left += (level - (double)x[j]) / ((double)x[j + 1] - (double)x[j]);
    j = pk;
    while (j < last && level < (double)x[j]) {
        ++j;
        if (j - pk > bound) return;
    }
    double right = (double)j;
    if ((double)x[j] < level) right -= (level - (double)x[j]) / ((double)x[j - 1] - (double)x[j]);
    const double width = right - left;
    if (!(a.min_w <= width && width <= a.max_w)) return;
    if (!(top > *a.mean * 2)) return;
    const int slot = atomicAdd(a.count, 1);
    if (slot < a.cap) a.out[slot] = pk;
}

// ascending sort of min(count, cap) survivors (cap <= 4096), padded with -1 after the valid entries
__global__ __launch_bounds__(1024) void k_sort(int64_t *buf, const int *count, int cap)
{
    __shared__ int64_t s[4096];
    const int n = min(*count, cap);
    for (int i = threadIdx.x; i < 4096; i += 1024) s[i] = i < n ? buf[i] : INT64_MAX;
    __syncthreads();
    for (int k = 2; k <= 4096; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < 4096; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const int64_t a = s[i], b = s[l];
                    if ((a > b) == up) { s[i] = b; s[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < cap; i += 1024) buf[i] = i < n ? s[i] : -1;
}

}  // namespace

// workspace layout (floats unless noted); returns bytes needed
size_t peaks_workspace_bytes(int n)
{
    const size_t n1 = (n + B1 - 1) / B1, n2 = (n + B2 - 1) / B2;
    return sizeof(float) * ((size_t)n + 2 * n1 + 2 * n2 + kMinBlocks + 16) + sizeof(double) * (n2 + 2) + 64;
}

// dev_out: int64[cap] (cap <= 4096); dev_count: int; dev_mean: double -- all inside `ws` after the tables
void launch_find_peaks(const float *d_spec, int n, double min_w, double max_w, double prominence, void *ws,
                       int64_t *d_out, int cap, int **d_count_out, double **d_mean_out, hipStream_t s)
{
    const int n1 = (n + B1 - 1) / B1, n2 = (n + B2 - 1) / B2;
    unsigned char *p = static_cast<unsigned char *>(ws);
    double *part = reinterpret_cast<double *>(p);  p += sizeof(double) * n2;
    double *mean = reinterpret_cast<double *>(p);  p += sizeof(double) * 2;
    float *x = reinterpret_cast<float *>(p);       p += sizeof(float) * (size_t)n;
    float *max1 = reinterpret_cast<float *>(p);    p += sizeof(float) * n1;
    float *min1 = reinterpret_cast<float *>(p);    p += sizeof(float) * n1;
    float *max2 = reinterpret_cast<float *>(p);    p += sizeof(float) * n2;
    float *min2 = reinterpret_cast<float *>(p);    p += sizeof(float) * n2;
    float *shift = reinterpret_cast<float *>(p);   p += sizeof(float) * kMinBlocks;
    int *count = reinterpret_cast<int *>(p);
    (void)hipMemsetAsync(count, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_min, dim3(kMinBlocks), dim3(256), 0, s, d_spec, n, shift);
    hipLaunchKernelGGL(k_prep, dim3(n2), dim3(256), 0, s, d_spec, n, shift, x, max1, min1, part);
    hipLaunchKernelGGL(k_tables, dim3(1), dim3(1024), 0, s, max1, min1, n1, max2, min2, n2, part, n, mean);
    PickArgs a{x, max1, min1, max2, min2, n, n1, n2, min_w, max_w, prominence, mean, d_out, count, cap};
    hipLaunchKernelGGL(k_pick, dim3((n + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_sort, dim3(1), dim3(1024), 0, s, d_out, count, cap);
    *d_count_out = count;
    *d_mean_out = mean;
}

}  // namespace rcfx
