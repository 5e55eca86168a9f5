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
#include <algorithm>
#include <cstdlib>

#include "rcf_internal.h"

namespace rcfx {

namespace {

constexpr int B1 = 32;      // level-1 block
constexpr int B2 = 1024;    // level-2 block (32 level-1 blocks)

constexpr int kMinBlocks = 256;
constexpr int kHeavyCap = 1 << 16;   // peaks k_pick may hand to k_pick_heavy (a spectrum has a handful)

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
    // ordered final sum: the partials go to LDS in one round trip (one thread reading them from global memory paid a
    // memory latency per element: 76 us at n2 = 1024), then ONE thread adds them in index order
    __shared__ double ps[1024];
    double a = 0.0;
    for (int base = 0; base < n2; base += 1024) {
        const int i = base + threadIdx.x;
        __syncthreads();
        ps[threadIdx.x] = i < n2 ? part[i] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = min(1024, n2 - base);
            for (int q = 0; q < cnt; ++q) a += ps[q];
        }
    }
    if (threadIdx.x == 0) *mean_out = a / (double)n;
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
    int *heavy;            // peaks whose walks are too long for one thread: finished by k_pick_heavy, one wave each
    int *heavy_count;
    int heavy_cap;
};
constexpr int kWalkBudget = 16;   // sequential steps a k_pick thread may spend on one peak before handing it over

// min over the stretch from p outward (dir = -1 / +1) up to, not including, the first sample > h
__device__ double walk_min(const PickArgs &a, int p, double h, int dir, int &budget)
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
    if (--budget < 0) return m;                                  // (the caller discards the value)
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
                if (--budget < 0) return m;
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
        if (--budget < 0) return m;
    }
    return m;
}

// ---- one wavefront per peak: the same quantities as walk_min / the width walks of k_pick, 64 positions per step
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

// min over the samples of level-1 block b1 from its near end (seen from the peak) up to, not including, the first
// sample > h; *stop = such a sample exists.  `from`: first sample to look at (inside the block).
__device__ __forceinline__ double wave_block_samples(const PickArgs &a, int from, int b1, double h, int dir, bool *stop)
{
    const int lane = threadIdx.x & 63;
    const int lo = b1 * B1, hi = min(lo + B1 - 1, a.n - 1);
    const int i = from + dir * lane;
    const bool in = lane < B1 && i >= lo && i <= hi;
    const double v = in ? (double)a.x[i] : h;
    const unsigned long long above = __ballot(in && v > h);
    const int first = above ? __ffsll((long long)above) - 1 : 64;
    *stop = above != 0;
    return wave_min(in && lane < first ? v : h);
}

__device__ double wave_walk_min(const PickArgs &a, int p, double h, int dir)
{
    const int lane = threadIdx.x & 63;
    const int last = a.n - 1;
    double m = h;
    int i = p + dir;
    if (i < 0 || i > last) return m;
    bool stop;
    // the rest of the peak's own level-1 block
    m = fmin(m, wave_block_samples(a, i, i / B1, h, dir, &stop));
    if (stop) return m;
    int b1 = i / B1 + dir;
    constexpr int R21 = B2 / B1;
    while (b1 >= 0 && b1 < a.n1) {
        const bool at_l2 = dir < 0 ? (b1 % R21 == R21 - 1) : (b1 % R21 == 0);
        if (at_l2) {
            // whole level-2 blocks, 64 per step
            int b2 = b1 / R21;
            bool found = false;
            while (b2 >= 0 && b2 < a.n2) {
                const int bb = b2 + dir * lane;
                const bool in = bb >= 0 && bb < a.n2;
                const bool hit = in && (double)a.max2[bb] > h;
                const unsigned long long hits = __ballot(hit);
                const int first = hits ? __ffsll((long long)hits) - 1 : 64;
                m = fmin(m, wave_min(in && lane < first ? (double)a.min2[bb] : h));
                if (hits) { b2 += dir * first; found = true; break; }
                b2 += dir * 64;
            }
            if (!found) return m;
            b1 = dir < 0 ? b2 * R21 + R21 - 1 : b2 * R21;
        }
        // level-1 blocks up to the next level-2 boundary (at most 32), one per lane
        const int left_in_l2 = dir < 0 ? (b1 % R21) + 1 : R21 - (b1 % R21);
        const int bb = b1 + dir * lane;
        const bool in = lane < left_in_l2 && bb >= 0 && bb < a.n1;
        const bool hit = in && (double)a.max1[bb] > h;
        const unsigned long long hits = __ballot(hit);
        const int first = hits ? __ffsll((long long)hits) - 1 : 64;
        m = fmin(m, wave_min(in && lane < first ? (double)a.min1[bb] : h));
        if (hits) {
            const int fb = b1 + dir * first;
            int from = dir < 0 ? fb * B1 + B1 - 1 : fb * B1;
            if (from > last) from = last;
            m = fmin(m, wave_block_samples(a, from, fb, h, dir, &stop));
            return m;                                             // the block holds a sample > h: the walk ends in it
        }
        b1 += dir * left_in_l2;
    }
    return m;
}

// first distance d >= 0 from pk (direction dir) at which !(level < x[pk + dir d]) or the array's end is reached
// (k_pick's width loops), capped: returns bound + 1 when every d <= bound is above the level
__device__ int wave_width_stop(const PickArgs &a, int pk, double level, int dir, int bound)
{
    const int lane = threadIdx.x & 63;
    const int last = a.n - 1;
    for (int d0 = 0; d0 <= bound; d0 += 64) {
        const int d = d0 + lane;
        const int j = pk + dir * d;
        const bool in = j >= 0 && j <= last;
        const bool end = in && (j == (dir < 0 ? 0 : last));
        const bool st = in && (end || !(level < (double)a.x[j]));
        const unsigned long long hits = __ballot(st);
        if (hits) return d0 + __ffsll((long long)hits) - 1;
    }
    return bound + 1;
}

// the decision both kernels share once a peak's two base minima are known
__device__ __forceinline__ void pick_finish(const PickArgs &a, int pk, double top, double left, double right)
{
    const double width = right - left;
    if (!(a.min_w <= width && width <= a.max_w)) return;
    if (!(top > *a.mean * 2)) return;
    const int slot = atomicAdd(a.count, 1);
    if (slot < a.cap) a.out[slot] = pk;
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
    // Nearly every local maximum of a noisy spectrum is settled within a few samples.  The few that are not -- a
    // carrier's summit walks hundreds of samples for its width and up to N / 1024 table entries for its bases, every
    // step a dependent load: 246 us for ONE thread at N = 2^20 -- are handed to k_pick_heavy, one wavefront each.
    // (second attempt, without a budget: only when the hand-over list is full -- nothing is ever dropped)
    for (int attempt = 0; attempt < 2; ++attempt) {
        int budget = attempt ? 0x7fffffff : kWalkBudget;
        const double lmin = walk_min(a, pk, top, -1, budget);
        const double rmin = budget >= 0 ? walk_min(a, pk, top, +1, budget) : 0.0;
        bool heavy = budget < 0;
        double left = 0.0, right = 0.0;
        if (!heavy) {
            const double prom = top - (lmin > rmin ? lmin : rmin);
            if (!(prom >= a.prominence)) return;
            const double level = top - prom * 0.5;
            // width at half prominence; walks are bounded: a side longer than max_w already fails the window
            const int bound = (int)fmin(a.max_w + 2.0, (double)a.n);
            int j = pk;
            while (j > 0 && level < (double)x[j]) {
                --j;
                if (pk - j > bound) return;
                if (--budget < 0) { heavy = true; break; }
            }
            if (!heavy) {
                left = (double)j;
                if ((double)x[j] < level) left += (level - (double)x[j]) / ((double)x[j + 1] - (double)x[j]);
                j = pk;
                while (j < last && level < (double)x[j]) {
                    ++j;
                    if (j - pk > bound) return;
                    if (--budget < 0) { heavy = true; break; }
                }
                if (!heavy) {
                    right = (double)j;
                    if ((double)x[j] < level) right -= (level - (double)x[j]) / ((double)x[j - 1] - (double)x[j]);
                }
            }
        }
        if (heavy) {
            const int slot = atomicAdd(a.heavy_count, 1);
            if (slot < a.heavy_cap) { a.heavy[slot] = pk; return; }
            continue;                                            // list full: finish it here, sequentially
        }
        pick_finish(a, pk, top, left, right);
        return;
    }
}

// one wavefront per handed-over peak; identical comparisons and float64 arithmetic, 64 positions per step
__global__ __launch_bounds__(64) void k_pick_heavy(PickArgs a)
{
    const int n_heavy = min(*a.heavy_count, a.heavy_cap);
    for (int it = blockIdx.x; it < n_heavy; it += gridDim.x) {
        const int pk = a.heavy[it];
        const float *x = a.x;
        const double top = (double)x[pk];
        const double lmin = wave_walk_min(a, pk, top, -1);
        const double rmin = wave_walk_min(a, pk, top, +1);
        const double prom = top - (lmin > rmin ? lmin : rmin);      // (every lane holds the same values: the
        if (!(prom >= a.prominence)) continue;                      //  branches below are wave-uniform)
        const double level = top - prom * 0.5;
        const int bound = (int)fmin(a.max_w + 2.0, (double)a.n);
        const int dl = wave_width_stop(a, pk, level, -1, bound);
        if (dl > bound) continue;
        const int dr = wave_width_stop(a, pk, level, +1, bound);
        if (dr > bound) continue;
        if ((threadIdx.x & 63) == 0) {
            int j = pk - dl;
            double left = (double)j;
            if ((double)x[j] < level) left += (level - (double)x[j]) / ((double)x[j + 1] - (double)x[j]);
            j = pk + dr;
            double right = (double)j;
            if ((double)x[j] < level) right -= (level - (double)x[j]) / ((double)x[j - 1] - (double)x[j]);
            pick_finish(a, pk, top, left, right);
        }
    }
}

// ascending sort of min(count, cap) survivors (cap <= 4096), padded with -1 after the valid entries
__global__ __launch_bounds__(1024) void k_sort(int64_t *buf, const int *count, int cap)
{
    __shared__ int64_t s[4096];
    const int n = min(*count, cap);
    int m = 2;                                             // sort only the power of two that holds the survivors:
    while (m < n) m <<= 1;                                 // a dozen peaks are 10 compare stages, not 78
    for (int i = threadIdx.x; i < 4096; i += 1024) s[i] = i < n ? buf[i] : INT64_MAX;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += 1024) {
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
    return sizeof(float) * ((size_t)n + 2 * n1 + 2 * n2 + kMinBlocks + 16) + sizeof(double) * (n2 + 2) + 64 +
           sizeof(int) * (kHeavyCap + 16);
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
    int *count = reinterpret_cast<int *>(p);       p += sizeof(int) * 4;
    int *heavy_count = count + 1;
    int *heavy = reinterpret_cast<int *>(p);
    (void)hipMemsetAsync(count, 0, 2 * sizeof(int), s);
    hipLaunchKernelGGL(k_min, dim3(kMinBlocks), dim3(256), 0, s, d_spec, n, shift);
    hipLaunchKernelGGL(k_prep, dim3(n2), dim3(256), 0, s, d_spec, n, shift, x, max1, min1, part);
    hipLaunchKernelGGL(k_tables, dim3(1), dim3(1024), 0, s, max1, min1, n1, max2, min2, n2, part, n, mean);
    int heavy_cap = kHeavyCap;
    if (const char *e = getenv("RCF_PEAKS_HEAVY_CAP")) heavy_cap = std::max(0, std::min(kHeavyCap, atoi(e)));   // tests: overflow path
    PickArgs a{x, max1, min1, max2, min2, n, n1, n2, min_w, max_w, prominence, mean, d_out, count, cap,
               heavy, heavy_count, heavy_cap};
    hipLaunchKernelGGL(k_pick, dim3((n + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_pick_heavy, dim3(1024), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_sort, dim3(1), dim3(1024), 0, s, d_out, count, cap);
    *d_count_out = count;
    *d_mean_out = mean;
}

}  // namespace rcfx
