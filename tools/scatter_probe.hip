// scatter_probe.hip -- what HBM rate does the PFB's channel-major store pattern allow on MI355X?
// Reads a 2^25-sample cf32 stream once (coalesced) and writes it back either linearly or as NB rings
// (bin-major) in SEG-frame pieces per bin, exactly like the PFB epilogue.  No arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void copy_lin(const float2* __restrict__ in, float2* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) out[i] = in[i];
}

// one workgroup (256 thr) handles FPW frames of 256 samples; per SEG-frame chunk: thread reads SEG samples
// (its column), stages through LDS, writes bin-major: lane -> (frame f = tid % SEG, bin k0 = tid / SEG)
template <int SEG, int NT, int REV>
__global__ __launch_bounds__(256) void copy_scatter(const float2* __restrict__ in, float2* __restrict__ out,
                                                   int n_frames, int fpw, long pitch) {
    __shared__ float2 buf[SEG * 258];
    const int tid = threadIdx.x;
    const long f0 = (long)blockIdx.x * fpw;
    const int kper = 256 / SEG;            // bins covered by one store instruction's 256 threads
    for (int ch = 0; ch < fpw; ch += SEG) {
        for (int f = 0; f < SEG; ++f) {
            long m = f0 + ch + f + 1;
            long idx = REV == 0 ? (m - 1) * 256 + tid : (REV == 1 ? m * 256 - tid : (tid == 0 ? m * 256 : m * 256 - 256 + tid));
            buf[f * 258 + tid] = in[idx];
        }
        __syncthreads();
        const int fl = tid % SEG, k0 = tid / SEG;
        for (int i = 0; i < SEG; ++i) {
            const int k = k0 + i * kper;
            float2 v = buf[fl * 258 + k];
            float2* dst = out + (long)k * pitch + (f0 + ch + fl);
            typedef float v2 __attribute__((ext_vector_type(2))); v2 vv; vv.x = v.x; vv.y = v.y;
            if (NT) __builtin_nontemporal_store(vv, (v2*)dst); else *dst = v;
        }
        __syncthreads();
    }
}

template <int SEG, int NT, int REV>
float run_scatter(const float2* in, float2* out, int n_frames, int fpw, long pitch, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 g(n_frames / fpw);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((copy_scatter<SEG, NT, REV>), g, dim3(256), 0, 0, in, out, n_frames, fpw, pitch);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_scatter<SEG, NT, REV>), g, dim3(256), 0, 0, in, out, n_frames, fpw, pitch);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    const size_t n = 1ull << 25; const int n_frames = n / 256; const long cap = 1 << 18;
    float2 *in, *out;
    CK(hipMalloc(&in, n * 8 + 4096)); CK(hipMalloc(&out, (size_t)256 * (cap + 1040) * 8));
    CK(hipMemset(in, 1, n * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(copy_lin, dim3(4096), dim3(256), 0, 0, in, out, n);
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_lin, dim3(4096), dim3(256), 0, 0, in, out, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("linear copy 8B/lane          : %.4f ms  %.0f GB/s\n", ms, 16.0 * n / ms / 1e6);
    {
        float t; const long pitch = cap + 80;
        t = run_scatter<16, 1, 0>(in, out, n_frames, 32, pitch, 10); printf("scatter seg=16 nt aligned ascending   : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        t = run_scatter<16, 1, 1>(in, out, n_frames, 32, pitch, 10); printf("scatter seg=16 nt x[mD - tid] (PFB)   : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        t = run_scatter<16, 1, 2>(in, out, n_frames, 32, pitch, 10); printf("scatter seg=16 nt ascending + stray   : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        t = run_scatter<16, 0, 0>(in, out, n_frames, 32, pitch, 10); printf("scatter seg=16 plain stores aligned   : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
    }
    return 0;
}
