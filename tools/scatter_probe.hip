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


template <int SEG, int LDSPAD, int TILED = 0>
__global__ __launch_bounds__(256) void copy_scatter_buf(const float2* __restrict__ in, float2* __restrict__ out,
                                                       int n_frames, int fpw, long pitch, long in_len) {
    extern __shared__ float2 dbuf[];
    constexpr int RS = 256 + LDSPAD;
    const int tid = threadIdx.x;
    const long f0 = (long)blockIdx.x * fpw;
    __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)(in_len * 8), 0x00020000);
    __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)(256 * pitch * 8), 0x00020000);
    for (int ch = 0; ch < fpw; ch += SEG) {
        const int vo = (int)(((f0 + ch + 1) * 256 - tid) * 8);
        u32x2 r[SEG];
#pragma unroll
        for (int f = 0; f < SEG; ++f) r[f] = __builtin_amdgcn_raw_buffer_load_b64(ir, vo, f * 2048, 0);
#pragma unroll
        for (int f = 0; f < SEG; ++f) dbuf[f * RS + tid + (LDSPAD ? (tid >> 4) : 0)] = make_float2(__uint_as_float(r[f].x), __uint_as_float(r[f].y));
        __syncthreads();
        const int fl = tid % SEG, k0 = tid / SEG;
        // TILED: [chunk][bin][SEG frames] -- one contiguous 256 * SEG * 8 bytes per chunk
        const int so = TILED ? (256 / SEG) * SEG * 8 : (int)((long)(256 / SEG) * pitch * 8);
        const int vo2 = TILED ? (int)((((f0 + ch) / SEG) * 256 + k0) * SEG * 8 + fl * 8) : (int)(((long)k0 * pitch + (f0 + ch + fl)) * 8);
#pragma unroll
        for (int i = 0; i < SEG; ++i) {
            const int k = k0 + i * (256 / SEG);
            float2 v = dbuf[fl * RS + k + (LDSPAD ? (k >> 4) : 0)];
            u32x2 o; o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y);
            __builtin_amdgcn_raw_buffer_store_b64(o, orr, vo2, i * so, 2);
        }
        __syncthreads();
    }
}
template <int SEG, int LDSPAD>
float run_buf(const float2* in, float2* out, int n_frames, int fpw, long pitch, long in_len, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 g(n_frames / fpw);
    size_t lds = (size_t)SEG * (256 + LDSPAD) * 8 + 2048;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((copy_scatter_buf<SEG, LDSPAD>), g, dim3(256), lds, 0, in, out, n_frames, fpw, pitch, in_len);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_scatter_buf<SEG, LDSPAD>), g, dim3(256), lds, 0, in, out, n_frames, fpw, pitch, in_len);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

template <int SEG, int LDSPAD, int TILED = 0>
float run_buf_rot(float2** ins, float2** outs, int nset, int n_frames, int fpw, long pitch, long in_len, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 g(n_frames / fpw);
    size_t lds = (size_t)SEG * (256 + LDSPAD) * 8 + 2048;
    for (int i = 0; i < nset; ++i) hipLaunchKernelGGL((copy_scatter_buf<SEG, LDSPAD, TILED>), g, dim3(256), lds, 0, ins[i], outs[i], n_frames, fpw, pitch, in_len);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_scatter_buf<SEG, LDSPAD, TILED>), g, dim3(256), lds, 0, ins[i % nset], outs[i % nset], n_frames, fpw, pitch, in_len);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
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
    if (getenv("RANDOM_INPUT")) {
        float *hbuf = (float *)malloc(n * 8);
        unsigned st = 12345u;
        for (size_t i = 0; i < 2 * n; ++i) { st = st * 1664525u + 1013904223u; hbuf[i] = (float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f; }
        CK(hipMemcpy(in, hbuf, n * 8, hipMemcpyHostToDevice));
        free(hbuf);
        printf("random input\n");
    } else CK(hipMemset(in, 1, n * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(copy_lin, dim3(4096), dim3(256), 0, 0, in, out, n);
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_lin, dim3(4096), dim3(256), 0, 0, in, out, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("linear copy 8B/lane          : %.4f ms  %.0f GB/s\n", ms, 16.0 * n / ms / 1e6);
    {
        float t; const long pitch = cap + 80;
        for (int fpw : {16, 32, 64, 128}) {
            t = run_scatter<16, 1, 1>(in, out, n_frames, fpw, pitch, 10);
            printf("scatter seg=16 nt x[mD - tid] frames/WG=%d : %.4f ms  %.0f GB/s\n", fpw, t, 16.0 * n / t / 1e6);
        }
    }
    {
        float t; const long pitch = cap + 80;
        t = run_buf<16, 0>(in, out, n_frames, 16, pitch, (long)n + 512, 10); printf("buffer ops, LDS stride 256, fpw16   : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        t = run_buf<16, 18>(in, out, n_frames, 16, pitch, (long)n + 512, 10); printf("buffer ops, LDS stride 274 padded     : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        t = run_buf<16, 18>(in, out, n_frames, 32, pitch, (long)n + 512, 10); printf("buffer ops, LDS stride 274, fpw32     : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
    }
    {
        const int NSET = 4; float2 *ins[NSET], *outs[NSET]; const long pitch = cap + 80;
        for (int i = 0; i < NSET; ++i) { CK(hipMalloc(&ins[i], n * 8 + 4096)); CK(hipMemset(ins[i], i + 1, n * 8)); CK(hipMalloc(&outs[i], (size_t)256 * (cap + 1040) * 8)); }
        float t = run_buf_rot<16, 18>(ins, outs, NSET, n_frames, 16, pitch, (long)n + 512, 12);
        printf("buffer ops, padded LDS, fpw16, rotating over %d x (256 MiB in, 512 MiB out) : %.4f ms  %.0f GB/s\n", NSET, t, 16.0 * n / t / 1e6);
        t = run_buf_rot<16, 18>(ins, outs, NSET, n_frames, 32, pitch, (long)n + 512, 12);
        printf("same, fpw32 : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        t = run_buf_rot<16, 18, 1>(ins, outs, NSET, n_frames, 16, pitch, (long)n + 512, 12);
        printf("TILED output [chunk][bin][16], fpw16, rotating : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        t = run_buf_rot<16, 18, 1>(ins, outs, NSET, n_frames, 32, pitch, (long)n + 512, 12);
        printf("TILED, fpw32, rotating : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
        CK(hipEventRecord(a));
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(copy_lin, dim3(4096), dim3(256), 0, 0, ins[i % NSET], outs[i % NSET], n);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&t, a, b)); t /= 12;
        printf("linear copy, rotating : %.4f ms  %.0f GB/s\n", t, 16.0 * n / t / 1e6);
    }
    return 0;
}
