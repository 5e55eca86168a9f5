// mfma_peak_probe.hip -- what the FP32 matrix instructions sustain on this chip with NON-ZERO operands (power-managed
// clock), bare and with the operand traffic of the FIR bank's inner loop (4 ds_read_b128 + 2 16-byte global loads per
// 1024 matrix-pipe cycles), for 1 / 2 / 4 waves per SIMD.  The calibration DESIGN.md quotes next to the 157.3 TF
// datasheet number.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o tools/mfma_peak_probe tools/mfma_peak_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s\n", (int)e_, #x); return 1; } } while (0)

// MODE 0: bare MFMA 16x16x4, operands in registers.  MODE 1: 16x16x4 with per-step operand loads (4 LDS + 2 global).
// MODE 2: bare 32x32x2.  MODE 3: 32x32x2 with per-step loads (4 LDS + 2 global per 16 MFMAs = 1024 cycles).
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters)
{
    __shared__ v4f lds[2048];                  // 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = (v4f){in[(i * 4) & 65535], in[(i * 4 + 1) & 65535], in[(i * 4 + 2) & 65535], in[(i * 4 + 3) & 65535]};
    __syncthreads();
    const v4f *g = reinterpret_cast<const v4f *>(in);
    float s0 = 0.f;
    if (MODE == 0 || MODE == 1) {
        v4f acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
        v4f a[4], b[2];
        for (int t = 0; t < 4; ++t) a[t] = lds[t * 64 + lane];
        for (int n = 0; n < 2; ++n) b[n] = g[(n * 64 + lane) & 16383];
        for (int it = 0; it < iters; ++it) {
            v4f an[4], bn[2];
            if (MODE == 1) {
                const int o = (it & 7) * 256;
                for (int t = 0; t < 4; ++t) an[t] = lds[o + t * 64 + lane];
                for (int n = 0; n < 2; ++n) bn[n] = g[((it & 63) * 128 + n * 64 + lane) & 16383];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[n * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u], b[n][u], acc[n * 4 + t], 0, 0, 0);
            if (MODE == 1) {
                __builtin_amdgcn_sched_barrier(0);
                for (int t = 0; t < 4; ++t) a[t] = an[t];
                for (int n = 0; n < 2; ++n) b[n] = (it & 1) ? bn[n] : -bn[n];
            } else {
                for (int t = 0; t < 4; ++t) a[t] = -a[t];
            }
        }
        v4f s = acc[0];
        for (int i = 1; i < 8; ++i) s += acc[i];
        s0 = s[0] + s[1] + s[2] + s[3];
    } else {
        v16f acc[2];
        for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        v4f a[4], b[2];      // two M-tiles x two steps of A, two steps of B
        for (int t = 0; t < 4; ++t) a[t] = lds[t * 64 + lane];
        for (int n = 0; n < 2; ++n) b[n] = g[(n * 64 + lane) & 16383];
        for (int it = 0; it < iters; ++it) {
            v4f an[4], bn[2];
            if (MODE == 3) {
                const int o = (it & 7) * 256;
                for (int t = 0; t < 4; ++t) an[t] = lds[o + t * 64 + lane];
                for (int n = 0; n < 2; ++n) bn[n] = g[((it & 63) * 128 + n * 64 + lane) & 16383];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st * 2 + t][u], b[st][u], acc[t], 0, 0, 0);
            if (MODE == 3) {
                __builtin_amdgcn_sched_barrier(0);
                for (int t = 0; t < 4; ++t) a[t] = an[t];
                for (int n = 0; n < 2; ++n) b[n] = (it & 1) ? bn[n] : -bn[n];
            } else {
                for (int t = 0; t < 4; ++t) a[t] = -a[t];
            }
        }
        for (int e = 0; e < 16; ++e) s0 += acc[0][e] + acc[1][e];
    }
    out[threadIdx.x + blockIdx.x * 256] = s0;
}

template <int MODE>
int run(const float *in, float *out, const char *name)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 10000;
    for (int blocks : {256, 512, 1024}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        // 1024 matrix-pipe cycles per iteration per wave: 32 x (16x16x4) or 16 x (32x32x2) = 65536 flop
        const double flop = (double)blocks * 4 * iters * 65536.0;
        printf("%-34s %d waves/SIMD: %8.3f ms  %6.1f TFLOP/s\n", name, blocks / 256, ms, flop / ms / 1e9);
    }
    return 0;
}

int main()
{
    float *in, *out;
    std::vector<float> h(65536);
    srand(1);
    for (auto &v : h) v = (float)(rand() % 20001) / 20000.0f - 0.5f;
    CK(hipMalloc(&in, 65536 * 4));
    CK(hipMalloc(&out, 256 * 4096 * 4));
    CK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
    if (run<0>(in, out, "16x16x4 bare")) return 1;
    if (run<1>(in, out, "16x16x4 + 4 LDS + 2 global / 32")) return 1;
    if (run<2>(in, out, "32x32x2 bare")) return 1;
    if (run<3>(in, out, "32x32x2 + 4 LDS + 2 global / 16")) return 1;
    return 0;
}
