// stage2_probe.hip -- where do the 19 us of the stage-2 launch (fir_small_kernel: 32 bins of a 256-bin tiled ring ->
// xlating FIR /3 + discriminator) go?  The same grid (32 channels x 86 tiles of 511 outputs, 256 threads, 17.5 KB LDS)
// and the same addresses -- one 128-byte line per 16-frame tile, tiles 16 NB + 80 samples apart -- with the work
// peeled off: MODE 0 nothing, 1 the sample loads only, 2 loads + the two output streams, 3 the same from a LINEAR
// source (what the loads would cost if a bin's stream were contiguous).
//   hipcc -O3 --offload-arch=gfx950 -o tools/stage2_probe tools/stage2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NB = 256, KB = 511, D = 3, T = 11;
constexpr long PITCH = 16 * NB + 80;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float2 *__restrict__ ring, float2 *__restrict__ iq, float *__restrict__ fm,
                                         const int *__restrict__ bins, int n_k, long ring_frames, long out_cap)
{
    extern __shared__ float2 xs[];
    const int tid = threadIdx.x, c = blockIdx.x, j0 = blockIdx.y * KB;
    if (MODE == 0) { if (tid == 1000) iq[0] = make_float2(0.f, 0.f); return; }
    if (j0 >= n_k) return;
    const int nj = min(KB, n_k - j0);
    const int bin = bins[c];
    const long s_first = (long)(j0 + 1) * D + 16;            // (any offset: a bin's frame index)
    const int len = nj * D + T;
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = tid + u * 256;
        const long s = s_first + (p < len ? p : len - 1);
        const long at = MODE == 3 ? (long)c * ring_frames + s : (s >> 4) * PITCH + 16 * bin + (s & 15);
        v[u] = ring[at];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = tid + u * 256;
        if (p < len) xs[p] = v[u];
    }
    __syncthreads();
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int j = tid + o * 256;
        if (j >= 1 && j <= nj) {
            float2 a = xs[j * D + T - 1];
            if (MODE >= 2) {
                iq[(long)c * out_cap + j0 + j] = a;
                fm[(long)c * out_cap + j0 + j] = a.x;
            } else {
                acc.x += a.x;
                acc.y += a.y;
            }
        }
    }
    if (MODE == 1 && acc.x == 12345.678f) iq[0] = acc;
}

template <int MODE>
float run(float2 *const *rings, int nset, float2 *iq, float *fm, const int *bins, int n_k, long ring_frames, long out_cap)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const dim3 grid(32, (n_k + KB - 1) / KB);
    const size_t lds = (size_t)(KB * D + 2 * T + KB + 1) * 8 + 264 * 4;
    for (int i = 0; i < 24; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), lds, 0, rings[i % nset], iq, fm, bins, n_k, ring_frames, out_cap);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 48; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), lds, 0, rings[i % nset], iq, fm, bins, n_k, ring_frames, out_cap);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 48 * 1e3f;
}

int main()
{
    const long frames = 1 << 17, n_k = frames / D, out_cap = 1 << 16;
    const size_t ring_samples = (size_t)(frames / 16 + 2) * PITCH;
    // twelve rings in turn: the 33.5 MB a launch touches must not come back out of the 256 MB Infinity Cache
    constexpr int NSET = 12;
    float2 *rings[NSET], *iq;
    float *fm;
    int *bins, hb[32];
    for (int i = 0; i < NSET; ++i) {
        CK(hipMalloc(&rings[i], ring_samples * 8));
        CK(hipMemset(rings[i], 1, ring_samples * 8));
    }
    CK(hipMalloc(&iq, 32 * out_cap * 8));
    CK(hipMalloc(&fm, 32 * out_cap * 4));
    CK(hipMalloc(&bins, 32 * 4));
    unsigned st = 7;
    for (int i = 0; i < 32; ++i) { st = st * 1664525u + 1013904223u; hb[i] = (i * 8 + (st >> 28)) % NB; }
    CK(hipMemcpy(bins, hb, sizeof(hb), hipMemcpyHostToDevice));
    printf("empty grid (2752 workgroups, 17.5 KB LDS)      : %.1f us\n", run<0>(rings, NSET, iq, fm, bins, (int)n_k, frames, out_cap));
    printf("sample loads only (33.5 MB in 128-byte lines)   : %.1f us\n", run<1>(rings, NSET, iq, fm, bins, (int)n_k, frames, out_cap));
    printf("loads + IQ and FM streams (16.8 MB written)     : %.1f us\n", run<2>(rings, NSET, iq, fm, bins, (int)n_k, frames, out_cap));
    printf("the same from a linear per-channel source       : %.1f us\n", run<3>(rings, NSET, iq, fm, bins, (int)n_k, frames, out_cap));
    return 0;
}
