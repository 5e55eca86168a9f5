// What the host link gives kernels that read and write pinned host memory in place (the real-time leg's conversion
// reads its input blocks that way, its gather writes the channels' host rings that way): read alone, write alone, both
// at once on two streams.  Bounded grids (128 workgroups), 16-byte accesses, 256 MB per direction and pass.
//   hipcc -O3 --offload-arch=gfx950 -o tools/pcie_duplex_probe tools/pcie_duplex_probe.hip && tools/pcie_duplex_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void rd(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        dst[i] = v;                       // device memory: the conversion writes what it read
        acc.x ^= v.x;
    }
    if (acc.x == 0x12345678u) dst[0] = acc;
}
__global__ void wr(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main()
{
    const size_t bytes = (size_t)256 << 20, n = bytes / 16;
    void *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_a, bytes));
    CK(hipMalloc(&d_b, bytes));
    CK(hipMemset(d_b, 1, bytes));
    for (size_t i = 0; i < bytes / 8; ++i) ((unsigned long long *)h_in)[i] = i * 0x9e3779b97f4a7c15ull;
    void *hd_in, *hd_out;
    CK(hipHostGetDevicePointer(&hd_in, h_in, 0));
    CK(hipHostGetDevicePointer(&hd_out, h_out, 0));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto run = [&](bool r, bool w, int grid) {
        const int reps = 8;
        for (int warm = 0; warm < 2; ++warm) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) {
                if (r) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, s1, (const uint4 *)hd_in, (uint4 *)d_a, n);
                if (w) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, s2, (uint4 *)hd_out, (const uint4 *)d_b, n);
            }
            CK(hipDeviceSynchronize());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (warm) printf("{\"grid\": %d, \"read\": %s, \"write\": %s, \"GBps_each_direction\": %.1f, \"GBps_total\": %.1f}\n", grid,
                             r ? "true" : "false", w ? "true" : "false", reps * bytes / dt / 1e9, (r + w) * reps * bytes / dt / 1e9);
        }
    };
    for (int grid : {128, 512}) { run(true, false, grid); run(false, true, grid); run(true, true, grid); }
    return 0;
}
