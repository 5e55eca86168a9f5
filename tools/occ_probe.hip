// How many workgroups of a given dynamic-LDS request are REALLY co-resident on a CU (the runtime's occupancy query
// divides 160 KB by the request; the hardware allocates LDS in granules).  Each workgroup spins for a fixed wall time;
// a grid of CUs x n workgroups takes ~1 spin if n fit per CU, ~2 spins if not.
// build: hipcc --offload-arch=gfx950 -O2 tools/occ_probe.hip -o tools/occ_probe ; run: tools/occ_probe [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern "C" __global__ void k(float *o, long long ticks)
{
    extern __shared__ float sm[];
    sm[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    o[threadIdx.x] = sm[(threadIdx.x * 7) % blockDim.x];
}
int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 320;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float *d;
    (void)hipMalloc(&d, 4096 * 4);
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long long ticks = 20000;            // 100 MHz clock: 200 us
    const size_t sizes[] = {32768, 32784, 33280, 34000, 40448, 40960, 40976, 41472, 42240, 53248, 53760, 53776, 53792, 54272, 54608,
                            69888, 73984, 80640, 81920, 81936};
    for (size_t lds : sizes) {
        printf("lds %6zu:", lds);
        for (int n = 1; n <= 5; ++n) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(cus * n), dim3(threads), lds, 0, d, ticks);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  n=%d %.2f", n, ms / 0.2f);
        }
        printf("\n");
    }
    return 0;
}
