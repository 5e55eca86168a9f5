// xcc_probe.hip -- which XCD does block b of a one-dimensional grid run on?  Every block reports the XCC_ID register.
//   hipcc -O2 --offload-arch=gfx950 -o tools/xcc_probe tools/xcc_probe.hip && gpurun -- tools/xcc_probe
// Measured on MI355X (SPX): block b runs on XCD b mod 8, exactly, also with the banks' 53 KB of LDS per workgroup -- what the
// filterbanks' chunk maps assume for locality and what pfb5_xcd_map_ok (pfb5.hip) verifies before the look-back hand-over
// is allowed to go through one XCD's L2.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
}
int main() {
    int n = 4096, *d, h[4096];
    hipMalloc(&d, n * sizeof(int));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(320), 53000, 0, d);
        hipMemcpy(h, d, n * sizeof(int), hipMemcpyDeviceToHost);
        int bad = 0; for (int b = 0; b < n; ++b) if (h[b] != h[b % 8]) ++bad;
        printf("rep %d first 16:", rep); for (int b = 0; b < 16; ++b) printf(" %d", h[b]); printf("  mismatches vs b%%8 rule: %d of %d\n", bad, n);
    }
    return 0;
}
