// What the memory system gives a kernel that reads a row-major matrix in column blocks -- tap_finalize_kernel's read side:
// a workgroup takes ROWS consecutive rows x W bytes of a [n_rows][PITCH bytes] matrix (requesting all of its pieces before
// it uses any) and writes them as one contiguous run.  W = 128 is the kernel's shape (16 taps x 8 bytes).  Neighbouring
// workgroups take neighbouring column blocks of the same rows, as the kernel's grid does.
//   hipcc -O3 --offload-arch=gfx950 -o tools/strided_read_probe tools/strided_read_probe.hip && tools/strided_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W, int ROWS>      // W bytes per piece, ROWS rows per workgroup; 256 threads, 8 bytes per lane and load
__global__ __launch_bounds__(256) void rd(const float2 *__restrict__ src, float2 *__restrict__ dst, int n_rows, int pitch_e, int n_cb)
{
    constexpr int LPR = W / 8;                    // lanes per row piece
    constexpr int RPL = 256 / LPR;                // rows covered by one load of the workgroup
    constexpr int NIT = ROWS / RPL;
    const int cb = blockIdx.x % n_cb, rb = blockIdx.x / n_cb;
    const int lane_c = threadIdx.x % LPR, lane_r = threadIdx.x / LPR;
    const int r0 = rb * ROWS;
    float2 z[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = r0 + lane_r + it * RPL;
        z[it] = r < n_rows ? src[(size_t)r * pitch_e + cb * LPR + lane_c] : make_float2(0.f, 0.f);
    }
    float2 *out = dst + ((size_t)blockIdx.x * ROWS) * LPR;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
        __builtin_nontemporal_store(z[it].x, &out[(size_t)(lane_r + it * RPL) * LPR + lane_c].x),
        __builtin_nontemporal_store(z[it].y, &out[(size_t)(lane_r + it * RPL) * LPR + lane_c].y);
}

template <int W, int ROWS>
void run(const float2 *src, float2 *dst, int n_rows, int pitch_e)
{
    const int n_cb = pitch_e * 8 / W, n_rb = (n_rows + ROWS - 1) / ROWS;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((rd<W, ROWS>), dim3(n_cb * n_rb), dim3(256), 0, 0, src, dst, n_rows, pitch_e, n_cb);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep >= 2 && ms / 10 < best) best = ms / 10;
    }
    const double bytes = (double)n_rows * pitch_e * 8;
    printf("{\"piece_bytes\": %d, \"rows_per_workgroup\": %d, \"ms\": %.4f, \"read_GBps\": %.0f, \"read_plus_write_GBps\": %.0f}\n", W, ROWS, best,
           bytes / (best * 1e-3) / 1e9, 2 * bytes / (best * 1e-3) / 1e9);
}

int main()
{
    const int n_rows = 41943, pitch_e = 1600;      // the 1600-bin bank's frames of a 2^25-sample block
    float2 *src, *dst;
    CK(hipMalloc(&src, (size_t)n_rows * pitch_e * 8));
    CK(hipMalloc(&dst, ((size_t)n_rows + 256) * pitch_e * 8));
    CK(hipMemset(src, 1, (size_t)n_rows * pitch_e * 8));
    run<128, 128>(src, dst, n_rows, pitch_e);
    run<128, 256>(src, dst, n_rows, pitch_e);
    run<256, 128>(src, dst, n_rows, pitch_e);
    run<512, 128>(src, dst, n_rows, pitch_e);
    run<1024, 128>(src, dst, n_rows, pitch_e);
    run<128, 32>(src, dst, n_rows, pitch_e);
    run<256, 32>(src, dst, n_rows, pitch_e);
    return 0;
}
