// hbm_mix_probe.hip -- what does the MI355X's memory system sustain for the read : write mixes the filterbank kernels
// have?  No arithmetic, fully coalesced 16-byte accesses, working sets far beyond the 256 MB Infinity Cache:
//   1 : 0  read only (sum kept in a register, one store per workgroup)
//   1 : 1  the 256 / 512 / 1024-bin banks (8 B read + 8 B written per input sample, critically sampled)
//   1 : 2  the 1600-bin reference-grid bank (OS = 2: 8 B read + 16 B written)
//   1 : 4  the 3200-bin bank at D = 800 (OS = 4: 8 B read + 32 B written)
//   0 : 1  write only
// Each mix is run with plain and with non-temporal stores, 256-thread workgroups, U independent 16-byte loads per thread
// in flight; the best of the variants is the "achievable" rate the kernels' roofline fractions can be read against.
//   hipcc -O3 --offload-arch=gfx950 -o tools/hbm_mix_probe tools/hbm_mix_probe.hip ; tools/hbm_mix_probe [json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// thread t of workgroup b handles 16-byte words  (b * U + u) * 256 + t, u < U  of the source, and writes word w to
// W destination planes (plane p at dst + p * n_words): the same word count read, W times as much written
template <int U, int W, bool NT, bool READ, class f4 = ::f4>
__global__ __launch_bounds__(256) void mix_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n_words)
{
    const size_t base = (size_t)blockIdx.x * U * 256 + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t w = base + (size_t)u * 256;
        if (READ) v[u] = w < n_words ? __builtin_nontemporal_load(src + w) : (f4)(0.f);
        else      v[u] = (f4)((float)threadIdx.x);
    }
    if (W == 0) {
        f4 s = (f4)(0.f);
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
        if (s.x + s.y == 12345.678f) dst[blockIdx.x] = s;                  // (never true: keeps the loads alive)
        return;
    }
#pragma unroll
    for (int p = 0; p < (W > 0 ? W : 1); ++p)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t w = base + (size_t)u * 256;
            if (w < n_words) {
                if (NT) __builtin_nontemporal_store(v[u], dst + (size_t)p * n_words + w);
                else    dst[(size_t)p * n_words + w] = v[u];
            }
        }
}

struct Result { std::string name; double rd, wr, ms, gbs; };

// launches alternate between two source blocks and two destination blocks (what the bench's ping-pong input buffers
// and two-block output ring do): the working set of a run is 2 x (read + written) -- 1 GB at 1 : 1 -- so that the
// 256 MB Infinity Cache holds nothing of the next launch
template <int U, int W, bool NT, bool READ, class T = f4>
Result run(const char *name, const void *src_, void *dst_, size_t n16, int reps)
{
    const size_t n_words = n16 * 16 / sizeof(T);
    const T *src = static_cast<const T *>(src_);
    T *dst = static_cast<T *>(dst_);
    const int grid = (int)((n_words + (size_t)U * 256 - 1) / ((size_t)U * 256));
#define LAUNCH(i) hipLaunchKernelGGL((mix_kernel<U, W, NT, READ, T>), dim3(grid), dim3(256), 0, 0, src + ((i) & 1) * n_words, dst + ((i) & 1) * n_words * (W > 0 ? W : 1), n_words)
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 6; ++i) LAUNCH(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) LAUNCH(i);
#undef LAUNCH
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    Result r;
    r.name = name;
    r.rd = READ ? 16.0 * n16 : 0.0;
    r.wr = 16.0 * n16 * W;
    r.ms = ms;
    r.gbs = (r.rd + r.wr) / (ms * 1e-3) / 1e9;
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return r;
}

int main(int argc, char **argv)
{
    const bool json = argc > 1 && !strcmp(argv[1], "json");
    const size_t n_words = (size_t)1 << 24;               // 2^24 x 16 B = 268 MB read: one 2^25-sample cf32 block
    f4 *src = nullptr, *dst = nullptr;
    CK(hipMalloc(&src, n_words * 16 * 2));
    CK(hipMalloc(&dst, n_words * 16 * 4 * 2));
    CK(hipMemset(src, 1, n_words * 16 * 2));
    CK(hipMemset(dst, 0, n_words * 16 * 4 * 2));
    const int reps = 30;
    std::vector<Result> rs;
    // chip warm-up: a bandwidth-bound launch settles only after ~100 launches
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((mix_kernel<4, 1, true, true>), dim3((int)(n_words / 1024)), dim3(256), 0, 0, src, dst, n_words);
    CK(hipDeviceSynchronize());
#define RUN(U, W, NT, READ, NAME) rs.push_back(run<U, W, NT, READ>(NAME, src, dst, n_words, reps))
    typedef float f2 __attribute__((ext_vector_type(2)));
#define RUN8(U, W, NT, READ, NAME) rs.push_back(run<U, W, NT, READ, f2>(NAME, src, dst, n_words, reps))
    RUN8(8, 1, false, true, "1:1 U8 8-byte"); RUN8(8, 1, true, true, "1:1 U8 8-byte nt"); RUN8(16, 1, false, true, "1:1 U16 8-byte");
    RUN8(8, 2, false, true, "1:2 U8 8-byte"); RUN8(8, 2, true, true, "1:2 U8 8-byte nt");
    RUN(4, 0, false, true, "1:0 U4");  RUN(8, 0, false, true, "1:0 U8");  RUN(16, 0, false, true, "1:0 U16");
    RUN(4, 1, false, true, "1:1 U4");  RUN(4, 1, true, true, "1:1 U4 nt"); RUN(8, 1, false, true, "1:1 U8"); RUN(8, 1, true, true, "1:1 U8 nt");
    RUN(16, 1, true, true, "1:1 U16 nt"); RUN(2, 1, false, true, "1:1 U2"); RUN(1, 1, false, true, "1:1 U1");
    RUN(4, 2, false, true, "1:2 U4");  RUN(4, 2, true, true, "1:2 U4 nt"); RUN(8, 2, false, true, "1:2 U8"); RUN(8, 2, true, true, "1:2 U8 nt");
    RUN(4, 4, false, true, "1:4 U4");  RUN(4, 4, true, true, "1:4 U4 nt"); RUN(8, 4, false, true, "1:4 U8"); RUN(8, 4, true, true, "1:4 U8 nt");
    RUN(4, 1, false, false, "0:1 U4"); RUN(4, 1, true, false, "0:1 U4 nt"); RUN(8, 1, true, false, "0:1 U8 nt");
    if (json) {
        printf("{\"what\": \"tools/hbm_mix_probe.hip: coalesced 16-byte reads of a 268 MB block and W x as many bytes written, no arithmetic, %d launches each after 200 warm-up launches; GB/s = (read + written) / launch time\",\n \"peak_GBps\": 8000, \"variants\": [", reps);
        for (size_t i = 0; i < rs.size(); ++i)
            printf("%s\n  {\"variant\": \"%s\", \"read_MB\": %.1f, \"written_MB\": %.1f, \"ms\": %.4f, \"GBps\": %.0f, \"frac_of_peak\": %.3f}", i ? "," : "",
                   rs[i].name.c_str(), rs[i].rd / 1e6, rs[i].wr / 1e6, rs[i].ms, rs[i].gbs, rs[i].gbs / 8000.0);
        printf("],\n \"best\": {");
        const char *mixes[] = {"1:0", "1:1", "1:2", "1:4", "0:1"};
        for (int m = 0; m < 5; ++m) {
            double best = 0;
            for (auto &r : rs) if (r.name.compare(0, 3, mixes[m]) == 0 && r.gbs > best) best = r.gbs;
            printf("%s\"%s\": {\"GBps\": %.0f, \"frac_of_peak\": %.3f}", m ? ", " : "", mixes[m], best, best / 8000.0);
        }
        printf("}}\n");
    } else {
        for (auto &r : rs) printf("%-18s read %7.1f MB written %7.1f MB  %.4f ms  %6.0f GB/s  = %.3f of 8 TB/s\n", r.name.c_str(), r.rd / 1e6, r.wr / 1e6, r.ms, r.gbs, r.gbs / 8000.0);
    }
    CK(hipFree(src));
    CK(hipFree(dst));
    return 0;
}
