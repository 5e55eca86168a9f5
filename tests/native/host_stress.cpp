// host_stress.cpp -- sanitizer driver for the CPU-visible part of librcf's host layer ("sanitizers").
// Built twice by tests/native/Makefile -- AddressSanitizer + UBSan, and ThreadSanitizer -- from the product sources
// rcf_design.cpp / rcf_peaks.cpp (no HIP device code, no GPU needed) and run by tests/test_native_sanitizers.py:
// several threads hammer the filter designs (window method, Parks-McClellan, resampler, de-emphasis, composite
// taps) and the float64 peak picker concurrently and compare every result with the single-threaded one.
// Any data race, out-of-bounds access or undefined behaviour fails the run.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../radiocapture-rf_amd/csrc/rcf_internal.h"

namespace rcfx {
// the two symbols the design / peak sources expect from rcf_handle.cpp
static thread_local char g_err[256];
void set_error(const char *fmt, ...) { (void)fmt; g_err[0] = 0; }
bool hip_ok(hipError_t e, const char *) { return e == hipSuccess; }
}  // namespace rcfx

using namespace rcfx;

struct Result {
    std::vector<float> lp2, firdes_hp, optfir, resamp, comp;
    std::vector<int64_t> peaks;
    double deemph[4];
    double mean;
    bool operator==(const Result &o) const
    {
        return lp2 == o.lp2 && firdes_hp == o.firdes_hp && optfir == o.optfir && resamp == o.resamp && comp == o.comp &&
               peaks == o.peaks && std::memcmp(deemph, o.deemph, sizeof(deemph)) == 0 && mean == o.mean;
    }
};

static std::vector<float> spectrum(int n, unsigned seed)
{
    std::vector<float> s(n);
    unsigned x = seed * 2654435761u + 1;
    for (int i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        s[i] = -40.0f + 5.0f * ((x >> 8) & 0xffff) / 65536.0f;
    }
    for (int c = 1500; c < n - 1500; c += 2300)
        for (int i = -200; i <= 200; ++i) s[c + i] += 300.0f * std::exp(-0.5f * (i / 30.0f) * (i / 30.0f));
    return s;
}

static Result work(unsigned seed)
{
    Result r;
    const double fs = 20e6 - 1e6 * (seed % 3);
    r.lp2 = design_low_pass_2(1.0, fs, 6250.0, 6250.0, 20.0, RCF_WIN_HAMMING);
    r.firdes_hp = design_firdes(RCF_FIR_HIGH_PASS, 1.0, 25000.0, 300.0, 30.0, RCF_WIN_HAMMING, 6.76);
    if (!design_optfir_low_pass(8.0, 25000.0, 6250.0, 8250.0, 0.1, 60.0, 2, r.optfir)) r.optfir.clear();
    r.resamp = design_resampler(8, 25);
    design_fm_deemph(25000.0, 75e-6, r.deemph, r.deemph + 2);
    float incr[2];
    design_composite(r.lp2.data(), (int)r.lp2.size(), 800, 5012500.0 + 12500.0 * seed, 20e6, r.comp, incr);
    r.comp.push_back(incr[0]);
    r.comp.push_back(incr[1]);
    const std::vector<float> sp = spectrum(16384, seed);
    r.peaks.assign(64, -1);
    const int64_t c = find_peaks_host(sp.data(), (int64_t)sp.size(), 20.48, 204.8, 1.0, r.peaks.data(), 64, &r.mean);
    r.peaks.resize((size_t)std::min<int64_t>(c, 64));
    return r;
}

// index arithmetic the kernels and the host share (rcf_internal.h): ring views, launch plans, kernel geometry
static int check_index_math()
{
    // StreamView::at against the layouts it stands for
    const int NB = 256;
    const int64_t pitch = pfb_tile_pitch(NB);
    if (pitch % 16 != 0 || pitch < 16 * NB) return 10;                    // whole lines, room for every bin's line
    for (int bin : {0, 1, 37, NB - 1}) {
        StreamView tiled{};
        tiled.base = nullptr; tiled.mask = 4095; tiled.origin = 0; tiled.stride = pitch; tiled.tshift = kPfbTileLog2;
        StreamView fm{};
        fm.mask = 4095; fm.origin = 0; fm.stride = 1600; fm.tshift = 0;
        StreamView lin{};
        lin.mask = ~0ull; lin.origin = -777; lin.stride = 1; lin.tshift = 0;
        for (int64_t n = 0; n < 3 * 4096; n += 7) {
            const uint64_t i = (uint64_t)n & 4095;
            if (tiled.at(n) + (uint64_t)bin * 16 != (i >> 4) * (uint64_t)pitch + (uint64_t)bin * 16 + (i & 15)) return 11;
            if (fm.at(n) != i * 1600) return 12;
            if (lin.at(n) != (uint64_t)(n + 777)) return 13;
        }
        // one bin's 16 frames of a tile are one 128-byte line, consecutive tiles never overlap
        if (tiled.at(15) - tiled.at(0) != 15 || tiled.at(16) - tiled.at(0) != (uint64_t)pitch) return 14;
    }
    // the matrix-core launch plan: always a legal (NT, parts), parts never more than the tap chunks
    for (int C : {8, 31, 32, 33, 256, 1024, 4096, 131072})
        for (int T : {65, 2909, 5817})
            for (int n_k : {1, 63, 64, 5243}) {
                const MfmaPlan pl = mfma_plan(C, n_k, T);
                if (pl.nt < 1 || pl.nt > 2 || pl.parts < 1 || pl.parts > 8) return 20;
                if (pl.parts > bank2_steps(T) / kM2ChunkSteps) return 21;
            }
    if (bank2_steps(2909) % kM2ChunkSteps != 0 || bank2_steps(2909) * 8 < 2909) return 22;
    // small-T kernel geometry: outputs per workgroup fit its 512 slots and its LDS tile
    for (int D : {1, 2, 3, 12, 40})
        for (int T : {1, 11, 69, 96, 97}) {
            const int kb = fir_small_outputs(D, T);
            if (T > 96 ? kb != 0 : (kb < 0 || kb > 511 || (kb > 0 && (kb < 32 || kb * D + T > 3300)))) return 30;
        }
    return 0;
}

int main()
{
    if (const int rc = check_index_math()) {
        std::fprintf(stderr, "host_stress: index arithmetic check %d failed\n", rc);
        return 3;
    }
    const int kThreads = 8, kRounds = 6;
    std::vector<Result> want;
    for (int t = 0; t < kThreads; ++t) want.push_back(work((unsigned)t));
    if (want[0].lp2.size() != 2909 || want[0].peaks.empty() || want[0].optfir.empty()) {
        std::fprintf(stderr, "host_stress: unexpected single-thread result (%zu taps, %zu peaks, %zu optfir)\n",
                     want[0].lp2.size(), want[0].peaks.size(), want[0].optfir.size());
        return 2;
    }
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < kThreads; ++t)
        th.emplace_back([&, t] {
            for (int i = 0; i < kRounds; ++i)
                if (!(work((unsigned)t) == want[(size_t)t])) bad.fetch_add(1);
        });
    for (auto &x : th) x.join();
    if (bad.load()) {
        std::fprintf(stderr, "host_stress: %d results differ between threads\n", bad.load());
        return 1;
    }
    std::printf("host_stress: %d threads x %d rounds identical to the single-threaded results\n", kThreads, kRounds);
    return 0;
}
