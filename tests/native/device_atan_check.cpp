// The discriminator kernels' fast_atan2f_gr (radiocapture-rf_amd/csrc/fast_atan2f_gr.hpp, written as selects) compiled for
// the host, against the oracle's restatement of gr::fast_atan2f's branches (oracle/rcf_oracle.c: ro_fast_atan2f), bit for
// bit: random arguments over many magnitudes, every octant boundary, signed zeros, equal magnitudes, the TAN_MAP_RES edge,
// the table's interval edges.  Prints "checked <n> mismatches <m>"; exit status 1 on any mismatch.
#define RCF_DEVFN static inline
#include "../../radiocapture-rf_amd/csrc/fast_atan2f_gr.hpp"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" float ro_fast_atan2f(float y, float x);

static float atan_table[257];
static void build_table()                     // as the library's host side builds the device table (fir.hip) and the oracle its own
{
    char buf[64];
    for (int i = 0; i < 257; ++i) {
        snprintf(buf, sizeof buf, "%.6e", i < 256 ? atan((double)i / 255.0) : M_PI / 4.0);
        atan_table[i] = (float)strtod(buf, NULL);
    }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}
static float rnd_unit() { return (float)((double)rnd() / 4294967296.0 * 2.0 - 1.0); }

int main(int argc, char **argv)
{
    const long n_random = argc > 1 ? atol(argv[1]) : 20000000L;
    build_table();
    long checked = 0, bad = 0;
    auto check = [&](float y, float x) {
        const float a = rcfx::fast_atan2f_gr(y, x, atan_table), b = ro_fast_atan2f(y, x);
        uint32_t ua, ub;
        memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
        ++checked;
        if (ua != ub && !(a != a && b != b)) {
            if (bad < 10) fprintf(stderr, "mismatch y=%.9g x=%.9g device-source %.9g (%08x) oracle %.9g (%08x)\n", y, x, a, ua, b, ub);
            ++bad;
        }
    };
    // special values in every sign combination
    std::vector<float> sp = {0.0f, -0.0f, 1.0f, -1.0f, 1e-30f, -1e-30f, 1e30f, -1e30f, 1.17549435e-38f, 1e-42f, -1e-42f,
                             0.003921569f, 0.0039215684f, 0.0039215693f, 0.5f, 2.0f, 255.0f, 1.0f / 255.0f, 3.4e38f};
    for (int i = 1; i <= 256; ++i) { sp.push_back((float)i / 255.0f); sp.push_back(nextafterf((float)i / 255.0f, 0.0f)); sp.push_back(nextafterf((float)i / 255.0f, 2.0f)); }
    for (float y : sp) for (float x : sp) for (int s = 0; s < 4; ++s) check((s & 1) ? -y : y, (s & 2) ? -x : x);
    // random arguments: similar magnitudes (every octant), one much smaller than the other (z below TAN_MAP_RES), wide range
    for (long i = 0; i < n_random; ++i) {
        const int kind = (int)(rnd() & 3);
        float y = rnd_unit(), x = rnd_unit();
        if (kind == 1) y *= 0.01f * (float)(rnd() & 1023) / 1024.0f;
        else if (kind == 2) x *= 0.01f * (float)(rnd() & 1023) / 1024.0f;
        else if (kind == 3) { y = ldexpf(y, (int)(rnd() % 200) - 100); x = ldexpf(x, (int)(rnd() % 200) - 100); }
        check(y, x);
        if ((i & 63) == 0) { check(y, y); check(y, -y); check(x, 0.0f); check(0.0f, x); check(-0.0f, x); check(y, -0.0f); }
    }
    printf("checked %ld mismatches %ld\n", checked, bad);
    return bad ? 1 : 0;
}
