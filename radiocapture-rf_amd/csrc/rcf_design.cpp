// Host-side filter / window design and GNU-Radio-faithful composite-tap construction.
//
// Follows the published GNU Radio 3.8 semantics of the calls the reference makes:
//   firdes.low_pass_2(...)                      /root/reference/rc_frontend/channel.py:33
//   window.blackmanharris(n)                    /root/reference/fft_vector.py:38
//   freq_xlating_fir_filter_ccc tap rotation    /root/reference/rc_frontend/channel.py:35
// Built with -ffp-contract=off: the float32 roundings below are part of the specified result
// (SURVEY.md 8(c) "[GR-spec]").
#include "rcf_internal.h"

#include <cmath>
#include <vector>

namespace rcfx {

static const double kPi = 3.14159265358979323846;

void design_window(int type, int n, float *w)
{
    const double M = (double)(n - 1);
    if (type == RCF_WIN_HAMMING) {
        for (int i = 0; i < n; ++i) w[i] = (float)(0.54 - 0.46 * std::cos((2.0 * kPi * i) / M));
        return;
    }
    // cosine-sum windows: GR passes the coefficients as C floats and evaluates in double
    double c[4] = {0, 0, 0, 0};
    if (type == RCF_WIN_BLACKMAN) {
        c[0] = (double)0.42f; c[1] = (double)0.5f; c[2] = (double)0.08f;
    } else {  // 4-term 92 dB Blackman-Harris
        c[0] = (double)0.35875f; c[1] = (double)0.48829f; c[2] = (double)0.14128f; c[3] = (double)0.01168f;
    }
    for (int i = 0; i < n; ++i) {
        double v = c[0] - c[1] * std::cos((2.0 * kPi * i) / M) + c[2] * std::cos((4.0 * kPi * i) / M);
        if (c[3] != 0.0) v -= c[3] * std::cos((6.0 * kPi * i) / M);
        w[i] = (float)v;
    }
}

int design_ntaps(double fs, double tw, double att_db)
{
    int n = (int)(att_db * fs / (22.0 * tw));
    return (n & 1) ? n : n + 1;
}

std::vector<float> design_low_pass_2(double gain, double fs, double fc, double tw, double att_db, int window)
{
    const int ntaps = design_ntaps(fs, tw, att_db);
    std::vector<float> w(ntaps), taps(ntaps);
    design_window(window, ntaps, w.data());
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2.0 * kPi * fc / fs;
    for (int n = -M; n <= M; ++n) {
        if (n == 0) taps[n + M] = (float)(fwT0 / kPi * w[n + M]);
        else        taps[n + M] = (float)(std::sin(n * fwT0) / (n * kPi) * w[n + M]);
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; ++n) fmax += 2.0 * taps[n + M];
    const double g = gain / fmax;
    for (int i = 0; i < ntaps; ++i) taps[i] = (float)(taps[i] * g);
    return taps;
}

// freq_xlating_fir_filter_ccc::build_composite_fir(): float32 fwT0, float32 (i * fwT0), cosf/sinf.
// Also returns the rotator increment exp(j * float32(-fwT0 * D)) as GR stores it (float32 pair).
void design_composite(const float *taps, int T, int D, double f0, double fs,
                      std::vector<float> &ctaps_interleaved, float incr[2])
{
    const float fwT0 = (float)(2.0 * kPi * f0 / fs);
    ctaps_interleaved.resize(2 * (size_t)T);
    for (unsigned i = 0; i < (unsigned)T; ++i) {
        const float th = (float)i * fwT0;
        ctaps_interleaved[2 * i]     = taps[i] * std::cos(th);   // cosf on a float argument
        ctaps_interleaved[2 * i + 1] = taps[i] * std::sin(th);
    }
    const float a = -fwT0 * (float)D;
    incr[0] = std::cos(a);
    incr[1] = std::sin(a);
}

}  // namespace rcfx
