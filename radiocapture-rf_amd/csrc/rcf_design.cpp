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

// gr-fft window.cc Izero(): power series of the modified Bessel function I0
static double izero(double x)
{
    double sum = 1, u = 1, halfx = x / 2.0;
    int n = 1;
    do {
        double temp = halfx / (double)n;
        n += 1;
        temp *= temp;
        u *= temp;
        sum += u;
    } while (u >= 1e-21 * sum);
    return sum;
}

void design_window(int type, int n, float *w, double beta)
{
    const double M = (double)(n - 1);
    if (type == RCF_WIN_KAISER) {
        const double ibeta = 1.0 / izero(beta), inm1 = 1.0 / M;
        for (int i = 0; i < n; ++i) {
            const double temp = 2 * i * inm1 - 1;
            w[i] = (float)(izero(beta * std::sqrt(1.0 - temp * temp)) * ibeta);
        }
        return;
    }
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

// window.cc max_attenuation()
double design_max_attenuation(int window, double beta)
{
    switch (window) {
    case RCF_WIN_HAMMING: return 53;
    case RCF_WIN_BLACKMAN: return 74;
    case RCF_WIN_KAISER: return beta / 0.1102 + 8.7;
    case RCF_WIN_BLACKMAN_HARRIS: return 92;
    default: return -1;
    }
}

// firdes::low_pass / firdes::high_pass: windowed ideal response, unit gain at DC / at fs/2
std::vector<float> design_firdes(int kind, double gain, double fs, double fc, double tw, int window, double beta)
{
    const int ntaps = design_ntaps(fs, tw, design_max_attenuation(window, beta));
    std::vector<float> w(ntaps), taps(ntaps);
    design_window(window, ntaps, w.data(), beta);
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2.0 * kPi * fc / fs;
    for (int n = -M; n <= M; ++n) {
        if (kind == RCF_FIR_LOW_PASS) {
            if (n == 0) taps[n + M] = (float)(fwT0 / kPi * w[n + M]);
            else        taps[n + M] = (float)(std::sin(n * fwT0) / (n * kPi) * w[n + M]);
        } else {
            if (n == 0) taps[n + M] = (float)((1 - (fwT0 / kPi)) * w[n + M]);
            else        taps[n + M] = (float)(-std::sin(n * fwT0) / (n * kPi) * w[n + M]);
        }
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; ++n)
        fmax += 2.0 * taps[n + M] * (kind == RCF_FIR_LOW_PASS ? 1.0 : std::cos(n * kPi));
    const double g = gain / fmax;
    for (int i = 0; i < ntaps; ++i) taps[i] = (float)(taps[i] * g);
    return taps;
}

// gr-analog fm_emph.py fm_deemph: H(s) = w_ca / (s + w_ca) through the bilinear transform
void design_fm_deemph(double fs, double tau, double b[2], double a[2])
{
    const double w_c = 1.0 / tau;
    const double w_ca = 2.0 * fs * std::tan(w_c / (2.0 * fs));
    const double k = -w_ca / (2.0 * fs);
    const double z1 = -1.0;
    const double p1 = (1.0 + k) / (1.0 - k);
    const double b0 = -k / (1.0 - k);
    b[0] = b0 * 1.0;
    b[1] = b0 * -z1;
    a[0] = 1.0;
    a[1] = -p1;
}

// rational_resampler.py design_filter(interpolation, decimation, fractional_bw = 0.4), ratio already reduced
std::vector<float> design_resampler(int interpolation, int decimation)
{
    const double beta = 7.0, halfband = 0.5, fractional_bw = 0.4;
    const double rate = (double)interpolation / (double)decimation;
    double trans_width, mid;
    if (rate >= 1.0) {
        trans_width = halfband - fractional_bw;
        mid = halfband - trans_width / 2.0;
    } else {
        trans_width = rate * (halfband - fractional_bw);
        mid = rate * halfband - trans_width / 2.0;
    }
    return design_firdes(RCF_FIR_LOW_PASS, interpolation, interpolation, mid, trans_width, RCF_WIN_KAISER, beta);
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
