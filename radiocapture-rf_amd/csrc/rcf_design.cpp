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

// What a bin of an exact-phase filterbank differs by from GNU Radio's channel at the same offset (SURVEY.md 8(c)
// [GR-spec] (i)): GR builds its composite taps as h[i] e^{j float32(i * fwT0)}, fwT0 = float32(2 pi f_k / fs) -- the
// product is rounded to float32, at |i fwT0| ~ 4-9e3 rad to a grid of 2.4-9.8e-4 rad -- where the bank has
// h[i] e^{j 2 pi k i / NB}.  d[i] = float32(i fwT0) - 2 pi k i / NB splits into a constant (the h^2-weighted mean:
// a rotation of the whole output, returned in *const_phase and carried by the tap's rotator) and an error filter
// g[i] = h[i] (e^{j (d[i] - const)} - 1) whose output is added to the channel: white input of power P leaks
// P |g|_2^2 into it.  *leak_l2 = |g|_2.
void design_tap_leakage(double fs, int n_bins, const float *taps, int T, int bin, double *leak_l2, double *const_phase)
{
    const int ks = bin < n_bins / 2 ? bin : bin - n_bins;
    const double f_k = (double)ks * fs / n_bins;
    const float fwT0 = (float)(2.0 * kPi * f_k / fs);
    std::vector<double> d((size_t)T);
    double sw = 0.0, swd = 0.0;
    for (unsigned i = 0; i < (unsigned)T; ++i) {
        const float th = (float)i * fwT0;
        // exact phase reduced before the subtraction: k i mod NB keeps the argument small
        const long long ki = ((long long)ks * (long long)i) % n_bins;
        const double ex = 2.0 * kPi * (double)ki / n_bins;
        d[i] = std::remainder((double)th - ex, 2.0 * kPi);
        const double w = (double)taps[i] * (double)taps[i];
        sw += w;
        swd += w * d[i];
    }
    const double c = sw > 0.0 ? swd / sw : 0.0;
    double l2 = 0.0;
    for (int i = 0; i < T; ++i) {
        const double e = d[i] - c;
        const double gr = (double)taps[i] * (std::cos(e) - 1.0), gi = (double)taps[i] * std::sin(e);
        l2 += gr * gr + gi * gi;
    }
    if (leak_l2) *leak_l2 = std::sqrt(l2);
    if (const_phase) *const_phase = c;
}

}  // namespace rcfx

// ---------------------------------------------------------------- Parks-McClellan (equiripple) design
// gr-filter's optfir.low_pass() = remezord() order estimate + pm_remez(order, bands, ampl, weights, "bandpass").
// This is the classical exchange algorithm for symmetric (type I / II) linear-phase filters on a dense grid of
// 16 points per extremum: barycentric Lagrange interpolation through the r + 1 trial extrema (Oppenheim &
// Schafer 7.131-7.133), search for the new alternating extrema of the weighted error, stop when the extremal
// errors agree to 1e-4, then frequency-sample the interpolant to get the taps.  All in double.
namespace rcfx {
namespace {

struct PmState {
    int r = 0;
    std::vector<double> grid, D, W, E, x, y, ad;
    std::vector<int> ext;
};

void pm_params(PmState &s)
{
    const int r = s.r;
    for (int i = 0; i <= r; ++i) s.x[i] = std::cos(2.0 * kPi * s.grid[s.ext[i]]);
    const int ld = (r - 1) / 15 + 1;                   // interleaved products keep the denominators in range
    for (int i = 0; i <= r; ++i) {
        double denom = 1.0;
        const double xi = s.x[i];
        for (int j = 0; j < ld; ++j)
            for (int k = j; k <= r; k += ld)
                if (k != i) denom *= 2.0 * (xi - s.x[k]);
        if (std::fabs(denom) < 1e-5) denom = 1e-5;
        s.ad[i] = 1.0 / denom;
    }
    double numer = 0, denom = 0, sign = 1;
    for (int i = 0; i <= r; ++i) {
        numer += s.ad[i] * s.D[s.ext[i]];
        denom += sign * s.ad[i] / s.W[s.ext[i]];
        sign = -sign;
    }
    const double delta = numer / denom;
    sign = 1;
    for (int i = 0; i <= r; ++i) {
        s.y[i] = s.D[s.ext[i]] - sign * delta / s.W[s.ext[i]];
        sign = -sign;
    }
}

double pm_response(const PmState &s, double freq)
{
    double numer = 0, denom = 0;
    const double xc = std::cos(2.0 * kPi * freq);
    for (int i = 0; i <= s.r; ++i) {
        double c = xc - s.x[i];
        if (std::fabs(c) < 1e-7) return s.y[i];
        c = s.ad[i] / c;
        denom += c;
        numer += c * s.y[i];
    }
    return numer / denom;
}

// new extremal set: local extrema of E on the grid, thinned to r + 1 alternating ones
bool pm_search(PmState &s)
{
    const int r = s.r, n = (int)s.grid.size();
    const std::vector<double> &E = s.E;
    std::vector<int> found;
    if ((E[0] > 0.0 && E[0] > E[1]) || (E[0] < 0.0 && E[0] < E[1])) found.push_back(0);
    for (int i = 1; i < n - 1; ++i)
        if ((E[i] >= E[i - 1] && E[i] > E[i + 1] && E[i] > 0.0) || (E[i] <= E[i - 1] && E[i] < E[i + 1] && E[i] < 0.0)) {
            if ((int)found.size() >= 2 * r) return false;
            found.push_back(i);
        }
    const int j = n - 1;
    if ((E[j] > 0.0 && E[j] > E[j - 1]) || (E[j] < 0.0 && E[j] < E[j - 1])) {
        if ((int)found.size() >= 2 * r) return false;
        found.push_back(j);
    }
    if ((int)found.size() < r + 1) return false;
    int extra = (int)found.size() - (r + 1);
    while (extra > 0) {
        const int k = (int)found.size();
        bool up = E[found[0]] > 0.0, alt = true;
        int l = 0;
        for (int q = 1; q < k; ++q) {
            if (std::fabs(E[found[q]]) < std::fabs(E[found[l]])) l = q;
            if (up && E[found[q]] < 0.0) up = false;
            else if (!up && E[found[q]] > 0.0) up = true;
            else { alt = false; break; }               // two neighbours of one sign: drop the smallest seen so far
        }
        if (alt && extra == 1) l = std::fabs(E[found[k - 1]]) < std::fabs(E[found[0]]) ? k - 1 : 0;
        found.erase(found.begin() + l);
        --extra;
    }
    for (int i = 0; i <= r; ++i) s.ext[i] = found[i];
    return true;
}

}  // namespace

// bands: edges in cycles/sample (0 .. 0.5), two per band; des: desired amplitude per band; weight per band
bool design_pm_remez(int numtaps, const std::vector<double> &bands, const std::vector<double> &des,
                     const std::vector<double> &weight, std::vector<double> &h)
{
    const int nb = (int)des.size(), density = 16;
    if (numtaps < 3 || nb < 1 || (int)bands.size() != 2 * nb || (int)weight.size() != nb) return false;
    PmState s;
    s.r = numtaps / 2 + (numtaps & 1);
    const int r = s.r;
    const double delf = 0.5 / (density * r);
    for (int b = 0; b < nb; ++b) {
        double lowf = bands[2 * b];
        const double highf = bands[2 * b + 1];
        const int k = (int)((highf - lowf) / delf + 0.5);
        if (k < 1) return false;
        for (int i = 0; i < k; ++i) {
            s.grid.push_back(lowf);
            s.D.push_back(des[b]);
            s.W.push_back(weight[b]);
            lowf += delf;
        }
        s.grid.back() = highf;
    }
    const int n = (int)s.grid.size();
    if (n < r + 2) return false;
    if ((numtaps & 1) == 0)                            // type II: A(w) = cos(w/2) P(w)
        for (int i = 0; i < n; ++i) {
            const double c = std::cos(kPi * s.grid[i]);
            s.D[i] /= c;
            s.W[i] *= c;
        }
    s.E.assign(n, 0.0);
    s.x.assign(r + 1, 0.0); s.y.assign(r + 1, 0.0); s.ad.assign(r + 1, 0.0);
    s.ext.resize(r + 1);
    for (int i = 0; i <= r; ++i) s.ext[i] = i * (n - 1) / r;
    for (int iter = 0; iter < 40; ++iter) {
        pm_params(s);
        for (int i = 0; i < n; ++i) s.E[i] = s.W[i] * (s.D[i] - pm_response(s, s.grid[i]));
        if (!pm_search(s)) return false;
        double mn = std::fabs(s.E[s.ext[0]]), mx = mn;
        for (int i = 1; i <= r; ++i) {
            const double v = std::fabs(s.E[s.ext[i]]);
            mn = std::min(mn, v);
            mx = std::max(mx, v);
        }
        if ((mx - mn) / mx < 0.0001) break;
    }
    pm_params(s);
    // frequency sampling of the converged interpolant
    std::vector<double> A(numtaps / 2 + 1);
    for (int i = 0; i <= numtaps / 2; ++i) {
        const double c = (numtaps & 1) ? 1.0 : std::cos(kPi * (double)i / numtaps);
        A[i] = pm_response(s, (double)i / numtaps) * c;
    }
    h.assign(numtaps, 0.0);
    const double M = (numtaps - 1) / 2.0;
    const int kmax = (numtaps & 1) ? (int)M : numtaps / 2 - 1;
    for (int t = 0; t < numtaps; ++t) {
        double val = A[0];
        const double xx = 2.0 * kPi * (t - M) / numtaps;
        for (int k = 1; k <= kmax; ++k) val += 2.0 * A[k] * std::cos(xx * k);
        h[t] = val / numtaps;
    }
    return true;
}

// gr-filter optfir.py: low_pass(gain, Fs, freq1, freq2, passband_ripple_db, stopband_atten_db, nextra_taps = 2)
bool design_optfir_low_pass(double gain, double fs, double f1, double f2, double ripple_db, double atten_db,
                            int nextra, std::vector<float> &taps)
{
    if (!(fs > 0) || !(f1 > 0) || !(f2 > f1) || !(f2 < fs / 2) || !(gain != 0)) return false;
    const double rr = std::pow(10.0, ripple_db / 20.0);
    const double dev_p = ((rr - 1.0) / (rr + 1.0)) / gain;       // remezord: deviation relative to the passband gain
    const double dev_s = std::pow(10.0, -atten_db / 20.0);
    const double c1 = f1 / fs, c2 = f2 / fs;
    // lporder(): Herrmann / Rabiner / Chan length estimate
    const double df = std::fabs(c2 - c1), ddp = std::log10(dev_p), dds = std::log10(dev_s);
    const double a1 = 5.309e-3, a2 = 7.114e-2, a3 = -4.761e-1, a4 = -2.66e-3, a5 = -5.941e-1, a6 = -4.278e-1;
    const double b1 = 11.01217, b2 = 0.5124401;
    const double dinf = ((a1 * ddp * ddp + a2 * ddp + a3) * dds) + (a4 * ddp * ddp + a5 * ddp + a6);
    const double ff = b1 + b2 * (ddp - dds);
    const double l = dinf / df - ff * df + 1;
    const int order = (int)std::ceil(l) - 1;
    const double mx = std::max(dev_p, dev_s);
    std::vector<double> h;
    if (!design_pm_remez(order + nextra + 1, {0.0, c1, c2, 0.5}, {gain, 0.0}, {mx / dev_p, mx / dev_s}, h)) return false;
    taps.resize(h.size());
    for (size_t i = 0; i < h.size(); ++i) taps[i] = (float)h[i];
    return true;
}

}  // namespace rcfx
