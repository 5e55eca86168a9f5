// rcf_design_abi.cpp -- C ABI of the host filter designs (rcf_design.cpp).
#include "rcf_plan.h"

namespace rcfx {


}  // namespace rcfx

using namespace rcfx;

// =================================================================== C ABI
extern "C" {

int rcf_design_low_pass_2(double gain, double fs, double fc, double tw, double att_db, int window, float *taps,
                          int cap)
{
    if (fs <= 0 || tw <= 0) { set_error("bad design arguments"); return RCF_EINVAL; }
    const int n = design_ntaps(fs, tw, att_db);
    if (!taps || cap < n) return -n;
    std::vector<float> t = design_low_pass_2(gain, fs, fc, tw, att_db, window);
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

int rcf_design_window(int window, int n, float *w)
{
    if (n < 2 || !w) { set_error("bad window arguments"); return RCF_EINVAL; }
    design_window(window, n, w);
    return RCF_OK;
}

int rcf_channel_params_ex(double samp_rate, int channel_rate, int decim_rule, int *decim, int *ntaps, double *out_rate)
{
    if (samp_rate <= 0 || channel_rate <= 0 || (decim_rule != RCF_DECIM_EXACT && decim_rule != RCF_DECIM_FLOOR)) {
        set_error("bad rates / decimation rule");
        return RCF_EINVAL;
    }
    const int q = (int)(samp_rate / channel_rate);
    if (q < 2 || ((q & 1) && decim_rule == RCF_DECIM_EXACT)) {
        set_error("int(fs/cr)/2 is not a positive integer for fs=%g cr=%d", samp_rate, channel_rate);
        return RCF_ERANGE;
    }
    if (decim) *decim = q / 2;
    if (ntaps) *ntaps = design_ntaps(samp_rate, channel_rate / 2.0, 20.0);
    if (out_rate) *out_rate = samp_rate / (q / 2);
    return RCF_OK;
}

int rcf_channel_params(double samp_rate, int channel_rate, int *decim, int *ntaps)
{
    return rcf_channel_params_ex(samp_rate, channel_rate, RCF_DECIM_EXACT, decim, ntaps, nullptr);
}

int rcf_design_firdes(int kind, double gain, double fs, double fc, double tw, int window, double beta, float *taps,
                      int cap)
{
    if (fs <= 0 || tw <= 0 || (kind != RCF_FIR_LOW_PASS && kind != RCF_FIR_HIGH_PASS) ||
        design_max_attenuation(window, beta) <= 0) {
        set_error("bad design arguments");
        return RCF_EINVAL;
    }
    const int n = design_ntaps(fs, tw, design_max_attenuation(window, beta));
    if (!taps || cap < n) return -n;
    std::vector<float> t = design_firdes(kind, gain, fs, fc, tw, window, beta);
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

int rcf_design_optfir_low_pass(double gain, double fs, double freq1, double freq2, double passband_ripple_db,
                               double stopband_atten_db, float *taps, int cap)
{
    std::vector<float> t;
    if (!design_optfir_low_pass(gain, fs, freq1, freq2, passband_ripple_db, stopband_atten_db, 2, t)) {
        set_error("equiripple design failed (bad band edges, or the exchange did not find its extrema)");
        return RCF_EINVAL;
    }
    const int n = (int)t.size();
    if (!taps || cap < n) return -n;
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

int rcf_design_fm_deemph(double fs, double tau, double btaps[2], double ataps[2])
{
    if (fs <= 0 || tau <= 0 || !btaps || !ataps) { set_error("bad de-emphasis arguments"); return RCF_EINVAL; }
    design_fm_deemph(fs, tau, btaps, ataps);
    return RCF_OK;
}

int rcf_design_resampler(int interpolation, int decimation, int *interp_out, int *decim_out, float *taps, int cap)
{
    if (interpolation < 1 || decimation < 1) { set_error("bad resampler ratio"); return RCF_EINVAL; }
    int a = interpolation, b = decimation;
    while (b) { const int t = a % b; a = b; b = t; }
    const int I = interpolation / a, D = decimation / a;
    if (interp_out) *interp_out = I;
    if (decim_out) *decim_out = D;
    std::vector<float> t = design_resampler(I, D);
    const int n = (int)t.size();
    if (!taps || cap < n) return -n;
    std::memcpy(taps, t.data(), sizeof(float) * (size_t)n);
    return n;
}

}  // extern "C"
