// rcf_peaks.cpp -- host peak picker: the arithmetic of /root/reference/fft_peak_detection.py:54-72.
//
//   data[i] += abs(min(data))            (float32, :58-59)
//   data_average = sum(data) / len(data) (sequential float64, :61 -- numpy-1.x scalar promotion)
//   scipy.signal.find_peaks(data, width=[min_w, max_w], prominence=p)   (:65)
//   keep peaks with data[line] > data_average * 2                        (:71)
//
// find_peaks is restated from the published SciPy algorithm (float64 throughout, like SciPy's
// Cython kernels): strict local maxima with plateau midpoints -> prominence by outward walks that
// stop at the first strictly higher sample -> width at half prominence with linear interpolation
// between the bracketing samples, bounded by the prominence bases.  Built -ffp-contract=off so every
// float64 operation rounds exactly as SciPy's does (bit-exact indices).
#include "rcf_internal.h"

#include <cmath>
#include <vector>

namespace rcfx {

int64_t find_peaks_host(const float *spectrum, int64_t n, double min_w, double max_w, double prominence,
                        int64_t *idx, int64_t cap, double *mean_out)
{
    if (mean_out) *mean_out = 0.0;
    if (n <= 0) return 0;
    float lo = spectrum[0];
    for (int64_t i = 1; i < n; ++i) lo = spectrum[i] < lo ? spectrum[i] : lo;
    const float lift = std::fabs(lo);
    std::vector<double> x((size_t)n);
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const float v = spectrum[i] + lift;     // float32 add
        x[(size_t)i] = (double)v;
        acc += (double)v;                       // left-to-right float64
    }
    const double mean = acc / (double)n;
    if (mean_out) *mean_out = mean;
    const double gate = mean * 2;

    int64_t found = 0;
    const int64_t last = n - 1;
    for (int64_t i = 1; i < last; ++i) {
        if (!(x[i - 1] < x[i])) continue;
        int64_t e = i + 1;                       // plateau scan
        while (e < last && x[e] == x[i]) ++e;
        if (!(x[e] < x[i])) continue;
        const int64_t pk = (i + e - 1) / 2;
        i = e;                                   // resume after the plateau (loop ++ moves past it)
        const double top = x[pk];
        // prominence: bases are the lowest samples between the peak and the first higher sample
        int64_t lbase = pk, rbase = pk;
        double lmin = top, rmin = top;
        for (int64_t j = pk; j >= 0 && x[j] <= top; --j)
            if (x[j] < lmin) { lmin = x[j]; lbase = j; }
        for (int64_t j = pk; j <= last && x[j] <= top; ++j)
            if (x[j] < rmin) { rmin = x[j]; rbase = j; }
        const double prom = top - (lmin > rmin ? lmin : rmin);
        if (!(prom >= prominence)) continue;
        // width at half prominence
        const double level = top - prom * 0.5;
        int64_t j = pk;
        while (lbase < j && level < x[j]) --j;
        double left = (double)j;
        if (x[j] < level) left += (level - x[j]) / (x[j + 1] - x[j]);
        j = pk;
        while (j < rbase && level < x[j]) ++j;
        double right = (double)j;
        if (x[j] < level) right -= (level - x[j]) / (x[j - 1] - x[j]);
        const double width = right - left;
        if (!(min_w <= width && width <= max_w)) continue;
        if (!(x[pk] > gate)) continue;
        if (found < cap && idx) idx[found] = pk;
        ++found;
    }
    return found;
}

}  // namespace rcfx
