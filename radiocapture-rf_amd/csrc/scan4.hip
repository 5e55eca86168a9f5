// scan4.hip -- four-step (N1 x N2) variant of the scan FFT for transforms that do not fit one
// workgroup's LDS (N > 16384, e.g. the 2^20-point scan of BASELINE config 3).
#include "fft_core.hpp"
#include "rcf_internal.h"

namespace rcfx {

bool scan4_supported(int N)
{
    (void)N;
    return false;
}

void launch_scan4_fft(const ScanLaunch &p, hipStream_t s)
{
    (void)p;
    (void)s;
}

}  // namespace rcfx
