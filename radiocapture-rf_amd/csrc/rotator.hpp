// rotator.hpp -- GNU Radio's rotator (phase *= incr in float32, renormalised every 512 calls) in closed form, shared
// by the FIR bank kernels (fir.hip) and the filterbank taps (tapfin.hip).  Both files are compiled without implicit
// FMA contraction: rotate() is an unfused float32 complex multiply in GNU Radio.  The products and sums are written as plain
// operators INSIDE bodies that switch contraction off (#pragma clang fp contract(off)): HIP's fmul_rn / fadd_rn intrinsics are
// header functions of their own, compiled under the translation unit's default -- in a unit built with contraction (pfb.hip,
// whose filterbank launch carries the stage-2 rider) the mul inside one and the add inside the other fuse after inlining.
#pragma once
#include "rcf_internal.h"

namespace rcfx {

namespace {

// sin / cos of a float64 angle of moderate size (|a| < 1e6 rad) to ~1e-16: Cody-Waite reduction by pi/2 and the
// Taylor polynomials on |x| <= pi/4.  The library sincos() carries a Payne-Hanek path and costs ~10x as much;
// with eight of them per lane it was a tenth of the matrix-core bank's time.
__device__ __forceinline__ void sincos_fast(double a, double &sn, double &cs)
{
#pragma clang fp contract(off)
    const double k = rint(a * 0.63661977236758134308);
    double x = fma(-k, 1.57079632679489655800e+00, a);
    x = fma(-k, 6.12323399573676603587e-17, x);
    const double x2 = x * x;
    double ps = -1.0 / 1307674368000.0;
    ps = fma(ps, x2, 1.0 / 6227020800.0);
    ps = fma(ps, x2, -1.0 / 39916800.0);
    ps = fma(ps, x2, 1.0 / 362880.0);
    ps = fma(ps, x2, -1.0 / 5040.0);
    ps = fma(ps, x2, 1.0 / 120.0);
    ps = fma(ps, x2, -1.0 / 6.0);
    const double S = fma(x * x2, ps, x);
    double pc = 1.0 / 20922789888000.0;
    pc = fma(pc, x2, -1.0 / 87178291200.0);
    pc = fma(pc, x2, 1.0 / 479001600.0);
    pc = fma(pc, x2, -1.0 / 3628800.0);
    pc = fma(pc, x2, 1.0 / 40320.0);
    pc = fma(pc, x2, -1.0 / 720.0);
    pc = fma(pc, x2, 1.0 / 24.0);
    pc = fma(pc, x2, -0.5);
    const double Cc = fma(x2, pc, 1.0);
    const int nq = (int)k & 3;
    sn = (nq & 1) ? Cc : S;
    cs = (nq & 1) ? S : Cc;
    if (nq == 1 || nq == 2) cs = -cs;
    if (nq >= 2) sn = -sn;
}

// GNU Radio's rotator (phase *= incr in float32, renormalised every 512 calls) in closed form: the
// float32 increment's true angle and magnitude drive a float64 model, rebased by the host every block.
template <class LT>
__device__ __forceinline__ float2 rotate_value(const LT &L, int64_t n, float vr, float vi)
{
#pragma clang fp contract(off)
    float pr = 1.f, pi = 0.f;
    bool have = false;
    if constexpr (LT::kHasRotRing) {
        if (L.rot_ring) {                                 // exact mode: the phase GNU Radio's iteration holds at output n
            const float2 p = L.rot_ring[(uint64_t)n & L.rot_mask];
            pr = p.x;
            pi = p.y;
            have = true;
        }
    }
    if (!have) {
        const int64_t dk = n - L.n_seg0;
        const int64_t r512 = n & ~(int64_t)511;
        const double ang = L.angle0 + (double)dk * L.dangle;
        const double lm = (r512 > L.n_seg0) ? (double)(n - r512) * L.dlogmag : L.logmag0 + (double)dk * L.dlogmag;
        double sn, cs;
        sincos_fast(ang, sn, cs);
        // |lm| is a few hundred times log|incr| ~ 1e-7: four series terms are exact to double rounding
        const double mag = fabs(lm) < 1e-3 ? 1.0 + lm * (1.0 + lm * (0.5 + lm * (1.0 / 6.0))) : exp(lm);
        pr = (float)(mag * cs);
        pi = (float)(mag * sn);
    }
    // rotator::rotate(): z = in * phase, float32 complex multiply, unfused
    float2 y;
    y.x = (vr * pr) - (vi * pi);
    y.y = (vr * pi) + (vi * pr);
    return y;
}

// The same model walked along outputs n0, n0 + step, n0 + 2 step, ...: the angle is linear in n, so after one
// sincos for the start and one for the step a phasor recurrence in float64 (four multiply-adds, ~2e-16 per step)
// replaces the polynomial pair per output.  tap_finalize_kernel rotates 11 rows per lane this way: with every bin of
// the 1600-bin bank tapped the per-output sincos was a third of that kernel's time (0.41 vs 0.27 ms idle).
template <class LT>
struct RotatorWalk {
    double c, s, cstep, sstep;
    // the magnitude model in float32: log|phase| is linear in n inside a renormalisation segment, |lm| < 512 log|incr| ~ 3e-5,
    // so 1 + lm in float32 carries it to ~1e-12 -- the float64 series and exp() of rotate_value() bought nothing here and
    // were a fifth of tap_finalize_kernel's vector instructions (round 5)
    float lm, dlm_step, dlm;
    int k512, step;              // n mod 512 (the next renormalisation resets the magnitude there)
    __device__ __forceinline__ RotatorWalk(const LT &L, int64_t n0, int step_) : step(step_)
    {
        sincos_fast(L.angle0 + (double)(n0 - L.n_seg0) * L.dangle, s, c);
        sincos_fast((double)step_ * L.dangle, sstep, cstep);
        dlm = (float)L.dlogmag;
        dlm_step = dlm * (float)step_;
        const int64_t r512 = n0 & ~(int64_t)511;
        k512 = (int)(n0 & 511);
        // inside the segment the host's model starts in: logmag0 + (n - n_seg0) dlogmag until n reaches the next multiple of 512
        lm = (r512 > L.n_seg0) ? (float)k512 * dlm : (float)(L.logmag0 + (double)(n0 - L.n_seg0) * L.dlogmag);
    }
    // rotate (vr, vi) by the phase at the current output
    __device__ __forceinline__ float2 rotate(const LT &, float vr, float vi) const
    {
#pragma clang fp contract(off)
        const float cf_ = (float)c, sf_ = (float)s;
        const float pr = fmaf(cf_, lm, cf_), pi = fmaf(sf_, lm, sf_);
        float2 y;
        y.x = (vr * pr) - (vi * pi);
        y.y = (vr * pi) + (vi * pr);
        return y;
    }
    __device__ __forceinline__ void advance()
    {
#pragma clang fp contract(off)
        const double c2 = fma(c, cstep, -(s * sstep));
        s = fma(s, cstep, c * sstep);
        c = c2;
        k512 += step;
        lm += dlm_step;
        if (k512 >= 512) {                       // crossed a renormalisation: the magnitude restarts at that multiple of 512
            k512 -= 512;
            lm = (float)k512 * dlm;
        }
    }
};

__device__ __forceinline__ void rotate_store(const ChanLaunch &L, int64_t k, float vr, float vi, uint64_t ring_mask)
{
    if (k < L.k_lo || k >= L.k_lo + L.n_k || k < L.k_abs0) return;
    const int64_t n = k - L.k_abs0;
    L.iq_ring[(uint64_t)n & ring_mask] = rotate_value(L, n, vr, vi);
}

}  // namespace

}  // namespace rcfx
