// fast_atan2f_gr.hpp -- gr::fast_atan2f (gr-runtime fast_atan2f.cc: 255-interval table + linear interpolation, octant
// fix-up) as every discriminator kernel runs it.  No HIP types in here: RCF_DEVFN is the function qualifier, so that the CPU
// suite can compile this very source with g++ and compare it with the oracle's restatement of GNU Radio's branches over
// millions of arguments (tests/test_device_atan_cpu.py) -- the device code is written as SELECTS, not as those branches.
#pragma once
#include <math.h>
#ifndef RCF_ATAN_PIN
#define RCF_ATAN_PIN 0
#endif
#ifndef RCF_DEVFN
#define RCF_DEVFN __device__ __forceinline__
#endif

namespace rcfx {

namespace {

// `lookup(index, t0, t1)` yields table[index] and table[index + 1], index in [0, 255]: a plain table (below) or wherever
// a kernel keeps the pairs (pfb5.hip: the spare LDS slots of its frame rows)
// RCP: the quotient as num * rcp(den) (v_rcp_f32: 1 ulp) instead of the correctly rounded division (eleven instructions with
// its scaling and fix-up): the argument of the table moves by <= 2 ulp, the angle by <= 2e-7 rad -- for the kernel that is
// bound by its vector instruction count (pfb5.hip: the discriminator fused into the bank).  Magnitudes below 1e-38 (denormal
// products: |bin| < 1e-19) are not handled -- v_rcp_f32 flushes them -- and come out as arbitrary angles; (0, 0) stays 0.
template <typename Lookup, bool RCP = false>
RCF_DEVFN float fast_atan2f_gr_lut(float y, float x, Lookup lookup)
{
#pragma clang fp contract(off)
    // The same values as gr::fast_atan2f's nested branches, written as selects: left as branches the compiler emits a
    // division on EACH side of (ya < xa) and both octant arms under exec masks -- with the lanes of a wavefront spread over
    // all octants every side runs, two 12-instruction divisions per call and ~15 scalar mask instructions around them (the
    // discriminator was half of tap_finalize_kernel's vector issue).  a - b == -(b - a) exactly, so every arm below is the
    // reference's own operation: base - PI = -(PI - base), -PI_2 + base = -(PI_2 - base), -PI_2 - base = -(PI_2 + base).
    const float TAN_MAP_RES = 0.003921569f;
    const float PI = 3.14159265358979323846f, PI_2 = 1.57079632679489661923f;
    const float ya = fabsf(y), xa = fabsf(x);
    const bool nonzero = (ya > 0.0f) || (xa > 0.0f);      // (0, 0) -> 0, selected at the end
    const bool y_small = ya < xa;
    const float num = y_small ? ya : xa, den = y_small ? xa : ya;
#if defined(__HIP_DEVICE_COMPILE__)
    const float z = RCP ? num * __builtin_amdgcn_rcpf(den) : num / den;
#else
    const float z = num / den;
#endif
    float alpha = z * 255.0f;
    const int index = ((int)alpha) & 0xff;                // (the NaN z of the (0, 0) case converts to 0: a valid table position)
    alpha = alpha - (float)index;
    float t0, t1;
    lookup(index, t0, t1);
#if defined(__HIP_DEVICE_COMPILE__) && RCF_ATAN_PIN
    asm volatile("" : "+v"(t0), "+v"(t1));                // (keeps the two table loads out of a conditional block: tools A/B)
#endif
    const float base = (z < TAN_MAP_RES) ? z : t0 + ((t1 - t0) * alpha);
    const bool xpos = x >= 0.0f, ypos = y >= 0.0f;
    const float r_x = xpos ? base : (PI - base);          // |x| > |y|
    const float r_y = PI_2 + (xpos ? -base : base);       // otherwise: PI_2 - base / PI_2 + base
    const float r = (xa > ya) ? r_x : r_y;
    const float angle = ypos ? r : -r;
    return nonzero ? angle : 0.0f;
}

RCF_DEVFN float fast_atan2f_gr(float y, float x, const float *tab)
{
    return fast_atan2f_gr_lut(y, x, [tab](int index, float &t0, float &t1) { t0 = tab[index]; t1 = tab[index + 1]; });
}

}  // namespace

}  // namespace rcfx
