// fft_core.hpp -- register-resident radix-2/4/8/16 DFTs and the LDS Stockham pass shared by the
// polyphase filterbank (pfb.hip) and the scan FFT (scan.hip).  gfx950 only: 64-wide wavefronts,
// 8-byte (ds_read_b64 / ds_write_b64) complex accesses, LDS rows padded by one complex per 16 so that
// the stride-R writes of the first pass spread over all banks (MI355X_MICROARCH.md LDS table:
// ds_write_b64 is serviced in 16-lane groups over 32 four-byte banks).
//
// Not a dense contraction: butterflies run on the vector ALU, no MFMA (BASELINE.json north_star).
#pragma once
#include <hip/hip_runtime.h>

namespace rcfx {

typedef float2 cf;

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
// RCF_EXPLICIT_FMA (pfb5.hip): that file is compiled without implicit contraction -- its template instantiations
// must round alike -- so the fused multiply-adds are spelled out there: four instructions per complex product
// instead of six.  Elsewhere the plain form stays: the compiler contracts it, and the scan kernels' SLP-packed
// (v_pk_*) version of it measures faster than the explicit-fma one.
#ifdef RCF_EXPLICIT_FMA
__device__ __forceinline__ cf cmul(cf a, cf b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cf cmulconj(cf a, cf b)   // a * conj(b)
{
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -(a.x * b.y)));
}
#else
__device__ __forceinline__ cf cmul(cf a, cf b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cf cmulconj(cf a, cf b)   // a * conj(b)
{
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
#endif
__device__ __forceinline__ cf cscale(cf a, float s) { return make_float2(a.x * s, a.y * s); }

// multiply by SIGN * i   (SIGN = -1: forward e^{-j..}, +1: inverse)
template <int SIGN>
__device__ __forceinline__ cf mul_si(cf a)
{
    return SIGN < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

// e^{SIGN * 2 pi i m / 16}, m = 0..15 (cos, |sin|) -- sign applied by the caller's template
__device__ __forceinline__ cf w16(int m, int sign)
{
    const float c[5] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f};
    int q = m & 3, o = (m >> 2) & 3;
    float cr = c[q], sr = c[4 - q];          // angle q*pi/8 in the first quadrant
    // rotate by o * 90 degrees
    float x, y;
    switch (o) {
        case 0: x = cr;  y = sr;  break;
        case 1: x = -sr; y = cr;  break;
        case 2: x = -cr; y = -sr; break;
        default: x = sr; y = -cr; break;
    }
    return make_float2(x, sign < 0 ? -y : y);
}

template <int SIGN>
__device__ __forceinline__ void dft2(cf &a, cf &b)
{
    cf t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// natural order in, natural order out
template <int SIGN>
__device__ __forceinline__ void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
    cf s02 = cadd(a0, a2), d02 = csub(a0, a2);
    cf s13 = cadd(a1, a3), d13 = mul_si<SIGN>(csub(a1, a3));
    a0 = cadd(s02, s13);
    a2 = csub(s02, s13);
    a1 = cadd(d02, d13);
    a3 = csub(d02, d13);
}

// In-register DFT of R points.  Input v[n] natural order; output frequency f is found in register
// Dft<R>::reg_of(f).  All indices are compile-time after unrolling.
template <int R, int SIGN> struct Dft;

template <int SIGN> struct Dft<1, SIGN> {
    static __device__ __forceinline__ void run(cf (&)[1]) {}
    static __device__ __forceinline__ constexpr int reg_of(int f) { return f; }
};
template <int SIGN> struct Dft<2, SIGN> {
    static __device__ __forceinline__ void run(cf (&v)[2]) { dft2<SIGN>(v[0], v[1]); }
    static __device__ __forceinline__ constexpr int reg_of(int f) { return f; }
};
template <int SIGN> struct Dft<4, SIGN> {
    static __device__ __forceinline__ void run(cf (&v)[4]) { dft4<SIGN>(v[0], v[1], v[2], v[3]); }
    static __device__ __forceinline__ constexpr int reg_of(int f) { return f; }
};
// 8 = 4 (n1) x 2 (n2): n = 2 n1 + n2, f = k1 + 4 k2
template <int SIGN> struct Dft<8, SIGN> {
    static __device__ __forceinline__ void run(cf (&v)[8])
    {
        dft4<SIGN>(v[0], v[2], v[4], v[6]);          // n2 = 0 -> k1 in v[2 k1]
        dft4<SIGN>(v[1], v[3], v[5], v[7]);          // n2 = 1 -> k1 in v[2 k1 + 1]
        const float h = 0.70710678118654752f;
        // W8^{k1}: k1=1: (h, S h), k1=2: S i, k1=3: (-h, S h)
        v[3] = cmul(v[3], make_float2(h, SIGN * h));
        v[5] = mul_si<SIGN>(v[5]);
        v[7] = cmul(v[7], make_float2(-h, SIGN * h));
        dft2<SIGN>(v[0], v[1]);
        dft2<SIGN>(v[2], v[3]);
        dft2<SIGN>(v[4], v[5]);
        dft2<SIGN>(v[6], v[7]);
    }
    // register t = 2 k1 + k2 holds f = k1 + 4 k2
    static __device__ __forceinline__ constexpr int reg_of(int f) { return 2 * (f & 3) + (f >> 2); }
};
// 16 = 4 (n1) x 4 (n2): n = 4 n1 + n2, f = k1 + 4 k2
template <int SIGN> struct Dft<16, SIGN> {
    static __device__ __forceinline__ void run(cf (&v)[16])
    {
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) dft4<SIGN>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
        // now v[4 k1 + n2] = A[n2][k1]; twiddle by W16^{n2 k1}
#pragma unroll
        for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
            for (int n2 = 1; n2 < 4; ++n2) v[4 * k1 + n2] = cmul(v[4 * k1 + n2], w16(n2 * k1, SIGN));
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4<SIGN>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    }
    // register t = 4 k1 + k2 holds f = k1 + 4 k2
    static __device__ __forceinline__ constexpr int reg_of(int f) { return 4 * (f & 3) + (f >> 2); }
};

// 5-point DFT, natural order in and out:  X[k] = sum_n x[n] e^{SIGN 2 pi i n k / 5}
template <int SIGN>
__device__ __forceinline__ void dft5(cf &x0, cf &x1, cf &x2, cf &x3, cf &x4)
{
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;     // cos(2 pi/5), cos(4 pi/5)
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;      // sin(2 pi/5), sin(4 pi/5)
    const cf t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    const cf m1 = make_float2(fmaf(c2, t2.x, fmaf(c1, t1.x, x0.x)), fmaf(c2, t2.y, fmaf(c1, t1.y, x0.y)));
    const cf m2 = make_float2(fmaf(c1, t2.x, fmaf(c2, t1.x, x0.x)), fmaf(c1, t2.y, fmaf(c2, t1.y, x0.y)));
    const cf r1 = make_float2(fmaf(s2, t4.x, s1 * t3.x), fmaf(s2, t4.y, s1 * t3.y));
    const cf r2 = make_float2(fmaf(-s1, t4.x, s2 * t3.x), fmaf(-s1, t4.y, s2 * t3.y));
    const cf i1 = mul_si<SIGN>(r1), i2 = mul_si<SIGN>(r2);                 // SIGN i (..)
    x0 = make_float2(x0.x + t1.x + t2.x, x0.y + t1.y + t2.y);
    x1 = cadd(m1, i1);
    x4 = csub(m1, i1);
    x2 = cadd(m2, i2);
    x3 = csub(m2, i2);
}

template <int SIGN> struct Dft<5, SIGN> {
    static __device__ __forceinline__ void run(cf (&v)[5]) { dft5<SIGN>(v[0], v[1], v[2], v[3], v[4]); }
    static __device__ __forceinline__ constexpr int reg_of(int f) { return f; }
};

// 5 M points, M = 4 here (any M coprime with 5 that Dft<M> covers): Good-Thomas prime-factor form -- no twiddles between the stages.
//   n = (M a + 5 b) mod 5M,  k = k1 (mod 5) = k2 (mod M):   X[k] = sum_b W_M^{b k2} sum_a W_5^{a k1} x[n(a, b)]
// Everything is register renaming after unrolling: stage 1 runs M five-point DFTs in place (register n(a, b)
// then holds Y[b][k1 = a]), stage 2 runs five M-point DFTs over b in place.
template <int M, int SIGN> struct DftPfa5 {
    static constexpr int R = 5 * M;
    static __device__ __forceinline__ void run(cf (&v)[R])
    {
#pragma unroll
        for (int b = 0; b < M; ++b)
            dft5<SIGN>(v[(5 * b) % R], v[(M + 5 * b) % R], v[(2 * M + 5 * b) % R], v[(3 * M + 5 * b) % R],
                       v[(4 * M + 5 * b) % R]);
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            cf w[M];
#pragma unroll
            for (int b = 0; b < M; ++b) w[b] = v[(M * a + 5 * b) % R];
            Dft<M, SIGN>::run(w);
#pragma unroll
            for (int b = 0; b < M; ++b) v[(M * a + 5 * b) % R] = w[b];
        }
    }
    // output k: k1 = k mod 5, k2 = k mod M; it sits where stage 2 left frequency k2 of row a = k1
    static __device__ __forceinline__ constexpr int reg_of(int f)
    {
        return (M * (f % 5) + 5 * Dft<M, SIGN>::reg_of(f % M)) % R;
    }
};
template <int SIGN> struct Dft<20, SIGN> : DftPfa5<4, SIGN> {};

// padded LDS index: one spare complex after every 16
__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 4); }
__host__ __device__ constexpr int lds_padded_len(int n) { return n + (n >> 4); }

// powers w^1..w^(R-1) from w (binary tree: depth <= 4 multiplications for R = 16)
template <int R>
__device__ __forceinline__ void twiddle_powers(cf w1, cf (&w)[R])
{
    w[0] = make_float2(1.f, 0.f);
    w[1] = w1;
#pragma unroll
    for (int t = 2; t < R; ++t) w[t] = (t & 1) ? cmul(w[t - 1], w1) : cmul(w[t / 2], w[t / 2]);
}

// One Stockham radix-R butterfly of an N-point transform held in (padded) LDS, in place:
//   read phase  : v[t] = buf[pad(j + t N/R)] * W_{Ns R}^{k t},  k = j mod Ns
//   (caller puts a barrier between read and write phases: every butterfly reads before any writes)
//   write phase : buf[pad((j / Ns) Ns R + k + f Ns)] = X[f]
template <int N, int R, int SIGN>
struct StockhamPass {
    // Padded indices are written as "pad(first) + t * constant" wherever the stride is a multiple of
    // 16 (pad(a + 16c) = pad(a) + 17c): the compiler then folds every access of a butterfly into ONE
    // address VGPR plus immediate offsets.  The naive pad(j + t*stride) form costs one address VGPR per
    // access (the shift hides the linearity), ~64 VGPRs across a two-pass kernel.
    static __device__ __forceinline__ void load(const cf *buf, int j, cf (&v)[R])
    {
        if constexpr ((N / R) % 16 == 0) {
            const cf *b = buf + lds_pad(j);
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = b[t * ((N / R) + (N / R) / 16)];
        } else {
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = buf[lds_pad(j + t * (N / R))];
        }
    }
    // tw: table of e^{SIGN 2 pi i n / N}, n in [0, N)
    static __device__ __forceinline__ void twiddle(const cf *tw, int Ns, int j, cf (&v)[R])
    {
        if (Ns == 1) return;
        const int k = j & (Ns - 1);
        cf w[R];
        twiddle_powers<R>(tw[k * (N / (Ns * R))], w);
#pragma unroll
        for (int t = 1; t < R; ++t) v[t] = cmul(v[t], w[t]);
    }
    template <int NS>
    static __device__ __forceinline__ void store_t(cf *buf, int j, cf (&v)[R])
    {
        const int k = j & (NS - 1);
        const int j0 = (j - k) * R + k;
        if constexpr (NS % 16 == 0) {
            cf *b = buf + lds_pad(j0);
#pragma unroll
            for (int f = 0; f < R; ++f) b[f * (NS + NS / 16)] = v[Dft<R, SIGN>::reg_of(f)];
        } else if constexpr (NS == 1 && R == 16) {
            cf *b = buf + lds_pad(j0);            // j0 = 16 j: (j0 + f) >> 4 == j for f < 16
#pragma unroll
            for (int f = 0; f < R; ++f) b[f] = v[Dft<R, SIGN>::reg_of(f)];
        } else {
#pragma unroll
            for (int f = 0; f < R; ++f) buf[lds_pad(j0 + f * NS)] = v[Dft<R, SIGN>::reg_of(f)];
        }
    }
    static __device__ __forceinline__ void store(cf *buf, int Ns, int j, cf (&v)[R])
    {
        const int k = j & (Ns - 1);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int f = 0; f < R; ++f) buf[lds_pad(j0 + f * Ns)] = v[Dft<R, SIGN>::reg_of(f)];
    }
};

}  // namespace rcfx
