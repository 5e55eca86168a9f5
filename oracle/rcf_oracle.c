#define _GNU_SOURCE
/*
 * rcf_oracle.c -- plain-C CPU restatement of the radiocapture-rf channelizer / discriminator /
 * scan hot path.  TEST INFRASTRUCTURE ONLY: loaded (ctypes) by tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py.  The product library never links or calls it.
 *
 * PARITY UNPINNED: GNU Radio 3.8 (gr-filter, gr-analog, gr-fft, gr-blocks), VOLK and FFTW are
 * un-vendored third-party dependencies of the reference and cannot be built or imported here; the
 * reference has no tests or golden vectors.  This file restates the published GR 3.8 block
 * semantics (SURVEY.md 8(c) "[GR-spec]") independently of oracle/grspec.py (numpy); the two are
 * checked against each other and against analytic known-answer tests.
 *
 * Structure mirrors the reference's CPU path on purpose (it is the timed "port" baseline):
 * one T-tap complex FIR + rotator per channel over the whole wideband stream
 * (/root/reference/rc_frontend/channel.py:31-38), one discriminator per channel
 * (/root/reference/p25_control_demod.py:120-121), and the fft_vector.py:37-60 scan chain followed
 * by fft_peak_detection.py:38-73.
 *
 * Build: see oracle/Makefile  (gcc -O3 -march=native -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RO_WIN_HAMMING 0
#define RO_WIN_BLACKMAN 2
#define RO_WIN_BLACKMAN_HARRIS 5

/* ---------------------------------------------------------------- windows (gr-fft window.cc) */
static void ro_window(int type, int ntaps, float *w)
{
    double M = (double)(ntaps - 1);
    for (int n = 0; n < ntaps; n++) {
        double v;
        if (type == RO_WIN_HAMMING) {
            v = 0.54 - 0.46 * cos((2.0 * M_PI * n) / M);
        } else if (type == RO_WIN_BLACKMAN) {
            /* coswindow(): float coefficients, double cosines */
            v = (double)0.42f - (double)0.5f * cos((2.0 * M_PI * n) / M)
              + (double)0.08f * cos((4.0 * M_PI * n) / M);
        } else { /* 92 dB 4-term Blackman-Harris */
            v = (double)0.35875f - (double)0.48829f * cos((2.0 * M_PI * n) / M)
              + (double)0.14128f * cos((4.0 * M_PI * n) / M)
              - (double)0.01168f * cos((6.0 * M_PI * n) / M);
        }
        w[n] = (float)v;
    }
}

int ro_window_f32(int type, int ntaps, float *w) { ro_window(type, ntaps, w); return 0; }

/* ---------------------------------------------------------------- firdes.low_pass_2 */
int ro_ntaps_windes(double fs, double tw, double att_db)
{
    int n = (int)(att_db * fs / (22.0 * tw));
    if ((n & 1) == 0) n++;
    return n;
}

/* returns ntaps (or -needed if cap too small) */
int ro_low_pass_2(double gain, double fs, double fc, double tw, double att_db, int wintype,
                  float *taps, int cap)
{
    int ntaps = ro_ntaps_windes(fs, tw, att_db);
    if (ntaps > cap) return -ntaps;
    float *w = (float *)malloc(sizeof(float) * (size_t)ntaps);
    ro_window(wintype, ntaps, w);
    int M = (ntaps - 1) / 2;
    double fwT0 = 2.0 * M_PI * fc / fs;
    for (int n = -M; n <= M; n++) {
        if (n == 0) taps[n + M] = (float)(fwT0 / M_PI * w[n + M]);
        else        taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w[n + M]);
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; n++) fmax += 2.0 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
    free(w);
    return ntaps;
}

/* ---------------------------------------------------------------- xlating FIR */
/* GR build_composite_fir(): ctaps interleaved re,im; incr[2] */
void ro_xlating_composite(const float *taps, int T, int D, double f0, double fs,
                          float *ctaps, float *incr)
{
    float fwT0 = (float)(2.0 * M_PI * f0 / fs);
    for (unsigned i = 0; i < (unsigned)T; i++) {
        float th = (float)i * fwT0;
        ctaps[2 * i]     = taps[i] * cosf(th);
        ctaps[2 * i + 1] = taps[i] * sinf(th);
    }
    float a = -fwT0 * (float)D;
    incr[0] = cosf(a);
    incr[1] = sinf(a);
}


/* dot products: cr = reversed composite taps, xs = first input sample of the window */
static void ro_dot_f64(const float *cr, const float *xs, int T, float *vr, float *vi)
{
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < T; j++) {
        double a = cr[2 * j], b = cr[2 * j + 1], c = xs[2 * j], d = xs[2 * j + 1];
        sr += a * c - b * d;
        si += a * d + b * c;
    }
    *vr = (float)sr; *vi = (float)si;
}

/* float32, 8 partial sums per component (the SIMD order a VOLK AVX2/FMA kernel uses); this is the
 * only function in the file allowed to contract mul+add into FMA (file is built -ffp-contract=off) */
__attribute__((optimize("fp-contract=fast")))
static void ro_dot_f32(const float *cr, const float *xs, int T, float *vr, float *vi)
{
    float sr[8] = {0}, si[8] = {0};
    int j = 0;
    for (; j + 8 <= T; j += 8) {
        for (int l = 0; l < 8; l++) {
            float a = cr[2 * (j + l)], b = cr[2 * (j + l) + 1];
            float c = xs[2 * (j + l)], d = xs[2 * (j + l) + 1];
            sr[l] += a * c - b * d;
            si[l] += a * d + b * c;
        }
    }
    float tr = 0.f, ti = 0.f;
    for (; j < T; j++) {
        float a = cr[2 * j], b = cr[2 * j + 1], c = xs[2 * j], d = xs[2 * j + 1];
        tr += a * c - b * d;
        ti += a * d + b * c;
    }
    for (int l = 0; l < 8; l++) { tr += sr[l]; ti += si[l]; }
    *vr = tr; *vi = ti;
}

typedef struct {
    float phase_re, phase_im;   /* rotator phase */
    uint32_t counter;           /* rotator call counter */
} ro_rot_state;

/*
 * One channel: y[n] = rot[n] * sum_i c[i] x[nD - i], n = n_first .. n_first + n_out - 1 where the
 * caller passes xh = pointer to sample index 0 of a buffer that has T-1 valid samples (history or
 * zeros) BEFORE it, i.e. xh[-(T-1)] is readable.  Samples are interleaved cf32.
 * acc_double != 0 -> accumulate the dot product in double (oracle mode); else float32 8-lane
 * partial sums (VOLK-like SIMD order; baseline mode).
 */
void ro_xlating_fir_ccc(const float *xh, int64_t n_first, int64_t n_out, int D,
                        const float *ctaps, int T, const float *incr, ro_rot_state *st,
                        float *y, int acc_double)
{
    /* reversed taps so the inner loop walks x forward */
    float *cr = (float *)malloc(sizeof(float) * 2 * (size_t)T);
    for (int j = 0; j < T; j++) {
        cr[2 * j] = ctaps[2 * (T - 1 - j)];
        cr[2 * j + 1] = ctaps[2 * (T - 1 - j) + 1];
    }
    float pr = st->phase_re, pi = st->phase_im;
    uint32_t cnt = st->counter;
    const float ir = incr[0], ii = incr[1];
    for (int64_t k = 0; k < n_out; k++) {
        const float *xs = xh + 2 * ((n_first + k) * (int64_t)D - (T - 1));
        float vr, vi;
        if (acc_double) ro_dot_f64(cr, xs, T, &vr, &vi);
        else            ro_dot_f32(cr, xs, T, &vr, &vi);
        /* rotator::rotate() */
        cnt++;
        {
            float m0 = vr * pr, m1 = vi * pi, m2 = vr * pi, m3 = vi * pr;
            y[2 * k] = m0 - m1;
            y[2 * k + 1] = m2 + m3;
        }
        {
            float m0 = pr * ir, m1 = pi * ii, m2 = pr * ii, m3 = pi * ir;
            float nr = m0 - m1, ni = m2 + m3;
            pr = nr; pi = ni;
        }
        if ((cnt % 512u) == 0) {
            float mag = hypotf(pr, pi);
            pr /= mag; pi /= mag;
        }
    }
    st->phase_re = pr; st->phase_im = pi; st->counter = cnt;
    free(cr);
}

/* ---------------------------------------------------------------- fast_atan2f + quad demod */
static float ro_atan_table[257];
static int ro_atan_table_ready = 0;
static void ro_atan_init(void)
{
    if (ro_atan_table_ready) return;
    /* GNU Radio's table is 257 literals of 7 significant digits: atan(i / 255) printed with %.6e (see grspec.py) */
    char buf[32];
    for (int i = 0; i < 257; i++) {
        snprintf(buf, sizeof buf, "%.6e", i < 256 ? atan((double)i / 255.0) : M_PI / 4.0);
        ro_atan_table[i] = (float)strtod(buf, NULL);
    }
    ro_atan_table_ready = 1;
}

float ro_fast_atan2f(float y, float x)
{
    ro_atan_init();
    const float TAN_MAP_RES = 0.003921569f;
    float y_abs = fabsf(y), x_abs = fabsf(x), z, base_angle, angle;
    if (!((y_abs > 0.0f) || (x_abs > 0.0f))) return 0.0f;
    z = (y_abs < x_abs) ? y_abs / x_abs : x_abs / y_abs;
    if (z < TAN_MAP_RES) {
        base_angle = z;
    } else {
        float alpha = z * 255.0f;
        int index = ((int)alpha) & 0xff;
        alpha -= (float)index;
        base_angle = ro_atan_table[index];
        base_angle += (ro_atan_table[index + 1] - ro_atan_table[index]) * alpha;
    }
    if (x_abs > y_abs) {
        if (x >= 0.0f) angle = (y >= 0.0f) ? base_angle : -base_angle;
        else {
            angle = (float)M_PI;
            angle = (y >= 0.0f) ? angle - base_angle : base_angle - angle;
        }
    } else {
        if (y >= 0.0f) {
            angle = (float)M_PI_2;
            angle = (x >= 0.0f) ? angle - base_angle : angle + base_angle;
        } else {
            angle = -(float)M_PI_2;
            angle = (x >= 0.0f) ? angle + base_angle : angle - base_angle;
        }
    }
    return angle;
}

/* prev[2] in/out: history sample */
void ro_quad_demod_cf(const float *x, int64_t n, float gain, float *prev, float *out)
{
    float br = prev[0], bi = prev[1];
    for (int64_t k = 0; k < n; k++) {
        float ar = x[2 * k], ai = x[2 * k + 1];
        float p0 = ar * br, p1 = ai * bi, p2 = ai * br, p3 = ar * bi;
        float tr = p0 + p1;     /* a * conj(b) */
        float ti = p2 - p3;
        out[k] = gain * ro_fast_atan2f(ti, tr);
        br = ar; bi = ai;
    }
    prev[0] = br; prev[1] = bi;
}

/* ---------------------------------------------------------------- channel bank (baseline shape) */
/*
 * The reference's structure: C independent flowgraphs, each running the full FIR over the whole
 * wideband stream, then a discriminator.  x has T-1 zero samples of history implied (first call).
 * ctaps: C x T interleaved; incr: C x 2; y: C x n_out interleaved; fm: C x n_out (may be NULL).
 * Threads: one channel per OpenMP task (GR: one thread per block/flowgraph).
 */
int ro_channel_bank(const float *x, int64_t n_in, int D, int T, int C,
                    const float *ctaps, const float *incr, const float *gains,
                    float *y, float *fm, int acc_double, int nthreads)
{
    int64_t n_out = n_in > 0 ? (n_in - 1) / D + 1 : 0;
    float *xp = (float *)calloc((size_t)(n_in + T - 1) * 2, sizeof(float));
    if (!xp) return -1;
    memcpy(xp + 2 * (size_t)(T - 1), x, sizeof(float) * 2 * (size_t)n_in);
    const float *xh = xp + 2 * (size_t)(T - 1);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int c = 0; c < C; c++) {
        ro_rot_state st = {1.0f, 0.0f, 0};
        ro_xlating_fir_ccc(xh, 0, n_out, D, ctaps + 2 * (size_t)c * T, T, incr + 2 * c, &st,
                           y + 2 * (size_t)c * n_out, acc_double);
        if (fm) {
            float prev[2] = {0.f, 0.f};
            ro_quad_demod_cf(y + 2 * (size_t)c * n_out, n_out, gains[c], prev, fm + (size_t)c * n_out);
        }
    }
    free(xp);
    return 0;
}

/*
 * cpu_baseline leg of bench.py: the same per-channel arithmetic, laid out so that the all-core figure is defensible.
 *   - n_threads workers, worker w pinned to cpu_ids[w] (one per PHYSICAL core; NULL = unpinned)
 *   - every worker owns a PRIVATE copy of the wideband tile, first-touched by itself -- the reference gives every
 *     channel flowgraph its own copy of the stream too (zeromq.pub_sink -> one sub_source per channel.py:29)
 *   - the tile (n_tile samples, periodic: its own tail is its history) is walked `passes` times per channel:
 *     passes * n_tile samples of signal per channel without passes * n_tile * 8 bytes of memory per thread
 *   - tiled == 0: the reference's structure -- channel outer, the whole stream per channel (cpt channels in turn)
 *     tiled != 0: "best CPU" -- time blocks of tile_block samples outer, the thread's cpt channels inner: the
 *     stream comes from DRAM once per thread instead of once per channel (NOT how GNU Radio runs; labelled so)
 *   - only the region between the two barriers is timed (allocation, first touch and thread start are outside)
 * Returns the wall seconds of the timed region; checksum[w] keeps the work observable; y0_last (optional, n_tile / D
 * complex) receives worker 0's first channel over the LAST pass, so that a test can hold what is timed against
 * ro_channel_bank on the same stream.
 */
#include <sched.h>
#include <time.h>
static double ro_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double ro_bank_bench(const float *tile, int64_t n_tile, int passes, int D, int T, int n_threads, int cpt,
                     const float *ctaps, const float *incr, const float *gains, const int *cpu_ids,
                     int tiled, int64_t tile_block, float *checksum, float *y0_last)
{
    if (n_tile % D || n_tile < T || n_threads < 1 || cpt < 1 || passes < 1) return -1.0;
    if (tile_block <= 0 || tile_block % D) tile_block = n_tile;
    const int64_t n_out = n_tile / D;
    double t_begin = 0.0, t_end = 0.0;
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads) reduction(+ : failed)
#endif
    {
#ifdef _OPENMP
        const int w = omp_get_thread_num();
#else
        const int w = 0;
#endif
        if (cpu_ids) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpu_ids[w], &set);
            (void)sched_setaffinity(0, sizeof set, &set);
        }
        float *xp = (float *)malloc(sizeof(float) * 2 * (size_t)(n_tile + T - 1));
        float *y = (float *)malloc(sizeof(float) * 2 * (size_t)n_out);
        float *fm = (float *)malloc(sizeof(float) * (size_t)n_out);
        if (!xp || !y || !fm) failed = 1;
        if (!failed) {
            /* history = the tile's own tail (periodic stream); first touch by this thread */
            memcpy(xp, tile + 2 * (size_t)(n_tile - (T - 1)), sizeof(float) * 2 * (size_t)(T - 1));
            memcpy(xp + 2 * (size_t)(T - 1), tile, sizeof(float) * 2 * (size_t)n_tile);
            memset(y, 0, sizeof(float) * 2 * (size_t)n_out);
            memset(fm, 0, sizeof(float) * (size_t)n_out);
        }
        const float *xh = xp + 2 * (size_t)(T - 1);
        float acc = 0.f;
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
        t_begin = ro_now();
#ifdef _OPENMP
#pragma omp barrier
#endif
        if (!failed) {
            ro_rot_state st[64];
            float prev[64][2];
            const int nc = cpt > 64 ? 64 : cpt;
            for (int c = 0; c < nc; c++) { st[c].phase_re = 1.f; st[c].phase_im = 0.f; st[c].counter = 0; prev[c][0] = prev[c][1] = 0.f; }
            for (int p = 0; p < passes; p++) {
                if (!tiled) {
                    for (int c = 0; c < nc; c++) {
                        const size_t ch = (size_t)w * cpt + c;
                        ro_xlating_fir_ccc(xh, 0, n_out, D, ctaps + 2 * ch * T, T, incr + 2 * ch, &st[c], y, 0);
                        ro_quad_demod_cf(y, n_out, gains[ch], prev[c], fm);
                        acc += fm[n_out - 1] + y[0];
                        if (y0_last && w == 0 && c == 0 && p == passes - 1) memcpy(y0_last, y, sizeof(float) * 2 * (size_t)n_out);
                    }
                } else {
                    for (int64_t s0 = 0; s0 < n_tile; s0 += tile_block) {
                        const int64_t k0 = s0 / D, k1 = (s0 + tile_block < n_tile ? s0 + tile_block : n_tile) / D;
                        for (int c = 0; c < nc; c++) {
                            const size_t ch = (size_t)w * cpt + c;
                            ro_xlating_fir_ccc(xh, k0, k1 - k0, D, ctaps + 2 * ch * T, T, incr + 2 * ch, &st[c], y + 2 * k0, 0);
                            ro_quad_demod_cf(y + 2 * k0, k1 - k0, gains[ch], prev[c], fm + k0);
                            acc += fm[k1 - 1] + y[2 * k0];
                            if (y0_last && w == 0 && c == 0 && p == passes - 1)
                                memcpy(y0_last + 2 * k0, y + 2 * k0, sizeof(float) * 2 * (size_t)(k1 - k0));
                        }
                    }
                }
            }
        }
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
        t_end = ro_now();
        if (checksum) checksum[w] = acc;
        free(xp); free(y); free(fm);
    }
    return failed ? -1.0 : t_end - t_begin;
}

/* streaming read bandwidth of the host with the same pinning: every worker sums its own first-touched buffer `reps`
 * times; returns bytes per second over all workers (what bounds the all-core run of the reference's structure) */
double ro_read_bandwidth(int64_t bytes_per_thread, int reps, int n_threads, const int *cpu_ids, float *sink)
{
    const int64_t n = bytes_per_thread / (int64_t)sizeof(float);
    double t_begin = 0.0, t_end = 0.0;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
#ifdef _OPENMP
        const int w = omp_get_thread_num();
#else
        const int w = 0;
#endif
        if (cpu_ids) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpu_ids[w], &set);
            (void)sched_setaffinity(0, sizeof set, &set);
        }
        float *b = (float *)malloc(sizeof(float) * (size_t)n);
        for (int64_t i = 0; b && i < n; i++) b[i] = (float)(i & 7);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
        t_begin = ro_now();
#ifdef _OPENMP
#pragma omp barrier
#endif
        for (int r = 0; b && r < reps; r++)
            for (int64_t i = 0; i + 16 <= n; i += 16) {
                s0 += b[i] + b[i + 4] + b[i + 8] + b[i + 12];
                s1 += b[i + 1] + b[i + 5] + b[i + 9] + b[i + 13];
                s2 += b[i + 2] + b[i + 6] + b[i + 10] + b[i + 14];
                s3 += b[i + 3] + b[i + 7] + b[i + 11] + b[i + 15];
            }
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
        t_end = ro_now();
        if (sink) sink[w] = s0 + s1 + s2 + s3;
        free(b);
    }
    return (double)bytes_per_thread * reps * n_threads / (t_end - t_begin);
}

/* ---------------------------------------------------------------- bounds on what cannot be pinned (tests only)
 * GNU Radio / VOLK leave three things to the build and the machine: the order in which a dot product is summed
 * (SIMD width of the VOLK kernel picked at run time), whether the compiler contracted the rotator's complex
 * multiply into fused multiply-adds, and the polynomial inside volk_32f_log2.  These restate the alternatives so that
 * tests/test_oracle_unpinned_bounds.py can bound how far each moves the outputs. */
static void ro_dot_mode(const float *cr, const float *xs, int T, int mode, float *vr, float *vi)
{
    if (mode == 0) { ro_dot_f32(cr, xs, T, vr, vi); return; }
    if (mode == 4) { ro_dot_f64(cr, xs, T, vr, vi); return; }
    if (mode == 1) {                                   /* strictly sequential float32 (VOLK generic) */
        float tr = 0.f, ti = 0.f;
        for (int j = 0; j < T; j++) {
            float a = cr[2 * j], b = cr[2 * j + 1], c = xs[2 * j], d = xs[2 * j + 1];
            float p0 = a * c, p1 = b * d, p2 = a * d, p3 = b * c;
            tr += p0 - p1;
            ti += p2 + p3;
        }
        *vr = tr; *vi = ti;
        return;
    }
    if (mode == 3) {                                   /* 16 lanes (AVX-512 width), lanes added in order at the end */
        float sr[16] = {0}, si[16] = {0};
        int j = 0;
        for (; j + 16 <= T; j += 16)
            for (int l = 0; l < 16; l++) {
                float a = cr[2 * (j + l)], b = cr[2 * (j + l) + 1], c = xs[2 * (j + l)], d = xs[2 * (j + l) + 1];
                float p0 = a * c, p1 = b * d, p2 = a * d, p3 = b * c;
                sr[l] += p0 - p1;
                si[l] += p2 + p3;
            }
        float tr = 0.f, ti = 0.f;
        for (; j < T; j++) {
            float a = cr[2 * j], b = cr[2 * j + 1], c = xs[2 * j], d = xs[2 * j + 1];
            float p0 = a * c, p1 = b * d, p2 = a * d, p3 = b * c;
            tr += p0 - p1;
            ti += p2 + p3;
        }
        for (int l = 0; l < 16; l++) { tr += sr[l]; ti += si[l]; }
        *vr = tr; *vi = ti;
        return;
    }
    /* mode 2: pairwise (tree) summation of the float32 products */
    float *pr = (float *)malloc(sizeof(float) * 2 * (size_t)T);
    for (int j = 0; j < T; j++) {
        float a = cr[2 * j], b = cr[2 * j + 1], c = xs[2 * j], d = xs[2 * j + 1];
        float p0 = a * c, p1 = b * d, p2 = a * d, p3 = b * c;
        pr[2 * j] = p0 - p1;
        pr[2 * j + 1] = p2 + p3;
    }
    for (int n = T; n > 1; n = (n + 1) / 2)
        for (int j = 0; j < n / 2; j++) {
            pr[2 * j] = pr[2 * (2 * j)] + pr[2 * (2 * j + 1)];
            pr[2 * j + 1] = pr[2 * (2 * j) + 1] + pr[2 * (2 * j + 1) + 1];
            if (j == n / 2 - 1 && (n & 1)) { pr[2 * (j + 1)] = pr[2 * (n - 1)]; pr[2 * (j + 1) + 1] = pr[2 * (n - 1) + 1]; }
        }
    *vr = pr[0]; *vi = pr[1];
    free(pr);
}

/* v[k] = sum_i ctaps[i] x[kD - i] (no rotator), dot product summed in `mode` (0: 8 lanes, 1: sequential, 2: pairwise,
 * 3: 16 lanes, 4: float64); x has T-1 zeros of history */
int ro_fir_summation(const float *x, int64_t n_in, int D, const float *ctaps, int T, int mode, float *v)
{
    const int64_t n_out = n_in > 0 ? (n_in - 1) / D + 1 : 0;
    float *xp = (float *)calloc((size_t)(n_in + T - 1) * 2, sizeof(float));
    float *cr = (float *)malloc(sizeof(float) * 2 * (size_t)T);
    if (!xp || !cr) { free(xp); free(cr); return -1; }
    memcpy(xp + 2 * (size_t)(T - 1), x, sizeof(float) * 2 * (size_t)n_in);
    for (int j = 0; j < T; j++) { cr[2 * j] = ctaps[2 * (T - 1 - j)]; cr[2 * j + 1] = ctaps[2 * (T - 1 - j) + 1]; }
    for (int64_t k = 0; k < n_out; k++) ro_dot_mode(cr, xp + 2 * (size_t)(k * D), T, mode, v + 2 * k, v + 2 * k + 1);
    free(xp); free(cr);
    return 0;
}

/* gr::blocks::rotator phases of outputs 0 .. n-1.  fma_mode 0: every product rounded (x86-64 baseline build: what the
 * oracle and the HIP kernels iterate); 1: the complex multiply contracted the way GCC / clang -ffp-contract=fast do
 * it on an FMA machine, re = fma(a, c, -(b d)), im = fma(a, d, b c) */
void ro_rotator_phases(const float *incr, int64_t n, int fma_mode, float *out)
{
    float pr = 1.f, pi = 0.f;
    const float ir = incr[0], ii = incr[1];
    for (int64_t k = 0; k < n; k++) {
        out[2 * k] = pr; out[2 * k + 1] = pi;
        float nr, ni;
        if (fma_mode) {
            float m1 = pi * ii, m3 = pi * ir;
            nr = fmaf(pr, ir, -m1);
            ni = fmaf(pr, ii, m3);
        } else {
            float m0 = pr * ir, m1 = pi * ii, m2 = pr * ii, m3 = pi * ir;
            nr = m0 - m1; ni = m2 + m3;
        }
        pr = nr; pi = ni;
        if (((uint32_t)(k + 1) % 512u) == 0) {
            float mag = hypotf(pr, pi);
            pr /= mag; pi /= mag;
        }
    }
}

int ro_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- FFT (float32, radix-2 DIT) */
static void ro_fft_inplace(float *re, float *im, int N, const float *twr, const float *twi)
{
    /* bit reversal */
    for (int i = 1, j = 0; i < N; i++) {
        int bit = N >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { float t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= N; len <<= 1) {
        int half = len >> 1, step = N / len;
        for (int i = 0; i < N; i += len) {
            for (int k = 0; k < half; k++) {
                float wr = twr[k * step], wi = twi[k * step];
                float ur = re[i + k], ui = im[i + k];
                float xr = re[i + k + half], xi = im[i + k + half];
                float vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
                re[i + k] = ur + vr; im[i + k] = ui + vi;
                re[i + k + half] = ur - vr; im[i + k + half] = ui - vi;
            }
        }
    }
}

/*
 * fft_vector.py:37-60 -- returns the one float32[N] vector of frame n_frames-1:
 *   blackmanharris(N) window -> forward FFT -> fftshift -> |X|^2 -> log10()+1 (log2f * 1/log2(10))
 *   -> float32 running sum over avg_len frames (add, emit, subtract oldest).
 * x: n_frames*N interleaved cf32.  N power of two.
 */
int ro_scan_chain(const float *x, int N, int n_frames, int avg_len, float *out)
{
    if (N < 2 || (N & (N - 1))) return -1;
    float *win = (float *)malloc(sizeof(float) * (size_t)N);
    float *twr = (float *)malloc(sizeof(float) * (size_t)N / 2);
    float *twi = (float *)malloc(sizeof(float) * (size_t)N / 2);
    float *re = (float *)malloc(sizeof(float) * (size_t)N);
    float *im = (float *)malloc(sizeof(float) * (size_t)N);
    float *sum = (float *)calloc((size_t)N, sizeof(float));
    float *ring = (float *)calloc((size_t)N * (size_t)avg_len, sizeof(float));
    if (!win || !twr || !twi || !re || !im || !sum || !ring) return -2;
    ro_window(RO_WIN_BLACKMAN_HARRIS, N, win);
    for (int k = 0; k < N / 2; k++) {
        twr[k] = (float)cos(-2.0 * M_PI * k / N);
        twi[k] = (float)sin(-2.0 * M_PI * k / N);
    }
    const float scale = (float)(1.0 / log2(10.0));
    const int half = (N + 1) / 2;
    for (int f = 0; f < n_frames; f++) {
        const float *xf = x + 2 * (size_t)f * N;
        for (int n = 0; n < N; n++) { re[n] = xf[2 * n] * win[n]; im[n] = xf[2 * n + 1] * win[n]; }
        ro_fft_inplace(re, im, N, twr, twi);
        float *slot = ring + (size_t)(f % avg_len) * N;   /* holds frame f-avg_len -> overwritten below */
        for (int k = 0; k < N; k++) {
            int src = (k + half) % N;                      /* fftshift: out[k] = X[(k+ceil(N/2)) % N] */
            float p = re[src] * re[src] + im[src] * im[src];
            float l2 = log2f(p);
            if (isinf(l2)) l2 = copysignf(127.0f, l2);
            float v = l2 * scale + 1.0f;
            float s = sum[k] + v;
            if (f == n_frames - 1) out[k] = s;
            /* subtract frame f-(avg_len-1): stored in slot (f+1)%avg_len */
            if (f - (avg_len - 1) >= 0) {
                float old = (avg_len == 1) ? v : ring[(size_t)((f + 1) % avg_len) * N + k];
                s -= old;
            }
            sum[k] = s;
            slot[k] = v;
        }
    }
    free(win); free(twr); free(twi); free(re); free(im); free(sum); free(ring);
    return 0;
}

/* ---------------------------------------------------------------- find_peaks restatement */
/*
 * fft_peak_detection.py:54-73 on a float32 spectrum:
 *   data += |min(data)| (float32); mean = sequential float64 sum / n;
 *   scipy.signal.find_peaks(data, width=[min_w,max_w], prominence=prom) in float64;
 *   keep data[line] > 2*mean.
 * Writes up to cap line indices (ascending); returns the count (may exceed cap).
 */
int64_t ro_peak_detect(const float *spectrum, int64_t n, double min_w, double max_w, double prom_min,
                       int64_t *lines, int64_t cap, double *mean_out)
{
    if (n <= 0) { if (mean_out) *mean_out = 0.0; return 0; }
    double *x = (double *)malloc(sizeof(double) * (size_t)n);
    float mn = spectrum[0];
    for (int64_t i = 1; i < n; i++) if (spectrum[i] < mn) mn = spectrum[i];
    float shift = fabsf(mn);
    double total = 0.0;
    for (int64_t i = 0; i < n; i++) {
        float v = spectrum[i] + shift;
        x[i] = (double)v;
        total += (double)v;
    }
    double mean = total / (double)n;
    if (mean_out) *mean_out = mean;
    int64_t count = 0;
    int64_t i = 1, i_max = n - 1;
    while (i < i_max) {
        if (x[i - 1] < x[i]) {
            int64_t ahead = i + 1;
            while (ahead < i_max && x[ahead] == x[i]) ahead++;
            if (x[ahead] < x[i]) {
                int64_t p = (i + ahead - 1) / 2;
                double h = x[p];
                /* prominence */
                int64_t j = p, lb = p, rb = p;
                double lmin = h, rmin = h;
                while (j >= 0 && x[j] <= h) { if (x[j] < lmin) { lmin = x[j]; lb = j; } j--; }
                j = p;
                while (j <= n - 1 && x[j] <= h) { if (x[j] < rmin) { rmin = x[j]; rb = j; } j++; }
                double prom = h - (lmin > rmin ? lmin : rmin);
                if (prom >= prom_min) {
                    double height = h - prom * 0.5;
                    j = p;
                    while (lb < j && height < x[j]) j--;
                    double lip = (double)j;
                    if (x[j] < height) lip += (height - x[j]) / (x[j + 1] - x[j]);
                    j = p;
                    while (j < rb && height < x[j]) j++;
                    double rip = (double)j;
                    if (x[j] < height) rip -= (height - x[j]) / (x[j - 1] - x[j]);
                    double w = rip - lip;
                    if (min_w <= w && w <= max_w && x[p] > mean * 2) {
                        if (count < cap) lines[count] = p;
                        count++;
                    }
                }
                i = ahead;
            }
        }
        i++;
    }
    free(x);
    return count;
}
