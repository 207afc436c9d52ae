/*
 * cpu_bench.c -- the oracle's CPU legs timed from C (no interpreter in the loop).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE: bench.py's cpu_baseline leg runs this binary on the GPU box's host cores and
 * quotes its figures next to the GPU's.  It times the plain-C restatements of the reference's CPU paths (oracle/o_*.c; each cites the
 * reference lines it follows) the way the reference's own CLI times them -- lib/test_clenabled.cc:1562-1691: one warm-up call, then N
 * timed calls around a steady clock -- on ONE core.  "kind": "port": scalar / auto-vectorised C, not FFTW / VOLK (absent from the image).
 *
 * usage: cpu_bench [seconds per leg, default 1.0]   -> one JSON object on stdout
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oracle.h"

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static unsigned long long lcg = 88172645463325252ull;
static double urand(void)
{
    lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17;
    return (double)(lcg >> 11) / 9007199254740992.0;
}
static float nrand(void) { return (float)(sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand())); }
static ocplx *crandn(size_t n)
{
    ocplx *p = (ocplx *)malloc(n * sizeof(ocplx));
    for (size_t i = 0; i < n; i++) { p[i].re = nrand(); p[i].im = nrand(); }
    return p;
}

/* repeat fn() for at least `seconds` after one warm-up call; returns million samples per second */
#define TIMED(seconds, nsamples, call, result)                         \
    do {                                                               \
        call;                                                          \
        long reps = 0;                                                 \
        const double t0 = now();                                       \
        double t1 = t0;                                                \
        while (t1 - t0 < (seconds)) { call; reps++; t1 = now(); }      \
        (result) = (double)reps * (double)(nsamples) / (t1 - t0) / 1e6; \
    } while (0)

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    printf("{");

    /* ---- BASELINE configs[0]: clMathOp complex multiply, 8192 x (1.0, 0.5), 1 warm-up + 200 timed testCPU calls
     * (lib/test_clenabled.cc:65,1596-1650; the loop body is lib/clMathOp_impl.cc:336-352) */
    {
        const size_t n = 8192;
        ocplx *a = (ocplx *)malloc(n * sizeof(ocplx)), *b = (ocplx *)malloc(n * sizeof(ocplx)), *c = (ocplx *)malloc(n * sizeof(ocplx));
        for (size_t i = 0; i < n; i++) { a[i].re = b[i].re = 1.0f; a[i].im = b[i].im = 0.5f; }
        oracle_mathop(O_DTYPE_COMPLEX, O_OP_MULTIPLY, n, a, b, c);
        const double t0 = now();
        for (int it = 0; it < 200; it++) oracle_mathop(O_DTYPE_COMPLEX, O_OP_MULTIPLY, n, a, b, c);
        const double dt = (now() - t0) / 200;
        int ok = 1;
        for (size_t i = 0; i < n; i++) ok &= (c[i].re == 0.75f && c[i].im == 1.0f);
        printf("\"config1_clMathOp_testCPU_8192\": {\"us_per_call\": %.3f, \"MSamples_per_s\": %.1f, \"iterations\": 200, \"items\": 8192, "
               "\"result_is_0.75+1.0j\": %s}",
               dt * 1e6, (double)n / dt / 1e6, ok ? "true" : "false");
        free(a); free(b); free(c);
    }

    const size_t n1 = (size_t)1 << 20;
    ocplx *x = crandn(n1 + 4096), *y = (ocplx *)malloc((n1 + 4096) * sizeof(ocplx));
    double r;

    TIMED(secs, n1, oracle_mathop(O_DTYPE_COMPLEX, O_OP_MULTIPLY, n1, x, x, y), r);           /* clMathOp_impl::testCPU */
    printf(", \"clMathOp_multiply_complex\": %.2f", r);
    TIMED(secs, n1, oracle_mathconst(O_DTYPE_COMPLEX, O_OP_MULTIPLY, 2.0f, n1, x, y), r);      /* clMathConst_impl::testCPU */
    printf(", \"clMathConst_multiply_complex\": %.2f", r);

    /* clFFT_impl::testCPU (lib/clFFT_impl.cc:464-518): 4096 points, Blackman window, shift; 64 frames per call */
    {
        float *w = (float *)malloc(4096 * sizeof(float));
        oracle_window(O_WIN_BLACKMAN, 4096, 6.76, w);
        const int nvec = 64;
        TIMED(secs, (size_t)nvec * 4096, oracle_fft_block(4096, 1, w, 1, O_DTYPE_COMPLEX, nvec, x, y, 0), r);
        printf(", \"clFFT_4096_blackman_shift\": %.2f", r);
        free(w);
    }

    /* fft_filter_ccf::filter / fir_filter_ccf::filterN / fir_filter_ccc::filterN (lib/fft_filter.cc:133-175, lib/fir_filter.cc:222-241,377-488) */
    {
        float taps[4096];
        const int nt = oracle_firdes_low_pass(1.0, 10e6, 1e6, 372000.0, O_WIN_HAMMING, 6.76, taps, 4096);
        const size_t m = 192 * 1024;
        oracle_fft_filter *f = oracle_fft_filter_new(1, taps, nt);
        TIMED(secs, m, oracle_fft_filter_filter(f, (int)m, x, y), r);
        printf(", \"clFilter_fft_65taps\": %.2f", r);
        oracle_fft_filter_free(f);
        TIMED(secs, m, oracle_fir_ccf_filterN(taps, nt, x, y, m, 1), r);
        printf(", \"clFilter_fir_65taps\": %.2f", r);
        ocplx ct[4096];
        for (int i = 0; i < nt; i++) { ct[i].re = (float)(taps[i] * cos(3.141592653589793 * i / 8)); ct[i].im = (float)(taps[i] * sin(3.141592653589793 * i / 8)); }
        TIMED(secs, m, oracle_fir_ccc_filterN(ct, nt, x, y, m, 1), r);
        printf(", \"clComplexFilter_fir_65ctaps\": %.2f", r);
        printf(", \"filter_ntaps\": %d", nt);
    }

    /* the channelizer's kernels restated (lib/clPolyphaseChannelizer_impl.cc:153-177): 64 channels x 32 taps per arm, buf_items 65536 */
    {
        static float taps[2048];
        const int nt = oracle_firdes_low_pass(1.0, 64.0, 0.5, 0.0753, O_WIN_HAMMING, 6.76, taps, 2048);
        for (int i = nt; i < 2048; i++) taps[i] = 0.0f;
        int map[64];
        for (int i = 0; i < 64; i++) map[i] = i;
        const int buf = 65536;
        TIMED(secs, buf, oracle_pfb_channelizer(taps, 2048, buf, 64, 64, map, 64, x, y, 0), r);
        printf(", \"clPolyphaseChannelizer_64x32_stream\": %.2f, \"pfb_design_ntaps\": %d", r, nt);
    }

    /* the X-engine kernel text restated (lib/clXEngine_impl.cc:708-817,859-867): 64 antennas x 8 of the 1024 channels x 1024 frames per call
     * (channels are independent: the rate per sample is the same) */
    {
        const int N = 64, F = 8, T = 1024;
        const size_t nb = (size_t)T * N * F * 2;
        int8_t *xi = (int8_t *)malloc(nb);
        for (size_t i = 0; i < nb; i++) xi[i] = (int8_t)((int)(urand() * 255.0) - 127);
        ocplx *v = (ocplx *)malloc(oracle_xengine_out_len(N, F, 1) * sizeof(ocplx));
        TIMED(secs, (size_t)N * F * T, oracle_xengine_ichar(N, F, 1, T, xi, v, 0, 0), r);
        printf(", \"clXEngine_64ant_1024ch_1024t_ichar\": {\"MSamples_per_s\": %.2f, \"sample\": \"64 antennas x 8 channels x 1024 frames per call (8 of the 1024 channels)\"}", r);
        free(xi); free(v);
    }
    printf(", \"cores\": 1, \"seconds_per_leg\": %.2f, \"kind\": \"port (plain C restatement timed from C: one warm-up call, then calls for the stated time, steady clock)\"}\n", secs);
    free(x); free(y);
    return 0;
}
