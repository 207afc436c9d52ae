/*
 * o_fft.c -- oracle (TEST INFRASTRUCTURE) for the clFFT block.  See oracle.h.
 *
 * The reference's CPU FFT is FFTW3f behind lib/fft.cc:146-186 (plan) and :239
 * (execute); FFTW is a third-party dependency that is not vendored, not pinned
 * to a version (lib/CMakeLists.txt:78) and absent from this image.  What it
 * computes is the unnormalised DFT  X[k] = sum_n x[n] exp(sign*2*pi*i*k*n/N);
 * that published definition is restated twice:
 *   oracle_fft_c2c_f32  radix-2 decimation-in-time in float arithmetic with
 *                       twiddles rounded from double (the class of result
 *                       FFTW3f gives; also the timed CPU baseline)
 *   oracle_fft_c2c_f64  the same definition in double, rounded once (truth
 *                       used to set tolerances)
 * Known answer held by the reference (lib/clFFT_impl.cc:361-455,
 * FFTValidationTest): N=2048, x[n]=(sin,cos)(2*pi*n/N) -> X[2047]=(0,2048).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static int ilog2_exact(int n)
{
    int l = 0;
    if (n < 1) return -1;
    while ((1 << l) < n) l++;
    return ((1 << l) == n) ? l : -1;
}

static unsigned bitrev(unsigned v, int bits)
{
    unsigned r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

/* direct O(N^2) DFT in double for sizes that are not a power of two (the reference leaves those to FFTW / clFFT's
 * mixed-radix plans; the definition is the same) */
static int dft_any(int n, int sign, const ocplx *in, ocplx *out)
{
    if (n < 1) return -1;
    double *c = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    ocplx *tmp = (ocplx *)malloc(sizeof(ocplx) * (size_t)n);
    if (!c || !tmp) { free(c); free(tmp); return -2; }
    double *sn = c + n;
    for (int k = 0; k < n; k++) {
        double a = sign * 2.0 * M_PI * (double)k / (double)n;
        c[k] = cos(a); sn[k] = sin(a);
    }
    for (int k = 0; k < n; k++) {
        double re = 0.0, im = 0.0;
        long long idx = 0;
        for (int i = 0; i < n; i++) {
            re += (double)in[i].re * c[idx] - (double)in[i].im * sn[idx];
            im += (double)in[i].re * sn[idx] + (double)in[i].im * c[idx];
            idx += k;
            if (idx >= n) idx -= n;
        }
        tmp[k].re = (float)re; tmp[k].im = (float)im;
    }
    memcpy(out, tmp, sizeof(ocplx) * (size_t)n);
    free(c); free(tmp);
    return 0;
}

int oracle_fft_c2c_f32(int n, int sign, const ocplx *in, ocplx *out)
{
    int lg = ilog2_exact(n);
    if (lg < 0) return dft_any(n, sign, in, out);
    ocplx *w = (ocplx *)malloc(sizeof(ocplx) * (size_t)(n / 2 + 1));
    if (!w) return -2;
    for (int k = 0; k < n / 2; k++) {
        double a = sign * 2.0 * M_PI * (double)k / (double)n;
        w[k].re = (float)cos(a);
        w[k].im = (float)sin(a);
    }
    if (in != out) {
        for (int i = 0; i < n; i++) out[bitrev((unsigned)i, lg)] = in[i];
    } else {
        for (int i = 0; i < n; i++) {
            int j = (int)bitrev((unsigned)i, lg);
            if (j > i) { ocplx t = out[i]; out[i] = out[j]; out[j] = t; }
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len) {
            for (int j = 0; j < half; j++) {
                ocplx t = w[j * step];
                ocplx u = out[base + j], v = out[base + j + half];
                float vr = v.re * t.re - v.im * t.im;
                float vi = v.re * t.im + v.im * t.re;
                out[base + j].re = u.re + vr;        out[base + j].im = u.im + vi;
                out[base + j + half].re = u.re - vr; out[base + j + half].im = u.im - vi;
            }
        }
    }
    free(w);
    return 0;
}

int oracle_fft_c2c_f64(int n, int sign, const ocplx *in, ocplx *out)
{
    int lg = ilog2_exact(n);
    if (lg < 0) return dft_any(n, sign, in, out);
    double *re = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *wr = (double *)malloc(sizeof(double) * (size_t)(n + 2));
    if (!re || !wr) { free(re); free(wr); return -2; }
    double *im = re + n, *wi = wr + n / 2 + 1;
    for (int k = 0; k < n / 2; k++) {
        double a = sign * 2.0 * M_PI * (double)k / (double)n;
        wr[k] = cos(a); wi[k] = sin(a);
    }
    for (int i = 0; i < n; i++) {
        unsigned j = bitrev((unsigned)i, lg);
        re[j] = in[i].re; im[j] = in[i].im;
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len) {
            for (int j = 0; j < half; j++) {
                double tr = wr[j * step], ti = wi[j * step];
                double vr = re[base + j + half] * tr - im[base + j + half] * ti;
                double vi = re[base + j + half] * ti + im[base + j + half] * tr;
                double ur = re[base + j], ui = im[base + j];
                re[base + j] = ur + vr;        im[base + j] = ui + vi;
                re[base + j + half] = ur - vr; im[base + j + half] = ui - vi;
            }
        }
    }
    for (int i = 0; i < n; i++) { out[i].re = (float)re[i]; out[i].im = (float)im[i]; }
    free(re); free(wr);
    return 0;
}

/*
 * Block semantics of one clFFT work() call over `nvec` frames of length n.
 * Follows clFFT_impl::testCPU, lib/clFFT_impl.cc:464-518 (= GNU Radio fft_vcc):
 *   - window: dst[i] = in[i]*w[i] (volk_32fc_32f_multiply_32fc)            :484
 *   - reverse + shift: input halves swapped while windowing, the window is
 *     indexed by the ORIGINAL input position                                :477-482,488-493
 *   - forward + shift: out[0..N-len) = X[len..N), out[N-len..N) = X[0..len),
 *     len = ceil(N/2)                                                       :503-507
 *   - no scaling in either direction (clFFT_impl.cc:121-122)
 * Real input (DTYPE_FLOAT) is a complex frame with zero imaginary part; the
 * full N-bin spectrum is produced (what lib/clFFT_impl.cc:608-630 builds from
 * the R2C half; its stale Nyquist bin, SURVEY App. B-14, is not reproduced).
 */
int oracle_fft_block(int n, int forward, const float *window, int shift, int dtype,
                     int nvec, const void *in, ocplx *out, int use_f64)
{
    if (n < 1) return -1;
    if (dtype != O_DTYPE_COMPLEX && dtype != O_DTYPE_FLOAT) return -1;
    ocplx *buf = (ocplx *)malloc(sizeof(ocplx) * 2 * (size_t)n);
    if (!buf) return -2;
    ocplx *res = buf + n;
    for (int v = 0; v < nvec; v++) {
        for (int i = 0; i < n; i++) {
            ocplx s;
            if (dtype == O_DTYPE_COMPLEX) s = ((const ocplx *)in)[(size_t)v * n + i];
            else { s.re = ((const float *)in)[(size_t)v * n + i]; s.im = 0.0f; }
            if (window) { s.re = s.re * window[i]; s.im = s.im * window[i]; }
            int dst = i;
            if (!forward && shift) {
                int half = n / 2; /* floor(N/2), :491 */
                dst = (i < half) ? i + (n - half) : i - half;
            }
            buf[dst] = s;
        }
        int rc = use_f64 ? oracle_fft_c2c_f64(n, forward ? -1 : 1, buf, res)
                         : oracle_fft_c2c_f32(n, forward ? -1 : 1, buf, res);
        if (rc) { free(buf); return rc; }
        ocplx *o = out + (size_t)v * n;
        if (forward && shift) {
            int len = (n + 1) / 2;
            memcpy(o, res + len, sizeof(ocplx) * (size_t)(n - len));
            memcpy(o + (n - len), res, sizeof(ocplx) * (size_t)len);
        } else {
            memcpy(o, res, sizeof(ocplx) * (size_t)n);
        }
    }
    free(buf);
    return 0;
}
