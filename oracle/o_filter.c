/*
 * o_filter.c -- oracle (TEST INFRASTRUCTURE) for clFilter / clComplexFilter:
 * the FFT fast-convolution kernel and the direct-form FIR kernels.
 * See oracle.h.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct oracle_fft_filter {
    int decim, ntaps, fftsize, nsamples;
    float *taps;
    ocplx *xtaps; /* forward transform of taps/fftsize, zero padded */
    ocplx *tail;  /* ntaps-1 samples carried to the next block */
    ocplx *a, *b; /* work buffers, fftsize each */
};

/* fft_filter_ccf::compute_sizes, lib/fft_filter.cc:72-97:
 * fftsize = 2 * 2^ceil(log2(ntaps)), nsamples = fftsize - ntaps + 1 */
static void size_for_taps(int ntaps, int *fftsize, int *nsamples)
{
    *fftsize = (int)(2 * pow(2.0, ceil(log((double)ntaps) / log(2.0))));
    *nsamples = *fftsize - ntaps + 1;
}

/* fft_filter_ccf::set_taps, lib/fft_filter.cc:38-69: taps scaled by 1/fftsize,
 * zero padded to fftsize, forward transformed; the tail is cleared. */
int oracle_fft_filter_set_taps(oracle_fft_filter *f, const float *taps, int ntaps)
{
    int fs, ns;
    if (!f || ntaps < 1) return -1;
    size_for_taps(ntaps, &fs, &ns);
    free(f->taps); free(f->xtaps); free(f->tail); free(f->a); free(f->b);
    f->ntaps = ntaps; f->fftsize = fs; f->nsamples = ns;
    f->taps = (float *)malloc(sizeof(float) * (size_t)ntaps);
    f->xtaps = (ocplx *)malloc(sizeof(ocplx) * (size_t)fs);
    f->tail = (ocplx *)calloc((size_t)(ntaps > 1 ? ntaps - 1 : 1), sizeof(ocplx));
    f->a = (ocplx *)malloc(sizeof(ocplx) * (size_t)fs);
    f->b = (ocplx *)malloc(sizeof(ocplx) * (size_t)fs);
    memcpy(f->taps, taps, sizeof(float) * (size_t)ntaps);
    float scale = (float)(1.0 / fs);
    for (int i = 0; i < fs; i++) {
        f->a[i].re = (i < ntaps) ? taps[i] * scale : 0.0f;
        f->a[i].im = 0.0f;
    }
    oracle_fft_c2c_f32(fs, -1, f->a, f->xtaps);
    return ns;
}

oracle_fft_filter *oracle_fft_filter_new(int decimation, const float *taps, int ntaps)
{
    oracle_fft_filter *f = (oracle_fft_filter *)calloc(1, sizeof(*f));
    if (!f) return NULL;
    f->decim = decimation < 1 ? 1 : decimation;
    if (oracle_fft_filter_set_taps(f, taps, ntaps) < 0) { free(f); return NULL; }
    return f;
}

void oracle_fft_filter_free(oracle_fft_filter *f)
{
    if (!f) return;
    free(f->taps); free(f->xtaps); free(f->tail); free(f->a); free(f->b);
    free(f);
}

int oracle_fft_filter_fftsize(const oracle_fft_filter *f) { return f->fftsize; }
int oracle_fft_filter_nsamples(const oracle_fft_filter *f) { return f->nsamples; }
int oracle_fft_filter_xformed_taps(const oracle_fft_filter *f, ocplx *out)
{
    memcpy(out, f->xtaps, sizeof(ocplx) * (size_t)f->fftsize);
    return f->fftsize;
}

/*
 * fft_filter_ccf::filter, lib/fft_filter.cc:133-175 (overlap-add):
 * per block of nsamples inputs: zero pad to fftsize, forward FFT, multiply by
 * the tap spectrum (volk_32fc_x2_multiply_32fc), inverse FFT, add the previous
 * tail into the first ntaps-1 outputs, emit every decim-th sample with the
 * phase carried in dec_ctr, keep the last ntaps-1 outputs as the new tail.
 * `nitems` is the OUTPUT count; nitems*decim inputs are consumed, rounded up to
 * whole blocks exactly as the reference loop does (so the caller must supply
 * ceil(nitems*decim/nsamples)*nsamples readable inputs).
 * Returns the number of outputs actually written.
 */
int oracle_fft_filter_filter(oracle_fft_filter *f, int nitems, const ocplx *in, ocplx *out)
{
    int dec_ctr = 0, written = 0;
    int nin = nitems * f->decim, tail = f->ntaps - 1;
    for (int i = 0; i < nin; i += f->nsamples) {
        memcpy(f->a, in + i, sizeof(ocplx) * (size_t)f->nsamples);
        for (int j = f->nsamples; j < f->fftsize; j++) { f->a[j].re = 0.0f; f->a[j].im = 0.0f; }
        oracle_fft_c2c_f32(f->fftsize, -1, f->a, f->b);
        for (int k = 0; k < f->fftsize; k++) {
            ocplx x = f->b[k], h = f->xtaps[k];
            f->a[k].re = x.re * h.re - x.im * h.im;
            f->a[k].im = x.re * h.im + x.im * h.re;
        }
        oracle_fft_c2c_f32(f->fftsize, +1, f->a, f->b);
        for (int j = 0; j < tail; j++) { f->b[j].re += f->tail[j].re; f->b[j].im += f->tail[j].im; }
        int j = dec_ctr;
        while (j < f->nsamples) { out[written++] = f->b[j]; j += f->decim; }
        dec_ctr = j - f->nsamples;
        memcpy(f->tail, f->b + f->nsamples, sizeof(ocplx) * (size_t)tail);
    }
    return written;
}

/*
 * fir_filter_ccf::filter / filterN / filterNdec, lib/fir_filter.cc:222-257:
 * taps are stored reversed (set_taps :174-196) and each output is the dot
 * product of the reversed taps with ntaps consecutive inputs starting at the
 * output's input index:  y[i] = sum_{j<K} taps[K-1-j] * in[i*decim + j].
 * With GNU Radio's history-prefixed buffer (in[K-1] is the newest sample of
 * output 0) that is y[m] = sum_k h[k] x[m*decim - k].  The accumulation order
 * is j ascending (volk_32fc_32f_dot_prod_32fc generic kernel order); the
 * SIMD kernels VOLK may pick instead differ in the last ulp only.
 */
int oracle_fir_ccf_filterN(const float *taps, int ntaps, const ocplx *in, ocplx *out, size_t n, int decim)
{
    if (decim < 1 || ntaps < 1) return -1;
    for (size_t i = 0; i < n; i++) {
        const ocplx *x = in + i * (size_t)decim;
        float sr = 0.0f, si = 0.0f;
        for (int j = 0; j < ntaps; j++) {
            float h = taps[ntaps - 1 - j];
            sr += x[j].re * h;
            si += x[j].im * h;
        }
        out[i].re = sr; out[i].im = si;
    }
    return 0;
}

/* fir_filter_ccc, lib/fir_filter.cc:377-488 (complex taps, full complex
 * multiply, volk_32fc_x2_dot_prod_32fc); the kernel the reference runs on the
 * device is lib/clComplexFilter_impl.cc:796-828 with the same indexing. */
int oracle_fir_ccc_filterN(const ocplx *taps, int ntaps, const ocplx *in, ocplx *out, size_t n, int decim)
{
    if (decim < 1 || ntaps < 1) return -1;
    for (size_t i = 0; i < n; i++) {
        const ocplx *x = in + i * (size_t)decim;
        float sr = 0.0f, si = 0.0f;
        for (int j = 0; j < ntaps; j++) {
            ocplx h = taps[ntaps - 1 - j];
            sr += x[j].re * h.re - x[j].im * h.im;
            si += x[j].re * h.im + x[j].im * h.re;
        }
        out[i].re = sr; out[i].im = si;
    }
    return 0;
}
