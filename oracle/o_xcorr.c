/*
 * o_xcorr.c -- CPU restatement of clxcorrelate_fft_vcf (TEST INFRASTRUCTURE ONLY, see oracle.h).
 *
 * Follows lib/clxcorrelate_fft_vcf_impl.cc:
 *   :1075-1097  per frame: the reference input (0) and every other input are used as given (input_type 1)
 *               or forward-FFT'd first (input_type 2, d_perform_fft_first :706-709)
 *   :886-910    MultConj kernel: b <- a * conj(b) with a = reference spectrum, b = signal spectrum
 *   :731,:1112  backward FFT with scale 1.0 (unnormalised)
 *   :912-935    ComplexToMag kernel: sqrt(fma(re, re, im*im))
 *   :1133-1140  host: the two halves of every output vector are exchanged (vlen_2 = fftSize/2)
 * PARITY UNPINNED by the reference (no test vectors, needs clFFT + a device); cross-checked in tests against the
 * circular cross-correlation definition evaluated in float64.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

int oracle_xcorr_fft(int n, int num_inputs, int input_type, int nframes, const ocplx *const *inputs, float *const *outputs,
                     int use_f64)
{
    if (n < 2 || (n & 1) || num_inputs < 2 || nframes < 0 || !inputs || !outputs) return -1;
    if (input_type != 1 && input_type != 2) return -1;
    int (*fft)(int, int, const ocplx *, ocplx *) = use_f64 ? oracle_fft_c2c_f64 : oracle_fft_c2c_f32;
    ocplx *ref = (ocplx *)malloc(sizeof(ocplx) * (size_t)n * 3);
    if (!ref) return -2;
    ocplx *sig = ref + n, *rev = sig + n;
    const int half = n / 2;
    int rc = 0;
    for (int i = 0; i < nframes && !rc; i++) {
        const ocplx *r_in = inputs[0] + (size_t)i * n;
        if (input_type == 2) rc = fft(n, -1, r_in, ref);
        else memcpy(ref, r_in, sizeof(ocplx) * (size_t)n);
        for (int s = 1; s < num_inputs && !rc; s++) {
            const ocplx *s_in = inputs[s] + (size_t)i * n;
            if (input_type == 2) rc = fft(n, -1, s_in, sig);
            else memcpy(sig, s_in, sizeof(ocplx) * (size_t)n);
            if (rc) break;
            for (int k = 0; k < n; k++) { /* MultConj, same operation order as the kernel text */
                const float a_r = ref[k].re, a_i = ref[k].im, b_r = sig[k].re, b_i = -sig[k].im;
                if (use_f64) {
                    sig[k].re = (float)((double)a_r * b_r - (double)a_i * b_i);
                    sig[k].im = (float)((double)a_r * b_i + (double)a_i * b_r);
                } else {
                    sig[k].re = (a_r * b_r) - (a_i * b_i);
                    sig[k].im = (a_r * b_i) + (a_i * b_r);
                }
            }
            rc = fft(n, +1, sig, rev);
            if (rc) break;
            float *out = outputs[s - 1] + (size_t)i * n;
            for (int k = 0; k < n; k++) {
                float m;
                if (use_f64) m = (float)sqrt((double)rev[k].re * rev[k].re + (double)rev[k].im * rev[k].im);
                else m = sqrtf(fmaf(rev[k].re, rev[k].re, rev[k].im * rev[k].im));
                out[k < half ? k + half : k - half] = m;
            }
        }
    }
    free(ref);
    return rc;
}
