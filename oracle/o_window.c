/*
 * o_window.c -- oracle (TEST INFRASTRUCTURE) for the window functions and the
 * windowed-sinc low-pass design the reference uses to build its benchmark
 * fixtures (lib/test_clenabled.cc:812,1831-1840).  See oracle.h.
 *
 * Pinned by values the survey recorded from the reference's own window.cc /
 * firdes.cc: blackman(4096)[1], [2048]; low_pass(1,10e6,
 * 1e6,372e3) -> 65 taps, t[0], t[32]; low_pass(1,64,0.5,0.0753) -> 2047 taps,
 * t[0], t[1023]; low_pass(1,300e3,48e3,5e3) -> 145 taps.
 */
#include "oracle.h"
#include <math.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* generalised cosine window, lib/window.cc:94-127: float coefficients, the
 * phase is formed in double ((2.0f*M_PI*n)/M), handed to cosf(), and the sum
 * is float arithmetic left to right. */
static void cosine_sum(int ntaps, const float *c, int nc, float *out)
{
    float M = (float)(ntaps - 1);
    for (int n = 0; n < ntaps; n++) {
        float acc = c[0];
        for (int q = 1; q < nc; q++) {
            float term = c[q] * cosf((float)(((2.0f * q) * M_PI * n) / M));
            acc = (q & 1) ? acc - term : acc + term;
        }
        out[n] = acc;
    }
}

/* modified Bessel I0 series used by the Kaiser window, lib/window.cc:33-49 */
static double bessel_i0(double x)
{
    double sum = 1.0, u = 1.0, half = x / 2.0;
    int n = 1;
    do {
        double t = half / (double)n;
        n++;
        u *= t * t;
        sum += u;
    } while (u >= 1e-21 * sum);
    return sum;
}

/* lib/window.cc:76-92 */
double oracle_window_max_attenuation(int type, double beta)
{
    switch (type) {
    case O_WIN_HAMMING: return 53;
    case O_WIN_HANN: return 44;
    case O_WIN_BLACKMAN: return 74;
    case O_WIN_RECTANGULAR: return 21;
    case O_WIN_KAISER: return beta / 0.1102 + 8.7;
    case O_WIN_BLACKMAN_HARRIS: return 92;
    case O_WIN_BARTLETT: return 27;
    case O_WIN_FLATTOP: return 93;
    }
    return -1;
}

/* window::build dispatch, lib/window.cc:353-367, and the per-type bodies:
 * rectangular :129-136, hamming :138-148 (double cos), hann :150-160,
 * blackman :166-170, blackman_harris(92) :190-201, kaiser :250-267,
 * bartlett :269-281, flattop :243-248. */
int oracle_window(int type, int ntaps, double beta, float *out)
{
    float M = (float)(ntaps - 1);
    switch (type) {
    case O_WIN_RECTANGULAR:
        for (int n = 0; n < ntaps; n++) out[n] = 1.0f;
        return 0;
    case O_WIN_HAMMING:
        for (int n = 0; n < ntaps; n++) out[n] = (float)(0.54 - 0.46 * cos((2 * M_PI * n) / M));
        return 0;
    case O_WIN_HANN:
        for (int n = 0; n < ntaps; n++) out[n] = (float)(0.5 - 0.5 * cos((2 * M_PI * n) / M));
        return 0;
    case O_WIN_BLACKMAN: {
        const float c[3] = { 0.42f, 0.5f, 0.08f };
        cosine_sum(ntaps, c, 3, out);
        return 0;
    }
    case O_WIN_BLACKMAN_HARRIS: {
        const float c[4] = { 0.35875f, 0.48829f, 0.14128f, 0.01168f };
        cosine_sum(ntaps, c, 4, out);
        return 0;
    }
    case O_WIN_FLATTOP: {
        const double s = 4.63867;
        const float c[5] = { (float)(1.0 / s), (float)(1.93 / s), (float)(1.29 / s), (float)(0.388 / s), (float)(0.028 / s) };
        cosine_sum(ntaps, c, 5, out);
        return 0;
    }
    case O_WIN_KAISER: {
        if (beta < 0) return -1;
        double ib = 1.0 / bessel_i0(beta), inm1 = 1.0 / (double)(ntaps - 1);
        for (int i = 0; i < ntaps; i++) {
            double t = 2 * i * inm1 - 1;
            out[i] = (float)(bessel_i0(beta * sqrt(1.0 - t * t)) * ib);
        }
        return 0;
    }
    case O_WIN_BARTLETT:
        for (int n = 0; n < ntaps / 2; n++) out[n] = 2 * n / M;
        for (int n = ntaps / 2; n < ntaps; n++) out[n] = 2 - 2 * n / M;
        return 0;
    }
    return -1;
}

/* firdes::compute_ntaps, lib/firdes.cc:674-686 */
int oracle_firdes_ntaps(double fs, double transition_width, int win_type, double beta)
{
    double a = oracle_window_max_attenuation(win_type, beta);
    int nt = (int)(a * fs / (22.0 * transition_width));
    if ((nt & 1) == 0) nt++;
    return nt;
}

/* firdes::low_pass, lib/firdes.cc:92-137: windowed sinc centred on M=(ntaps-1)/2,
 * tap stored as float, DC gain normalised using the float taps. Returns ntaps,
 * or -1 if `cap` is too small / arguments fail sanity_check_1f (:706-717). */
int oracle_firdes_low_pass(double gain, double fs, double cutoff, double transition_width,
                           int win_type, double beta, float *taps, int cap)
{
    if (fs <= 0.0 || cutoff <= 0.0 || cutoff > fs / 2 || transition_width <= 0) return -1;
    int nt = oracle_firdes_ntaps(fs, transition_width, win_type, beta);
    if (nt > cap) return -1;
    if (oracle_window(win_type, nt, beta, taps) != 0) return -1; /* taps[] holds w[] for now */
    int M = (nt - 1) / 2;
    double w0 = 2 * M_PI * cutoff / fs;
    for (int n = -M; n <= M; n++) {
        float w = taps[n + M];
        if (n == 0) taps[n + M] = (float)(w0 / M_PI * w);
        else        taps[n + M] = (float)(sin(n * w0) / (n * M_PI) * w);
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < nt; i++) taps[i] = (float)(taps[i] * gain);
    return nt;
}
