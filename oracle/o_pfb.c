/*
 * o_pfb.c -- oracle (TEST INFRASTRUCTURE) for clPolyphaseChannelizer.
 * See oracle.h.
 *
 * PARITY UNPINNED: the reference has no CPU implementation, no test vectors
 * and no compilable source for this block (only an OpenCL kernel string and a
 * clFFT call).  This file restates those two device steps; the tests cross
 * check it against the independent closed form of SURVEY App. A.4 evaluated in
 * float64 by numpy (tests/golden/gen_golden.py).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/*
 * One general_work() call (lib/clPolyphaseChannelizer_impl.cc:83-109).
 *   in   : history-prefixed input, ntaps-1 old samples first; sample index
 *          i*R - k + ntaps - 1 is read for i < buf_items/R, k < ntaps
 *   step 1  filterpfb2 (:156-167): a_j(i) = sum_{k=j,j+M,..<K} in[i*R-k+K-1]*taps[k]
 *           (fma, k ascending), stored at v_i[(j + i*(M-R)) % M]
 *   step 2  M-point BACKWARD DFT, scale 1 (:100, :208-225):
 *           u_i[c] = sum_m v_i[m] exp(+2*pi*i*m*c/M)
 *   step 3  channel_map (:169-177): out[i*nmap + q] = u_i[ch_map[q]]
 * use_f64=0: float fma accumulation + direct float64-twiddle DFT with float
 * accumulate (M need not be a power of two, e.g. the reference flowgraph's
 * M=3); use_f64=1: everything in double, rounded once.
 */
int oracle_pfb_channelizer(const float *taps, int ntaps, int buf_items, int nch, int ninputs_per_iter,
                           const int *ch_map, int nmap, const ocplx *in, ocplx *out, int use_f64)
{
    int M = nch, R = ninputs_per_iter, K = ntaps;
    if (M < 1 || R < 1 || K < 1 || buf_items % M != 0) return -1; /* :59-62 */
    for (int q = 0; q < nmap; q++) if (ch_map[q] < 0 || ch_map[q] >= M) return -1;
    int nsteps = buf_items / R;
    double *vr = (double *)malloc(sizeof(double) * 2 * (size_t)M);
    double *wr = (double *)malloc(sizeof(double) * 2 * (size_t)M);
    if (!vr || !wr) { free(vr); free(wr); return -2; }
    double *vi = vr + M, *wi = wr + M;
    for (int m = 0; m < M; m++) { wr[m] = cos(2.0 * M_PI * m / M); wi[m] = sin(2.0 * M_PI * m / M); }
    for (int i = 0; i < nsteps; i++) {
        for (int j = 0; j < M; j++) {
            int slot = (int)(((long long)j + (long long)i * (M - R)) % M);
            if (use_f64) {
                double sr = 0, si = 0;
                for (int k = j; k < K; k += M) {
                    ocplx x = in[(size_t)i * R - k + K - 1];
                    sr += (double)x.re * taps[k]; si += (double)x.im * taps[k];
                }
                vr[slot] = sr; vi[slot] = si;
            } else {
                float sr = 0, si = 0;
                for (int k = j; k < K; k += M) {
                    ocplx x = in[(size_t)i * R - k + K - 1];
                    sr = fmaf(x.re, taps[k], sr); si = fmaf(x.im, taps[k], si);
                }
                vr[slot] = sr; vi[slot] = si;
            }
        }
        for (int q = 0; q < nmap; q++) {
            int c = ch_map[q];
            if (use_f64) {
                double sr = 0, si = 0;
                for (int m = 0; m < M; m++) {
                    int t = (int)(((long long)m * c) % M);
                    sr += vr[m] * wr[t] - vi[m] * wi[t];
                    si += vr[m] * wi[t] + vi[m] * wr[t];
                }
                out[(size_t)i * nmap + q].re = (float)sr; out[(size_t)i * nmap + q].im = (float)si;
            } else {
                float sr = 0, si = 0;
                for (int m = 0; m < M; m++) {
                    int t = (int)(((long long)m * c) % M);
                    float cr = (float)wr[t], ci = (float)wi[t], xr = (float)vr[m], xi = (float)vi[m];
                    sr += xr * cr - xi * ci;
                    si += xr * ci + xi * cr;
                }
                out[(size_t)i * nmap + q].re = sr; out[(size_t)i * nmap + q].im = si;
            }
        }
    }
    free(vr); free(wr);
    return 0;
}
