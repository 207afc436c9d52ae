/*
 * o_xengine.c -- oracle (TEST INFRASTRUCTURE) for clXEngine (FX-correlator
 * X-engine).  See oracle.h.
 *
 * PARITY UNPINNED: the reference has no CPU X-engine; its only implementation
 * is an OpenCL kernel string, and its golden-data test is compiled out with the
 * data files absent (lib/test-clxengine.cc:493-535).  This file restates the
 * kernel text; the tests cross check it against exact int64 sums and the
 * closed-form cases of SURVEY section 8c item 5.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

size_t oracle_xengine_out_len(int ninputs, int nchan, int npol)
{
    /* matrix_flat_length, triangular order: lib/clXEngine_impl.cc:183,204-207 */
    size_t nb = (size_t)(ninputs + 1) * (size_t)ninputs / 2;
    return (size_t)nchan * nb * (size_t)npol * (size_t)npol;
}

/* baseline index -> (station1 >= station2), lib/clXEngine_impl.cc:742-750;
 * the float sqrt of the reference is replaced by the exact integer inverse of
 * k = s1(s1+1)/2 + s2 (identical for every k the block can produce). */
static void baseline_stations(int k, int *s1, int *s2)
{
    int a = (int)(-0.5 + sqrt(0.25 + 2.0 * (double)k));
    while ((a + 1) * (a + 2) / 2 <= k) a++;
    while (a * (a + 1) / 2 > k) a--;
    *s1 = a;
    *s2 = k - a * (a + 1) / 2;
}

/* cxmac, lib/clXEngine_impl.cc:728-737 (non-FMA form): acc += z0 * conj(z1) */
static inline void cxmac(ocplx *acc, ocplx z0, ocplx z1)
{
    acc->re += z0.re * z1.re + z0.im * z1.im;
    acc->im += z0.im * z1.re - z0.re * z1.im;
}

/*
 * XCorrelate kernel, lib/clXEngine_impl.cc:739-808.
 * in  : [t][station][chan][pol] complex float, element index
 *       t*frame_size + (s*F + f)*npol + p, frame_size = F*N*npol (:189,766-767)
 * out : [f][k][pol*pol] with k = s1(s1+1)/2 + s2; pol order XX,XY,YX,YY (:786-808)
 * accumulate != 0 reproduces pipeline integration ("+=" :785-796).
 */
int oracle_xengine_cf32(int ninputs, int nchan, int npol, int ntime, const ocplx *in, ocplx *out, int accumulate)
{
    if (ninputs < 2 || (npol != 1 && npol != 2)) return -1; /* :106-109 */
    int nb = (ninputs + 1) * ninputs / 2;
    size_t frame = (size_t)nchan * ninputs * npol;
    for (int f = 0; f < nchan; f++) {
        for (int k = 0; k < nb; k++) {
            int s1, s2;
            baseline_stations(k, &s1, &s2);
            ocplx xx = {0, 0}, xy = {0, 0}, yx = {0, 0}, yy = {0, 0};
            for (int t = 0; t < ntime; t++) {
                size_t i1 = (size_t)t * frame + ((size_t)s1 * nchan + f) * npol;
                size_t i2 = (size_t)t * frame + ((size_t)s2 * nchan + f) * npol;
                cxmac(&xx, in[i1], in[i2]);
                if (npol == 2) {
                    cxmac(&xy, in[i1], in[i2 + 1]);
                    cxmac(&yx, in[i1 + 1], in[i2]);
                    cxmac(&yy, in[i1 + 1], in[i2 + 1]);
                }
            }
            size_t o = ((size_t)f * nb + k) * (size_t)(npol * npol);
            ocplx r[4] = { xx, xy, yx, yy };
            for (int q = 0; q < npol * npol; q++) {
                if (accumulate) { out[o + q].re += r[q].re; out[o + q].im += r[q].im; }
                else out[o + q] = r[q];
            }
        }
    }
    return 0;
}

/*
 * IChar path: CharToComplex (lib/clXEngine_impl.cc:859-867,
 * x = (float)int8 * 0.007874015748031496063) followed by XCorrelate.
 * exact=0 : float conversion + float accumulation in t order (the reference's
 *           arithmetic)
 * exact=1 : integer sums S = sum (I1*I2+Q1*Q2) + j(Q1*I2-I1*Q2) in int64, one
 *           scale by (1/127)^2 in double, rounded once -- the
 *           bit-exact target for an integer-MFMA implementation.
 */
int oracle_xengine_ichar(int ninputs, int nchan, int npol, int ntime, const int8_t *in, ocplx *out,
                         int accumulate, int exact)
{
    if (ninputs < 2 || (npol != 1 && npol != 2)) return -1;
    int nb = (ninputs + 1) * ninputs / 2;
    size_t frame = (size_t)nchan * ninputs * npol;
    const float kf = (float)0.007874015748031496063;
    const double kd = 0.007874015748031496063;
    for (int f = 0; f < nchan; f++) {
        for (int k = 0; k < nb; k++) {
            int s1, s2;
            baseline_stations(k, &s1, &s2);
            size_t o = ((size_t)f * nb + k) * (size_t)(npol * npol);
            for (int p1 = 0; p1 < npol; p1++) {
                for (int p2 = 0; p2 < npol; p2++) {
                    ocplx r;
                    if (exact) {
                        int64_t sr = 0, si = 0;
                        for (int t = 0; t < ntime; t++) {
                            const int8_t *a = in + 2 * ((size_t)t * frame + ((size_t)s1 * nchan + f) * npol + p1);
                            const int8_t *b = in + 2 * ((size_t)t * frame + ((size_t)s2 * nchan + f) * npol + p2);
                            sr += (int64_t)a[0] * b[0] + (int64_t)a[1] * b[1];
                            si += (int64_t)a[1] * b[0] - (int64_t)a[0] * b[1];
                        }
                        r.re = (float)((double)sr * kd * kd);
                        r.im = (float)((double)si * kd * kd);
                    } else {
                        ocplx acc = {0, 0};
                        for (int t = 0; t < ntime; t++) {
                            const int8_t *a = in + 2 * ((size_t)t * frame + ((size_t)s1 * nchan + f) * npol + p1);
                            const int8_t *b = in + 2 * ((size_t)t * frame + ((size_t)s2 * nchan + f) * npol + p2);
                            ocplx z0 = { (float)a[0] * kf, (float)a[1] * kf };
                            ocplx z1 = { (float)b[0] * kf, (float)b[1] * kf };
                            cxmac(&acc, z0, z1);
                        }
                        r = acc;
                    }
                    size_t q = (size_t)p1 * npol + p2;
                    if (accumulate) { out[o + q].re += r.re; out[o + q].im += r.im; }
                    else out[o + q] = r;
                }
            }
        }
    }
    return 0;
}

/*
 * Packed 4-bit path: CharToComplex packed variant (lib/clXEngine_impl.cc:831-857)
 * + float8 XCorrelate (:605-706).  Host layout per frame: [station][chan]{X,Y}
 * bytes; high nibble = real, low nibble = imag; two's-complement LUT with code
 * 8 -> 0 (:833); scale 1/7 (:835).  Output order as above with npol = 2.
 */
int oracle_xengine_packed4(int ninputs, int nchan, int ntime, const uint8_t *in, ocplx *out, int accumulate)
{
    static const int lut[16] = { 0, 1, 2, 3, 4, 5, 6, 7, 0, -7, -6, -5, -4, -3, -2, -1 };
    const float kf = (float)0.142857142857142857143;
    if (ninputs < 2) return -1;
    int nb = (ninputs + 1) * ninputs / 2;
    size_t frame = (size_t)nchan * ninputs * 2; /* bytes per time step */
    for (int f = 0; f < nchan; f++) {
        for (int k = 0; k < nb; k++) {
            int s1, s2;
            baseline_stations(k, &s1, &s2);
            ocplx acc[4] = { {0, 0}, {0, 0}, {0, 0}, {0, 0} };
            for (int t = 0; t < ntime; t++) {
                const uint8_t *a = in + (size_t)t * frame + ((size_t)s1 * nchan + f) * 2;
                const uint8_t *b = in + (size_t)t * frame + ((size_t)s2 * nchan + f) * 2;
                ocplx ax = { (float)lut[a[0] >> 4] * kf, (float)lut[a[0] & 15] * kf };
                ocplx ay = { (float)lut[a[1] >> 4] * kf, (float)lut[a[1] & 15] * kf };
                ocplx bx = { (float)lut[b[0] >> 4] * kf, (float)lut[b[0] & 15] * kf };
                ocplx by = { (float)lut[b[1] >> 4] * kf, (float)lut[b[1] & 15] * kf };
                cxmac(&acc[0], ax, bx); cxmac(&acc[1], ax, by);
                cxmac(&acc[2], ay, bx); cxmac(&acc[3], ay, by);
            }
            size_t o = ((size_t)f * nb + k) * 4;
            for (int q = 0; q < 4; q++) {
                if (accumulate) { out[o + q].re += acc[q].re; out[o + q].im += acc[q].im; }
                else out[o + q] = acc[q];
            }
        }
    }
    return 0;
}

/*
 * Host frame gather of work_processor(), lib/clXEngine_impl.cc:987-1061:
 * copies frames [0,nframes) of every input stream into the frame buffer at
 * time slots frame0.. in [t][station][chan][pol] order.
 *   npol==1          : inputs[i] is antenna i                         :991-1006
 *   npol==2 cf32/i8  : inputs[i] = X of antenna i, inputs[i+N] = Y;
 *                      interleaved X,Y per channel                     :1009-1023,1038-1058
 *   packed           : inputs[i] already holds X,Y bytes per channel   :1025-1037
 */
int oracle_xengine_gather(int dtype, int ninputs, int nchan, int npol, int nframes, int frame0,
                          const void *const *inputs, void *frame_buffer)
{
    size_t esz;
    if (dtype == O_DTYPE_COMPLEX) esz = 8;
    else if (dtype == O_DTYPE_BYTE) esz = 2;
    else if (dtype == O_DTYPE_PACKEDXY) { esz = 1; npol = 2; }
    else return -1;
    size_t frame_elems = (size_t)nchan * ninputs * npol;
    char *dst = (char *)frame_buffer;
    for (int b = 0; b < nframes; b++) {
        size_t base = frame_elems * (size_t)(frame0 + b) * esz;
        for (int i = 0; i < ninputs; i++) {
            if (npol == 1 || dtype == O_DTYPE_PACKEDXY) {
                size_t row = (size_t)nchan * npol * esz;
                memcpy(dst + base + (size_t)i * row, (const char *)inputs[i] + (size_t)b * row, row);
            } else {
                const char *x = (const char *)inputs[i] + (size_t)b * nchan * esz;
                const char *y = (const char *)inputs[i + ninputs] + (size_t)b * nchan * esz;
                char *row = dst + base + (size_t)i * nchan * 2 * esz;
                for (int c = 0; c < nchan; c++) {
                    memcpy(row + (size_t)c * 2 * esz, x + (size_t)c * esz, esz);
                    memcpy(row + (size_t)c * 2 * esz + esz, y + (size_t)c * esz, esz);
                }
            }
        }
    }
    return 0;
}
