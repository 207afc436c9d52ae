/*
 * o_elem.c -- oracle (TEST INFRASTRUCTURE) for the remaining elementwise blocks.
 * See oracle.h.  Each function restates the reference's device kernel text / CPU twin.
 * PARITY UNPINNED by the reference (it holds no vectors for these blocks); the tests compare with
 * float64 numpy evaluations of the same formulas.
 */
#include "oracle.h"
#include <math.h>

/* kind codes match include/mi355_clenabled.h MI355_ELEM_* */
int oracle_elem(int kind, float p0, float p1, size_t n, const void *in0, const void *in1, void *out0, void *out1)
{
    const float *a = (const float *)in0, *b = (const float *)in1;
    const ocplx *z = (const ocplx *)in0;
    float *o = (float *)out0, *o1 = (float *)out1;
    for (size_t i = 0; i < n; i++) {
        switch (kind) {
        case 1: /* clLog_impl::testCPU, lib/clLog_impl.cc:200-214: n*log10(a)+k */
            o[i] = p0 * log10f(a[i]) + p1;
            break;
        case 2: { /* op_snr kernel, lib/clSNR_impl.cc:110-112 */
            float t = a[i] / b[i];
            o[i] = fabsf(p0 * log10f(t) + p1);
            break;
        }
        case 3: /* complextomag, lib/clComplexToMag_impl.cc:144-148 */
            o[i] = sqrtf(z[i].im * z[i].im + z[i].re * z[i].re);
            break;
        case 4: /* complextoarg (double path), lib/clComplexToArg_impl.cc:145-147 */
            o[i] = (float)atan2((double)z[i].im, (double)z[i].re);
            break;
        case 5: /* complextomagphase, lib/clComplexToMagPhase_impl.cc:155-160 */
            o[i] = sqrtf(z[i].im * z[i].im + z[i].re * z[i].re);
            o1[i] = (float)atan2((double)z[i].im, (double)z[i].re);
            break;
        case 6: { /* magphasetocomplex (double path), lib/clMagPhaseToComplex_impl.cc:175-191 */
            double mag = (double)a[i], ph = (double)b[i];
            ((ocplx *)out0)[i].re = (float)(mag * cos(ph));
            ((ocplx *)out0)[i].im = (float)(mag * sin(ph));
            break;
        }
        case 7: { /* quadDemod (double path), lib/clQuadratureDemod_impl.cc:125-141; input has 1 item of history */
            double ar = z[i + 1].re, ai = z[i + 1].im, br = z[i].re, bi = -1.0 * (double)z[i].im;
            double re = ar * br - ai * bi, im = ar * bi + ai * br;
            o[i] = (float)((double)p0 * atan2(im, re));
            break;
        }
        default: return -1;
        }
    }
    return 0;
}
