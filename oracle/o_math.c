/*
 * o_math.c -- oracle (TEST INFRASTRUCTURE) for the clMathOp / clMathConst
 * elementwise family.  See oracle.h for the rules on who may use this.
 */
#include "oracle.h"

/*
 * Two-input elementwise operator.
 * Follows lib/clMathOp_impl.cc:336-352 (testCPU, complex multiply) and the
 * device kernel text lib/clMathOp_impl.cc:119-236 for the other operators
 * and data types:
 *   complex x      : (ar*br - ai*bi, ar*bi + ai*br)         :195-200 / :347-348
 *   complex +,-    : componentwise                           :203-208
 *   complex x conj : b.imag negated first, then as multiply  :226-231
 *   float / int    : c = a op b                              :134-147, :167-174
 * int arithmetic wraps modulo 2^32 (OpenCL int semantics); done in uint32 here
 * so the C is well defined.
 */
int oracle_mathop(int dtype, int op, size_t n, const void *a, const void *b, void *c)
{
    size_t i;
    if (dtype == O_DTYPE_COMPLEX) {
        const ocplx *x = (const ocplx *)a, *y = (const ocplx *)b;
        ocplx *z = (ocplx *)c;
        for (i = 0; i < n; i++) {
            float ar = x[i].re, ai = x[i].im, br = y[i].re, bi = y[i].im;
            switch (op) {
            case O_OP_MULTIPLY:
                z[i].re = ar * br - ai * bi;
                z[i].im = ar * bi + ai * br;
                break;
            case O_OP_ADD:      z[i].re = ar + br; z[i].im = ai + bi; break;
            case O_OP_SUBTRACT: z[i].re = ar - br; z[i].im = ai - bi; break;
            case O_OP_MULTIPLY_CONJUGATE:
                bi = -1.0f * bi;
                z[i].re = ar * br - ai * bi;
                z[i].im = ar * bi + ai * br;
                break;
            default: return -1;
            }
        }
        return 0;
    }
    if (dtype == O_DTYPE_FLOAT) {
        const float *x = (const float *)a, *y = (const float *)b;
        float *z = (float *)c;
        for (i = 0; i < n; i++) {
            switch (op) {
            case O_OP_MULTIPLY: z[i] = x[i] * y[i]; break;
            case O_OP_ADD:      z[i] = x[i] + y[i]; break;
            case O_OP_SUBTRACT: z[i] = x[i] - y[i]; break;
            default: return -1;
            }
        }
        return 0;
    }
    if (dtype == O_DTYPE_INT) {
        const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
        uint32_t *z = (uint32_t *)c;
        for (i = 0; i < n; i++) {
            switch (op) {
            case O_OP_MULTIPLY: z[i] = x[i] * y[i]; break;
            case O_OP_ADD:      z[i] = x[i] + y[i]; break;
            case O_OP_SUBTRACT: z[i] = x[i] - y[i]; break;
            default: return -1;
            }
        }
        return 0;
    }
    return -1;
}

/*
 * One-input elementwise operator with a real scalar k.
 * Follows lib/clMathConst_impl.cc:275-301 (testCPU: copy and multiply) and the
 * kernel text :120-223:
 *   complex x k / + k / - k : the REAL k is applied to both components  :190-201
 *   complex conjugate       : (re, -1.0*im)                             :203-218
 *   float / int             : c = a op k                                :131-141, :155-166
 * EMPTY_W_COPY is restated as a plain copy (what testCPU does, :286-291); the
 * reference's device kernel falls through into MULTIPLY, a
 * defect that is not reproduced.  For DTYPE_INT the scalar is (int)k.
 */
int oracle_mathconst(int dtype, int op, float k, size_t n, const void *a, void *c)
{
    size_t i;
    if (dtype == O_DTYPE_COMPLEX) {
        const ocplx *x = (const ocplx *)a;
        ocplx *z = (ocplx *)c;
        for (i = 0; i < n; i++) {
            switch (op) {
            case O_OP_MULTIPLY: z[i].re = x[i].re * k; z[i].im = x[i].im * k; break;
            case O_OP_ADD:      z[i].re = x[i].re + k; z[i].im = x[i].im + k; break;
            case O_OP_SUBTRACT: z[i].re = x[i].re - k; z[i].im = x[i].im - k; break;
            case O_OP_CONJUGATE: z[i].re = x[i].re; z[i].im = -1.0f * x[i].im; break;
            case O_OP_EMPTY_W_COPY: z[i] = x[i]; break;
            default: return -1;
            }
        }
        return 0;
    }
    if (dtype == O_DTYPE_FLOAT) {
        const float *x = (const float *)a;
        float *z = (float *)c;
        for (i = 0; i < n; i++) {
            switch (op) {
            case O_OP_MULTIPLY: z[i] = x[i] * k; break;
            case O_OP_ADD:      z[i] = x[i] + k; break;
            case O_OP_SUBTRACT: z[i] = x[i] - k; break;
            case O_OP_EMPTY_W_COPY: z[i] = x[i]; break;
            default: return -1;
            }
        }
        return 0;
    }
    if (dtype == O_DTYPE_INT) {
        const uint32_t *x = (const uint32_t *)a;
        uint32_t *z = (uint32_t *)c;
        uint32_t ki = (uint32_t)(int32_t)k;
        for (i = 0; i < n; i++) {
            switch (op) {
            case O_OP_MULTIPLY: z[i] = x[i] * ki; break;
            case O_OP_ADD:      z[i] = x[i] + ki; break;
            case O_OP_SUBTRACT: z[i] = x[i] - ki; break;
            case O_OP_EMPTY_W_COPY: z[i] = x[i]; break;
            default: return -1;
            }
        }
        return 0;
    }
    return -1;
}
