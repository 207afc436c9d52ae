/*
 * oracle.h -- CPU restatement of the gr-clenabled streaming-DSP hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so.  The product path (gr-clenabled_amd/) never links,
 * imports or calls anything declared here and fails loudly when the HIP
 * library is missing.
 *
 * Every function restates, in plain C, the algorithm of one reference
 * function; the reference file:line it follows is cited at each definition
 * (paths relative to the reference repository root).
 *
 * Pin status (see DESIGN.md section "Oracle"):
 *   - the reference cannot be compiled in this image (its CPU mirrors need
 *     <config.h>, <gnuradio/...>, <volk/volk.h>, <fftw3.h>, <boost/...>, none
 *     of which exist here), so there is no oracle/_ref build;
 *   - math ops, FFT, window, firdes, fft_filter sizes: pinned by the
 *     reference's own known-answer tests and by the outputs of the reference
 *     files recorded in SURVEY.md section 8(c) (tests/golden/kat.json);
 *   - polyphase channelizer and X-engine: the reference holds nothing for
 *     them (no test vectors, no CPU implementation, no compilable source), so
 *     they cannot be pinned by reference-held data; they are pinned by
 *     implementations that are not this repository's -- scipy.signal.upfirdn
 *     per channel, numpy.einsum + numpy.tril_indices, scipy.signal.correlate
 *     (tests/golden/independent_golden.npz) -- and cross-checked against
 *     float64 closed forms.
 *
 * cpu_bench.c (same directory) times these restatements from C for bench.py's
 * cpu_baseline leg; it is measurement infrastructure under the same rule.
 */
#ifndef CLENABLED_ORACLE_H
#define CLENABLED_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } ocplx; /* == gr_complex == SComplex (include/clenabled/clSComplex.h:12-17) */

/* data-type / operator codes (include/clenabled/GRCLBase.h:57-62, clMathOpTypes.h:11-20) */
enum { O_DTYPE_COMPLEX = 1, O_DTYPE_FLOAT = 2, O_DTYPE_INT = 3, O_DTYPE_SHORT = 4, O_DTYPE_BYTE = 5, O_DTYPE_PACKEDXY = 6 };
enum { O_OP_MULTIPLY = 1, O_OP_ADD = 2, O_OP_SUBTRACT = 3, O_OP_CONJUGATE = 4, O_OP_MULTIPLY_CONJUGATE = 5,
       O_OP_EMPTY_W_COPY = 254, O_OP_EMPTY = 255 };

/* ---- elementwise family ------------------------------------------------ */
int oracle_mathop(int dtype, int op, size_t n, const void *a, const void *b, void *c);
int oracle_mathconst(int dtype, int op, float k, size_t n, const void *a, void *c);

/* ---- windows / filter design ------------------------------------------- */
enum { O_WIN_HAMMING = 0, O_WIN_HANN = 1, O_WIN_BLACKMAN = 2, O_WIN_RECTANGULAR = 3, O_WIN_KAISER = 4,
       O_WIN_BLACKMAN_HARRIS = 5, O_WIN_BARTLETT = 6, O_WIN_FLATTOP = 7 };
int    oracle_window(int type, int ntaps, double beta, float *out);
double oracle_window_max_attenuation(int type, double beta);
int    oracle_firdes_ntaps(double fs, double transition_width, int win_type, double beta);
int    oracle_firdes_low_pass(double gain, double fs, double cutoff, double transition_width,
                              int win_type, double beta, float *taps, int cap);

/* ---- FFT ---------------------------------------------------------------- */
/* unnormalised DFT, sign=-1 forward / +1 backward, float arithmetic (stand-in for FFTW3f plan execute) */
int oracle_fft_c2c_f32(int n, int sign, const ocplx *in, ocplx *out);
/* same definition evaluated in float64 then rounded: the mathematical truth used for tolerances */
int oracle_fft_c2c_f64(int n, int sign, const ocplx *in, ocplx *out);
/* block-level semantics of clFFT (window, shift, direction, real input), nvec frames */
int oracle_fft_block(int n, int forward, const float *window /* NULL or n */, int shift, int dtype,
                     int nvec, const void *in, ocplx *out, int use_f64);

/* ---- FIR / FFT filters --------------------------------------------------- */
typedef struct oracle_fft_filter oracle_fft_filter;
oracle_fft_filter *oracle_fft_filter_new(int decimation, const float *taps, int ntaps);
void oracle_fft_filter_free(oracle_fft_filter *f);
int  oracle_fft_filter_set_taps(oracle_fft_filter *f, const float *taps, int ntaps); /* returns nsamples */
int  oracle_fft_filter_fftsize(const oracle_fft_filter *f);
int  oracle_fft_filter_nsamples(const oracle_fft_filter *f);
int  oracle_fft_filter_xformed_taps(const oracle_fft_filter *f, ocplx *out);
int  oracle_fft_filter_filter(oracle_fft_filter *f, int nitems, const ocplx *in, ocplx *out);

int oracle_fir_ccf_filterN(const float *taps, int ntaps, const ocplx *in, ocplx *out, size_t n, int decim);
int oracle_fir_ccc_filterN(const ocplx *taps, int ntaps, const ocplx *in, ocplx *out, size_t n, int decim);

/* ---- polyphase channelizer ----------------------------------------------- */
int oracle_pfb_channelizer(const float *taps, int ntaps, int buf_items, int nch, int ninputs_per_iter,
                           const int *ch_map, int nmap, const ocplx *in /* history-prefixed */, ocplx *out,
                           int use_f64);

/* ---- X-engine --------------------------------------------------------------- */
size_t oracle_xengine_out_len(int ninputs, int nchan, int npol);
/* float path: in is [t][station][chan][pol] gr_complex */
int oracle_xengine_cf32(int ninputs, int nchan, int npol, int ntime, const ocplx *in, ocplx *out, int accumulate);
/* IChar path: in is [t][station][chan][pol]{I,Q} int8; exact=1 -> int64 sums then one scale */
int oracle_xengine_ichar(int ninputs, int nchan, int npol, int ntime, const int8_t *in, ocplx *out,
                         int accumulate, int exact);
/* packed 4-bit: in is [t][station][chan]{X byte, Y byte}; npol forced 2 */
int oracle_xengine_packed4(int ninputs, int nchan, int ntime, const uint8_t *in, ocplx *out, int accumulate);
/* host frame gather of work_processor() */
int oracle_xengine_gather(int dtype, int ninputs, int nchan, int npol, int nframes, int frame0,
                          const void *const *inputs, void *frame_buffer);

/* ---- remaining elementwise blocks; kind = MI355_ELEM_* code ---- */
int oracle_elem(int kind, float p0, float p1, size_t n, const void *in0, const void *in1, void *out0, void *out1);

/* ---- frequency-domain cross-correlator: inputs[num_inputs] of [nframes][n], outputs[num_inputs-1] ---- */
int oracle_xcorr_fft(int n, int num_inputs, int input_type, int nframes, const ocplx *const *inputs, float *const *outputs,
                     int use_f64);

#ifdef __cplusplus
}
#endif
#endif
