/*
 * mi355_clenabled.h -- C ABI of the MI355X (gfx950) streaming-DSP hot path that
 * replaces gr-clenabled's OpenCL runtime shim and kernel bodies.
 *
 * This is the drop-in boundary: the reference's block implementations
 * (lib/cl*_impl.cc) inherit GRCLBase (include/clenabled/GRCLBase.h:77-141,
 * lib/GRCLBase.cpp:17-474) and call cl::CommandQueue / clFFT from work();
 * a gr-clenabled maintainer binds the entry points below instead (see
 * INTEGRATION.md for the exact _impl stubs).  Plain C, opaque handles, raw
 * pointers and sizes only; nothing here throws or calls exit().
 *
 * Conventions
 *   - every function returns MI355_OK (0) or a negative MI355_ERR_* code;
 *     mi355_last_error() gives the thread's last diagnostic string;
 *   - gr_complex == { float re, im } interleaved, 8 bytes
 *     (include/clenabled/clSComplex.h:12-17);
 *   - `*_work(...)`     take HOST pointers, exactly what GNU Radio hands a
 *     block's work(); the call stages through pinned double buffers
 *     (H2D / kernel / D2H overlapped on two HIP streams) and returns when the
 *     output is complete -- the contract of the reference's blocking
 *     enqueueReadBuffer (e.g. lib/clMathOp_impl.cc:438);
 *   - `*_work_dev(...)` take DEVICE pointers plus a hipStream_t (as void*; NULL
 *     = HIP's default stream; pass mi355_ctx_stream() for the context's own)
 *     and only enqueue: this is the device-resident path chained blocks and
 *     the benchmarks use;
 *   - there is NO CPU fallback: OCLTYPE_CPU (3) is refused with
 *     MI355_ERR_UNSUPPORTED and every call fails loudly without a gfx950 GPU.
 */
#ifndef MI355_CLENABLED_H
#define MI355_CLENABLED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK               0
#define MI355_ERR_INVALID_ARG (-1)
#define MI355_ERR_NO_DEVICE   (-2)
#define MI355_ERR_UNSUPPORTED (-3)
#define MI355_ERR_HIP         (-4)
#define MI355_ERR_NOMEM       (-5)
#define MI355_ERR_STATE       (-6)

/* data types: include/clenabled/GRCLBase.h:57-62 */
#define MI355_DTYPE_COMPLEX  1
#define MI355_DTYPE_FLOAT    2
#define MI355_DTYPE_INT      3
#define MI355_DTYPE_SHORT    4
#define MI355_DTYPE_BYTE     5
#define MI355_DTYPE_PACKEDXY 6
/* device selection: include/clenabled/GRCLBase.h:64-70 */
#define MI355_OCLTYPE_GPU         1
#define MI355_OCLTYPE_ACCELERATOR 2
#define MI355_OCLTYPE_CPU         3
#define MI355_OCLTYPE_ANY         4
#define MI355_DEVSEL_FIRST    1
#define MI355_DEVSEL_SPECIFIC 2
/* operators: include/clenabled/clMathOpTypes.h:11-20 */
#define MI355_OP_MULTIPLY           1
#define MI355_OP_ADD                2
#define MI355_OP_SUBTRACT           3
#define MI355_OP_COMPLEX_CONJUGATE  4
#define MI355_OP_MULTIPLY_CONJUGATE 5
#define MI355_OP_EMPTY_W_COPY     254
#define MI355_OP_EMPTY            255
/* FFT direction as GRC passes it (clFFT's enum; lib/clFFT_impl.cc:84-89,
 * grc/clenabled_clFFT.block.yml:37-41) */
#define MI355_FFT_FORWARD  (-1)
#define MI355_FFT_BACKWARD   1

typedef struct mi355_ctx     mi355_ctx;
typedef struct mi355_mathop  mi355_mathop;
typedef struct mi355_mathconst mi355_mathconst;
typedef struct mi355_fft     mi355_fft;
typedef struct mi355_filter  mi355_filter;
typedef struct mi355_pfb     mi355_pfb;
typedef struct mi355_xengine mi355_xengine;
typedef struct mi355_elem    mi355_elem;
typedef struct mi355_xcorr_fft mi355_xcorr_fft;

/* ---------------------------------------------------------------------------
 * Runtime: replaces GRCLBase::InitOpenCL / cleanup (lib/GRCLBase.cpp:17-369,
 * 423-474) and the ctor arguments (include/clenabled/GRCLBase.h:136-137).
 * ------------------------------------------------------------------------- */
const char *mi355_strerror(int code);
const char *mi355_last_error(void);
const char *mi355_version(void);
/* Diagnostics sink.  The reference writes through GNU Radio's logger (GR_LOG_INFO / GR_LOG_ERROR, lib/clXEngine_impl.cc:107,
 * 137,257) and to std::cout when setDebug is on (lib/GRCLBase.cpp:96-120); a C library has neither, so the block layer hands
 * its logger in here.  Process wide; fn == NULL restores the default (DEBUG/INFO lines of a context created with debug != 0 go
 * to stderr, errors are only kept for mi355_last_error()).  The callback receives every such line and every error message at
 * MI355_LOG_ERROR, on the calling thread, with no lock held; `message` is valid only during the call. */
#define MI355_LOG_DEBUG 0
#define MI355_LOG_INFO  1
#define MI355_LOG_WARN  2
#define MI355_LOG_ERROR 3
typedef void (*mi355_log_fn)(void *user, int level, const char *message);
int mi355_set_log_callback(mi355_log_fn fn, void *user);
/* number of gfx950 devices visible, or a negative error */
int mi355_device_count(void);
/* ocl_type 1/2/4 -> HIP device; 3 (CPU) -> MI355_ERR_UNSUPPORTED.
 * dev_selector FIRST -> ordinal 0; SPECIFIC -> ordinal dev_id (platform_id must
 * be 0: there is one HIP platform). */
int mi355_ctx_create(int ocl_type, int dev_selector, int platform_id, int dev_id, int debug, mi355_ctx **out);
int mi355_ctx_destroy(mi355_ctx *ctx);
int mi355_ctx_device(const mi355_ctx *ctx);
/* the context's compute stream (hipStream_t) */
void *mi355_ctx_stream(mi355_ctx *ctx);
int mi355_ctx_synchronize(mi355_ctx *ctx);
/* device memory helpers for callers without their own allocator (C++ harness) */
int mi355_malloc(mi355_ctx *ctx, size_t bytes, void **dptr);
int mi355_free(mi355_ctx *ctx, void *dptr);
int mi355_memcpy_h2d(mi355_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int mi355_memcpy_d2h(mi355_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* Device-side strided block copy, dst[b][r][0..width) = src[b][r][0..width) for b < nblocks, r < rows, with independent
 * row pitches and block strides (bytes).  Used to pack the per-peer channel slices of the X-engine's all-to-all corner
 * turn (the reference has no multi-device path). */
int mi355_pack3d_dev(mi355_ctx *ctx, void *dst_dev, const void *src_dev, size_t width_bytes, size_t rows, size_t nblocks,
                     size_t src_pitch, size_t src_block_stride, size_t dst_pitch, size_t dst_block_stride, void *stream);

/* ---------------------------------------------------------------------------
 * clMathOp: c = a (op) b.  Replaces clMathOp_impl::processOpenCL
 * (lib/clMathOp_impl.cc:361-442) and its kernels (:104-238).
 * dtype COMPLEX/FLOAT/INT; op MULTIPLY/ADD/SUBTRACT (+MULTIPLY_CONJUGATE for
 * complex).  max_items is a sizing hint (0 -> 8192, :80-84); buffers grow.
 * ------------------------------------------------------------------------- */
int mi355_mathop_create(mi355_ctx *ctx, int dtype, int op, size_t max_items, mi355_mathop **out);
int mi355_mathop_destroy(mi355_mathop *h);
int mi355_mathop_work(mi355_mathop *h, size_t nitems, const void *a, const void *b, void *c);
int mi355_mathop_work_dev(mi355_mathop *h, size_t nitems, const void *a, const void *b, void *c, void *stream);

/* ---------------------------------------------------------------------------
 * clMathConst: c = a (op) k with a REAL scalar k applied to both components of
 * a complex item.  Replaces clMathConst_impl::processOpenCL
 * (lib/clMathConst_impl.cc:311-361), kernels (:100-225), k()/set_k()
 * (lib/clMathConst_impl.h:107-108).  op MULTIPLY/ADD/SUBTRACT,
 * COMPLEX_CONJUGATE (complex only), EMPTY_W_COPY (copy), EMPTY (no-op launch).
 * ------------------------------------------------------------------------- */
int mi355_mathconst_create(mi355_ctx *ctx, int dtype, int op, float k, size_t max_items, mi355_mathconst **out);
int mi355_mathconst_destroy(mi355_mathconst *h);
int mi355_mathconst_set_k(mi355_mathconst *h, float k);
int mi355_mathconst_get_k(const mi355_mathconst *h, float *k);
int mi355_mathconst_work(mi355_mathconst *h, size_t nitems, const void *a, void *c);
int mi355_mathconst_work_dev(mi355_mathconst *h, size_t nitems, const void *a, void *c, void *stream);

/* ---------------------------------------------------------------------------
 * clFFT: per frame Y = [fftshift] FFT_N( x .* window ), unnormalised in both
 * directions.  Replaces the clFFT plan + MultiplyFloat kernel + host fftshift
 * of clFFT_impl (ctor lib/clFFT_impl.cc:65-151, processOpenCL :526-634).
 * fft_size: any power of two 2..32768 (one fused kernel), 65536..1048576 (two passes over 16-column tiles), 2097152..16777216
 * (four passes; 2^24 is clFFT's own single-precision limit); lengths 2^a 3^b 5^c 7^d 11^e 13^f up to 15360 (14336 / 13312 / 11264 with a factor 7 / 13 / 11) that are not a power of two in
 * one pass by a mixed-radix kernel (the lengths clFFT's radix-3/5/7 plans cover; its workgroup shape is measured once per length
 * and process at create, about 20 ms), such lengths up to 921600 (= N1 x N2, both at most 960) in two passes; any other size 3..8388608 by chirp-z over the power-of-two kernels (the reference
 * leaves those to clFFT, which refuses prime factors above 13);
 * larger sizes return MI355_ERR_UNSUPPORTED.  The shift of an odd-sized frame follows
 * clFFT_impl::testCPU (len = ceil(N/2), :503-507).  window: NULL/0 or exactly fft_size floats
 * (:74-76).  dtype COMPLEX or FLOAT (real input, complex output).
 * `nvec` = number of frames per stream (= noutput_items of work(), :637-654).
 * ------------------------------------------------------------------------- */
int mi355_fft_create(mi355_ctx *ctx, int fft_size, int direction, const float *window, int window_len,
                     int dtype, int num_streams, int shift, mi355_fft **out);
int mi355_fft_destroy(mi355_fft *h);
/* Which path a length takes, as text ("one pass", "two tile passes 256 x 512", "mixed radix 10 x 10 x 10",
 * "chirp-z, m = 8192 (fused)", ...): planning only, no device is touched.  Returns MI355_OK, or MI355_ERR_UNSUPPORTED with the
 * reason in buf for a length mi355_fft_create would refuse. */
int mi355_fft_plan_text(int fft_size, char *buf, int buf_len);
int mi355_fft_work(mi355_fft *h, int nvec, const void *const *in_streams, void *const *out_streams);
/* Concurrency: sizes up to 32768 are stateless (any number of work_dev calls of one handle may be in flight on any streams).
 * 65536 points and more, and the non-power-of-two sizes above 2048, go through ONE per-handle workspace: calls from different threads or
 * streams are accepted and serialised on it (the later stream waits for the earlier call's kernels); use one handle per
 * stream for overlap.  Sizes: powers of two 2 .. 16777216, any other length 3 .. 8388608 (chirp-z); larger ones return
 * MI355_ERR_UNSUPPORTED. */
int mi355_fft_work_dev(mi355_fft *h, int nvec, const void *in, void *out, void *stream);

/* ---------------------------------------------------------------------------
 * clFilter / clComplexFilter: decimating FIR on a complex stream,
 *     y[m] = sum_k h[k] * x[m*decimation - k]
 * Replaces clFilter_impl (ctor lib/clFilter_impl.cc:50-83, filterGPUTimeDomain
 * :504-589, filterGPUFrequencyDomain :591-681, set_taps2 :441-479) and
 * clComplexFilter_impl::filterGPU (lib/clComplexFilter_impl.cc:959-1030).
 * `in` is GNU Radio's history-prefixed buffer (set_history(ntaps), :78):
 * in[ntaps-1] is x[0]; noutput*decimation + ntaps - 1 samples are read.
 * use_time = 0 : fused overlap-save fast convolution (FFT -> xH -> IFFT in LDS); the transform size is chosen
 *                for throughput (>= the reference's 2*2^ceil(log2 ntaps), lib/fft_filter.cc:72-97); a filter longer
 *                than 2048 taps is partitioned into ceil(ntaps/2048) segments whose spectra are applied to delayed
 *                input blocks and summed before ONE inverse transform per block (same y; 10-25x the direct form's rate)
 * use_time = 1 : direct-form tap dot product
 * complex_taps = 1 : taps is ntaps gr_complex (clComplexFilter), else floats.
 * ------------------------------------------------------------------------- */
int mi355_filter_create(mi355_ctx *ctx, int decimation, const void *taps, int ntaps, int complex_taps,
                        int use_time, mi355_filter **out);
int mi355_filter_destroy(mi355_filter *h);
int mi355_filter_set_taps(mi355_filter *h, const void *taps, int ntaps);
int mi355_filter_ntaps(const mi355_filter *h);
int mi355_filter_get_taps(const mi355_filter *h, void *taps_out, int cap);
/* FFT size the fast-convolution kernel runs (0 in time-domain mode; 4096 for partitioned filters of more than 2048 taps) */
int mi355_filter_fftsize(const mi355_filter *h);
int mi355_filter_work(mi355_filter *h, size_t noutput_items, const void *in_with_history, void *out);
int mi355_filter_work_dev(mi355_filter *h, size_t noutput_items, const void *in_with_history, void *out, void *stream);

/* ---------------------------------------------------------------------------
 * clPolyphaseChannelizer: M-branch polyphase filterbank + M-point backward DFT
 * + channel map, one multiplexed output stream.  Replaces
 * clPolyphaseChannelizer_impl (ctor lib/clPolyphaseChannelizer_impl.cc:47-67,
 * general_work :83-109, kernels :153-177, clFFT plan :208-225).
 * One call consumes buf_items inputs and produces nmap*buf_items/ninputs_per_iter
 * outputs.  `in` is history-prefixed (set_history(ntaps), :63) and must hold
 * buf_items - ninputs_per_iter + ntaps samples.
 * ------------------------------------------------------------------------- */
int mi355_pfb_create(mi355_ctx *ctx, const float *taps, int ntaps, int buf_items, int num_channels,
                     int ninputs_per_iter, const int *ch_map, int nmap, mi355_pfb **out);
int mi355_pfb_destroy(mi355_pfb *h);
int mi355_pfb_noutput(const mi355_pfb *h);
int mi355_pfb_ninput(const mi355_pfb *h);
int mi355_pfb_work(mi355_pfb *h, const void *in_with_history, void *out);
int mi355_pfb_work_dev(mi355_pfb *h, const void *in_with_history, void *out, void *stream);
/* nbuf consecutive buffers in one launch (general_work() offered nbuf output multiples): in holds
 * nbuf * buf_items - ninputs_per_iter + ntaps samples, out nbuf * noutput().  Same samples as nbuf single calls. */
int mi355_pfb_work_dev_n(mi355_pfb *h, int nbuf, const void *in_dev, void *out_dev, void *stream);

/* ---------------------------------------------------------------------------
 * clXEngine: V[f][k][pol2] = sum_t x_s1(t,f) conj(x_s2(t,f)), k = s1(s1+1)/2+s2.
 * Replaces clXEngine_impl's device side: xcorrelate() overloads
 * (lib/clXEngine_impl.h:150-201), kernels (lib/clXEngine_impl.cc:605-916),
 * accumulator zero/"+=" for pipeline integration (:289-292,785-796), and the
 * host frame gather of work_processor (:987-1061).
 * data_type COMPLEX (cf32), BYTE (int8 I,Q) or PACKEDXY (4-bit, npol forced 2).
 * Input layout [t][station][chan][pol]; output matrix_flat_length =
 * nchan * ninputs(ninputs+1)/2 * npol^2 gr_complex, triangular order.
 * ------------------------------------------------------------------------- */
int mi355_xengine_create(mi355_ctx *ctx, int data_type, int npol, int num_inputs, int num_channels,
                         int integration, mi355_xengine **out);
int mi355_xengine_destroy(mi355_xengine *h);
size_t mi355_xengine_input_bytes(const mi355_xengine *h);
size_t mi355_xengine_output_items(const mi355_xengine *h);
/* accumulate=0: out = V ; accumulate=1: out += V (pipeline integration) */
int mi355_xengine_xcorrelate(mi355_xengine *h, const void *in_host, void *out_host, int accumulate);
/* (device-pointer calls: a handle may be used from several streams -- launches that share the handle's partial-sum workspace are ordered by the
 * library when the stream changes: behind one of the context's own streams with an event, behind a caller's stream -- which may have been
 * destroyed since, and is therefore never touched again -- by waiting for the device.  A stream may be destroyed as soon as the caller is done
 * with it.  Batched launches that need no workspace -- no time ranges -- are not ordered at all.) */
int mi355_xengine_xcorrelate_dev(mi355_xengine *h, const void *in_dev, void *out_dev, int accumulate, void *stream);
/* Multi-GPU form (no counterpart in the reference, which runs one X-engine on one device): the input is the
 * receive buffer of the all-to-all corner turn, [group][t][station in group][chan][pol], stations_per_group stations per
 * sending rank; it is read in place.  IChar geometries of the fused path only (<= 64 rows, rows of whole 16-byte pieces),
 * otherwise MI355_ERR_UNSUPPORTED. */
int mi355_xengine_xcorrelate_grouped_dev(mi355_xengine *h, const void *in_dev, void *out_dev, int accumulate,
                                         int stations_per_group, void *stream);
/* Batched form: nint integration windows in ONE launch -- what the worker thread of the reference does one window at a time
 * (the per-integration loop of lib/clXEngine_impl.cc:1234-1299).  in_dev holds nint windows back to back, each in the reference's
 * frame layout (stations_per_group == 0 or num_inputs), or -- the receive buffer of one all-to-all over nint windows --
 * [group][window][t][station in group][chan][pol]; out_dev receives nint matrices back to back (accumulate: each += its window).
 * With few channels per device (the channel slab of one rank of an 8-GPU X-engine) one window cannot fill the device and pays two
 * dispatches; a batch runs (window x column slice x time range) workgroups, and once nint * slices >= CUs no partial sums at all.
 * Other sample formats / geometries run the windows one after the other (group-major input of several windows: UNSUPPORTED). */
int mi355_xengine_xcorrelate_n_dev(mi355_xengine *h, int nint, const void *in_dev, void *out_dev, int accumulate,
                                   int stations_per_group, void *stream);
/* Which kernels the handle's LAST device-side call ran (no counterpart in the reference, whose one kernel per data type is fixed at construction,
 * lib/clXEngine_impl.cc:605-916): the routes differ 2 x in speed and depend on geometry, alignment, window count and environment switches, so a
 * caller (and the tests) can assert the one it expects.  The first use of a route on a handle is also logged at MI355_LOG_DEBUG. */
typedef struct mi355_xe_route {
    char kernel[64];          /* e.g. "k_xe_i8_lines", "k_xe_i8_lines<split>", "k_xe_i8_fused", "k_xe_i8_fused+k_xe_i8_reduce", "k_xe_turn_lds+k_xe_corr_sb" */
    int launches;             /* launches the last call was cut into (mi355_xengine_xcorrelate_n_dev splits window counts between the good ones) */
    int windows;              /* integration windows of the last launch */
    int workgroups;           /* of the last launch (0: not recorded for this route) */
    int units_per_workgroup;  /* persistent forms: units a workgroup runs one after the other */
    int tsplit;               /* time ranges per window (1: none) */
    int in_launch_reduce;     /* the time ranges are combined by the kernel's own tail (0: by a second kernel, or no ranges) */
    int touches;              /* early touches of the slow lines: distance in K blocks (0: off) */
    int pace;                 /* pacing of a line's workgroups, half K blocks (0: off) */
} mi355_xe_route;
int mi355_xengine_last_route(const mi355_xengine *h, mi355_xe_route *out);
/* Double-buffered asynchronous form of the host path: replaces the reference's pinned double
 * buffers + worker thread (lib/clXEngine_impl.cc:304-382 start(), :1234-1299 runThread()).
 * submit() copies the integration window into a pinned slot and enqueues H2D + kernels + D2H on that
 * slot's stream (acc_host != NULL: out = acc + V, pipeline integration); at most two are in flight.
 * wait() blocks for the OLDEST one and writes its matrix. */
int mi355_xengine_submit(mi355_xengine *h, const void *in_host, const void *acc_host);
int mi355_xengine_wait(mi355_xengine *h, void *out_host);
int mi355_xengine_pending(const mi355_xengine *h);
/* Zero-copy form of submit(): acquire() hands out the pinned frame buffer of the next free slot (input_bytes() long, the
 * reference's pinned char_input / complex_input, lib/clXEngine_impl.cc:325-362); the block gathers its frames straight
 * into it (mi355_xengine_gather or its own copies) and submit_acquired() enqueues H2D + kernels + D2H.  MI355_ERR_STATE
 * when two integrations are in flight.  Between acquire() and submit_acquired() a plain submit() is refused. */
int mi355_xengine_acquire(mi355_xengine *h, void **frame_buffer);
int mi355_xengine_submit_acquired(mi355_xengine *h, const void *accumulator_or_null);
/* host gather: copy frames [0,nframes) of each input stream into time slots
 * frame0.. of a frame buffer laid out as the reference's pinned host buffer */
int mi355_xengine_gather(const mi355_xengine *h, int nframes, int frame0, const void *const *inputs, void *frame_buffer);
/* ---- clXEngine over several devices of ONE process.  The reference picks one device per block (devId,
 * lib/GRCLBase.cpp:115-134) and a GNU Radio flowgraph is one process: this handle owns `world` device contexts and runs the FX correlator's
 * corner turn between them.  Rank r = device_ids[r] (a device may appear more than once: the ranks then share it) ingests antenna group r --
 * frames [window][t][num_inputs/world stations][chan][pol]{I,Q}, the reference's frame layout of lib/clXEngine_impl.cc:987-1061 -- and
 * produces channels [r F/W, (r+1) F/W) of the reference's [chan][baseline][pol^2] matrix (:786-808) for each of `windows` integration
 * windows per exchange.  IChar (int8 I/Q) with num_inputs * npol <= 64 rows, or 64 inputs x 2 polarisations with channel slabs of whole 32-channel
 * lines and enough windows per exchange to fill the device (8 ranks x 1024 channels: 8); world must divide num_inputs and num_channels.
 * Per exchange: one strided device copy packs a rank's frames into per-destination blocks (mi355_pack3d_dev), `world` peer copies
 * (hipMemcpyPeerAsync: xGMI) deliver them, and mi355_xengine_xcorrelate_n_dev reads the receive buffer in place; two slots, an exchange and
 * a compute stream per rank, so exchange k+1 runs under correlation k.  (gr-clenabled_amd/shard.py is the same pipeline with one process per
 * device and an RCCL all-to-all.) */
typedef struct mi355_xengine_shard mi355_xengine_shard;
int mi355_xengine_shard_create(int world, const int *device_ids, int npol, int num_inputs, int num_channels, int integration, int windows,
                               mi355_xengine_shard **out);
int mi355_xengine_shard_destroy(mi355_xengine_shard *h);
int mi355_xengine_shard_world(const mi355_xengine_shard *h);
int mi355_xengine_shard_device(const mi355_xengine_shard *h, int rank);
size_t mi355_xengine_shard_frames_bytes(const mi355_xengine_shard *h);  /* bytes of one rank's frames per exchange */
size_t mi355_xengine_shard_slab_items(const mi355_xengine_shard *h);    /* complex floats of one rank's matrix per window */
void *mi355_xengine_shard_stream(mi355_xengine_shard *h, int rank);     /* the rank's compute stream (hipStream_t): producers of frames_dev go here */
/* the rank's compute stream waits for everything enqueued so far on `stream` (hipStream_t of that device): the other way to order a producer */
int mi355_xengine_shard_wait_stream(mi355_xengine_shard *h, int rank, void *stream);
/* enqueue one exchange + correlation: frames_dev[r] / out_dev[r] live on device_ids[r] (out: windows x slab_items complex floats);
 * submit only ENQUEUES, so the packing copy of this exchange may still be reading frames_dev[r] after the next submit has returned: rewrite a
 * rank's frames only from work enqueued on mi355_xengine_shard_stream(rank) AFTER that next submit (it is ordered behind the next correlation,
 * which is behind this exchange's copies), or after mi355_xengine_shard_synchronize -- a producer on any other stream is not ordered behind the
 * pack.  An error return may leave the exchange half enqueued: synchronize and resubmit. */
int mi355_xengine_shard_submit_dev(mi355_xengine_shard *h, const void *const *frames_dev, void *const *out_dev, int accumulate);
int mi355_xengine_shard_synchronize(mi355_xengine_shard *h);
/* host form: `windows` windows in the reference's layout [window][t][station][chan][pol] -> [window][chan][baseline][pol^2]; every rank
 * copies its antenna group over its own host link and its slab back; blocking (lib/clXEngine_impl.h:179-201 over `world` devices) */
int mi355_xengine_shard_xcorrelate(mi355_xengine_shard *h, const void *in_host, void *out_host, int accumulate);
/* Streaming host form -- what a block that gathers frames all the time uses; the sharded counterpart of mi355_xengine_acquire / submit_acquired /
 * wait, i.e. of the reference's pinned frame buffers (lib/clXEngine_impl.cc:325-362) and worker thread (:1234-1299).  acquire(): a PINNED buffer of
 * input_bytes() = `windows` integration windows in the reference's frame layout, to be filled by the caller (mi355_xengine_gather writes this
 * layout); submit_acquired(): per rank, on the rank's own stream, the asynchronous upload of its antenna group out of that buffer (W host links at
 * once), exchange, correlation, download of its slab into a pinned result -- enqueue only; wait(): blocks for the OLDEST exchange, writes `windows`
 * matrices [window][chan][baseline][pol^2].  Two exchanges may be in flight (MI355_ERR_STATE beyond that), so the gather and upload of exchange
 * k+1 overlap the devices' work on exchange k. */
int mi355_xengine_shard_windows(const mi355_xengine_shard *h);
size_t mi355_xengine_shard_input_bytes(const mi355_xengine_shard *h);
int mi355_xengine_shard_acquire(mi355_xengine_shard *h, void **frame_buffer);
int mi355_xengine_shard_submit_acquired(mi355_xengine_shard *h);
int mi355_xengine_shard_wait(mi355_xengine_shard *h, void *out_host);
int mi355_xengine_shard_pending(const mi355_xengine_shard *h);
/* Self-test of the IChar scale (lib/clXEngine_impl.cc:859-867: every sample / 127, i.e. every sum / 16129): the device evaluates the
 * single-precision form the matrix stores use and the double expression (float)((double)S * (1/127) * (1/127)) for EVERY sum S with
 * |S| <= 2^24 (the range the single-precision form is used in) and counts the sums where the two floats differ; *mismatches must be 0. */
int mi355_xengine_selftest_scale(mi355_ctx *ctx, long long *mismatches);

/* ---------------------------------------------------------------------------
 * Remaining elementwise family.  One handle type; `kind` selects the block:
 *   LOG10       float -> float            c = p0*log10(a) + p1            clLog   (lib/clLog_impl.cc:113-147)
 *   SNR         float,float -> float      c = |p0*log10(a/b) + p1|        clSNR   (lib/clSNR_impl.cc:98-116)
 *   C2MAG       complex -> float          sqrt(im^2+re^2)                 clComplexToMag (:138-148)
 *   C2ARG       complex -> float          (float)atan2((double)im,(double)re)   clComplexToArg (:136-151)
 *   C2MAGPHASE  complex -> float,float    both of the above               clComplexToMagPhase (:150-164)
 *   MAGPHASE2C  float,float -> complex    (mag*cos ph, mag*sin ph) in double    clMagPhaseToComplex (:170-191)
 *   QUADDEMOD   complex -> float          p0*atan2(a[i+1]*conj(a[i])) in double, input carries 1 item of
 *                                         history (set_history(2))        clQuadratureDemod (:81,118-146)
 * Unused in/out pointers are NULL.  n = output items.
 * ------------------------------------------------------------------------- */
#define MI355_ELEM_LOG10      1
#define MI355_ELEM_SNR        2
#define MI355_ELEM_C2MAG      3
#define MI355_ELEM_C2ARG      4
#define MI355_ELEM_C2MAGPHASE 5
#define MI355_ELEM_MAGPHASE2C 6
#define MI355_ELEM_QUADDEMOD  7
int mi355_elem_create(mi355_ctx *ctx, int kind, float p0, float p1, mi355_elem **out);
int mi355_elem_destroy(mi355_elem *h);
int mi355_elem_history(const mi355_elem *h);
int mi355_elem_work(mi355_elem *h, size_t n, const void *in0, const void *in1, void *out0, void *out1);
int mi355_elem_work_dev(mi355_elem *h, size_t n, const void *in0, const void *in1, void *out0, void *out1, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Frequency-domain cross-correlator, replaces clxcorrelate_fft_vcf:
 *   make(fftSize, num_inputs, openCLPlatformType, devSelector, platformId, devId, input_type)
 *                                                  include/clenabled/clxcorrelate_fft_vcf.h:50
 *   work(): lib/clxcorrelate_fft_vcf_impl.cc:1058-1143.
 * Input 0 is the reference signal.  For every frame (vector of fft_size complex items) and every other
 * input s = 1..num_inputs-1:
 *     out[s-1][frame] = halfswap( | IFFT_unscaled( X0 * conj(Xs) ) | )        (float, fft_size items)
 * where X = the input itself (input_type 1, spectra) or its forward FFT (input_type 2, time series) and
 * halfswap exchanges the two halves of the vector (:1133-1140).  inputs[] holds num_inputs pointers to
 * [nframes][fft_size] complex, outputs[] num_inputs-1 pointers to [nframes][fft_size] float.
 * fft_size: powers of two 16..4096 run as ONE fused kernel; every other even size up to 4194304 (the reference hands fftSize to
 * clFFT, lib/clxcorrelate_fft_vcf_impl.cc:711-737) runs the reference's steps one after the other over the clFFT transforms of
 * this library; an odd size is refused (the reference would leave the last output of every vector unwritten).  num_inputs 2..32.
 * ------------------------------------------------------------------------------------------------ */
int mi355_xcorr_fft_create(mi355_ctx *ctx, int fft_size, int num_inputs, int input_type, mi355_xcorr_fft **out);
int mi355_xcorr_fft_destroy(mi355_xcorr_fft *h);
int mi355_xcorr_fft_work(mi355_xcorr_fft *h, int nframes, const void *const *inputs, void *const *outputs);
int mi355_xcorr_fft_work_dev(mi355_xcorr_fft *h, int nframes, const void *const *d_inputs, void *const *d_outputs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_CLENABLED_H */
