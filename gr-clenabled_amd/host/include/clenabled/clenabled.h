// gr::clenabled block API for the hot path, MI355X build.  The make() signatures are the
// reference's public headers verbatim (positional order as lib/*_impl.cc defines and GRC emits):
//   clMathOp.h:42, clMathConst.h:51-54, clFFT.h:54-55 (+ lib/clFFT_impl.cc:34-36 order),
//   clFilter.h:52-61, clComplexFilter.h:706-709, clPolyphaseChannelizer.h:48-49, clXEngine.h:48-52.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "gr_compat.h"

// include/clenabled/GRCLBase.h:57-70, clMathOpTypes.h:11-20
#define DTYPE_COMPLEX 1
#define DTYPE_FLOAT 2
#define DTYPE_INT 3
#define DTYPE_SHORT 4
#define DTYPE_BYTE 5
#define DTYPE_PACKEDXY 6
#define OCLTYPE_GPU 1
#define OCLTYPE_ACCELERATOR 2
#define OCLTYPE_CPU 3
#define OCLTYPE_ANY 4
#define OCLDEVICESELECTOR_FIRST 1
#define OCLDEVICESELECTOR_SPECIFIC 2
#define MATHOP_MULTIPLY 1
#define MATHOP_ADD 2
#define MATHOP_SUBTRACT 3
#define MATHOP_COMPLEX_CONJUGATE 4
#define MATHOP_MULTIPLY_CONJUGATE 5
#define MATHOP_EMPTY 255
#define MATHOP_EMPTY_W_COPY 254
#define CLFFT_FORWARD (-1)
#define CLFFT_BACKWARD 1
#define CLXCORR_TRIANGULAR_ORDER 1
#define CLXCORR_FULL_MATRIX 2

namespace gr {
namespace clenabled {

struct XComplex { float real = 0.0f, imag = 0.0f; };  // lib/clXEngine_impl.h:34-38

class clMathOp : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clMathOp> sptr;
    static sptr make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, int operatorType,
                     int setDebug = 0);
    virtual int testOpenCL(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items) = 0;
};

class clMathConst : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clMathConst> sptr;
    static sptr make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, float fValue,
                     int operatorType, int setDebug = 0);
    virtual float k() const = 0;
    virtual void set_k(float newValue) = 0;
    virtual int testOpenCL(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items) = 0;
};

class clFFT : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clFFT> sptr;
    // positional order of lib/clFFT_impl.cc:34-36 (what GRC passes; the reference header's names differ)
    static sptr make(int fftSize, int clFFTDir, const std::vector<float> &window, int idataType, int openCLPlatformType,
                     int devSelector, int platformId, int devId, int setDebug = 0, int num_streams = 1, bool shift = false);
    // counts SAMPLES like the reference's test hook (lib/clFFT_impl.cc:520-524)
    virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

const bool DEFAULT_USE_TIME_DOMAIN_SETTING = false;  // clFilter.h:32

class clFilter : virtual public gr::sync_decimator {
public:
    typedef std::shared_ptr<clFilter> sptr;
    static sptr make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                     const std::vector<float> &taps, int nthreads = 1, int setDebug = 0,
                     bool use_time = DEFAULT_USE_TIME_DOMAIN_SETTING);
    virtual void set_taps2(const std::vector<float> &taps) = 0;
    virtual std::vector<float> taps() const = 0;
    virtual void set_nthreads(int n) = 0;
    virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

class clComplexFilter : virtual public gr::sync_decimator {
public:
    typedef std::shared_ptr<clComplexFilter> sptr;
    static sptr make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                     const std::vector<gr_complex> &taps, int nthreads = 1, int setDebug = 0);
    virtual void set_taps2(const std::vector<gr_complex> &taps) = 0;
    virtual std::vector<gr_complex> taps() const = 0;
    virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

class clPolyphaseChannelizer : virtual public gr::block {
public:
    typedef std::shared_ptr<clPolyphaseChannelizer> sptr;
    static sptr make(int openCLPlatformType, int devSelector, int platformId, int devId, const std::vector<float> &taps,
                     int buf_items, int num_channels, int ninputs_per_iter, const std::vector<int> &ch_map, int setDebug = 0);
};

class clXEngine : virtual public gr::block {
public:
    typedef std::shared_ptr<clXEngine> sptr;
    static sptr make(int openCLPlatformType, int devSelector, int platformId, int devId, bool setDebug, int data_type,
                     int polarization, int num_inputs, int output_format, int first_channel, int num_channels, int integration,
                     std::vector<std::string> antenna_list, bool output_file = false, std::string file_base = "",
                     int rollover_size_mb = 0, bool internal_synchronizer = false, long sync_timestamp = 0,
                     std::string object_name = "", double starting_chan_center_freq = 0.0, double channel_width = 0.0,
                     bool disable_output = false, int pipeline_integration = 0);
    // lib/clXEngine_impl.h:176-201
    virtual long get_input_buffer_size() = 0;
    virtual long get_output_buffer_size() = 0;
    virtual void xcorrelate(XComplex *input_matrix, XComplex *cross_correlation) = 0;
    virtual void xcorrelate(char *input_matrix, XComplex *cross_correlation) = 0;
    // asynchronous, double-buffered form (what start()/runThread() do with a worker thread in the
    // reference, lib/clXEngine_impl.cc:304-382,1234-1299): at most two integrations in flight
    virtual void submit(const void *input_matrix, const XComplex *accumulator = nullptr) = 0;
    virtual void wait(XComplex *cross_correlation) = 0;
    // work_test(): the scheduler-free entry the reference's CLI times (lib/clXEngine_impl.cc:1144-1150 ->
    // work_processor :918-1142): gathers up to noutput_items frames of every input stream into the
    // integration window; a full window is correlated asynchronously, and the PREVIOUS result is
    // delivered (result handler = the "xcorr" PDU port in standalone mode; file sink + JSON sidecar).
    virtual int work_test(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    // stands for message_port_pub(pmt::mp("xcorr"), cons("triang_matrix", c32vector)) (:1076-1077)
    typedef void (*result_handler_t)(void *user, const XComplex *matrix, size_t matrix_flat_length);
    virtual void set_result_handler(result_handler_t fn, void *user) = 0;
    virtual long integrations_delivered() const = 0;
    // frames of every input stream -> the frame buffer, lib/clXEngine_impl.cc:987-1061
    virtual int gather_frames(int nframes, int frame0, gr_vector_const_void_star &input_items, void *frame_buffer) = 0;
    // stream-tag synchroniser state (internal_synchronizer = true, lib/clXEngine_impl.cc:1158-1226)
    virtual bool synchronized() const = 0;
    virtual uint64_t sync_tag() const = 0;
};

// ---- remaining elementwise family; make() signatures of include/clenabled/cl<Name>.h:49 ----
#define MI355_DECLARE_SYNC_BLOCK(NAME, ...)                                                                          \
    class NAME : virtual public gr::sync_block {                                                                      \
    public:                                                                                                           \
        typedef std::shared_ptr<NAME> sptr;                                                                           \
        static sptr make(__VA_ARGS__);                                                                                \
        virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items,                            \
                               gr_vector_void_star &output_items) = 0;                                                \
    }
MI355_DECLARE_SYNC_BLOCK(clLog, int openCLPlatformType, int devSelector, int platformId, int devId, float nValue, float kValue,
                         int setDebug = 0);
MI355_DECLARE_SYNC_BLOCK(clSNR, int openCLPlatformType, int devSelector, int platformId, int devId, float nValue, float kValue,
                         int setDebug = 0);
MI355_DECLARE_SYNC_BLOCK(clComplexToMag, int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug = 0);
MI355_DECLARE_SYNC_BLOCK(clComplexToArg, int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug = 0);
MI355_DECLARE_SYNC_BLOCK(clComplexToMagPhase, int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug = 0);
MI355_DECLARE_SYNC_BLOCK(clMagPhaseToComplex, int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug = 0);
MI355_DECLARE_SYNC_BLOCK(clQuadratureDemod, float gain, int openCLPlatformType, int devSelector, int platformId, int devId,
                         int setDebug = 0);
#undef MI355_DECLARE_SYNC_BLOCK

// ---- frequency-domain cross-correlator, include/clenabled/clxcorrelate_fft_vcf.h:50 ----
// io: num_inputs vectors of fftSize complex in (input 0 = reference), num_inputs-1 vectors of fftSize float out
class clxcorrelate_fft_vcf : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clxcorrelate_fft_vcf> sptr;
    // input_type 1 = the inputs are spectra, 2 = time series (forward FFT first)
    static sptr make(int fftSize, int num_inputs, int openCLPlatformType, int devSelector, int platformId, int devId,
                     int input_type = 1);
    virtual int work_test(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
