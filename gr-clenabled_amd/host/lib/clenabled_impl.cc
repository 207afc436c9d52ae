// _impl classes of the hot-path blocks: every work() is a call into the C ABI (mi355_clenabled.h).
// Structure follows the reference's lib/cl*_impl.cc with the GRCLBase base class replaced by
// MI355Base; constructor-time errors throw the same exception types as the reference.
#include <clenabled/clenabled.h>
#include <mi355_clenabled.h>

#include <mutex>
#include <stdexcept>

namespace gr {
namespace clenabled {

namespace {
// stands where "public GRCLBase" was (include/clenabled/GRCLBase.h:77-141)
class MI355Base {
protected:
    mi355_ctx *d_ctx = nullptr;
    bool debugMode;
    MI355Base(int openCLPlatformType, int devSelector, int platformId, int devId, bool setDebug) : debugMode(setDebug)
    {
        // the reference prints and exit(0)s on failure (lib/GRCLBase.cpp:239-257); throw instead
        chk(mi355_ctx_create(openCLPlatformType, devSelector, platformId, devId, setDebug ? 1 : 0, &d_ctx), "mi355_ctx_create");
    }
    ~MI355Base() { mi355_ctx_destroy(d_ctx); }
    static void chk(int rc, const char *what)
    {
        if (rc != MI355_OK) throw std::runtime_error(std::string(what) + ": " + mi355_strerror(rc) + ": " + mi355_last_error());
    }
    static size_t dsize(int idataType)
    {
        switch (idataType) {  // lib/clMathOp_impl.cc:35-47
        case DTYPE_COMPLEX: return sizeof(gr_complex);
        case DTYPE_INT: return sizeof(int);
        default: return sizeof(float);
        }
    }
};

class clMathOp_impl : public clMathOp, public MI355Base {
    mi355_mathop *d_h = nullptr;
public:
    clMathOp_impl(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, int operatorType, bool setDebug)
        : gr::sync_block("clMathOp"), MI355Base(openCLPlatformType, devSelector, platformId, devId, setDebug)
    {
        chk(mi355_mathop_create(d_ctx, idataType, operatorType, 8192, &d_h), "mi355_mathop_create");
    }
    ~clMathOp_impl() override { mi355_mathop_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        chk(mi355_mathop_work(d_h, (size_t)noutput_items, in[0], in[1], out[0]), "mi355_mathop_work");
        return noutput_items;
    }
    int testOpenCL(int n, gr_vector_int &, gr_vector_const_void_star &in, gr_vector_void_star &out) override { return work(n, in, out); }
};

class clMathConst_impl : public clMathConst, public MI355Base {
    mi355_mathconst *d_h = nullptr;
public:
    clMathConst_impl(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, float fValue,
                     int operatorType, bool setDebug)
        : gr::sync_block("clMathConst"), MI355Base(openCLPlatformType, devSelector, platformId, devId, setDebug)
    {
        chk(mi355_mathconst_create(d_ctx, idataType, operatorType, fValue, 8192, &d_h), "mi355_mathconst_create");
    }
    ~clMathConst_impl() override { mi355_mathconst_destroy(d_h); }
    float k() const override { float v = 0; mi355_mathconst_get_k(d_h, &v); return v; }
    void set_k(float v) override { chk(mi355_mathconst_set_k(d_h, v), "mi355_mathconst_set_k"); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        chk(mi355_mathconst_work(d_h, (size_t)noutput_items, in[0], out[0]), "mi355_mathconst_work");
        return noutput_items;
    }
    int testOpenCL(int n, gr_vector_int &, gr_vector_const_void_star &in, gr_vector_void_star &out) override { return work(n, in, out); }
};

class clFFT_impl : public clFFT, public MI355Base {
    mi355_fft *d_h = nullptr;
    int d_fft_size;
public:
    clFFT_impl(int fftSize, int clFFTDir, const std::vector<float> &window, int idataType, int openCLPlatformType, int devSelector,
               int platformId, int devId, bool setDebug, int num_streams, bool shift)
        : gr::sync_block("clFFT"), MI355Base(openCLPlatformType, devSelector, platformId, devId, setDebug), d_fft_size(fftSize)
    {
        if (!(window.empty() || window.size() == (size_t)fftSize))  // lib/clFFT_impl.cc:74-76
            throw std::runtime_error("OpenCL FFT: window not the same length as fft_size\n");
        chk(mi355_fft_create(d_ctx, fftSize, clFFTDir, window.empty() ? nullptr : window.data(), (int)window.size(), idataType,
                             num_streams, shift ? 1 : 0, &d_h), "mi355_fft_create");
    }
    ~clFFT_impl() override { mi355_fft_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        chk(mi355_fft_work(d_h, noutput_items, in.data(), out.data()), "mi355_fft_work");  // noutput_items = vectors (:637-654)
        return noutput_items;
    }
    int testOpenCL(int nsamples, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        return work(nsamples / d_fft_size, in, out) * d_fft_size;
    }
};

template <class Base, class Tap>
class filter_impl_t : public Base, public MI355Base {
protected:
    mi355_filter *d_h = nullptr;
    std::mutex d_lock;
    bool d_updated = false;
public:
    filter_impl_t(const char *name, int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                  const std::vector<Tap> &taps, bool setDebug, bool complex_taps, bool use_time)
        : gr::sync_decimator(name, decimation), MI355Base(openclPlatform, devSelector, platformId, devId, setDebug)
    {
        chk(mi355_filter_create(d_ctx, decimation, taps.data(), (int)taps.size(), complex_taps, use_time, &d_h), "mi355_filter_create");
        this->set_history(taps.size());  // lib/clFilter_impl.cc:78
    }
    ~filter_impl_t() override { mi355_filter_destroy(d_h); }
    void set_taps2(const std::vector<Tap> &taps) override
    {
        std::lock_guard<std::mutex> g(d_lock);  // lib/clFilter_impl.cc:443
        chk(mi355_filter_set_taps(d_h, taps.data(), (int)taps.size()), "mi355_filter_set_taps");
        d_updated = true;
    }
    std::vector<Tap> taps() const override
    {
        std::vector<Tap> t(mi355_filter_ntaps(d_h));
        mi355_filter_get_taps(d_h, t.data(), (int)t.size());
        return t;
    }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        std::lock_guard<std::mutex> g(d_lock);
        if (d_updated) {  // lib/clFilter_impl.cc:774-789: new history first, produce nothing this call
            this->set_history(mi355_filter_ntaps(d_h));
            d_updated = false;
            return 0;
        }
        chk(mi355_filter_work(d_h, (size_t)noutput_items, in[0], out[0]), "mi355_filter_work");
        return noutput_items;
    }
    int testOpenCL(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override { return work(n, in, out); }
};

class clFilter_impl : public filter_impl_t<clFilter, float> {
public:
    clFilter_impl(int p, int s, int pl, int d, int decim, const std::vector<float> &taps, bool dbg, bool use_time)
        : gr::sync_decimator("clFilter", decim), filter_impl_t("clFilter", p, s, pl, d, decim, taps, dbg, false, use_time) {}
    void set_nthreads(int) override {}  // lib/clFilter_impl.cc:413-415 only configured the CPU FFTW plan
};

class clComplexFilter_impl : public filter_impl_t<clComplexFilter, gr_complex> {
public:
    clComplexFilter_impl(int p, int s, int pl, int d, int decim, const std::vector<gr_complex> &taps, bool dbg)
        : gr::sync_decimator("clComplexFilter", decim), filter_impl_t("clComplexFilter", p, s, pl, d, decim, taps, dbg, true, true) {}
};

class clPolyphaseChannelizer_impl : public clPolyphaseChannelizer, public MI355Base {
    mi355_pfb *d_h = nullptr;
    int d_buf_items;
public:
    clPolyphaseChannelizer_impl(int p, int s, int pl, int d, const std::vector<float> &taps, int buf_items, int num_channels,
                                int ninputs_per_iter, const std::vector<int> &ch_map, bool dbg)
        : gr::block("clPolyphaseChannelizer"), MI355Base(p, s, pl, d, dbg), d_buf_items(buf_items)
    {
        if (num_channels <= 0 || buf_items % num_channels != 0)  // lib/clPolyphaseChannelizer_impl.cc:59-62
            throw std::invalid_argument("buf_items must be a multiple of num_channels");
        chk(mi355_pfb_create(d_ctx, taps.data(), (int)taps.size(), buf_items, num_channels, ninputs_per_iter, ch_map.data(),
                             (int)ch_map.size(), &d_h), "mi355_pfb_create");
        set_history(taps.size());                     // :63
        set_output_multiple(mi355_pfb_noutput(d_h));  // :64
    }
    ~clPolyphaseChannelizer_impl() override { mi355_pfb_destroy(d_h); }
    void forecast(int, gr_vector_int &req) override { req[0] = mi355_pfb_ninput(d_h); }
    int general_work(int, gr_vector_int &, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        chk(mi355_pfb_work(d_h, in[0], out[0]), "mi355_pfb_work");
        // consume_each(d_buf_items) with a real scheduler (:105)
        return mi355_pfb_noutput(d_h);
    }
};

class clXEngine_impl : public clXEngine, public MI355Base {
    mi355_xengine *d_h = nullptr;
    int d_pipeline_integration;
    long d_in_items;
public:
    clXEngine_impl(int p, int s, int pl, int d, bool dbg, int data_type, int polarization, int num_inputs, int num_channels,
                   int integration, int pipeline_integration)
        : gr::block("clXEngine"), MI355Base(p, s, pl, d, dbg), d_pipeline_integration(pipeline_integration)
    {
        if (num_inputs < 2)  // lib/clXEngine_impl.cc:106-109
            throw std::out_of_range("Please specify at least 2 inputs to correlate.");
        chk(mi355_xengine_create(d_ctx, data_type, polarization, num_inputs, num_channels, integration, &d_h), "mi355_xengine_create");
        const int npol = data_type == DTYPE_PACKEDXY ? 2 : polarization;
        d_in_items = (long)num_inputs * num_channels * npol * integration;
    }
    ~clXEngine_impl() override { mi355_xengine_destroy(d_h); }
    void forecast(int n, gr_vector_int &req) override { for (auto &r : req) r = n; }
    int general_work(int, gr_vector_int &, gr_vector_const_void_star &, gr_vector_void_star &) override { return 0; }
    long get_input_buffer_size() override { return d_in_items; }
    long get_output_buffer_size() override { return (long)mi355_xengine_output_items(d_h); }
    void xcorrelate(XComplex *in, XComplex *out) override
    {
        chk(mi355_xengine_xcorrelate(d_h, in, out, d_pipeline_integration > 1), "mi355_xengine_xcorrelate");
    }
    void xcorrelate(char *in, XComplex *out) override
    {
        chk(mi355_xengine_xcorrelate(d_h, in, out, d_pipeline_integration > 1), "mi355_xengine_xcorrelate");
    }
    void submit(const void *in, const XComplex *acc) override { chk(mi355_xengine_submit(d_h, in, acc), "mi355_xengine_submit"); }
    void wait(XComplex *out) override { chk(mi355_xengine_wait(d_h, out), "mi355_xengine_wait"); }
    int gather_frames(int nframes, int frame0, gr_vector_const_void_star &in, void *fb) override
    {
        chk(mi355_xengine_gather(d_h, nframes, frame0, in.data(), fb), "mi355_xengine_gather");
        return nframes;
    }
};
}  // namespace

clMathOp::sptr clMathOp::make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, int operatorType,
                              int setDebug)
{
    return sptr(new clMathOp_impl(idataType, openCLPlatformType, devSelector, platformId, devId, operatorType, setDebug == 1));
}
clMathConst::sptr clMathConst::make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, float fValue,
                                    int operatorType, int setDebug)
{
    return sptr(new clMathConst_impl(idataType, openCLPlatformType, devSelector, platformId, devId, fValue, operatorType, setDebug == 1));
}
clFFT::sptr clFFT::make(int fftSize, int clFFTDir, const std::vector<float> &window, int idataType, int openCLPlatformType,
                        int devSelector, int platformId, int devId, int setDebug, int num_streams, bool shift)
{
    return sptr(new clFFT_impl(fftSize, clFFTDir, window, idataType, openCLPlatformType, devSelector, platformId, devId, setDebug == 1,
                               num_streams, shift));
}
clFilter::sptr clFilter::make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                              const std::vector<float> &taps, int, int setDebug, bool use_time)
{
    return sptr(new clFilter_impl(openclPlatform, devSelector, platformId, devId, decimation, taps, setDebug == 1, use_time));
}
clComplexFilter::sptr clComplexFilter::make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                                            const std::vector<gr_complex> &taps, int, int setDebug)
{
    return sptr(new clComplexFilter_impl(openclPlatform, devSelector, platformId, devId, decimation, taps, setDebug == 1));
}
clPolyphaseChannelizer::sptr clPolyphaseChannelizer::make(int openCLPlatformType, int devSelector, int platformId, int devId,
                                                          const std::vector<float> &taps, int buf_items, int num_channels,
                                                          int ninputs_per_iter, const std::vector<int> &ch_map, int setDebug)
{
    return sptr(new clPolyphaseChannelizer_impl(openCLPlatformType, devSelector, platformId, devId, taps, buf_items, num_channels,
                                                ninputs_per_iter, ch_map, setDebug == 1));
}
clXEngine::sptr clXEngine::make(int openCLPlatformType, int devSelector, int platformId, int devId, bool setDebug, int data_type,
                                int polarization, int num_inputs, int, int, int num_channels, int integration,
                                std::vector<std::string>, bool, std::string, int, bool, long, std::string, double, double, bool,
                                int pipeline_integration)
{
    return sptr(new clXEngine_impl(openCLPlatformType, devSelector, platformId, devId, setDebug, data_type, polarization, num_inputs,
                                   num_channels, integration, pipeline_integration));
}

}  // namespace clenabled
}  // namespace gr
