// _impl classes of the hot-path blocks: every work() is a call into the C ABI (mi355_clenabled.h).
// Structure follows the reference's lib/cl*_impl.cc with the GRCLBase base class replaced by
// MI355Base; constructor-time errors throw the same exception types as the reference.
#include <clenabled/clenabled.h>
#include <mi355_clenabled.h>

#include <cstdio>
#include <mutex>
#include <stdexcept>

namespace gr {
namespace clenabled {

namespace {
// The library's diagnostics (mi355_set_log_callback) go where the reference's go: GNU Radio's logger (GR_LOG_INFO / GR_LOG_ERROR,
// lib/clXEngine_impl.cc:107,137,257).  Stand-alone build: stderr, one "clenabled :level: text" line each, like the logger's format.
void log_sink(void *, int level, const char *message)
{
#ifdef MI355_WITH_GNURADIO
    // (configure_default_loggers + the GR_LOG_* macros exist in 3.9's log4cpp logger and in 3.10's spdlog one alike)
    static gr::logger_ptr logger, debug_logger;
    static const bool configured = gr::configure_default_loggers(logger, debug_logger, "clenabled");
    (void)configured;
    const std::string text(message);
    switch (level) {
    case MI355_LOG_DEBUG: GR_LOG_DEBUG(debug_logger, text); break;
    case MI355_LOG_INFO: GR_LOG_INFO(logger, text); break;
    case MI355_LOG_WARN: GR_LOG_WARN(logger, text); break;
    default: GR_LOG_ERROR(logger, text); break;
    }
#else
    static const char *const names[] = {"debug", "info", "warning", "error"};
    fprintf(stderr, "clenabled :%s: %s\n", names[level < 0 ? 0 : level > 3 ? 3 : level], message);
#endif
}

// stands where "public GRCLBase" was (include/clenabled/GRCLBase.h:77-141)
class MI355Base {
protected:
    mi355_ctx *d_ctx = nullptr;
    bool debugMode;
    MI355Base(int openCLPlatformType, int devSelector, int platformId, int devId, bool setDebug) : debugMode(setDebug)
    {
        static std::once_flag once;
        std::call_once(once, [] { mi355_set_log_callback(&log_sink, nullptr); });
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
        : gr::sync_block("clMathOp", gr::io_signature::make(2, 2, (int)dsize(idataType)), gr::io_signature::make(1, 1, (int)dsize(idataType))),  // lib/clMathOp_impl.cc:63-65
          MI355Base(openCLPlatformType, devSelector, platformId, devId, setDebug)
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
        : gr::sync_block("clMathConst", gr::io_signature::make(1, 1, (int)dsize(idataType)), gr::io_signature::make(1, 1, (int)dsize(idataType))),
          MI355Base(openCLPlatformType, devSelector, platformId, devId, setDebug)
    {
        chk(mi355_mathconst_create(d_ctx, idataType, operatorType, fValue, 8192, &d_h), "mi355_mathconst_create");
    }
    ~clMathConst_impl() override { mi355_mathconst_destroy(d_h); }
    float k() const override { float v = 0; mi355_mathconst_get_k(d_h, &v); return v; }
    void set_k(float v) override { chk(mi355_mathconst_set_k(d_h, v), "mi355_mathconst_set_k"); }
#if defined(MI355_WITH_GNURADIO) && defined(GR_CTRLPORT)
    // ControlPort: the constant as a readable and writable real "Constant" (lib/clMathConst_impl.cc:377-401)
    void setup_rpc() override
    {
        const pmt::pmt_t lo = pmt::from_double(-4.29e9), hi = pmt::from_double(4.29e9), def = pmt::from_double(0);
        add_rpc_variable(rpcbasic_sptr(new rpcbasic_register_get<clMathConst, float>(alias(), "Constant", &clMathConst::k, lo, hi, def, "", "Constant",
                                                                                      RPC_PRIVLVL_MIN, DISPTIME | DISPOPTSTRIP)));
        add_rpc_variable(rpcbasic_sptr(new rpcbasic_register_set<clMathConst, float>(alias(), "Constant", &clMathConst::set_k, lo, hi, def, "", "Constant",
                                                                                      RPC_PRIVLVL_MIN, DISPNULL)));
    }
#endif
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
        : gr::sync_block("clFFT", gr::io_signature::make(1, num_streams, fftSize * (int)dsize(idataType)), gr::io_signature::make(1, num_streams, fftSize * (int)sizeof(gr_complex))),  // lib/clFFT_impl.cc:68-70
          MI355Base(openCLPlatformType, devSelector, platformId, devId, setDebug), d_fft_size(fftSize)
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
        : gr::sync_decimator(name, gr::io_signature::make(1, 1, (int)sizeof(gr_complex)), gr::io_signature::make(1, 1, (int)sizeof(gr_complex)), decimation),
          MI355Base(openclPlatform, devSelector, platformId, devId, setDebug)
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
        : gr::sync_decimator("clFilter", gr::io_signature::make(1, 1, (int)sizeof(gr_complex)), gr::io_signature::make(1, 1, (int)sizeof(gr_complex)), decim),
          filter_impl_t("clFilter", p, s, pl, d, decim, taps, dbg, false, use_time) {}
    void set_nthreads(int) override {}  // lib/clFilter_impl.cc:413-415 only configured the CPU FFTW plan
};

class clComplexFilter_impl : public filter_impl_t<clComplexFilter, gr_complex> {
public:
    clComplexFilter_impl(int p, int s, int pl, int d, int decim, const std::vector<gr_complex> &taps, bool dbg)
        : gr::sync_decimator("clComplexFilter", gr::io_signature::make(1, 1, (int)sizeof(gr_complex)), gr::io_signature::make(1, 1, (int)sizeof(gr_complex)), decim),
          filter_impl_t("clComplexFilter", p, s, pl, d, decim, taps, dbg, true, true) {}
};

class clPolyphaseChannelizer_impl : public clPolyphaseChannelizer, public MI355Base {
    mi355_pfb *d_h = nullptr;
    int d_buf_items;
public:
    clPolyphaseChannelizer_impl(int p, int s, int pl, int d, const std::vector<float> &taps, int buf_items, int num_channels,
                                int ninputs_per_iter, const std::vector<int> &ch_map, bool dbg)
        : gr::block("clPolyphaseChannelizer", gr::io_signature::make(1, 1, (int)sizeof(gr_complex)), gr::io_signature::make(1, 1, (int)sizeof(gr_complex))),  // lib/clPolyphaseChannelizer_impl.cc:50-51
          MI355Base(p, s, pl, d, dbg), d_buf_items(buf_items)
    {
        if (num_channels <= 0 || buf_items % num_channels != 0)  // lib/clPolyphaseChannelizer_impl.cc:59-62
            throw std::invalid_argument("buf_items must be a multiple of num_channels");
        chk(mi355_pfb_create(d_ctx, taps.data(), (int)taps.size(), buf_items, num_channels, ninputs_per_iter, ch_map.data(),
                             (int)ch_map.size(), &d_h), "mi355_pfb_create");
        set_history(taps.size());                     // :63
        set_output_multiple(mi355_pfb_noutput(d_h));  // :64
    }
    ~clPolyphaseChannelizer_impl() override { mi355_pfb_destroy(d_h); }
    // ninput_items_required = ninputs_per_iter * noutput_items / nmap + history() - num_channels (:78-82); for one output multiple
    // that is mi355_pfb_ninput(); every further multiple needs buf_items more
    void forecast(int noutput_items, gr_vector_int &req) override
    {
        const int per = mi355_pfb_noutput(d_h), k = noutput_items > per ? noutput_items / per : 1;
        req[0] = mi355_pfb_ninput(d_h) + (k - 1) * d_buf_items;
    }
    // The reference handles exactly one buffer per call (:83-109).  The scheduler calls with a multiple of the output multiple;
    // all whole buffers it offers are processed (same samples, fewer launches), consume_each(k * buf_items), k * noutput() returned.
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        const int per = mi355_pfb_noutput(d_h), k = noutput_items > per ? noutput_items / per : 1;
        for (int b = 0; b < k; b++)
            chk(mi355_pfb_work(d_h, (const gr_complex *)in[0] + (size_t)b * d_buf_items, (gr_complex *)out[0] + (size_t)b * per), "mi355_pfb_work");
        consume_each(k * d_buf_items);  // :105
        return k * per;
    }
};

class clXEngine_impl : public clXEngine, public MI355Base {
    mi355_xengine *d_h = nullptr;
    // several devices behind ONE block (set_shard_devices / MI355_XENGINE_DEVICES): antenna groups in, channel slabs out, the corner turn
    // between the devices inside mi355_xengine_shard_* -- the reference has one device per block (devId, lib/GRCLBase.cpp:115-134)
    bool d_sharded = false;                         // set_shard_devices() with two or more devices
    mi355_xengine_shard *d_shard = nullptr;         // one window per call: xcorrelate(char*, XComplex*), pipeline integration -- created at first use
    // the STREAMING path of work_test(): d_shard_windows integration windows per exchange, gathered straight into the handle's pinned frame slots
    // (mi355_xengine_shard_acquire / submit_acquired / wait: every device uploads its antenna group over its own link, asynchronously, while the next
    // windows are gathered) -- created at the first streamed window
    mi355_xengine_shard *d_shard_stream = nullptr;
    std::vector<int> d_shard_ids;
    int d_shard_windows = 4, d_batch_fill = 0;
    char *d_batch_base = nullptr;
    std::vector<long> d_batch_first;                // first frame numbers of the windows gathered into the current batch
    std::vector<std::vector<long>> d_shard_pending_first;  // ... of the exchanges in flight
    std::vector<XComplex> d_result_batch;
    int d_data_type;
    int d_npol, d_num_inputs, d_num_channels, d_integration, d_pipeline_integration, d_first_channel;
    long d_in_items;
    size_t d_matrix_len, d_in_bytes;
    // integration window being filled: the pinned frame buffer of the next free slot (the reference's pinned
    // char_input/complex_input, :325-362), acquired when a window starts -- no second host copy at submit time
    char *d_frames = nullptr;
    std::vector<char> d_frames_sync;  // pipeline-integration mode uses the synchronous call and a plain buffer
    std::vector<XComplex> d_result, d_accum;
    int d_tracker = 0, d_pipe_count = 0;
    long d_delivered = 0, d_frame_counter = 0;
    result_handler_t d_handler = nullptr;
    void *d_handler_user = nullptr;
    bool d_disable_output;
    // file sink (lib/clXEngine_impl.cc:393-465)
    bool d_output_file, d_rollover = false, d_wrote_json = false;
    std::string d_file_base, d_filename, d_object_name, d_antenna_json;
    size_t d_rollover_bytes = 0, d_bytes_written = 0;
    int d_rollover_index = 1;
    long d_sync_timestamp;
    double d_start_freq, d_chan_width;
    FILE *d_fp = nullptr;
    std::mutex d_lock;
    bool d_use_synchronizer = false, d_synchronized = false, d_publish = true;
    uint64_t d_sync_tag = 0;
    std::vector<uint64_t> d_tag_list;

    bool open_file()
    {
        d_wrote_json = false;
        d_filename = d_file_base;
        if (d_rollover) {  // "_001", "_002", ... (:406-413)
            char idx[16];
            snprintf(idx, sizeof idx, "_%03d", d_rollover_index++);
            d_filename += idx;
        }
        if (d_fp) fclose(d_fp);
        d_fp = fopen(d_filename.c_str(), "wb");
        d_bytes_written = 0;
        return d_fp != nullptr;
    }
    void write_json(long seq_num)
    {
        // same keys, order and formats as lib/clXEngine_impl.cc:438-465
        FILE *f = fopen((d_filename + ".json").c_str(), "w");
        if (!f) return;
        const long integ_frames = d_pipeline_integration < 2 ? (long)d_integration : (long)d_integration * d_pipeline_integration;
        const int nb = (d_num_inputs + 1) * d_num_inputs / 2;
        fprintf(f,
                "{\n\"sync_timestamp\":%ld,\n\"first_seq_num\":%ld,\n\"object_name\":\"%s\",\n\"num_baselines\":%d,\n"
                "\"first_channel\":%d,\n\"first_channel_center_freq\":%f,\n\"channels\":%d,\n\"channel_width\":%f,\n"
                "\"polarizations\":%d,\n\"antennas\":%d,\n\"antenna_names\":%s,\n\"ntime\":%ld,\n\"samples_per_block\":%ld,\n"
                "\"bytes_per_block\":%ld,\n\"data_type\":\"cf32_le\",\n\"data_format\": \"triangular order\"\n}\n",
                d_sync_timestamp, seq_num, d_object_name.c_str(), nb, d_first_channel, d_start_freq, d_num_channels, d_chan_width, d_npol,
                d_num_inputs, d_antenna_json.c_str(), integ_frames, (long)d_matrix_len, (long)(d_matrix_len * sizeof(XComplex)));
        fclose(f);
        d_wrote_json = true;
    }
    void deliver(const XComplex *m, long first_frame)
    {
        d_delivered++;
        if (d_disable_output) return;
        if (d_fp) {
            if (d_rollover && d_bytes_written >= d_rollover_bytes) open_file();  // :1264-1267
            if (!d_wrote_json) write_json(first_frame);
            if (fwrite(m, d_matrix_len * sizeof(XComplex), 1, d_fp) == 1) d_bytes_written += d_matrix_len * sizeof(XComplex);
        } else {
            // message_port_pub("xcorr", cons("triang_matrix", c32vector)) -- :1070-1094; a C callback may be set instead / as well
            if (d_handler) d_handler(d_handler_user, m, d_matrix_len);
            if (d_publish) sched::publish_c32vector(this, "xcorr", "triang_matrix", (const gr_complex *)m, d_matrix_len);
        }
    }
    void collect_shard()
    {
        chk(mi355_xengine_shard_wait(d_shard_stream, d_result_batch.data()), "mi355_xengine_shard_wait");
        const std::vector<long> firsts = d_shard_pending_first.front();
        d_shard_pending_first.erase(d_shard_pending_first.begin());
        for (size_t w = 0; w < firsts.size(); w++) deliver(d_result_batch.data() + w * d_matrix_len, firsts[w]);
    }
    bool shard_streaming() const { return d_sharded && d_pipeline_integration <= 1; }
    mi355_xengine_shard *sync_shard()  // the one-window-per-call handle (the streaming path has its own, with windows_per_exchange windows)
    {
        if (!d_shard)
            chk(mi355_xengine_shard_create((int)d_shard_ids.size(), d_shard_ids.data(), d_npol, d_num_inputs, d_num_channels, d_integration, 1, &d_shard),
                "mi355_xengine_shard_create");
        return d_shard;
    }
    void collect_one()
    {
        chk(mi355_xengine_wait(d_h, d_result.data()), "mi355_xengine_wait");
        deliver(d_result.data(), d_pending_first.front());
        d_pending_first.erase(d_pending_first.begin());
    }
    std::vector<long> d_pending_first;  // first frame number of each integration in flight

public:
    static int item_bytes(int data_type) { return data_type == DTYPE_COMPLEX ? (int)sizeof(gr_complex) : data_type == DTYPE_BYTE ? 2 : 1; }
    clXEngine_impl(int p, int s, int pl, int d, bool dbg, int data_type, int polarization, int num_inputs, int first_channel,
                   int num_channels, int integration, const std::vector<std::string> &antennas, bool output_file,
                   const std::string &file_base, int rollover_size_mb, bool internal_synchronizer, long sync_timestamp,
                   const std::string &object_name, double start_freq, double chan_width, bool disable_output, int pipeline_integration)
        : gr::block("clXEngine",  // lib/clXEngine_impl.cc:88-90: up to num_inputs * polarization streams of one channel row each, no stream output
                    gr::io_signature::make(2, num_inputs * (data_type == DTYPE_PACKEDXY ? 1 : polarization),
                                           num_channels * (data_type == DTYPE_PACKEDXY ? 2 : item_bytes(data_type))),
                    gr::io_signature::make(0, 0, 0)),
          MI355Base(p, s, pl, d, dbg), d_data_type(data_type), d_npol(data_type == DTYPE_PACKEDXY ? 2 : polarization),
          d_num_inputs(num_inputs), d_num_channels(num_channels), d_integration(integration),
          d_pipeline_integration(pipeline_integration), d_first_channel(first_channel), d_disable_output(disable_output),
          d_output_file(output_file && !disable_output), d_file_base(file_base), d_object_name(object_name),
          d_sync_timestamp(sync_timestamp), d_start_freq(start_freq), d_chan_width(chan_width)
    {
        if (num_inputs < 2)  // lib/clXEngine_impl.cc:106-109
            throw std::out_of_range("Please specify at least 2 inputs to correlate.");
        chk(mi355_xengine_create(d_ctx, data_type, polarization, num_inputs, num_channels, integration, &d_h), "mi355_xengine_create");
        d_in_items = (long)num_inputs * num_channels * d_npol * integration;
        d_in_bytes = mi355_xengine_input_bytes(d_h);
        d_matrix_len = mi355_xengine_output_items(d_h);
        if (d_pipeline_integration > 1) d_frames_sync.resize(d_in_bytes);
        d_result.resize(d_matrix_len);
        if (d_pipeline_integration > 1) d_accum.assign(d_matrix_len, XComplex());
        d_antenna_json = "[";  // :141-165
        if (antennas.size() > 1)
            for (size_t i = 0; i < antennas.size(); i++) d_antenna_json += "\"" + antennas[i] + "\"" + (i + 1 < antennas.size() ? "," : "");
        d_antenna_json += "]";
        if (d_output_file) {
            if (rollover_size_mb > 0) { d_rollover = true; d_rollover_bytes = (size_t)rollover_size_mb * 1000000; }  // :121-125
            if (!open_file()) throw std::runtime_error("[X-Engine] can't open file: " + d_filename);                 // :130-137
        }
        sched::register_out(this, "xcorr");  // :294-295
        sched::register_out(this, "sync");
        d_use_synchronizer = internal_synchronizer;
        d_tag_list.assign((size_t)num_inputs, 0);
        if (d_use_synchronizer) {  // :297-301: the SNAP boards send 16-time-step packets
            sched::no_tag_propagation(this);
            set_output_multiple(16);
        }
        if (const char *e = getenv("MI355_XENGINE_DEVICES")) {  // "0,1,2,3": an unmodified flowgraph's block over several devices
            std::vector<int> ids;
            for (const char *q = e; *q;) {
                char *end = nullptr;
                const long v = strtol(q, &end, 10);
                if (end == q) break;
                ids.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
            }
            if (ids.size() > 1) {
                // the variable is process-wide ("flowgraphs that are not edited"): a block it does not fit -- complex input, more than 64 rows, counts the
                // ranks do not divide -- keeps its one device and says so; only the explicit call throws
                try { set_shard_devices(ids); }
                catch (const std::exception &ex) {
                    mi355_xengine_shard_destroy(d_shard);
                    mi355_xengine_shard_destroy(d_shard_stream);
                    d_shard = d_shard_stream = nullptr;
                    d_sharded = false;
                    log_sink(nullptr, MI355_LOG_WARN, (std::string("MI355_XENGINE_DEVICES ignored for this clXEngine block: ") + ex.what()).c_str());
                }
            }
        }
    }
    ~clXEngine_impl() override
    {
        try { stop(); } catch (...) {}
        mi355_xengine_shard_destroy(d_shard_stream);
        mi355_xengine_shard_destroy(d_shard);
        mi355_xengine_destroy(d_h);
    }
    void set_shard_devices(const std::vector<int> &device_ids, int windows_per_exchange = 4) override
    {
        std::lock_guard<std::mutex> g(d_lock);
        if (d_tracker != 0 || mi355_xengine_pending(d_h) > 0) throw std::runtime_error("[X-Engine] set_shard_devices: an integration is in progress");
        if (d_data_type != DTYPE_BYTE) throw std::invalid_argument("[X-Engine] several devices: IChar (byte) input only");
        if (d_batch_fill != 0 || (d_shard_stream && mi355_xengine_shard_pending(d_shard_stream) > 0))
            throw std::runtime_error("[X-Engine] set_shard_devices: an exchange is in progress (stop() first)");
        if (windows_per_exchange < 1) throw std::invalid_argument("[X-Engine] set_shard_devices: windows_per_exchange must be >= 1");
        mi355_xengine_shard_destroy(d_shard_stream);
        d_shard_stream = nullptr;
        mi355_xengine_shard_destroy(d_shard);
        d_shard = nullptr;
        d_shard_ids = device_ids;
        d_shard_windows = windows_per_exchange;
        if (const char *e = getenv("MI355_XENGINE_SHARD_WINDOWS")) d_shard_windows = atoi(e) > 0 ? atoi(e) : d_shard_windows;
        d_sharded = false;
        if (device_ids.size() < 2) return;  // back to the one device of make()
        // (the streaming handle is created here: it validates devices and geometry -- 64 inputs x 2 polarisations need enough windows per exchange)
        chk(mi355_xengine_shard_create((int)device_ids.size(), device_ids.data(), d_npol, d_num_inputs, d_num_channels, d_integration, d_shard_windows,
                                       &d_shard_stream), "mi355_xengine_shard_create");
        d_result_batch.resize((size_t)d_shard_windows * d_matrix_len);
        d_sharded = true;
        d_frames_sync.resize(d_in_bytes);
        if (d_accum.size() != d_matrix_len) d_accum.assign(d_matrix_len, XComplex());
    }
    int shard_devices() const override { return d_sharded ? (int)d_shard_ids.size() : 1; }
    bool stop() override
    {
        std::lock_guard<std::mutex> g(d_lock);
        while (mi355_xengine_pending(d_h) > 0) collect_one();
        if (d_shard_stream) {
            while (mi355_xengine_shard_pending(d_shard_stream) > 0) collect_shard();
            // whole windows of a batch that did not fill up: through the block's one device, from the pinned buffer they were gathered into
            for (int w = 0; w < d_batch_fill; w++) {
                chk(mi355_xengine_xcorrelate(d_h, d_batch_base + (size_t)w * d_in_bytes, d_result.data(), 0), "mi355_xengine_xcorrelate");
                deliver(d_result.data(), d_batch_first[(size_t)w]);
            }
            d_batch_fill = 0;
            d_batch_first.clear();
        }
        if (d_fp) { fclose(d_fp); d_fp = nullptr; }
        return true;
    }
    void forecast(int n, gr_vector_int &req) override { for (auto &r : req) r = n; }
    // lib/clXEngine_impl.cc:1152-1232.  With the tag synchroniser on, nothing is correlated until the first tag of every input
    // carries the same timestamp: inputs that are behind are advanced by (highest - own) items (timestamps step with the items),
    // capped at noutput_items, and 0 is returned; once aligned, the timestamp is published on "sync" and normal work begins.
    // first tag of an input in the current window: get_tags_in_window(tags, input, 0, 1), lib/clXEngine_impl.cc:1173-1175
    // (a protected member of gr::block, hence a member here)
    bool first_tag(int input, uint64_t &value)
    {
#ifdef MI355_WITH_GNURADIO
        std::vector<gr::tag_t> tags;
        this->get_tags_in_window(tags, (unsigned)input, 0, 1);
        if (tags.empty()) return false;
        value = pmt::to_uint64(tags[0].value);
        return true;
#else
        return shim_first_tag(input, value);
#endif
    }
    int general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        if (d_use_synchronizer && !d_synchronized) {
            uint64_t highest = 0, first = 0;
            bool aligned = true;
            for (int i = 0; i < d_num_inputs; i++) {
                uint64_t t = 0;
                if (!first_tag(i, t)) {  // no tag in the window yet: cannot decide, consume nothing
                    return 0;
                }
                if (i == 0) first = t;
                else if (t != first) aligned = false;
                d_tag_list[i] = t;
                if (t > highest) highest = t;
            }
            if (!aligned) {
                for (int i = 0; i < d_num_inputs; i++) {
                    uint64_t n = highest - d_tag_list[i];
                    if (n > (uint64_t)noutput_items) n = (uint64_t)noutput_items;
                    consume(i, (int)n);
                }
                return 0;
            }
            d_synchronized = true;
            d_sync_tag = highest;
            if (d_fp && !d_wrote_json) write_json((long)highest);
            sched::publish_u64(this, "sync", "synctimestamp", highest);  // :1203-1204
            log_sink(nullptr, MI355_LOG_INFO, ("Synchronized on timestamp " + std::to_string(highest)).c_str());  // :1206-1208
        }
        const int done = work_test(noutput_items, in, out);
        consume_each(done);  // :1230
        return done;
    }
    bool synchronized() const override { return d_synchronized; }
    uint64_t sync_tag() const override { return d_sync_tag; }
    int work_test(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &) override
    {
        std::lock_guard<std::mutex> g(d_lock);
        const int remaining = d_integration - d_tracker;
        const int n = noutput_items > remaining ? remaining : noutput_items;  // :925-934
        if (d_tracker == 0) {  // a new integration window starts: get the buffer it is gathered into
            if (shard_streaming()) {
                if (d_batch_fill == 0) {  // ... and a new exchange: the next pinned frame slot of the sharded handle
                    if (!d_shard_stream) {
                        chk(mi355_xengine_shard_create((int)d_shard_ids.size(), d_shard_ids.data(), d_npol, d_num_inputs, d_num_channels, d_integration,
                                                       d_shard_windows, &d_shard_stream), "mi355_xengine_shard_create");
                        d_result_batch.resize((size_t)d_shard_windows * d_matrix_len);
                    }
                    if (mi355_xengine_shard_pending(d_shard_stream) == 2) collect_shard();
                    void *fb = nullptr;
                    chk(mi355_xengine_shard_acquire(d_shard_stream, &fb), "mi355_xengine_shard_acquire");
                    d_batch_base = (char *)fb;
                    d_batch_first.clear();
                }
                d_frames = d_batch_base + (size_t)d_batch_fill * d_in_bytes;
            } else if (d_pipeline_integration > 1 || d_sharded) d_frames = d_frames_sync.data();
            else {
                if (mi355_xengine_pending(d_h) == 2) collect_one();  // previous result goes out before the swap (:1070-1094)
                void *fb = nullptr;
                chk(mi355_xengine_acquire(d_h, &fb), "mi355_xengine_acquire");
                d_frames = (char *)fb;
            }
        }
        chk(mi355_xengine_gather(d_h, n, d_tracker, in.data(), d_frames), "mi355_xengine_gather");
        d_tracker += n;
        d_frame_counter += n;
        if (d_tracker == d_integration) {
            const long first = d_frame_counter - d_integration;
            if (shard_streaming()) {
                // every device takes its antenna group over its own link out of the pinned slot (asynchronous uploads on the ranks' streams), the
                // devices turn the corner and correlate their channel slabs, d_shard_windows windows per exchange; one exchange stays in flight
                // under the gather of the next (the reference overlaps with a worker thread, lib/clXEngine_impl.cc:1234-1299)
                d_batch_first.push_back(first);
                if (++d_batch_fill == d_shard_windows) {
                    chk(mi355_xengine_shard_submit_acquired(d_shard_stream), "mi355_xengine_shard_submit_acquired");
                    d_shard_pending_first.push_back(d_batch_first);
                    d_batch_fill = 0;
                    if (mi355_xengine_shard_pending(d_shard_stream) == 2) collect_shard();
                }
            } else if (d_pipeline_integration > 1) {
                // device "+=" into the running matrix, read back every pipeline_integration windows (:785-796,1250-1285)
                if (d_sharded) chk(mi355_xengine_shard_xcorrelate(sync_shard(), d_frames, d_accum.data(), 1), "mi355_xengine_shard_xcorrelate");
                else
                chk(mi355_xengine_xcorrelate(d_h, d_frames, d_accum.data(), 1), "mi355_xengine_xcorrelate");
                if (++d_pipe_count >= d_pipeline_integration) {
                    deliver(d_accum.data(), first - (long)d_integration * (d_pipeline_integration - 1));
                    d_accum.assign(d_matrix_len, XComplex());
                    d_pipe_count = 0;
                }
            } else {
                chk(mi355_xengine_submit_acquired(d_h, nullptr), "mi355_xengine_submit_acquired");
                d_pending_first.push_back(first);
                if (mi355_xengine_pending(d_h) == 2) collect_one();  // keep one in flight: overlap with the next window's gather
            }
            d_tracker = 0;
        }
        return n;
    }
    void set_result_handler(result_handler_t fn, void *user) override { d_handler = fn; d_handler_user = user; }
    long integrations_delivered() const override { return d_delivered; }
    long get_input_buffer_size() override { return d_in_items; }
    long get_output_buffer_size() override { return (long)d_matrix_len; }
    void xcorrelate(XComplex *in, XComplex *out) override
    {
        chk(mi355_xengine_xcorrelate(d_h, in, out, d_pipeline_integration > 1), "mi355_xengine_xcorrelate");
    }
    void xcorrelate(char *in, XComplex *out) override
    {
        if (d_sharded) chk(mi355_xengine_shard_xcorrelate(sync_shard(), in, out, d_pipeline_integration > 1), "mi355_xengine_shard_xcorrelate");
        else
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

// remaining elementwise family: one implementation over mi355_elem_* (kind selects the block)
template <class Base>
class elem_impl_t : public Base, public MI355Base {
    mi355_elem *d_h = nullptr;
    int d_nin, d_nout;
public:
    // item sizes as in the reference's constructors (e.g. lib/clComplexToMagPhase_impl.cc:49-50: complex in, two floats out)
    static int in_size(int kind)
    {
        return (kind == MI355_ELEM_LOG10 || kind == MI355_ELEM_SNR || kind == MI355_ELEM_MAGPHASE2C) ? (int)sizeof(float) : (int)sizeof(gr_complex);
    }
    static int out_size(int kind) { return kind == MI355_ELEM_MAGPHASE2C ? (int)sizeof(gr_complex) : (int)sizeof(float); }
    elem_impl_t(const char *name, int kind, int nin, int nout, float p0, float p1, int openCLPlatformType, int devSelector,
                int platformId, int devId, bool setDebug)
        : gr::sync_block(name, gr::io_signature::make(nin, nin, in_size(kind)), gr::io_signature::make(nout, nout, out_size(kind))),
          MI355Base(openCLPlatformType, devSelector, platformId, devId, setDebug), d_nin(nin), d_nout(nout)
    {
        chk(mi355_elem_create(d_ctx, kind, p0, p1, &d_h), "mi355_elem_create");
        this->set_history((unsigned)mi355_elem_history(d_h));  // clQuadratureDemod: 2 (lib/clQuadratureDemod_impl.cc:81)
    }
    ~elem_impl_t() override { mi355_elem_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        chk(mi355_elem_work(d_h, (size_t)noutput_items, in[0], d_nin > 1 ? in[1] : nullptr, out[0], d_nout > 1 ? out[1] : nullptr),
            "mi355_elem_work");
        return noutput_items;
    }
    int testOpenCL(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        return work(noutput_items, in, out);
    }
};

class clxcorrelate_fft_vcf_impl : public clxcorrelate_fft_vcf, public MI355Base {
    mi355_xcorr_fft *d_h = nullptr;
public:
    clxcorrelate_fft_vcf_impl(int fftSize, int num_inputs, int openCLPlatformType, int devSelector, int platformId, int devId,
                              int input_type)
        : gr::sync_block("clxcorrelate_fft_vcf", gr::io_signature::make(2, num_inputs, (int)sizeof(gr_complex) * fftSize),
                         gr::io_signature::make(1, num_inputs - 1, (int)sizeof(float) * fftSize)),  // lib/clxcorrelate_fft_vcf_impl.cc:701-702
          MI355Base(openCLPlatformType, devSelector, platformId, devId, false)
    {
        chk(mi355_xcorr_fft_create(d_ctx, fftSize, num_inputs, input_type, &d_h), "mi355_xcorr_fft_create");
    }
    ~clxcorrelate_fft_vcf_impl() override { mi355_xcorr_fft_destroy(d_h); }
    int work(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        chk(mi355_xcorr_fft_work(d_h, noutput_items, in.data(), out.data()), "mi355_xcorr_fft_work");  // vectors (:1058-1143)
        return noutput_items;
    }
    int work_test(int noutput_items, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        return work(noutput_items, in, out);  // lib/clxcorrelate_fft_vcf_impl.cc:982-1056 is the same data path
    }
};
}  // namespace

#define MI355_ELEM_MAKE(NAME, KIND, NIN, NOUT, P0, P1, ...)                                                              \
    {                                                                                                                     \
        return sched::adopt(new elem_impl_t<NAME>(#NAME, KIND, NIN, NOUT, P0, P1, openCLPlatformType, devSelector, platformId, devId, \
                                          setDebug == 1));                                                                \
    }
clLog::sptr clLog::make(int openCLPlatformType, int devSelector, int platformId, int devId, float nValue, float kValue, int setDebug)
MI355_ELEM_MAKE(clLog, MI355_ELEM_LOG10, 1, 1, nValue, kValue)
clSNR::sptr clSNR::make(int openCLPlatformType, int devSelector, int platformId, int devId, float nValue, float kValue, int setDebug)
MI355_ELEM_MAKE(clSNR, MI355_ELEM_SNR, 2, 1, nValue, kValue)
clComplexToMag::sptr clComplexToMag::make(int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug)
MI355_ELEM_MAKE(clComplexToMag, MI355_ELEM_C2MAG, 1, 1, 0.f, 0.f)
clComplexToArg::sptr clComplexToArg::make(int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug)
MI355_ELEM_MAKE(clComplexToArg, MI355_ELEM_C2ARG, 1, 1, 0.f, 0.f)
clComplexToMagPhase::sptr clComplexToMagPhase::make(int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug)
MI355_ELEM_MAKE(clComplexToMagPhase, MI355_ELEM_C2MAGPHASE, 1, 2, 0.f, 0.f)
clMagPhaseToComplex::sptr clMagPhaseToComplex::make(int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug)
MI355_ELEM_MAKE(clMagPhaseToComplex, MI355_ELEM_MAGPHASE2C, 2, 1, 0.f, 0.f)
clQuadratureDemod::sptr clQuadratureDemod::make(float gain, int openCLPlatformType, int devSelector, int platformId, int devId,
                                                int setDebug)
MI355_ELEM_MAKE(clQuadratureDemod, MI355_ELEM_QUADDEMOD, 1, 1, gain, 0.f)
#undef MI355_ELEM_MAKE

clxcorrelate_fft_vcf::sptr clxcorrelate_fft_vcf::make(int fftSize, int num_inputs, int openCLPlatformType, int devSelector,
                                                      int platformId, int devId, int input_type)
{
    return sched::adopt(new clxcorrelate_fft_vcf_impl(fftSize, num_inputs, openCLPlatformType, devSelector, platformId, devId, input_type));
}

clMathOp::sptr clMathOp::make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, int operatorType,
                              int setDebug)
{
    return sched::adopt(new clMathOp_impl(idataType, openCLPlatformType, devSelector, platformId, devId, operatorType, setDebug == 1));
}
clMathConst::sptr clMathConst::make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, float fValue,
                                    int operatorType, int setDebug)
{
    return sched::adopt(new clMathConst_impl(idataType, openCLPlatformType, devSelector, platformId, devId, fValue, operatorType, setDebug == 1));
}
clFFT::sptr clFFT::make(int fftSize, int clFFTDir, const std::vector<float> &window, int idataType, int openCLPlatformType,
                        int devSelector, int platformId, int devId, int setDebug, int num_streams, bool shift)
{
    return sched::adopt(new clFFT_impl(fftSize, clFFTDir, window, idataType, openCLPlatformType, devSelector, platformId, devId, setDebug == 1,
                               num_streams, shift));
}
clFilter::sptr clFilter::make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                              const std::vector<float> &taps, int, int setDebug, bool use_time)
{
    return sched::adopt(new clFilter_impl(openclPlatform, devSelector, platformId, devId, decimation, taps, setDebug == 1, use_time));
}
clComplexFilter::sptr clComplexFilter::make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                                            const std::vector<gr_complex> &taps, int, int setDebug)
{
    return sched::adopt(new clComplexFilter_impl(openclPlatform, devSelector, platformId, devId, decimation, taps, setDebug == 1));
}
clPolyphaseChannelizer::sptr clPolyphaseChannelizer::make(int openCLPlatformType, int devSelector, int platformId, int devId,
                                                          const std::vector<float> &taps, int buf_items, int num_channels,
                                                          int ninputs_per_iter, const std::vector<int> &ch_map, int setDebug)
{
    return sched::adopt(new clPolyphaseChannelizer_impl(openCLPlatformType, devSelector, platformId, devId, taps, buf_items, num_channels,
                                                ninputs_per_iter, ch_map, setDebug == 1));
}
clXEngine::sptr clXEngine::make(int openCLPlatformType, int devSelector, int platformId, int devId, bool setDebug, int data_type,
                                int polarization, int num_inputs, int /*output_format: the kernel always writes triangular
                                order, lib/clXEngine_impl.cc:208-211*/, int first_channel, int num_channels, int integration,
                                std::vector<std::string> antenna_list, bool output_file, std::string file_base, int rollover_size_mb,
                                bool internal_synchronizer, long sync_timestamp,
                                std::string object_name, double starting_chan_center_freq, double channel_width, bool disable_output,
                                int pipeline_integration)
{
    return sched::adopt(new clXEngine_impl(openCLPlatformType, devSelector, platformId, devId, setDebug, data_type, polarization, num_inputs,
                                   first_channel, num_channels, integration, antenna_list, output_file, file_base, rollover_size_mb,
                                   internal_synchronizer, sync_timestamp, object_name, starting_chan_center_freq, channel_width, disable_output,
                                   pipeline_integration));
}

}  // namespace clenabled
}  // namespace gr
