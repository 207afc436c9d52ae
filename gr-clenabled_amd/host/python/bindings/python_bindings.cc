// pybind11 module `clenabled_python`: the gr::clenabled block classes of the MI355X build.
//
// Takes the place of the reference's python/bindings/python_bindings.cc:57-95 + the per-block *_python.cc files: every block is
// constructed through its static make() (py::init(&X::make)), positional order exactly the make() order, so the `make:`
// templates of grc/clenabled_*.block.yml construct the same objects.  Keyword names follow the C++ parameter names of
// clenabled.h -- for clFFT that is the .cc order of the reference (its generated binding carries the header's mislabelled
// names; GRC passes positionally).
// With GNU Radio (MI355_WITH_GNURADIO, set by CMake when find_package(Gnuradio) succeeds) the classes derive from GNU
// Radio's bound gr::sync_block / gr::block / gr::basic_block, i.e. they connect in flowgraphs; without it the same source
// builds a stand-alone module (no flowgraph, work()/general_work() callable on numpy buffers) used by tests/test_pybind.py.
#include <pybind11/complex.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <clenabled/clenabled.h>

#include <string>
#include <type_traits>

namespace py = pybind11;
using namespace gr::clenabled;

#ifdef MI355_WITH_GNURADIO
#define SYNC_BASES , gr::sync_block, gr::block, gr::basic_block
#define DECIM_BASES , gr::sync_decimator, gr::sync_block, gr::block, gr::basic_block
#define BLOCK_BASES , gr::block, gr::basic_block
#else
#define SYNC_BASES
#define DECIM_BASES
#define BLOCK_BASES
#endif

namespace {
// numpy buffers -> the pointer vectors work() takes (stand-alone use and tests; inside GNU Radio the scheduler calls work()).
// The scheduler guarantees buffer sizes; a Python caller does not, so every buffer is checked against the block's own io
// signature, history and decimation before a pointer reaches work(): C-contiguous, and at least as many bytes as the call reads
// or writes.
void need(bool ok, const std::string &what)
{
    if (!ok) throw py::value_error(what);
}
template <class B> size_t item_size(const B &b, bool input, size_t k)
{
    auto sig = input ? b.input_signature() : b.output_signature();
    return (size_t)sig->sizeof_stream_item((int)k);
}
gr_vector_const_void_star in_ptrs(const std::vector<py::array> &a)
{
    gr_vector_const_void_star v;
    for (auto &x : a) {
        need((x.flags() & py::array::c_style) != 0, "input buffers must be C-contiguous numpy arrays");
        v.push_back(x.data());
    }
    return v;
}
gr_vector_void_star out_ptrs(std::vector<py::array> &a)
{
    gr_vector_void_star v;
    for (auto &x : a) {
        need((x.flags() & py::array::c_style) != 0 && x.writeable(), "output buffers must be writable C-contiguous numpy arrays");
        v.push_back(x.mutable_data());
    }
    return v;
}
template <class B> unsigned decimation_of(const B &b)
{
    if constexpr (std::is_base_of<gr::sync_decimator, B>::value) return b.decimation();
    else return 1;
}
template <class B> int call_work(B &b, int noutput_items, const std::vector<py::array> &in, std::vector<py::array> out)
{
    need(noutput_items >= 0, "noutput_items is negative");
    auto i = in_ptrs(in);
    auto o = out_ptrs(out);
    const size_t items_in = (size_t)noutput_items * decimation_of(b) + (b.history() > 0 ? b.history() - 1 : 0);
    for (size_t k = 0; k < in.size(); k++)
        need((size_t)in[k].nbytes() >= items_in * item_size(b, true, k),
             "input " + std::to_string(k) + " holds fewer than noutput_items * decimation + history - 1 items");
    for (size_t k = 0; k < out.size(); k++)
        need((size_t)out[k].nbytes() >= (size_t)noutput_items * item_size(b, false, k),
             "output " + std::to_string(k) + " holds fewer than noutput_items items");
    return b.work(noutput_items, i, o);
}
template <class B> int call_general_work(B &b, int noutput_items, const std::vector<py::array> &in, std::vector<py::array> out)
{
    need(noutput_items >= 0, "noutput_items is negative");
    auto i = in_ptrs(in);
    auto o = out_ptrs(out);
    gr_vector_int n(in.size(), 0), req(in.size(), 0);
    for (size_t k = 0; k < in.size(); k++) {  // items, not array elements: one item = sizeof_stream_item bytes
        const size_t isz = item_size(b, true, k);
        n[k] = isz ? (int)((size_t)in[k].nbytes() / isz) : 0;
    }
    b.forecast(noutput_items, req);
    for (size_t k = 0; k < in.size(); k++)
        need(n[k] >= req[k], "input " + std::to_string(k) + " holds fewer items than forecast() asks for");
    for (size_t k = 0; k < out.size(); k++)
        need((size_t)out[k].nbytes() >= (size_t)noutput_items * item_size(b, false, k),
             "output " + std::to_string(k) + " holds fewer than noutput_items items");
    return b.general_work(noutput_items, n, i, o);
}
}  // namespace

PYBIND11_MODULE(clenabled_python, m)
{
    m.doc() = "gr-clenabled blocks, MI355X build (HIP kernels behind the reference's block API)";
#ifdef MI355_WITH_GNURADIO
    py::module::import("gnuradio.gr");  // the base classes' bindings
    m.attr("with_gnuradio") = true;
#else
    m.attr("with_gnuradio") = false;
#endif
    // include/clenabled/GRCLBase.h:57-70, clMathOpTypes.h:11-20, clFFT.h:28-29
    m.attr("DTYPE_COMPLEX") = DTYPE_COMPLEX; m.attr("DTYPE_FLOAT") = DTYPE_FLOAT; m.attr("DTYPE_INT") = DTYPE_INT;
    m.attr("DTYPE_SHORT") = DTYPE_SHORT; m.attr("DTYPE_BYTE") = DTYPE_BYTE; m.attr("DTYPE_PACKEDXY") = DTYPE_PACKEDXY;
    m.attr("OCLTYPE_GPU") = OCLTYPE_GPU; m.attr("OCLTYPE_ACCELERATOR") = OCLTYPE_ACCELERATOR; m.attr("OCLTYPE_CPU") = OCLTYPE_CPU;
    m.attr("OCLTYPE_ANY") = OCLTYPE_ANY; m.attr("OCLDEVICESELECTOR_FIRST") = OCLDEVICESELECTOR_FIRST;
    m.attr("OCLDEVICESELECTOR_SPECIFIC") = OCLDEVICESELECTOR_SPECIFIC;
    m.attr("MATHOP_MULTIPLY") = MATHOP_MULTIPLY; m.attr("MATHOP_ADD") = MATHOP_ADD; m.attr("MATHOP_SUBTRACT") = MATHOP_SUBTRACT;
    m.attr("MATHOP_COMPLEX_CONJUGATE") = MATHOP_COMPLEX_CONJUGATE; m.attr("MATHOP_MULTIPLY_CONJUGATE") = MATHOP_MULTIPLY_CONJUGATE;
    m.attr("CLFFT_FORWARD") = CLFFT_FORWARD; m.attr("CLFFT_BACKWARD") = CLFFT_BACKWARD;
    m.attr("CLXCORR_TRIANGULAR_ORDER") = CLXCORR_TRIANGULAR_ORDER; m.attr("CLXCORR_FULL_MATRIX") = CLXCORR_FULL_MATRIX;

    py::class_<clMathOp SYNC_BASES, std::shared_ptr<clMathOp>>(m, "clMathOp")
        .def(py::init(&clMathOp::make), py::arg("idataType"), py::arg("openCLPlatformType"), py::arg("devSelector"), py::arg("platformId"),
             py::arg("devId"), py::arg("operatorType"), py::arg("setDebug") = 0)
        .def("work", &call_work<clMathOp>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"));

    py::class_<clMathConst SYNC_BASES, std::shared_ptr<clMathConst>>(m, "clMathConst")
        .def(py::init(&clMathConst::make), py::arg("idataType"), py::arg("openCLPlatformType"), py::arg("devSelector"), py::arg("platformId"),
             py::arg("devId"), py::arg("fValue"), py::arg("operatorType"), py::arg("setDebug") = 0)
        .def("k", &clMathConst::k)          // GRC callback set_k(${const}); also what the reference exposes on ControlPort
        .def("set_k", &clMathConst::set_k, py::arg("newValue"))
        .def("work", &call_work<clMathConst>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"));

    py::class_<clFFT SYNC_BASES, std::shared_ptr<clFFT>>(m, "clFFT")
        .def(py::init(&clFFT::make), py::arg("fftSize"), py::arg("clFFTDir"), py::arg("window"), py::arg("idataType"),
             py::arg("openCLPlatformType"), py::arg("devSelector"), py::arg("platformId"), py::arg("devId"), py::arg("setDebug") = 0,
             py::arg("num_streams") = 1, py::arg("shift") = false)
        .def("work", &call_work<clFFT>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"));

    py::class_<clFilter DECIM_BASES, std::shared_ptr<clFilter>>(m, "clFilter")
        .def(py::init(&clFilter::make), py::arg("openclPlatform"), py::arg("devSelector"), py::arg("platformId"), py::arg("devId"),
             py::arg("decimation"), py::arg("taps"), py::arg("nthreads") = 1, py::arg("setDebug") = 0, py::arg("use_time") = DEFAULT_USE_TIME_DOMAIN_SETTING)
        .def("set_taps2", &clFilter::set_taps2, py::arg("taps"))
        .def("taps", &clFilter::taps)
        .def("set_nthreads", &clFilter::set_nthreads, py::arg("n"))
        .def("work", &call_work<clFilter>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"));

    py::class_<clComplexFilter DECIM_BASES, std::shared_ptr<clComplexFilter>>(m, "clComplexFilter")
        .def(py::init(&clComplexFilter::make), py::arg("openclPlatform"), py::arg("devSelector"), py::arg("platformId"), py::arg("devId"),
             py::arg("decimation"), py::arg("taps"), py::arg("nthreads") = 1, py::arg("setDebug") = 0)
        .def("set_taps2", &clComplexFilter::set_taps2, py::arg("taps"))
        .def("taps", &clComplexFilter::taps)
        .def("work", &call_work<clComplexFilter>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"));

    py::class_<clPolyphaseChannelizer BLOCK_BASES, std::shared_ptr<clPolyphaseChannelizer>>(m, "clPolyphaseChannelizer")
        .def(py::init(&clPolyphaseChannelizer::make), py::arg("openCLPlatformType"), py::arg("devSelector"), py::arg("platformId"),
             py::arg("devId"), py::arg("taps"), py::arg("buf_items"), py::arg("num_channels"), py::arg("ninputs_per_iter"), py::arg("ch_map"),
             py::arg("setDebug") = 0)
        .def("general_work", &call_general_work<clPolyphaseChannelizer>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"));

    py::class_<clXEngine BLOCK_BASES, std::shared_ptr<clXEngine>>(m, "clXEngine")
        .def(py::init(&clXEngine::make), py::arg("openCLPlatformType"), py::arg("devSelector"), py::arg("platformId"), py::arg("devId"),
             py::arg("setDebug"), py::arg("data_type"), py::arg("polarization"), py::arg("num_inputs"), py::arg("output_format"),
             py::arg("first_channel"), py::arg("num_channels"), py::arg("integration"), py::arg("antenna_list"), py::arg("output_file") = false,
             py::arg("file_base") = "", py::arg("rollover_size_mb") = 0, py::arg("internal_synchronizer") = false, py::arg("sync_timestamp") = 0,
             py::arg("object_name") = "", py::arg("starting_chan_center_freq") = 0.0, py::arg("channel_width") = 0.0,
             py::arg("disable_output") = false, py::arg("pipeline_integration") = 0)
        .def("get_input_buffer_size", &clXEngine::get_input_buffer_size)
        .def("get_output_buffer_size", &clXEngine::get_output_buffer_size)
        .def("integrations_delivered", &clXEngine::integrations_delivered)
        .def("set_shard_devices", &clXEngine::set_shard_devices, py::arg("device_ids"), py::arg("windows_per_exchange") = 4)  // not in the reference: several devices behind one block
        .def("shard_devices", &clXEngine::shard_devices)
        .def("synchronized", &clXEngine::synchronized)
        .def("general_work", &call_general_work<clXEngine>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"))
        .def("stop", [](clXEngine &x) { return x.stop(); });  // (a member of the virtual base: no pointer-to-member through it)

    // ---- the remaining elementwise family and the reference correlator: python/bindings/clLog_python.cc etc.
#define MI355_BIND_ELEM(NAME, ...)                                                                                     \
    py::class_<NAME SYNC_BASES, std::shared_ptr<NAME>>(m, #NAME)                                                       \
        .def(py::init(&NAME::make), __VA_ARGS__)                                                                       \
        .def("work", &call_work<NAME>, py::arg("noutput_items"), py::arg("input_items"), py::arg("output_items"))
#define MI355_DEV_ARGS py::arg("openCLPlatformType"), py::arg("devSelector"), py::arg("platformId"), py::arg("devId")
    MI355_BIND_ELEM(clLog, MI355_DEV_ARGS, py::arg("nValue"), py::arg("kValue"), py::arg("setDebug") = 0);
    MI355_BIND_ELEM(clSNR, MI355_DEV_ARGS, py::arg("nValue"), py::arg("kValue"), py::arg("setDebug") = 0);
    MI355_BIND_ELEM(clComplexToMag, MI355_DEV_ARGS, py::arg("setDebug") = 0);
    MI355_BIND_ELEM(clComplexToArg, MI355_DEV_ARGS, py::arg("setDebug") = 0);
    MI355_BIND_ELEM(clComplexToMagPhase, MI355_DEV_ARGS, py::arg("setDebug") = 0);
    MI355_BIND_ELEM(clMagPhaseToComplex, MI355_DEV_ARGS, py::arg("setDebug") = 0);
    MI355_BIND_ELEM(clQuadratureDemod, py::arg("gain"), MI355_DEV_ARGS, py::arg("setDebug") = 0);
    MI355_BIND_ELEM(clxcorrelate_fft_vcf, py::arg("fftSize"), py::arg("num_inputs"), MI355_DEV_ARGS, py::arg("input_type") = 1);
#undef MI355_DEV_ARGS
#undef MI355_BIND_ELEM
}
