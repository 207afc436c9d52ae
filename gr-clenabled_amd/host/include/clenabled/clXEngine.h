// gr::clenabled::clXEngine, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clXEngine.h:48-52
#pragma once
#include "GRCLBase.h"
#include "gr_compat.h"

constexpr int CLXCORR_TRIANGULAR_ORDER = 1, CLXCORR_FULL_MATRIX = 2;  // output orders, lib/clXEngine_impl.h:28-29

namespace gr {
namespace clenabled {

struct XComplex { float real = 0.0f, imag = 0.0f; };  // lib/clXEngine_impl.h:34-38

class CLENABLED_API clXEngine : virtual public gr::block {
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
    // Not in the reference (one device per block, devId: lib/GRCLBase.cpp:115-134): run THIS block over several devices of the process --
    // device r ingests antenna group r over its own host link, the devices exchange (corner turn over xGMI) and device r correlates channel
    // slab r; results are identical to the one-device block's.  IChar input, num_inputs * polarization <= 64, the device count must divide
    // num_inputs and num_channels.  Also set by the environment variable MI355_XENGINE_DEVICES="0,1,2,3" for flowgraphs that are not edited.
    // An empty or one-element list returns to the device of make().  Streamed frames (work / work_test) go to the devices windows_per_exchange
    // integration windows at a time (MI355_XENGINE_SHARD_WINDOWS overrides): gathered straight into pinned memory, uploaded by every device over its
    // own link while the next windows are gathered; results come out in order, windows_per_exchange at a time (stop() flushes a partial batch).
    virtual void set_shard_devices(const std::vector<int> &device_ids, int windows_per_exchange = 4) = 0;
    virtual int shard_devices() const = 0;
    // stream-tag synchroniser state (internal_synchronizer = true, lib/clXEngine_impl.cc:1158-1226)
    virtual bool synchronized() const = 0;
    virtual uint64_t sync_tag() const = 0;
};

}  // namespace clenabled
}  // namespace gr
