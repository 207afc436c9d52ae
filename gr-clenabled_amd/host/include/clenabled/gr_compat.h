// Minimal block scaffolding used when GNU Radio's headers are not available (this image has none).
// With GNU Radio present, define MI355_WITH_GNURADIO and the real gr::sync_block / gr::block /
// gr::sync_decimator, gr_complex and vector typedefs are used instead; the class bodies are identical.
#pragma once
#ifdef MI355_WITH_GNURADIO
#include <gnuradio/block.h>
#include <gnuradio/sync_block.h>
#include <gnuradio/sync_decimator.h>
#else
#include <complex>
#include <memory>
#include <string>
#include <vector>
typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;
namespace gr {
// just enough of the scheduler-facing surface for standalone (CLI / test) use of the blocks
class basic_block_shim {
    std::string d_name;
    unsigned d_history = 1;
    int d_output_multiple = 1;
public:
    explicit basic_block_shim(const std::string &name) : d_name(name) {}
    virtual ~basic_block_shim() {}
    const std::string &name() const { return d_name; }
    unsigned history() const { return d_history; }
    void set_history(unsigned h) { d_history = h; }
    int output_multiple() const { return d_output_multiple; }
    void set_output_multiple(int m) { d_output_multiple = m; }
    virtual bool start() { return true; }
    virtual bool stop() { return true; }
};
class sync_block : public basic_block_shim {
public:
    using basic_block_shim::basic_block_shim;
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};
class sync_decimator : public sync_block {
    unsigned d_decimation;
public:
    sync_decimator(const std::string &name, unsigned decimation) : sync_block(name), d_decimation(decimation) {}
    unsigned decimation() const { return d_decimation; }
};
class block : public basic_block_shim {
public:
    using basic_block_shim::basic_block_shim;
    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required) = 0;
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items) = 0;
};
}  // namespace gr
#endif
