// Scheduler-facing base classes of the block layer.
//
// With GNU Radio installed (CMake finds it and defines MI355_WITH_GNURADIO) these are GNU Radio's own gr::block /
// gr::sync_block / gr::sync_decimator, gr::io_signature, message ports (pmt) and stream tags, and the blocks of
// lib/clenabled_impl.cc are ordinary out-of-tree blocks.  This image has no GNU Radio, so the same block code also compiles
// against the small stand-alone classes below: the same constructor shape (name, input signature, output signature), the
// same calls (set_history, set_output_multiple, consume / consume_each, message_port_register_out) -- they RECORD what the
// block asked of the scheduler so that the CLI and the tests can act as the scheduler and check it.  Nothing here computes.
#pragma once
#ifdef MI355_WITH_GNURADIO
#include <gnuradio/block.h>
#include <gnuradio/io_signature.h>
#include <gnuradio/logger.h>
#include <gnuradio/sptr_magic.h>
#include <gnuradio/sync_block.h>
#include <gnuradio/sync_decimator.h>
#include <pmt/pmt.h>
#else
#include <complex>
#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <vector>
typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;
namespace gr {
class io_signature {
    int d_min, d_max;
    std::vector<int> d_sizes;
public:
    typedef std::shared_ptr<io_signature> sptr;
    static constexpr int IO_INFINITE = -1;
    io_signature(int mn, int mx, const std::vector<int> &sizes) : d_min(mn), d_max(mx), d_sizes(sizes) {}
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return std::make_shared<io_signature>(min_streams, max_streams, std::vector<int>{sizeof_stream_item});
    }
    static sptr makev(int min_streams, int max_streams, const std::vector<int> &sizes)
    {
        return std::make_shared<io_signature>(min_streams, max_streams, sizes);
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int i) const { return d_sizes.empty() ? 0 : d_sizes[(size_t)i < d_sizes.size() ? i : d_sizes.size() - 1]; }
};

// one published message of the stand-alone build: port, key of the pair, and either a complex vector or an integer
struct shim_message {
    std::string port, key;
    std::vector<gr_complex> c32;
    uint64_t u64 = 0;
};

class basic_block_shim {
    std::string d_name;
    io_signature::sptr d_in, d_out;
    unsigned d_history = 1;
    int d_output_multiple = 1;
    std::vector<long> d_consumed;  // per input, since the last reset_consumed()
    std::vector<std::string> d_out_ports;
    std::deque<shim_message> d_messages;
    std::vector<uint64_t> d_first_tags;  // what get_tags_in_window(i, 0, 1) would return, set by the caller acting as scheduler

protected:
    basic_block_shim() {}  // (interface classes; the _impl constructs the virtual base)

public:
    basic_block_shim(const std::string &name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(in), d_out(out) {}
    virtual ~basic_block_shim() {}
    const std::string &name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    unsigned history() const { return d_history; }
    void set_history(unsigned h) { d_history = h; }
    int output_multiple() const { return d_output_multiple; }
    void set_output_multiple(int m) { d_output_multiple = m; }
    virtual bool start() { return true; }
    virtual bool stop() { return true; }
    // general_work() side of the contract
    void consume(int which_input, int how_many)
    {
        if ((size_t)which_input >= d_consumed.size()) d_consumed.resize(which_input + 1, 0);
        d_consumed[which_input] += how_many;
    }
    void consume_each(int how_many)
    {
        const int n = d_in ? (d_in->max_streams() > 0 ? d_in->max_streams() : (int)d_consumed.size()) : 1;
        for (int i = 0; i < (n > 0 ? n : 1); i++) consume(i, how_many);
    }
    long nitems_consumed(int which_input) const { return (size_t)which_input < d_consumed.size() ? d_consumed[which_input] : 0; }
    void reset_consumed() { d_consumed.assign(d_consumed.size(), 0); }
    // message ports
    void message_port_register_out(const std::string &port) { d_out_ports.push_back(port); }
    const std::vector<std::string> &message_ports_out() const { return d_out_ports; }
    void shim_publish(shim_message &&m) { d_messages.push_back(std::move(m)); }
    bool pop_message(shim_message &m)
    {
        if (d_messages.empty()) return false;
        m = std::move(d_messages.front());
        d_messages.pop_front();
        return true;
    }
    // stream tags: the first tag value of every input in the current window (set by the caller acting as scheduler)
    void set_first_tags(const std::vector<uint64_t> &t) { d_first_tags = t; }

protected:
    // reading tags is a protected service of gr::block (get_tags_in_window): only the block itself may ask
    bool shim_first_tag(int which_input, uint64_t &value) const
    {
        if ((size_t)which_input >= d_first_tags.size()) return false;
        value = d_first_tags[which_input];
        return true;
    }
};
class sync_block : public basic_block_shim {
protected:
    sync_block() {}
public:
    using basic_block_shim::basic_block_shim;
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};
class sync_decimator : public sync_block {
    unsigned d_decimation = 1;
protected:
    sync_decimator() {}
public:
    sync_decimator(const std::string &name, io_signature::sptr in, io_signature::sptr out, unsigned decimation)
        : sync_block(name, in, out), d_decimation(decimation) {}
    unsigned decimation() const { return d_decimation; }
};
class block : public basic_block_shim {
protected:
    block() {}
public:
    using basic_block_shim::basic_block_shim;
    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required) = 0;
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items) = 0;
};
}  // namespace gr
#endif

// ---- the few scheduler services the blocks use, spelled once for both builds ------------------------------------
namespace gr {
namespace clenabled {
namespace sched {
#ifdef MI355_WITH_GNURADIO
inline void register_out(gr::basic_block *b, const char *port) { b->message_port_register_out(pmt::mp(port)); }
// message_port_pub(port, cons(intern(key), c32vector)) -- lib/clXEngine_impl.cc:1076-1077
inline void publish_c32vector(gr::basic_block *b, const char *port, const char *key, const gr_complex *v, size_t n)
{
    b->message_port_pub(pmt::mp(port), pmt::cons(pmt::string_to_symbol(key), pmt::init_c32vector(n, v)));
}
// message_port_pub(port, cons(intern(key), uint64)) -- lib/clXEngine_impl.cc:1203-1204
inline void publish_u64(gr::basic_block *b, const char *port, const char *key, uint64_t value)
{
    b->message_port_pub(pmt::mp(port), pmt::cons(pmt::intern(key), pmt::from_uint64(value)));
}
// (reading the first tag of an input, get_tags_in_window, is protected in gr::block: a member of clXEngine_impl does it)
// blocks are handed to the scheduler through GNU Radio's initial-sptr registry, as every make() of the reference does
template <class T> std::shared_ptr<T> adopt(T *p) { return gnuradio::get_initial_sptr(p); }
inline void no_tag_propagation(gr::block *b) { b->set_tag_propagation_policy(gr::block::TPP_DONT); }
#else
inline void register_out(gr::basic_block_shim *b, const char *port) { b->message_port_register_out(port); }
inline void publish_c32vector(gr::basic_block_shim *b, const char *port, const char *key, const gr_complex *v, size_t n)
{
    gr::shim_message m;
    m.port = port;
    m.key = key;
    m.c32.assign(v, v + n);
    b->shim_publish(std::move(m));
}
inline void publish_u64(gr::basic_block_shim *b, const char *port, const char *key, uint64_t value)
{
    gr::shim_message m;
    m.port = port;
    m.key = key;
    m.u64 = value;
    b->shim_publish(std::move(m));
}
template <class T> std::shared_ptr<T> adopt(T *p) { return std::shared_ptr<T>(p); }
inline void no_tag_propagation(gr::basic_block_shim *) {}
#endif
}  // namespace sched
}  // namespace clenabled
}  // namespace gr
