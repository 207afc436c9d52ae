#pragma once
#include <gnuradio/io_signature.h>
#include <gnuradio/logger.h>
#include <gnuradio/types.h>
#include <pmt/pmt.h>
#include <memory>
#include <string>
namespace gr {
class basic_block : public std::enable_shared_from_this<basic_block> {
protected:
    gr::logger_ptr d_logger, d_debug_logger;
    basic_block(void) {}
    basic_block(const std::string &name, gr::io_signature::sptr input_signature, gr::io_signature::sptr output_signature);
public:
    virtual ~basic_block();
    std::string name() const;
    std::string alias() const;
    gr::io_signature::sptr input_signature() const;
    gr::io_signature::sptr output_signature() const;
    void message_port_register_in(pmt::pmt_t port_id);
    void message_port_register_out(pmt::pmt_t port_id);
    void message_port_pub(pmt::pmt_t port_id, pmt::pmt_t msg);
    virtual void setup_rpc() {}
};
typedef std::shared_ptr<basic_block> basic_block_sptr;
}  // namespace gr
