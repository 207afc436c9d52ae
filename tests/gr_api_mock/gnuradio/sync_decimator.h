#pragma once
#include <gnuradio/sync_block.h>
namespace gr {
class sync_decimator : public sync_block {
protected:
    sync_decimator(void) {}
    sync_decimator(const std::string &name, gr::io_signature::sptr input_signature, gr::io_signature::sptr output_signature,
                   unsigned decimation);
public:
    unsigned decimation() const;
    void set_decimation(unsigned decimation);
    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override;
};
}  // namespace gr
