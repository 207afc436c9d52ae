#pragma once
#include <gnuradio/block.h>
namespace gr {
class sync_block : public block {
protected:
    sync_block(void) {}
    sync_block(const std::string &name, gr::io_signature::sptr input_signature, gr::io_signature::sptr output_signature);
public:
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;
    int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) override;
};
}  // namespace gr
