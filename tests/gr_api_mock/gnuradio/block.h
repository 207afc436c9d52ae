#pragma once
#include <gnuradio/basic_block.h>
#include <gnuradio/tags.h>
namespace gr {
class block : public basic_block {
public:
    enum work_return_t { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2, TPP_CUSTOM = 3 };
    ~block() override;
    unsigned history() const;
    void set_history(unsigned history);
    virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required);
    virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                             gr_vector_void_star &output_items);
    virtual bool start();
    virtual bool stop();
    void set_output_multiple(int multiple);
    int output_multiple() const;
    void consume(int which_input, int how_many_items);
    void consume_each(int how_many_items);
    void produce(int which_output, int how_many_items);
    uint64_t nitems_read(unsigned int which_input);
    uint64_t nitems_written(unsigned int which_output);
    tag_propagation_policy_t tag_propagation_policy();
    void set_tag_propagation_policy(tag_propagation_policy_t p);
protected:
    block(void) {}
    block(const std::string &name, gr::io_signature::sptr input_signature, gr::io_signature::sptr output_signature);
    void add_item_tag(unsigned int which_output, const tag_t &tag);
    void get_tags_in_range(std::vector<tag_t> &v, unsigned int which_input, uint64_t abs_start, uint64_t abs_end);
    void get_tags_in_window(std::vector<tag_t> &v, unsigned int which_input, uint64_t rel_start, uint64_t rel_end);
};
typedef std::shared_ptr<block> block_sptr;
}  // namespace gr
