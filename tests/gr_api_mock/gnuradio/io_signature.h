#pragma once
#include <memory>
#include <vector>
namespace gr {
class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    static constexpr int IO_INFINITE = -1;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item);
    static sptr makev(int min_streams, int max_streams, const std::vector<int> &sizeof_stream_items);
    int min_streams() const;
    int max_streams() const;
    int sizeof_stream_item(int index) const;
};
}  // namespace gr
