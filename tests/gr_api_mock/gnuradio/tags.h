#pragma once
#include <pmt/pmt.h>
namespace gr {
struct tag_t {
    uint64_t offset;
    pmt::pmt_t key, value, srcid;
};
}  // namespace gr
