// gr::clenabled::clComplexFilter, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clComplexFilter.h:706-709
#pragma once
#include "GRCLBase.h"
#include "gr_compat.h"

namespace gr {
namespace clenabled {

class CLENABLED_API clComplexFilter : virtual public gr::sync_decimator {
public:
    typedef std::shared_ptr<clComplexFilter> sptr;
    static sptr make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                     const std::vector<gr_complex> &taps, int nthreads = 1, int setDebug = 0);
    virtual void set_taps2(const std::vector<gr_complex> &taps) = 0;
    virtual std::vector<gr_complex> taps() const = 0;
    virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
