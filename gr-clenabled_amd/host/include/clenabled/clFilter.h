// gr::clenabled::clFilter, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clFilter.h:52-61
#pragma once
#include "GRCLBase.h"
#include "gr_compat.h"

namespace gr {
namespace clenabled {

const bool DEFAULT_USE_TIME_DOMAIN_SETTING = false;  // clFilter.h:32: overlap-save unless asked otherwise

class CLENABLED_API clFilter : virtual public gr::sync_decimator {
public:
    typedef std::shared_ptr<clFilter> sptr;
    static sptr make(int openclPlatform, int devSelector, int platformId, int devId, int decimation,
                     const std::vector<float> &taps, int nthreads = 1, int setDebug = 0,
                     bool use_time = DEFAULT_USE_TIME_DOMAIN_SETTING);
    virtual void set_taps2(const std::vector<float> &taps) = 0;
    virtual std::vector<float> taps() const = 0;
    virtual void set_nthreads(int n) = 0;
    virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
