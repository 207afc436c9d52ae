// gr::clenabled::clMagPhaseToComplex, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clMagPhaseToComplex.h:49
#pragma once
#include "GRCLBase.h"
#include "gr_compat.h"

namespace gr {
namespace clenabled {

class CLENABLED_API clMagPhaseToComplex : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clMagPhaseToComplex> sptr;
    static sptr make(int openCLPlatformType, int devSelector, int platformId, int devId, int setDebug = 0);
    virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
