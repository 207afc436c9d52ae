// gr::clenabled::clMathOp, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clMathOp.h:42
#pragma once
#include "GRCLBase.h"
#include "clMathOpTypes.h"
#include "gr_compat.h"

namespace gr {
namespace clenabled {

class CLENABLED_API clMathOp : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clMathOp> sptr;
    static sptr make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, int operatorType,
                     int setDebug = 0);
    // the timing hook of the reference's CLI (lib/clMathOp_impl.h:64-79)
    virtual int testOpenCL(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
