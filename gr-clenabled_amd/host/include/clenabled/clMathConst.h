// gr::clenabled::clMathConst, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clMathConst.h:51-54
#pragma once
#include "GRCLBase.h"
#include "clMathOpTypes.h"
#include "gr_compat.h"

namespace gr {
namespace clenabled {

class CLENABLED_API clMathConst : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clMathConst> sptr;
    static sptr make(int idataType, int openCLPlatformType, int devSelector, int platformId, int devId, float fValue,
                     int operatorType, int setDebug = 0);
    virtual float k() const = 0;
    virtual void set_k(float newValue) = 0;
    virtual int testOpenCL(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
