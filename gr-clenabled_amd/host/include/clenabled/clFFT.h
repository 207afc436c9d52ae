// gr::clenabled::clFFT, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clFFT.h:54-55
#pragma once
#include "GRCLBase.h"
#include "gr_compat.h"

// Direction codes.  make() takes the clFFT LIBRARY's codes, which is what GRC passes (grc/clenabled_clFFT.block.yml:37-41:
// "-1" forward, "1" reverse) and what lib/clFFT_impl.cc:84-89 compares against; the reference's own header also defines
// CLFFT_FWD / CLFFT_REV (clFFT.h:28-29), which nothing reads -- kept for source compatibility only.
constexpr int CLFFT_FORWARD = -1, CLFFT_BACKWARD = 1;
constexpr int CLFFT_FWD = 1, CLFFT_REV = 2;

namespace gr {
namespace clenabled {

class CLENABLED_API clFFT : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clFFT> sptr;
    // positional order of lib/clFFT_impl.cc:34-36 (what GRC passes; the reference header's parameter NAMES differ, App. B-1).  The
    // defaults are the reference header's (include/clenabled/clFFT.h:54-55: the 8th positional argument defaults to 4): a seven-argument
    // caller compiles against both headers, and its seventh argument lands where the reference's implementation reads it.
    static sptr make(int fftSize, int clFFTDir, const std::vector<float> &window, int idataType, int openCLPlatformType,
                     int devSelector, int platformId, int devId = 4, int setDebug = 0, int num_streams = 1, bool shift = false);
    // counts SAMPLES like the reference's test hook (lib/clFFT_impl.cc:520-524)
    virtual int testOpenCL(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
