// gr::clenabled::clxcorrelate_fft_vcf, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clxcorrelate_fft_vcf.h:50
#pragma once
#include "GRCLBase.h"
#include "gr_compat.h"

namespace gr {
namespace clenabled {

// io: num_inputs vectors of fftSize complex in (input 0 = reference), num_inputs-1 vectors of fftSize float out
class CLENABLED_API clxcorrelate_fft_vcf : virtual public gr::sync_block {
public:
    typedef std::shared_ptr<clxcorrelate_fft_vcf> sptr;
    // input_type 1 = the inputs are spectra, 2 = time series (forward FFT first)
    static sptr make(int fftSize, int num_inputs, int openCLPlatformType, int devSelector, int platformId, int devId,
                     int input_type = 1);
    virtual int work_test(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};

}  // namespace clenabled
}  // namespace gr
