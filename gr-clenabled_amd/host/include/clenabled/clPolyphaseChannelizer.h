// gr::clenabled::clPolyphaseChannelizer, MI355X build -- public header, same include path and make() signature as the reference's
// include/clenabled/clPolyphaseChannelizer.h:48-49
#pragma once
#include "GRCLBase.h"
#include "gr_compat.h"

namespace gr {
namespace clenabled {

class CLENABLED_API clPolyphaseChannelizer : virtual public gr::block {
public:
    typedef std::shared_ptr<clPolyphaseChannelizer> sptr;
    static sptr make(int openCLPlatformType, int devSelector, int platformId, int devId, const std::vector<float> &taps,
                     int buf_items, int num_channels, int ninputs_per_iter, const std::vector<int> &ch_map, int setDebug = 0);
};

}  // namespace clenabled
}  // namespace gr
