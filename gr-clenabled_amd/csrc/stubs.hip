// TEMPORARY: entry points not implemented yet return MI355_ERR_UNSUPPORTED.
#include "common.h"
#define STUB(sig) extern "C" int sig { mi355_set_error("not implemented yet"); return MI355_ERR_UNSUPPORTED; }
STUB(mi355_pfb_create(mi355_ctx *, const float *, int, int, int, int, const int *, int, mi355_pfb **))
STUB(mi355_pfb_destroy(mi355_pfb *))
STUB(mi355_pfb_noutput(const mi355_pfb *))
STUB(mi355_pfb_ninput(const mi355_pfb *))
STUB(mi355_pfb_work(mi355_pfb *, const void *, void *))
STUB(mi355_pfb_work_dev(mi355_pfb *, const void *, void *, void *))
STUB(mi355_xengine_create(mi355_ctx *, int, int, int, int, int, mi355_xengine **))
STUB(mi355_xengine_destroy(mi355_xengine *))
extern "C" size_t mi355_xengine_input_bytes(const mi355_xengine *) { return 0; }
extern "C" size_t mi355_xengine_output_items(const mi355_xengine *) { return 0; }
STUB(mi355_xengine_xcorrelate(mi355_xengine *, const void *, void *, int))
STUB(mi355_xengine_xcorrelate_dev(mi355_xengine *, const void *, void *, int, void *))
STUB(mi355_xengine_gather(const mi355_xengine *, int, int, const void *const *, void *))
