// TEMPORARY: entry points not implemented yet return MI355_ERR_UNSUPPORTED.
#include "common.h"
#define STUB(sig) extern "C" int sig { mi355_set_error("not implemented yet"); return MI355_ERR_UNSUPPORTED; }
STUB(mi355_xengine_create(mi355_ctx *, int, int, int, int, int, mi355_xengine **))
STUB(mi355_xengine_destroy(mi355_xengine *))
extern "C" size_t mi355_xengine_input_bytes(const mi355_xengine *) { return 0; }
extern "C" size_t mi355_xengine_output_items(const mi355_xengine *) { return 0; }
STUB(mi355_xengine_xcorrelate(mi355_xengine *, const void *, void *, int))
STUB(mi355_xengine_xcorrelate_dev(mi355_xengine *, const void *, void *, int, void *))
STUB(mi355_xengine_gather(const mi355_xengine *, int, int, const void *const *, void *))
