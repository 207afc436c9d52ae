// Symbol visibility of libgnuradio-clenabled-mi355 (stands where the reference's include/clenabled/api.h:25-31 is).
#pragma once
#ifdef MI355_WITH_GNURADIO
#include <gnuradio/attributes.h>
#define MI355_VIS_EXPORT __GR_ATTR_EXPORT
#define MI355_VIS_IMPORT __GR_ATTR_IMPORT
#else
#define MI355_VIS_EXPORT __attribute__((visibility("default")))
#define MI355_VIS_IMPORT __attribute__((visibility("default")))
#endif
#if defined(gnuradio_clenabled_EXPORTS) || defined(gnuradio_clenabled_mi355_EXPORTS)
#define CLENABLED_API MI355_VIS_EXPORT
#else
#define CLENABLED_API MI355_VIS_IMPORT
#endif
