#pragma once
#include <gnuradio/gr_complex.h>
#include <vector>
typedef std::vector<int> gr_vector_int;
typedef std::vector<void *> gr_vector_void_star;
typedef std::vector<const void *> gr_vector_const_void_star;
