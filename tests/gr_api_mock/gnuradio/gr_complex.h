#pragma once
#include <complex>
typedef std::complex<float> gr_complex;
