// The 8-byte interleaved complex sample every buffer of the boundary carries (reference: include/clenabled/clSComplex.h:12-17;
// layout-identical to gr_complex = std::complex<float>, which tests/test_abi.py asserts on the C side).
#pragma once
typedef struct ComplexStruct { float real, imag; } SComplex;
#ifdef __cplusplus
static_assert(sizeof(SComplex) == 8, "SComplex is two packed floats");
#endif
