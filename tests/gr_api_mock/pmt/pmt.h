#pragma once
#include <complex>
#include <cstdint>
#include <memory>
#include <string>
namespace pmt {
class pmt_base;
typedef std::shared_ptr<pmt_base> pmt_t;
pmt_t mp(const char *s);
pmt_t mp(const std::string &s);
pmt_t intern(const std::string &s);
pmt_t string_to_symbol(const std::string &s);
pmt_t cons(const pmt_t &x, const pmt_t &y);
pmt_t from_uint64(uint64_t x);
uint64_t to_uint64(pmt_t x);
pmt_t from_double(double x);
pmt_t init_c32vector(size_t k, const std::complex<float> *data);
}  // namespace pmt
