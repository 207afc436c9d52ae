#pragma once
#include <memory>
namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p);
template <class T, class... Args> std::shared_ptr<T> make_block_sptr(Args &&...args);
}  // namespace gnuradio
