// ear/helpers/assert.hpp — always-on assertion (libear include/ear/helpers/assert.hpp:7-18)
#pragma once
#include <string>
#include "../exceptions.hpp"

namespace ear {
  inline void _assert_impl(bool condition, const std::string &message) {
    if (!condition) throw internal_error("internal error: " + message);
  }
}  // namespace ear
#define ear_assert(condition, message) ear::_assert_impl((condition), (message))
