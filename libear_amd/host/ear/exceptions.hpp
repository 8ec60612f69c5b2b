// ear/exceptions.hpp — error types of the ear:: namespace, same names and
// hierarchy as libear's include/ear/exceptions.hpp:8-43, so user code that
// catches libear's exceptions keeps working on the HIP path.
#pragma once
#include <stdexcept>
#include <string>

namespace ear {
  /// errors inside the library (always-on assertions, HIP runtime failures)
  class internal_error : public std::runtime_error {
   public:
    explicit internal_error(const std::string &what) : std::runtime_error(what) {}
  };
  /// invariants on parameters are not met
  class invalid_argument : public std::invalid_argument {
   public:
    explicit invalid_argument(const std::string &what) : std::invalid_argument(what) {}
  };
  class not_implemented : public std::runtime_error {
   public:
    explicit not_implemented(const std::string &what)
        : std::runtime_error("not implemented: " + what) {}
  };
  /// invalid ADM metadata
  class adm_error : public std::invalid_argument {
   public:
    explicit adm_error(const std::string &what) : std::invalid_argument("ADM error: " + what) {}
  };
  /// an unknown loudspeaker layout is requested
  class unknown_layout : public std::invalid_argument {
   public:
    explicit unknown_layout(const std::string &what) : std::invalid_argument("unknown layout: " + what) {}
  };
}  // namespace ear
