// ear/hip.hpp — glue between the C++ mirror classes and the C ABI (earhip.h):
// status -> exception translation and the process-wide default context.
#pragma once
#include <cstdlib>
#include <string>

#include "earhip.h"
#include "exceptions.hpp"

namespace ear {
  namespace hip {
    /// Translate an earhip status into libear's exception types
    /// (include/earhip.h: 1 -> ear::invalid_argument, 2/3 -> ear::internal_error).
    inline void check(int status) {
      if (status == EARHIP_OK) return;
      const std::string msg = earhip_last_error();
      if (status == EARHIP_INVALID_ARGUMENT) throw invalid_argument(msg);
      throw internal_error(msg);
    }

    /// One device + stream.  Not thread-safe, like libear's stateful DSP objects.
    class Context {
     public:
      explicit Context(int device = 0, void *stream = nullptr) {
        check(earhip_ctx_create(device, stream, &ctx_));
      }
      ~Context() { earhip_ctx_destroy(ctx_); }
      Context(const Context &) = delete;
      Context &operator=(const Context &) = delete;
      earhip_ctx *get() const { return ctx_; }
      /// bit-exact libear arithmetic in the gain kernels (slower)
      void set_strict(bool strict) { check(earhip_ctx_set_strict(ctx_, strict ? 1 : 0)); }
      void synchronize() { check(earhip_ctx_synchronize(ctx_)); }

     private:
      earhip_ctx *ctx_ = nullptr;
    };

    /// Context used by the drop-in classes whose libear signature has no place
    /// for one.  Device from $EARHIP_DEVICE (default 0).  Throws
    /// ear::internal_error when no GPU is present: there is no CPU fallback.
    inline Context &default_context() {
      static Context ctx([] {
        const char *e = std::getenv("EARHIP_DEVICE");
        return e ? std::atoi(e) : 0;
      }());
      return ctx;
    }
  }  // namespace hip
}  // namespace ear
