// ear/dsp/variable_block_size.hpp — libear include/ear/dsp/variable_block_size.hpp:17-40
// over earhip_vbs_*.  Exceptions thrown by process_func propagate to the caller
// of process(), as they do in libear.
#pragma once
#include <cstddef>
#include <exception>
#include <functional>
#include "../hip.hpp"

namespace ear {
  namespace dsp {
    class VariableBlockSizeAdapter {
     public:
      using ProcessFunc = void(const float *const *in, float *const *out);

      VariableBlockSizeAdapter(size_t block_size, size_t num_channels_in, size_t num_channels_out,
                               std::function<ProcessFunc> process_func)
          : fn_(std::move(process_func)) {
        hip::check(earhip_vbs_create(block_size, num_channels_in, num_channels_out, &trampoline,
                                     this, &h_));
      }
      /// The same with the adapter's FIFO buffers in device-reachable host memory of `ctx`: an ObjectsRenderer
      /// called from process_func takes its no-staging path (0.085 instead of 0.12 ms per 512-sample block at
      /// 1024 objects).  The adapter must be destroyed before the context.
      VariableBlockSizeAdapter(size_t block_size, size_t num_channels_in, size_t num_channels_out,
                               std::function<ProcessFunc> process_func, hip::Context &ctx)
          : fn_(std::move(process_func)) {
        hip::check(earhip_vbs_create_pinned(ctx.get(), block_size, num_channels_in, num_channels_out, &trampoline,
                                            this, &h_));
      }
      ~VariableBlockSizeAdapter() { earhip_vbs_destroy(h_); }
      VariableBlockSizeAdapter(const VariableBlockSizeAdapter &) = delete;
      VariableBlockSizeAdapter &operator=(const VariableBlockSizeAdapter &) = delete;

      void process(size_t nsamples, const float *const *in, float *const *out) {
        const int status = earhip_vbs_process(h_, nsamples, in, out);
        if (pending_) {
          std::exception_ptr e = pending_;
          pending_ = nullptr;
          std::rethrow_exception(e);
        }
        hip::check(status);
      }
      int get_delay() const { return earhip_vbs_get_delay(h_); }

     private:
      static int trampoline(const float *const *in, float *const *out, void *user) {
        auto *self = static_cast<VariableBlockSizeAdapter *>(user);
        try {
          self->fn_(in, out);
          return EARHIP_OK;
        } catch (...) {
          self->pending_ = std::current_exception();
          return EARHIP_INTERNAL_ERROR;
        }
      }
      std::function<ProcessFunc> fn_;
      earhip_vbs *h_ = nullptr;
      std::exception_ptr pending_;
    };
  }  // namespace dsp
}  // namespace ear
