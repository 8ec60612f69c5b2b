// ear/dsp/delay_buffer.hpp — libear include/ear/dsp/delay_buffer.hpp:12-30 over earhip_delay_*
#pragma once
#include <cstddef>
#include "../hip.hpp"

namespace ear {
  namespace dsp {
    /// A multi-channel delay buffer
    class DelayBuffer {
     public:
      DelayBuffer(size_t nchannels, size_t nsamples) {
        hip::check(earhip_delay_create(hip::default_context().get(), nchannels, nsamples, &h_));
      }
      ~DelayBuffer() { earhip_delay_destroy(h_); }
      DelayBuffer(const DelayBuffer &) = delete;
      DelayBuffer &operator=(const DelayBuffer &) = delete;
      void process(size_t nsamples, const float *const *input, float *const *output) {
        hip::check(earhip_delay_process(h_, nsamples, input, output));
      }
      int get_delay() const { return earhip_delay_get_delay(h_); }

     private:
      earhip_delay *h_ = nullptr;
    };
  }  // namespace dsp
}  // namespace ear
