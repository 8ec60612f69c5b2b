// ear/dsp/ptr_adapter.hpp — the `float *const *` view that every process() call of this
// library takes, built over planar storage (one contiguous run of samples per channel).
// Mirrors the interface of libear's PtrAdapterT (include/ear/dsp/ptr_adapter.hpp:10-40):
// same class names, set_eigen(mat, offset), ptrs(), non-copyable because it aliases
// memory it does not own.
#pragma once
#include <cstddef>
#include <memory>
#include "../helpers/assert.hpp"

namespace ear {
  namespace dsp {

    template <typename PtrT = float *>
    class PtrAdapterT {
      std::unique_ptr<PtrT[]> channel_;  // channel_[c] = first sample of channel c
      size_t count_;

      // fill from any callable mapping a channel index to its first sample
      template <typename FirstSample>
      void assign(FirstSample first) {
        for (size_t c = 0; c != count_; ++c) channel_[c] = first(c);
      }

     public:
      explicit PtrAdapterT(size_t nchannels) : channel_(new PtrT[nchannels]()), count_(nchannels) {}
      PtrAdapterT(PtrAdapterT const &) = delete;
      void operator=(PtrAdapterT const &) = delete;

      /// Columns of a column-major matrix are the channels (anything with cols()
      /// and col(c).data(), e.g. an Eigen matrix or block); `offset` skips samples.
      template <typename Matrix>
      void set_eigen(Matrix &&mat, size_t offset = 0) {
        ear_assert(static_cast<size_t>(mat.cols()) == count_, "wrong number of channels");
        assign([&](size_t c) { return mat.col(c).data() + offset; });
      }

      /// Eigen-free variant: channel c starts at base + c * stride + offset.
      void set_planar(PtrT base, size_t stride, size_t offset = 0) {
        assign([=](size_t c) { return base + c * stride + offset; });
      }

      PtrT *ptrs() { return channel_.get(); }
      size_t size() const { return count_; }
    };

    using PtrAdapter = PtrAdapterT<float *>;
    using PtrAdapterConst = PtrAdapterT<const float *>;

  }  // namespace dsp
}  // namespace ear
