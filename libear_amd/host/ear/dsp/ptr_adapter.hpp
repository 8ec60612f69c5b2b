// ear/dsp/ptr_adapter.hpp — channel-pointer array over planar storage
// (libear include/ear/dsp/ptr_adapter.hpp:10-40).  set_eigen works with any
// column-major matrix type offering cols() and col(i).data(); set_planar is the
// Eigen-free equivalent.
#pragma once
#include <cstddef>
#include <vector>
#include "../helpers/assert.hpp"

namespace ear {
  namespace dsp {
    template <typename PtrT = float *>
    class PtrAdapterT {
     public:
      explicit PtrAdapterT(size_t nchannels) : _ptrs(nchannels) {}

      template <typename T>
      void set_eigen(T &&mat, size_t offset = 0) {
        ear_assert((size_t)mat.cols() == _ptrs.size(), "wrong number of channels");
        for (size_t i = 0; i < _ptrs.size(); i++) _ptrs[i] = mat.col(i).data() + offset;
      }
      /// channel c at base + c * stride + offset
      void set_planar(PtrT base, size_t stride, size_t offset = 0) {
        for (size_t i = 0; i < _ptrs.size(); i++) _ptrs[i] = base + i * stride + offset;
      }
      PtrT *ptrs() { return _ptrs.data(); }

      PtrAdapterT(const PtrAdapterT &) = delete;
      PtrAdapterT &operator=(const PtrAdapterT &) = delete;

     private:
      std::vector<PtrT> _ptrs;
    };
    using PtrAdapter = PtrAdapterT<float *>;
    using PtrAdapterConst = PtrAdapterT<const float *>;
  }  // namespace dsp
}  // namespace ear
