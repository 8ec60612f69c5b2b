// ear/fft.hpp — FFT plugin interface, same shape as libear's include/ear/fft.hpp:14-67,
// plus get_fft_hip(): an implementation backed by the device transform
// (earhip_fft_*), for users of the plugin point and for parity tests.
#pragma once
#include <complex>
#include <memory>

#include "hip.hpp"

namespace ear {
  class FFTWorkBuf {
   public:
    virtual ~FFTWorkBuf() {}
  };

  template <typename Real>
  class FFTPlan {
   public:
    using Complex = std::complex<Real>;
    /// r2c: n_fft reals -> n_fft/2+1 unpacked bins, un-normalised
    virtual void transform_forward(Real *input, Complex *output, FFTWorkBuf &workbuf) const = 0;
    /// c2r: n_fft/2+1 bins -> n_fft reals, un-normalised
    virtual void transform_reverse(Complex *input, Real *output, FFTWorkBuf &workbuf) const = 0;
    virtual std::unique_ptr<FFTWorkBuf> alloc_workbuf() const = 0;
    virtual ~FFTPlan() {}
  };

  template <typename Real>
  class FFTImpl {
   public:
    virtual std::shared_ptr<FFTPlan<Real>> plan(size_t n_fft) const = 0;
    virtual ~FFTImpl() {}
  };

  namespace hip {
    class FFTWorkBufHip : public FFTWorkBuf {};

    class FFTPlanHip : public FFTPlan<float> {
     public:
      explicit FFTPlanHip(size_t n_fft) {
        check(earhip_fft_plan_create(default_context().get(), n_fft, &plan_));
      }
      ~FFTPlanHip() override { earhip_fft_plan_destroy(plan_); }
      void transform_forward(float *input, Complex *output, FFTWorkBuf &) const override {
        check(earhip_fft_forward(plan_, input, reinterpret_cast<float *>(output)));
      }
      void transform_reverse(Complex *input, float *output, FFTWorkBuf &) const override {
        check(earhip_fft_reverse(plan_, reinterpret_cast<const float *>(input), output));
      }
      std::unique_ptr<FFTWorkBuf> alloc_workbuf() const override {
        return std::unique_ptr<FFTWorkBuf>(new FFTWorkBufHip);
      }

     private:
      earhip_fft_plan *plan_ = nullptr;
    };

    class FFTHip : public FFTImpl<float> {
     public:
      std::shared_ptr<FFTPlan<float>> plan(size_t n_fft) const override {
        if (n_fft % 2 != 0) throw internal_error("internal error: n_fft must be even");
        return std::make_shared<FFTPlanHip>(n_fft);
      }
    };
  }  // namespace hip

  /// Device-backed FFT implementation (every even size in [2, 8192], any factorisation, like libear's kissfft).
  inline FFTImpl<float> &get_fft_hip() {
    static hip::FFTHip fft;
    return fft;
  }

  /// libear's accessor of its always-available implementation (include/ear/fft.hpp:64-67), so that application
  /// code written against libear compiles unchanged: here it IS the device transform (same contract:
  /// un-normalised r2c / c2r, n/2 + 1 unpacked bins).  float only, like the rest of the DSP path.
  template <typename Real>
  FFTImpl<Real> &get_fft_kiss();
  template <>
  inline FFTImpl<float> &get_fft_kiss<float>() {
    return get_fft_hip();
  }
}  // namespace ear
