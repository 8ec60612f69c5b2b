// oracle/ref_kiss_capi.cpp — C entry points over the REFERENCE's own vendored
// kissfft.hh (compiled in place from /root/reference/submodules/kissfft, never
// copied).  Builds into oracle/_ref/libref_kiss.so (git-ignored).  Used only to
// check the oracle's FFT restatement bit-for-bit and to generate the golden
// vectors under tests/golden/ (see tests/golden/make_golden.py).
//
// The rest of the reference's hot path cannot be compiled in this image:
// every libear header reaches ear/export.hpp -> generated/export.hpp (a CMake
// product), and the .cpp files need Eigen and Boost, which are absent.
#include <complex>
#include <cstddef>

#include "kissfft.hh"

extern "C" {
void ref_kiss_cfft_f32(size_t n, int inverse, const float *in, float *out) {
  kissfft<float> f(n, inverse != 0);
  f.transform(reinterpret_cast<const std::complex<float> *>(in),
              reinterpret_cast<std::complex<float> *>(out));
}
void ref_kiss_cfft_f64(size_t n, int inverse, const double *in, double *out) {
  kissfft<double> f(n, inverse != 0);
  f.transform(reinterpret_cast<const std::complex<double> *>(in),
              reinterpret_cast<std::complex<double> *>(out));
}
// real transform of 2*nhalf reals through an nhalf-point plan; packed output
// (Nyquist in out[0].imag), nhalf complex values
void ref_kiss_rfft_packed_f32(size_t nhalf, const float *in, float *out) {
  kissfft<float> f(nhalf, false);
  f.transform_real(in, reinterpret_cast<std::complex<float> *>(out));
}
}
