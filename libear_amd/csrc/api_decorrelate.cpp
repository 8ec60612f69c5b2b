// api_decorrelate.cpp — decorrelator FIR design on the native side (setup path,
// runs once; libear src/decorrelate.cpp:16-97): all-pass random-phase FIR from
// mt19937(id), id = rank of the channel name among the layout's names, inverse
// DFT in double, compensation delay (size-1)/2.  Host code: the reference runs
// this on the CPU as well; its output feeds the device decorrelators.
#include <cmath>
#include <complex>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "common.h"

using namespace earhip;

namespace {
const int kDecorrelatorSize = 512;  // decorrelate.cpp:53

// decorrelate.cpp:31-51
std::vector<double> design_basic(int id, int size) {
  const double pi = 3.14159265358979323846264338327950288;
  std::mt19937 eng((std::mt19937::result_type)id);
  std::vector<std::complex<double>> fd(size);
  fd[0] = 1.0;
  for (int i = 0; i < size / 2 - 1; i++) {
    const double u = eng() / 4294967296.0;  // :18-21
    fd[i + 1] = std::polar(1.0, 2.0 * pi * u);
  }
  fd[size / 2] = 1.0;
  for (int i = 1; i < size / 2; i++) fd[size - i] = std::conj(fd[i]);
  // un-normalised inverse DFT, radix-2 decimation in time in double, then / size
  std::vector<std::complex<double>> a(size);
  int bits = 0;
  while ((1 << bits) < size) bits++;
  for (int i = 0; i < size; i++) {
    int r = 0;
    for (int b = 0; b < bits; b++)
      if (i & (1 << b)) r |= 1 << (bits - 1 - b);
    a[r] = fd[i];
  }
  for (int len = 2; len <= size; len <<= 1)
    for (int s = 0; s < size; s += len)
      for (int k = 0; k < len / 2; k++) {
        const std::complex<double> w = std::polar(1.0, 2.0 * pi * k / len);
        const std::complex<double> t = w * a[s + k + len / 2];
        a[s + k + len / 2] = a[s + k] - t;
        a[s + k] += t;
      }
  std::vector<double> h(size);
  for (int i = 0; i < size; i++) h[i] = a[i].real() / size;
  return h;
}
}  // namespace

extern "C" {

int earhip_decorrelator_size(void) { return kDecorrelatorSize; }

// decorrelate.cpp:97
int earhip_decorrelator_compensation_delay(void) { return (kDecorrelatorSize - 1) / 2; }

// decorrelate.cpp:31-51 (double precision, `size` must be a power of two >= 4)
int earhip_design_decorrelator_basic(int decorrelator_id, int size, double *out) {
  return guarded([&] {
    require(out != nullptr, "out must not be NULL");
    require(size >= 4 && is_pow2((size_t)size), "size must be a power of two >= 4");
    const auto h = design_basic(decorrelator_id, size);
    std::memcpy(out, h.data(), sizeof(double) * size);
  });
}

// decorrelate.cpp:55-90: one 512-tap float filter per channel; the filter id of
// a channel is the number of channel names that sort before its name.
int earhip_design_decorrelators(int n_channels, const char *const *channel_names, float *out) {
  return guarded([&] {
    require(n_channels >= 1 && channel_names != nullptr && out != nullptr, "NULL or empty argument");
    std::vector<std::string> names(n_channels);
    for (int c = 0; c < n_channels; c++) {
      require(channel_names[c] != nullptr, "channel name must not be NULL");
      names[c] = channel_names[c];
    }
    for (int c = 0; c < n_channels; c++) {
      int id = 0;
      for (auto &n : names)
        if (n < names[c]) id++;
      const auto h = design_basic(id, kDecorrelatorSize);
      for (int i = 0; i < kDecorrelatorSize; i++) out[(size_t)c * kDecorrelatorSize + i] = (float)h[i];
    }
  });
}

}  // extern "C"
