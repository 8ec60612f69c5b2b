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
#include "layout_table.h"

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
// ear::getLayout (src/bs2051.cpp:13-22): unknown names are an error
const LayoutEntry &find_layout(const char *name) {
  require(name != nullptr, "layout name must not be NULL");
  for (int i = 0; i < kNumLayouts; i++)
    if (std::strcmp(kLayouts[i].name, name) == 0) return kLayouts[i];
  throw Error{EARHIP_UNKNOWN_LAYOUT, std::string("unknown layout: ") + name};
}

void design_for_names(const std::vector<std::string> &names, float *out) {
  for (size_t c = 0; c < names.size(); c++) {
    int id = 0;
    for (auto &n : names)
      if (n < names[c]) id++;
    const auto h = design_basic(id, kDecorrelatorSize);
    for (int i = 0; i < kDecorrelatorSize; i++) out[c * kDecorrelatorSize + i] = (float)h[i];
  }
}
}  // namespace

extern "C" {

int earhip_layout_count(void) { return kNumLayouts; }

const char *earhip_layout_name(int index) {
  return index >= 0 && index < kNumLayouts ? kLayouts[index].name : nullptr;
}

int earhip_layout_num_channels(const char *layout, int *n_channels) {
  return guarded([&] {
    require(n_channels != nullptr, "n_channels must not be NULL");
    *n_channels = find_layout(layout).n;
  });
}

int earhip_layout_channel(const char *layout, int index, const char **name, double *azimuth,
                          double *elevation, int *is_lfe) {
  return guarded([&] {
    const LayoutEntry &L = find_layout(layout);
    require(index >= 0 && index < L.n, "channel index out of range");
    const LayoutChannel &c = L.channels[index];
    if (name) *name = c.name;
    if (azimuth) *azimuth = c.azimuth;
    if (elevation) *elevation = c.elevation;
    if (is_lfe) *is_lfe = c.is_lfe ? 1 : 0;
  });
}

int earhip_layout_channel_ranges(const char *layout, int index, double azimuth_range[2], double elevation_range[2]) {
  return guarded([&] {
    const LayoutEntry &L = find_layout(layout);
    require(index >= 0 && index < L.n, "channel index out of range");
    const LayoutChannel &c = L.channels[index];
    if (azimuth_range) azimuth_range[0] = c.az_lo, azimuth_range[1] = c.az_hi;
    if (elevation_range) elevation_range[0] = c.el_lo, elevation_range[1] = c.el_hi;
  });
}

// designDecorrelators(getLayout(name)) / designDecorrelators(getLayout(name).withoutLfe())
int earhip_design_decorrelators_for_layout(const char *layout, int without_lfe, float *out) {
  return guarded([&] {
    require(out != nullptr, "out must not be NULL");
    const LayoutEntry &L = find_layout(layout);
    std::vector<std::string> names;
    for (int c = 0; c < L.n; c++)
      if (!(without_lfe && L.channels[c].is_lfe)) names.push_back(L.channels[c].name);
    design_for_names(names, out);
  });
}

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
    design_for_names(names, out);
  });
}

}  // extern "C"
