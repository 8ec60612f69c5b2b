// ear/decorrelate.hpp — decorrelator design, libear include/ear/decorrelate.hpp:16-34.
// libear takes a Layout; the hot path needs only the channel names (the filter
// id of a channel is the rank of its name, src/decorrelate.cpp:55-68).
#pragma once
#include <string>
#include <vector>

#include "hip.hpp"

namespace ear {
  inline std::vector<std::vector<float>> designDecorrelators(
      const std::vector<std::string> &channel_names) {
    const int n = (int)channel_names.size();
    const int taps = earhip_decorrelator_size();
    std::vector<const char *> names(n);
    for (int i = 0; i < n; i++) names[i] = channel_names[i].c_str();
    std::vector<float> flat((size_t)n * taps);
    hip::check(earhip_design_decorrelators(n, names.data(), flat.data()));
    std::vector<std::vector<float>> out(n);
    for (int i = 0; i < n; i++)
      out[i].assign(flat.begin() + (size_t)i * taps, flat.begin() + (size_t)(i + 1) * taps);
    return out;
  }
  /// designDecorrelators(getLayout(name)) — or of getLayout(name).withoutLfe() — by BS.2051 layout
  /// name (the native side holds the channel-name table, src/bs2051_layouts.cpp)
  inline std::vector<std::vector<float>> designDecorrelators(const char *layout_name,
                                                             bool without_lfe = false) {
    int n = 0;
    hip::check(earhip_layout_num_channels(layout_name, &n));
    std::vector<std::string> names;
    for (int i = 0; i < n; i++) {
      const char *name = nullptr;
      int lfe = 0;
      hip::check(earhip_layout_channel(layout_name, i, &name, nullptr, nullptr, &lfe));
      if (!(without_lfe && lfe)) names.push_back(name);
    }
    return designDecorrelators(names);
  }
  inline std::vector<double> designDecorrelatorBasic(int decorrelatorId, int size) {
    std::vector<double> out(size);
    hip::check(earhip_design_decorrelator_basic(decorrelatorId, size, out.data()));
    return out;
  }
  inline int decorrelatorCompensationDelay() { return earhip_decorrelator_compensation_delay(); }
}  // namespace ear
