// ear/decorrelate.hpp — decorrelator design, libear include/ear/decorrelate.hpp:16-34: designDecorrelator(layout,
// channelIdx), designDecorrelators(layout), decorrelatorCompensationDelay(), plus forms that take only what the
// design needs — the channel names (the filter id of a channel is the rank of its name, src/decorrelate.cpp:55-68)
// or a BS.2051 layout name.
#pragma once
#include <string>
#include <vector>

#include "hip.hpp"
#include "layout.hpp"

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

  /// libear's own signatures (include/ear/decorrelate.hpp:16-27).  T = double gives the design before its cast to
  /// float (src/decorrelate.cpp:55-80).
  template <typename T = float>
  std::vector<T> designDecorrelator(const Layout &layout, size_t channelIdx) {
    const std::vector<std::string> names = layout.channelNames();
    if (channelIdx >= names.size()) throw invalid_argument("channel index out of range");
    int id = 0;  // the number of channel names that sort before this one (:63-65)
    for (auto &n : names)
      if (n < names[channelIdx]) id++;
    const std::vector<double> basic = designDecorrelatorBasic(id, earhip_decorrelator_size());
    return std::vector<T>(basic.begin(), basic.end());
  }
  template <typename T = float>
  std::vector<std::vector<T>> designDecorrelators(const Layout &layout) {
    std::vector<std::vector<T>> out;
    for (size_t i = 0; i < layout.channels().size(); i++) out.push_back(designDecorrelator<T>(layout, i));
    return out;
  }
}  // namespace ear
