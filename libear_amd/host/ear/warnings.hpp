// ear/warnings.hpp — libear's warning callback (include/ear/warnings.hpp:6-38, src/warnings.cpp:5-11): the gain
// calculators report what they ignore through it.  Same type names and codes; header-only.
#pragma once
#include <cstdio>
#include <functional>
#include <string>

namespace ear {
  struct Warning {
    enum class Code {
      FREQ_SPEAKERLABEL_LFE_MISMATCH = 1,  ///< LFE indication from frequency element does not match speakerLabel
      FREQ_NOT_LFE,                        ///< frequency indication present but does not indicate an LFE channel
      FREQ_IGNORED,                        ///< frequency information is not implemented; ignoring
      HOA_SCREENREF_NOT_IMPLEMENTED,       ///< screenRef for HOA is not implemented; ignoring
      HOA_NFCREFDIST_NOT_IMPLEMENTED,      ///< nfcRefDist is not implemented; ignoring
    };
    Code code;
    std::string message;
  };

  /// passed into `calculate` calls, called with any warnings
  using WarningCB = std::function<void(const Warning &warning)>;

  /// prints to stderr with the prefix `libear: warning: `
  inline void default_warning_cb_fn(const Warning &warning) {
    std::fputs("libear: warning: ", stderr);
    std::fputs(warning.message.c_str(), stderr);
    std::fputc('\n', stderr);
  }
  static const WarningCB default_warning_cb = default_warning_cb_fn;
}  // namespace ear
