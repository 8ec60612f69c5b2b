// ear/warnings.hpp — the warning callback of libear's gain calculators (interface: include/ear/warnings.hpp:6-38,
// default printer: src/warnings.cpp:5-11).  Header-only here; the codes keep libear's names and values so that
// application code that switches on them is unaffected.
#pragma once
#include <cstdio>
#include <functional>
#include <string>

namespace ear {
  struct Warning {
    // 1: LFE indication from the frequency element does not match speakerLabel; 2: frequency indication present
    // but not that of an LFE channel; 3: frequency information ignored; 4, 5: HOA screenRef / nfcRefDist ignored
    enum class Code : int {
      FREQ_SPEAKERLABEL_LFE_MISMATCH = 1, FREQ_NOT_LFE = 2, FREQ_IGNORED = 3,
      HOA_SCREENREF_NOT_IMPLEMENTED = 4, HOA_NFCREFDIST_NOT_IMPLEMENTED = 5
    };
    Code code;            // for programs that act on a warning
    std::string message;  // for people: complete without the code
  };

  // what `calculate` calls take and report through
  typedef std::function<void(const Warning &)> WarningCB;

  // libear's default: one line on stderr, "libear: warning: <message>"
  static const WarningCB default_warning_cb = [](const Warning &w) { std::fprintf(stderr, "libear: warning: %s\n", w.message.c_str()); };
}  // namespace ear
