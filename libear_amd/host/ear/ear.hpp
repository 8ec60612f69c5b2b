// ear/ear.hpp — the umbrella header of libear (include/ear/ear.hpp): layouts (getLayout / loadLayouts), decorrelator
// design, gain calculators with their metadata, warnings and exception types.
#pragma once
#include "layout.hpp"  // (+ exceptions through hip.hpp; ear/bs2051.hpp is an alias of this header)
#include "metadata.hpp"
#include "warnings.hpp"
#include "decorrelate.hpp"
#include "gain_calculators.hpp"
