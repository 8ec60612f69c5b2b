// ear/screen.hpp — the reference screen of ADM metadata (libear include/ear/screen.hpp:8-24, default
// src/screen.cpp:4-6).  Carried by the metadata for source compatibility; screen-related rendering (screenRef) is
// refused by the gain calculators, as in libear.  libear's boost::variant is a struct that converts from either
// alternative here.
#pragma once
#include "layout.hpp"

namespace ear {
  struct PolarScreen {
    double aspectRatio;
    PolarPosition centrePosition;
    double widthAzimuth;
  };
  struct CartesianScreen {
    double aspectRatio;
    CartesianPosition centrePosition;
    double widthX;
  };
  struct Screen {
    Screen(PolarScreen s) : isCartesian(false), polar(s), cartesian{0.0, CartesianPosition(), 0.0} {}
    Screen(CartesianScreen s) : isCartesian(true), polar{0.0, PolarPosition(), 0.0}, cartesian(s) {}
    bool isCartesian;
    PolarScreen polar;
    CartesianScreen cartesian;
  };
  /// 16:9 (1.78), straight ahead at distance 1, 58 degrees wide
  inline Screen getDefaultScreen() { return PolarScreen{1.78, PolarPosition(0.0, 0.0, 1.0), 58.0}; }
}  // namespace ear
