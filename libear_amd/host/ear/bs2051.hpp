// ear/bs2051.hpp — libear's include/ear/bs2051.hpp:8-11: loadLayouts() and getLayout() live with the layout
// classes here (ear/layout.hpp, over the native BS.2051 table of earhip group H); this header keeps libear's
// include path working.
#pragma once
#include "layout.hpp"
