// ear/common_types.hpp — libear's include/ear/common_types.hpp:8-28: PolarPosition / CartesianPosition are
// defined with the layout classes (ear/layout.hpp), the Position variant of the metadata in ear/metadata.hpp;
// this header keeps libear's include path working.
#pragma once
#include "layout.hpp"
