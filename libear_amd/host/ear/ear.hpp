// ear/ear.hpp — libear's umbrella header (include/ear/ear.hpp:1-5)
#pragma once
#include "bs2051.hpp"
#include "decorrelate.hpp"
#include "exceptions.hpp"
#include "gain_calculators.hpp"
